# A/B timing of two builds of the library in one session: bash experiments/ab.sh libA.so libB.so
for i in 1 2 3; do
  for L in "$@"; do
    NDT2D_HIP_LIB=$PWD/experiments/bin/$L python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-particles 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$L', d['ms_per_step'], d['roofline']['kernel_ms'] if 'kernel_ms' in d['roofline'] else '')"
  done
done
