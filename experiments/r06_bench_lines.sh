# the three committed bench records of the round (stdout line + --detail-file), then the GPU suite
cd $GRAFT_REPO_ROOT
O=gpurun_out/prof_r06b; mkdir -p $O
python3 bench.py --detail-file $PWD/$O/bench_detail.json > $O/bench.json 2> $O/bench.err
python3 bench.py --steps 20 --warmup 5 --detail-file $PWD/$O/bench_driver_flags_detail.json > $O/bench_driver_flags.json 2>> $O/bench.err
NDT2D_BENCH_BACKEND=gloo python3 bench.py --gpus 8 --steps 4 --warmup 2 --detail-file $PWD/$O/bench_8ranks_one_gpu_gloo_detail.json > $O/bench_8ranks_one_gpu_gloo.json 2>> $O/bench.err
wc -c $O/bench.json $O/bench_driver_flags.json $O/bench_8ranks_one_gpu_gloo.json
python3 -c "
import json
for n in ('bench','bench_driver_flags','bench_8ranks_one_gpu_gloo'):
    d=json.load(open('$O/'+n+'.json')); print(n, d['value'], d['ms_per_step'], d['roofline']['frac'], d['library'])"
python -m pytest tests -m gpu -q > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; grep -n "passed\|failed" $O/gpu_tests.log | tail -3
