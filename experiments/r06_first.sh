# round 6, first GPU call: the GPU suite, then the host side of the mapper's cycle on the box's EPYC
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06a
mkdir -p $O
cd $R
python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?" > $O/rc.txt
for t in 1 0; do experiments/bin/host_build_phases $t 0; NOIL=1 experiments/bin/host_build_phases $t 0; done > $O/host_build_phases.txt 2>&1
experiments/bin/cycle_breakdown 2000 1 > $O/cycle_toy.txt 2>&1
experiments/bin/cycle_breakdown 2000 0 > $O/cycle_big.txt 2>&1
ndt_2d_amd/ndt2d_latency_probe > $O/probe.json 2> $O/probe.err
python bench.py --steps 20 --warmup 5 > $O/bench_driver_flags.json 2> $O/bench.err
wc -c $O/bench_driver_flags.json
tail -3 $O/gpu_tests.log; cat $O/rc.txt $O/host_build_phases.txt $O/cycle_toy.txt $O/cycle_big.txt; cat $O/probe.json
