# round 6: the whole GPU suite, smoke(), the default bench line and fresh cycle timelines in one call
cd $GRAFT_REPO_ROOT
TAG=${1:-r06v}
O=gpurun_out/$TAG; mkdir -p $O
python3 -m pytest tests -m gpu -q -x > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?"
tail -3 $O/gpu_tests.log
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log
python3 bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; wc -c $O/bench.json
bash experiments/r06_cycle.sh $TAG > $O/cycle.log 2>&1
tail -42 $O/cycle.log
