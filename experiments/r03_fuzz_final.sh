# the fuzz batteries and the soak on the round's final code: mapper cycles on a persistent matcher
# (incl. the scoreScan / matchScan pair: searches launched ahead, collected and dropped), random
# cases, large lattices, wide windows (each prints "N failures")
cd $GRAFT_REPO_ROOT
O=gpurun_out/fuzz_final; rm -rf $O; mkdir -p $O
for s in 50 51 52 53 54 55 56 57; do
  timeout 600 python experiments/fuzz_cycle.py $((s * 100000)) 5000 > $O/cycle_$s.txt 2>&1
done
timeout 1500 python experiments/fuzz_more.py 200000 20000 > $O/random_cases.txt 2>&1
timeout 900 python experiments/fuzz_large.py 30000 1500 > $O/large.txt 2>&1
timeout 900 python experiments/fuzz_wide.py 30000 1500 > $O/wide.txt 2>&1
timeout 900 python experiments/soak_r03.py 45 > $O/soak.txt 2>&1
grep -h "failures\|identical\|soak ok\|differs" $O/*.txt
