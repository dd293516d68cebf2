# the fuzz batteries on the round's final code: mapper cycles on a persistent matcher, random cases,
# large lattices, wide windows (each prints "N failures")
cd $GRAFT_REPO_ROOT
O=gpurun_out/fuzz_final; mkdir -p $O
for s in 0 1 2 3 4 5 6 7 8 9 10 11; do
  timeout 600 python experiments/fuzz_cycle.py $((s * 100000)) 5000 > $O/cycle_$s.txt 2>&1
done
timeout 1500 python experiments/fuzz_more.py 100000 20000 > $O/random_cases.txt 2>&1
timeout 900 python experiments/fuzz_large.py 20000 1500 > $O/large.txt 2>&1
timeout 900 python experiments/fuzz_wide.py 20000 1500 > $O/wide.txt 2>&1
grep -h "failures" $O/*.txt
