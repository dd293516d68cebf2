cd $GRAFT_REPO_ROOT
O=gpurun_out/r06d; mkdir -p $O
NDT2D_HIP_LIB=$PWD/experiments/bin/lane_paths.so python experiments/lane_paths.py 2 > $O/lane_paths_cfg2.json 2> $O/lane_paths.err
cat $O/lane_paths_cfg2.json
# particle kernel: four- against eight-wave groups around cfg-3's size and an 8-GPU share of cfg-5
for w in 0 1; do NDT2D_POSES_EIGHT_WAVES=$w python experiments/particles_split_ab.py > $O/particles_eight_$w.txt 2>&1; done
python experiments/particles_split_ab.py > $O/particles_default.txt 2>&1
tail -n 12 $O/particles_eight_0.txt $O/particles_eight_1.txt $O/particles_default.txt
# cfg-5: what the scoring kernel gains from particles sorted by grid tile (upper bound of VERDICT r05 item 6)
python experiments/particles_sorted_ab.py > $O/particles_sorted.txt 2>&1
cat $O/particles_sorted.txt
