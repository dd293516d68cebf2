# timeline of the device NDT build's addScans (kernels + copies) at cfg-3 / cfg-5
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace -d $GRAFT_REPO_ROOT/gpurun_out/r05b/build_tl -o t -- python3 $GRAFT_REPO_ROOT/experiments/build_ab.py > /dev/null 2>&1
python3 $GRAFT_REPO_ROOT/experiments/rocpd_timeline.py $GRAFT_REPO_ROOT/gpurun_out/r05b/build_tl/t_results.db 300 > /tmp/tl.txt
awk '/points_kernel/{c++} c==4' /tmp/tl.txt | head -24
awk '/MEMORY_COPY_HOST_TO_DEVICE 6/{c++} c==4 && !p {p=1; print "--- around the 4th 6 MB upload"} ' /tmp/tl.txt
grep -n "MEMORY_COPY" /tmp/tl.txt | sed -n 20,34p
