cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06c
for i in 1 2; do
for t in 1 0; do experiments/bin/host_build_phases $t 0; NOIL=1 experiments/bin/host_build_phases $t 0; ILALL=1 experiments/bin/host_build_phases $t 0; done
done > gpurun_out/r06c/host_build_phases2.txt 2>&1
cat gpurun_out/r06c/host_build_phases2.txt
