# Copies what experiments/profile_r04.sh left under gpurun_out/prof_<tag>/ into profiles/ as the
# committed round-4 artefacts (the three bench lines re-written as indented JSON objects).
TAG=${1:-r04}
O=gpurun_out/prof_$TAG
cp $O/pmc.json profiles/r04_pmc.json
cp $O/pmc_summary.txt profiles/r04_pmc_summary.txt
cp $O/shares_summary.txt profiles/r04_shares_summary.txt
cp $O/kernel_stats.csv profiles/r04_kernel_stats.csv
python3 - "$O" <<'PY'
import json, sys
O = sys.argv[1]
for src, dst in (("bench.json", "r04_bench.json"), ("bench_driver_flags.json", "r04_bench_driver_flags.json"),
                 ("bench_8ranks_one_gpu_gloo.json", "r04_bench_8ranks_one_gpu_gloo.json")):
    line = json.loads(open(O + "/" + src).read().strip().splitlines()[-1])
    json.dump(line, open("profiles/" + dst, "w"), indent=1)
    r = line["roofline"]
    print("%-36s value %.4g  ms/step %.4f  kernel %.4f  frac %.3f  pmc_matches_source %s" %
          (dst, line["value"], line["ms_per_step"], r["kernel_ms_avg"], r["frac"], r["pmc_matches_source"]))
PY
