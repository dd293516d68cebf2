# Copies what experiments/profile_<tag>.sh left under gpurun_out/prof_<tag>/ into profiles/ as the
# committed round artefacts (the three bench lines re-written as indented JSON objects).
TAG=${1:-r04}
O=gpurun_out/prof_$TAG
cp $O/pmc.json profiles/${TAG}_pmc.json
cp $O/pmc_summary.txt profiles/${TAG}_pmc_summary.txt
cp $O/shares_summary.txt profiles/${TAG}_shares_summary.txt
cp $O/kernel_stats.csv profiles/${TAG}_kernel_stats.csv
[ -f $O/multi_device_summary.json ] && cp $O/multi_device_summary.json profiles/${TAG}_multi_device_summary.json
python3 - "$O" "$TAG" <<'PY'
import json, sys
O, TAG = sys.argv[1], sys.argv[2]
for src, dst in (("bench.json", TAG + "_bench.json"), ("bench_driver_flags.json", TAG + "_bench_driver_flags.json"),
                 ("bench_8ranks_one_gpu_gloo.json", TAG + "_bench_8ranks_one_gpu_gloo.json")):
    line = json.loads(open(O + "/" + src).read().strip().splitlines()[-1])
    json.dump(line, open("profiles/" + dst, "w"), indent=1)
    r = line["roofline"]
    print("%-36s value %.4g  ms/step %.4f  kernel %.4f  frac %.3f  pmc_matches_source %s" %
          (dst, line["value"], line["ms_per_step"], r["kernel_ms_avg"], r["frac"], r["pmc_matches_source"]))
PY
