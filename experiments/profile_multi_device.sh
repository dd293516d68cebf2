# The multi-device matcher through the plain-C probe on the one GPU available: step times per
# device list / exchange, and a kernel trace of three contexts driven by one host thread (do the
# three searches of a step overlap in time?).  Leaves gpurun_out/multi_device/{summary.json,overlap.txt}.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/multi_device
rm -rf $O && mkdir -p $O
P=$R/ndt_2d_amd/ndt2d_latency_probe
for spec in "0 host" "0 rccl" "0,0 host" "0,0,0 host" "0,0,0,0,0,0,0,0 host"; do
  set -- $spec
  $P --devices $1 --exchange $2 2>/dev/null | grep '^{"devices"' >> $O/lines.jsonl
done
rocprofv3 --kernel-trace --output-format csv -d $O/kt -- $P --devices 0,0,0 --exchange host > $O/kt.log 2>&1
python3 - <<'PY'
import csv, glob, json, os
O = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/multi_device"
lines = [json.loads(l) for l in open(O + "/lines.jsonl")]
json.dump({"source": "experiments/profile_multi_device.sh: ndt2d_latency_probe --devices <ids> --exchange <mode>, one MI355X "
                     "(several contexts on ONE GPU compete for it: these are not scaling numbers)", "runs": lines},
          open(O + "/summary.json", "w"), indent=1)
rows = []
for path in glob.glob(O + "/kt/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        if "match_lane_compact_kernel" in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Stream_Id", r.get("Queue_Id", "?"))))
rows.sort()
# the last cfg-4 step: the three longest-running recent kernels
big = [r for r in rows if r[1] - r[0] > 5_000_000][-3:]
with open(O + "/overlap.txt", "w") as f:
    f.write("last cfg-4 step of `ndt2d_latency_probe --devices 0,0,0 --exchange host` (rocprofv3 --kernel-trace): the three\n"
            "devices' search kernels, start / end in us relative to the first start, queue\n")
    if big:
        t0 = min(b[0] for b in big)
        for b in big:
            f.write("  start %9.1f  end %9.1f  (%.2f ms)  queue %s\n" % ((b[0] - t0) / 1e3, (b[1] - t0) / 1e3, (b[1] - b[0]) / 1e6, b[2]))
        lo, hi = max(b[0] for b in big), min(b[1] for b in big)
        f.write("  all three in flight together for %.2f ms of the step's %.2f ms\n" % (max(0, hi - lo) / 1e6, (max(b[1] for b in big) - t0) / 1e6))
print(open(O + "/overlap.txt").read())
for l in lines:
    print(l["devices"], l["exchange_requested"], l["cfg2"]["step_ms"], l["cfg4"]["step_ms"], l["cfg4"]["variant"][:18])
PY
