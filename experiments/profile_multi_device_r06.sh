# Round 6: what dealing a call to N device contexts costs, through the plain-C probe on the one
# GPU available (ndt2d_latency_probe --devices <ids> --workload all): cfg-2 / cfg-4 searches and
# cfg-5's 1,000,000-particle measure, per device list -- call time, when each device's launch
# had been queued (fan-out), every share measured alone.  Contexts that share ONE GPU run their
# shares one after the other: call - sum(shares) is the dealing overhead there; on N distinct
# GPUs it would be call - slowest share.  Leaves gpurun_out/r06/multi_device/summary.json.
#     bash experiments/profile_multi_device_r06.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06/multi_device
rm -rf $O && mkdir -p $O
P=$R/ndt_2d_amd/ndt2d_latency_probe
for spec in "0 host" "0,0 host" "0,0,0,0 host" "0,0,0,0,0,0,0,0 host" "0 rccl"; do
  set -- $spec
  $P --devices $1 --exchange $2 --workload all 2>> $O/probe.err | grep '^{"devices"' >> $O/lines.jsonl
done
python3 - <<'PY'
import json, os
O = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r06/multi_device"
lines = [json.loads(l) for l in open(O + "/lines.jsonl")]
json.dump({"source": "experiments/profile_multi_device_r06.sh: ndt2d_latency_probe --devices <ids> --exchange <mode> --workload all, "
                     "one MI355X; several contexts on ONE GPU share it (their shares run one after the other): "
                     "call_minus_sum_of_shares_us is the dealing overhead here, call_minus_slowest_share_us what it would be "
                     "on distinct GPUs if the shares overlapped perfectly; fanout_us = when each device's launch had been "
                     "queued, from the call's start", "runs": lines}, open(O + "/summary.json", "w"), indent=1)
for l in lines:
    ov = l.get("dealing_overhead")
    if ov:
        print(l["devices"], l["exchange_requested"], "dealing overhead: search %.1f -> %.1f us (+%.1f), pf_measure %.1f -> %.1f us (+%.1f)" % (
            ov["search"]["one_device_call_us"], ov["search"]["dealt_call_us"], ov["search"]["difference_us"],
            ov["pf_measure"]["one_device_call_us"], ov["pf_measure"]["dealt_call_us"], ov["pf_measure"]["difference_us"]))
    for w in ("cfg2", "cfg4", "cfg5"):
        c = l.get(w, {})
        print(l["devices"], l["exchange_requested"], w, "call %.3f ms" % c.get("step_ms", c.get("call_ms", 0)),
              "fanout last %.1f skew %.1f" % (c.get("fanout_last_us", -1), c.get("fanout_skew_us", -1)),
              "slowest %.3f sum %.3f" % (c.get("slowest_share_ms", -1), c.get("sum_of_shares_ms", -1)),
              "call-sum %.1f us" % c.get("call_minus_sum_of_shares_us", float("nan")), c.get("variant", "")[:16])
PY
tail -5 $O/probe.err
