#!/usr/bin/env python3
"""Turns the rocprofv3 passes of experiments/profile_r02.sh into pmc.json (per kernel:
every counter's average per dispatch, the kernel's average duration in the trace-only pass
and under the counters) and prints a readable summary.

    python3 experiments/pmc_to_json.py gpurun_out/prof_<tag>  > pmc_summary.txt
"""
import collections
import csv
import glob
import json
import os
import sys

KERNELS = ("match_lane_compact_parts_kernel", "match_lane_gather6_parts_kernel", "match_lane_gather6_kernel",
           "match_lane_combine_kernel", "match_lane_compact_kernel", "match_lane_kernel", "match_kernel", "match_small_kernel", "score_poses_compact_kernel",
           "outer_table_kernel", "match_reduce_kernel", "score_one_kernel")


def base(name):
    if "ndt2d" not in name:
        return None
    for k in KERNELS:
        if ("::" + k + "<") in name or ("::" + k + "(") in name or name.endswith("::" + k):
            return k
    return None


def main(root):
    out = collections.defaultdict(dict)
    for pass_dir in sorted(glob.glob(os.path.join(root, "*"))):
        if not os.path.isdir(pass_dir):
            continue
        pname = os.path.basename(pass_dir)
        for path in glob.glob(os.path.join(pass_dir, "**", "*counter_collection.csv"), recursive=True):
            agg = collections.defaultdict(lambda: collections.defaultdict(list))
            for r in csv.DictReader(open(path)):
                k = base(r["Kernel_Name"])
                if k:
                    agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            for k, counters in agg.items():
                for c, vals in counters.items():
                    out[k][c] = sum(vals) / len(vals)
                    out[k].setdefault("dispatches_per_pass", {})[pname] = len(vals)
        for path in glob.glob(os.path.join(pass_dir, "**", "*kernel_trace.csv"), recursive=True):
            dur = collections.defaultdict(list)
            for r in csv.DictReader(open(path)):
                k = base(r["Kernel_Name"])
                if k:
                    dur[k].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
            for k, vals in dur.items():
                out[k].setdefault("avg_duration_ns", {})[pname] = sum(vals) / len(vals)
    doc = {"source": "experiments/profile_r0N.sh: rocprofv3 --kernel-trace [--pmc ...] on "
                     "`python3 bench.py --steps 4 --warmup 2 --prewarm 0 --no-cpu-baseline --no-default-search --no-anchors` "
                     "(kt: --stats pass, --steps 200 --warmup 10 behind the 0.5 s pre-warm), one MI355X; counters are averages "
                     "per dispatch; FETCH_SIZE / WRITE_SIZE in KiB as rocprofv3 reports them",
           "kernels": out}
    with open(os.path.join(root, "pmc.json"), "w") as f:
        json.dump(doc, f, indent=1, sort_keys=True)
    for k, v in sorted(out.items()):
        print(k)
        for c, val in sorted(v.items()):
            if isinstance(val, dict):
                print("   %-28s %s" % (c, json.dumps(val, sort_keys=True)))
            else:
                print("   %-28s %.6g" % (c, val))
        if "SQ_INSTS_VALU" in v and "SQ_BUSY_CU_CYCLES" in v:
            # VALU issue: 4 cycles per wave-instruction on one of 4 SIMDs per CU;
            # SQ_BUSY_CU_CYCLES sums the busy cycles of the 256 CUs
            print("   %-28s %.4f" % ("valu_issue_frac = INSTS_VALU*4/(4*BUSY_CU_CYCLES)",
                                     v["SQ_INSTS_VALU"] * 4.0 / (4.0 * v["SQ_BUSY_CU_CYCLES"])))


if __name__ == "__main__":
    main(sys.argv[1])
