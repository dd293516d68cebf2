#!/usr/bin/env python3
"""Adds to <root>/pmc.json (experiments/pmc_to_json.py) what round 4's bench line needs:

  * "shares": per 1-of-8 share of the two 8-GPU workloads, the counters of its ONE launch on
    one GPU -- from rocprofv3 --pmc passes of `python3 bench.py --profile-shares` under
    <root>/shares/<pass>/ (that command launches share 0 as a warm-up and then shares 0..7 in
    order: the LAST eight dispatches of the kernel are the shares);
  * "source_sha256": the hash of the kernel sources the counters were taken with
    (bench.py --print-source-hash), which bench.py compares with the sources it runs.

    python3 experiments/pmc_shares.py gpurun_out/prof_<tag> <sha256>
"""
import collections
import csv
import glob
import json
import os
import sys

KERNEL_OF = {"cfg4": "match_lane_compact_kernel", "cfg5": "score_poses_compact_kernel"}


def short(name):
    for k in KERNEL_OF.values():
        if ("::" + k + "<") in name or ("::" + k + "(") in name or name.endswith("::" + k):
            return k
    return None


def main(root, sha):
    path = os.path.join(root, "pmc.json")
    with open(path) as f:
        doc = json.load(f)
    shares = {w: [dict() for _ in range(8)] for w in KERNEL_OF}
    for csv_path in sorted(glob.glob(os.path.join(root, "shares", "*", "**", "*counter_collection.csv"), recursive=True)):
        # kernel -> counter -> {dispatch id: value}
        per = collections.defaultdict(lambda: collections.defaultdict(dict))
        meta = collections.defaultdict(dict)
        for r in csv.DictReader(open(csv_path)):
            k = short(r["Kernel_Name"])
            if k is None:
                continue
            d = int(r["Dispatch_Id"])
            per[k][r["Counter_Name"]][d] = per[k][r["Counter_Name"]].get(d, 0.0) + float(r["Counter_Value"])
            meta[k][d] = (float(r["End_Timestamp"]) - float(r["Start_Timestamp"]), int(r["Scratch_Size"]),
                          int(r["VGPR_Count"]), int(r["SGPR_Count"]))
        for w, k in KERNEL_OF.items():
            for counter, by_dispatch in per.get(k, {}).items():
                ids = sorted(by_dispatch)
                if len(ids) < 8:
                    continue
                for r, d in enumerate(ids[-8:]):
                    shares[w][r][counter] = by_dispatch[d]
                    shares[w][r].setdefault("duration_ns_under_counters", {})[counter] = meta[k][d][0]
                    shares[w][r]["scratch_size"], shares[w][r]["vgprs"], shares[w][r]["sgprs"] = meta[k][d][1:]
    doc["shares"] = shares
    doc["shares_source"] = ("rocprofv3 --kernel-trace --pmc ... -- python3 bench.py --profile-shares: one launch per "
                            "1-of-8 share (cfg-4: theta steps r, r + 8, ...; cfg-5: particle range r), one MI355X")
    doc["source_sha256"] = sha
    with open(path, "w") as f:
        json.dump(doc, f, indent=1, sort_keys=True)
    for w in shares:
        print("shares of", w)
        for r, sh in enumerate(shares[w]):
            print("   %d  %s" % (r, "  ".join("%s=%.6g" % (c, v) for c, v in sorted(sh.items())
                                              if not isinstance(v, dict))))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
