#!/usr/bin/env python3
"""Dynamic VALU instruction table of match_lane_compact_kernel at cfg-2: how often a wave takes each
path of the search (profiles/r06_lane_paths.json, counted by a -DNDT2D_LANE_PATHS build:
experiments/lane_paths.py) times the path's instruction list (read off the kernel's ISA: hipcc -S of
csrc/ndt2d_match_lane.hip, the eight unrolled copies of lane_beams<8> are identical up to
registers), checked class by class against the hardware's counters (profiles/r0N_pmc.json:
SQ_INSTS_VALU_*), priced with the measured issue cycles (profiles/r05_ubench_issue.json).

    python3 experiments/lane_path_mix.py [profiles/r06_lane_paths.json] [profiles/r06_pmc.json] > profiles/r06_lane_path_mix.json

What it answers: (1) VERDICT r05 item 3 -- which instructions the launch spends and what could still
go; (2) item 8 -- how many of the instructions in the two counter classes that mix 2- and 4-cycle
forms (INT32, and the unclassified rest) ARE 2-cycle forms: on the measured paths exactly, in the
per-item remainder (set-up, reduction, pre-test: the part no path counter covers) bracketed."""
import json
import os
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
paths_file = sys.argv[1] if len(sys.argv) > 1 else os.path.join(R, "profiles", "r06_lane_paths.json")
pmc_file = sys.argv[2] if len(sys.argv) > 2 else os.path.join(R, "profiles", "r06_pmc.json")
issue = json.load(open(os.path.join(R, "profiles", "r05_ubench_issue.json")))
CYC = issue["cycles"]
CLS = {k: ([c for c in v.get("counted_by", []) if not c.startswith("(")] or ["other"])[0]
       for k, v in issue["instructions"].items()}

# (instruction form as profiles/r05_ubench_issue.json names it, executions per path execution)
P = {
    # one look-up group of 8 beams: K + D, the byte select, the map byte; max over the group, wave test
    "lookup_group (per 8 beams)": [("v_add_f64", 8), ("v_perm_b32", 8), ("v_max_u32", 1), ("v_max3_u32", 3),
                                    ("v_cmp_le_u32 vcc", 1)],
    # a group some lane of which is live: every beam's live test
    "live_group (8 live tests)": [("v_cmp_le_u32 vcc", 8)],
    # a beam with a live lane: near-boundary tests (x, y), occupancy
    "live_beam": [("v_cmp_gt_u16 vcc", 2), ("v_perm_b32", 1), ("v_and_b32", 1), ("v_cmp_le_u32 vcc", 1)],
    # exact evaluation, interior of a cell: points_inner, look-up cell -> rank -> record, Cell::score's exponent, test
    "exact_evaluation": [("v_add_f64", 2),
                         ("v_lshrrev_b32_sdwa", 2), ("v_mul_u32_u24", 1), ("v_add3_u32", 1), ("v_cndmask_b32 sgpr", 1),
                         ("v_lshl_add_u32", 1), ("v_mad_u32_u24", 1),
                         ("v_add_f64", 5), ("v_mul_f64", 6), ("v_cmp_lt_f64 vcc", 1)],
    # ... a lane near a boundary: the reference's own index arithmetic instead of the look-up cell (replaces 5)
    "reference_index (extra)": [("v_add_f64", 2), ("v_mul_f64", 2), ("v_cmp_lt_f64 vcc", 4), ("v_cvt_u32_f64", 2),
                                ("v_mad_u64_u32", 1), ("v_mov_b32 from sgpr", 1), ("v_cndmask_b32 vcc", 1),
                                ("v_lshrrev_b32_sdwa", -2), ("v_mul_u32_u24", -1), ("v_add3_u32", -1),
                                ("v_cndmask_b32 sgpr", -1)],
    # exp() and the sum
    "exp_and_add": [("v_max_f64", 1), ("v_mul_f64", 1), ("v_rndne_f64", 1), ("v_fmac_f64", 2), ("v_fma_f64", 11),
                    ("v_cvt_i32_f64", 1), ("v_cmp_lt_f64 vcc", 1), ("v_ldexp_f64", 1), ("v_add_f64", 1)],
    # skip state refreshed after a group that added something
    "skip_refresh": [("v_frexp_exp_i32_f64", 1), ("v_add_u32", 2), ("v_mov_b32", 2), ("v_cvt_f64_i32", 1),
                     ("v_fma_f64", 1), ("v_cmp_lt_f64 vcc", 2), ("v_cndmask_b32 vcc", 2), ("v_ceil_f64", 1),
                     ("v_cvt_i32_f64", 1), ("v_max_u32", 1), ("v_min_u32", 1), ("v_lshlrev_b32", 1)],
}

doc = json.load(open(paths_file))
n = doc["counts"]
execs = {
    "lookup_group (per 8 beams)": n["beams_in_lookup_groups"] / 8.0,
    "live_group (8 live tests)": n["groups_with_a_live_lane"],
    "live_beam": n["beams_with_a_live_lane"],
    "exact_evaluation": n["exact_evaluations"],
    "reference_index (extra)": n["evaluations_by_reference_index"],
    "exp_and_add": n["evaluations_with_exp"],
    "skip_refresh": n["skip_refreshes"],
}
pmc = json.load(open(pmc_file))["kernels"]["match_lane_compact_kernel"] if os.path.exists(pmc_file) else None

by_class, two_cycle, rows, total = {}, {}, [], 0.0
for name, lst in P.items():
    per = sum(k for _, k in lst)
    cyc = sum(k * CYC[f] for f, k in lst)
    rows.append({"path": name, "executions": execs[name], "valu_per_execution": per,
                 "valu": execs[name] * per, "cycles_per_instruction": cyc / per if per else None})
    total += execs[name] * per
    for f, k in lst:
        c = CLS.get(f, "other")
        by_class[c] = by_class.get(c, 0.0) + execs[name] * k
        if CYC[f] == 2.0:
            two_cycle[c] = two_cycle.get(c, 0.0) + execs[name] * k
out = {"what": "match_lane_compact_kernel at cfg-2: path executions (profiles/%s) x the paths' instruction lists "
               "(ISA of csrc/ndt2d_match_lane.hip); experiments/lane_path_mix.py" % os.path.basename(paths_file),
       "paths": rows, "valu_on_counted_paths": total, "by_class_on_counted_paths": by_class,
       "two_cycle_on_counted_paths": two_cycle, "items": n["items"], "wave_beams": doc["wave_beams"]}
if pmc:
    tot = pmc["SQ_INSTS_VALU"]
    measured = {"ADD_F64": pmc.get("SQ_INSTS_VALU_ADD_F64"), "MUL_F64": pmc.get("SQ_INSTS_VALU_MUL_F64"),
                "FMA_F64": pmc.get("SQ_INSTS_VALU_FMA_F64"), "CVT": pmc.get("SQ_INSTS_VALU_CVT"),
                "INT32": pmc.get("SQ_INSTS_VALU_INT32"), "INT64": pmc.get("SQ_INSTS_VALU_INT64")}
    classified = sum(v for v in measured.values() if v)
    measured["other"] = tot - classified - sum(pmc.get(k, 0) or 0 for k in ("SQ_INSTS_VALU_ADD_F32", "SQ_INSTS_VALU_MUL_F32",
                                                                          "SQ_INSTS_VALU_FMA_F32", "SQ_INSTS_VALU_TRANS_F32",
                                                                          "SQ_INSTS_VALU_TRANS_F64"))
    rest = tot - total
    out["pmc"] = {"file": os.path.basename(pmc_file), "SQ_INSTS_VALU": tot, "by_class": measured}
    out["remainder"] = {"valu": rest, "share": rest / tot, "per_item": rest / n["items"],
                        "what": "per-item work no path counter covers: item placement, offsets, the pre-test of 12 x 64 beams, "
                                "the wave reduction of the item's record (12 values), staging"}
    out["class_check"] = {c: {"counted_paths": by_class.get(c, 0.0), "pmc": measured.get(c),
                              "paths_over_pmc": (by_class.get(c, 0.0) / measured[c]) if measured.get(c) else None}
                          for c in sorted(set(by_class) | set(k for k, v in measured.items() if v))}
    # The two mixed classes: 2-cycle instructions on the counted paths are known; in the remainder of the class
    # (measured minus counted paths) every instruction may be a 2- or a 4-cycle form.
    mixed = {}
    for c in ("INT32", "other"):
        m = measured[c]
        known = by_class.get(c, 0.0)
        two_known = two_cycle.get(c, 0.0)
        rem = max(m - known, 0.0)
        mixed[c] = {"instructions": m, "on_counted_paths": known, "two_cycle_on_counted_paths": two_known,
                    "remainder": rem,
                    "mean_cycles_low": (4.0 * (known - two_known) + 2.0 * two_known + 2.0 * rem) / m,
                    "mean_cycles_high": (4.0 * (known - two_known) + 2.0 * two_known + 4.0 * rem) / m}
        # the remainder priced with the kernel's static in-loop mix of that class (r0N_valu_mix.json) when there is one
        mixed[c]["mean_cycles"] = 0.5 * (mixed[c]["mean_cycles_low"] + mixed[c]["mean_cycles_high"])
    out["mixed_classes"] = mixed
    fixed = sum((measured[c] or 0) * 4.0 for c in ("ADD_F64", "MUL_F64", "FMA_F64", "CVT", "INT64"))
    lo = fixed + sum(mixed[c]["instructions"] * mixed[c]["mean_cycles_low"] for c in mixed)
    hi = fixed + sum(mixed[c]["instructions"] * mixed[c]["mean_cycles_high"] for c in mixed)
    out["cycles_per_instruction"] = {"low": lo / tot, "high": hi / tot, "mid": 0.5 * (lo + hi) / tot}
print(json.dumps(out, indent=1))
