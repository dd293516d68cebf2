#!/usr/bin/env python3
"""gpurun_out/r05/ubench/ubench_issue.jsonl (experiments/ubench_issue.hip) -> profiles/r05_ubench_issue.json:
per instruction the measured issue cost in shader cycles, the table bench.py's roofline prices with.

    python3 experiments/ubench_issue_to_json.py gpurun_out/r05/ubench/ubench_issue.jsonl > profiles/r05_ubench_issue.json
"""
import json
import sys

rows = [json.loads(line) for line in open(sys.argv[1]) if line.strip()]
device, rows = rows[0], rows[1:]
table = {}
for r in rows:
    e = table.setdefault(r["instruction"], {"pmc_class": r["pmc_class"], "ns_per_wave_instruction": {}, "clock_mhz": {}})
    e["ns_per_wave_instruction"][str(r["waves_per_simd"])] = round(r["ns_launch"], 4)
    e["clock_mhz"][str(r["waves_per_simd"])] = round(r["memtime_mhz"], 1)
cycles = {}
for name, e in table.items():
    # the issue cost: launch time per wave-instruction per SIMD x the shader clock of that same
    # launch (s_memtime over s_memrealtime), at the occupancies that hide the instruction's latency
    vals = [e["ns_per_wave_instruction"][w] * e["clock_mhz"][w] * 1e-3 for w in ("4", "6", "8")]
    e["cycles_measured"] = round(min(vals), 3)
    nominal = min((2.0, 4.0, 8.0, 16.0), key=lambda c: abs(c - e["cycles_measured"]) / c)
    e["cycles_nominal"] = nominal if abs(nominal - e["cycles_measured"]) / nominal < 0.2 else round(e["cycles_measured"], 1)
    if e["pmc_class"] != "mix" and "(dst" not in name:
        cycles[name] = e["cycles_nominal"]
# which SQ_INSTS_VALU_* counter counts the instruction (second argument: pmc_classes.json of the same run)
if len(sys.argv) > 2:
    counted = json.load(open(sys.argv[2]))
    for name, e in table.items():
        c = counted.get(name, {})
        e["counted_by"] = sorted(k for k in c if k not in ("SQ_INSTS_VALU", "busy_cu_cycles_per_inst_x1024")) or ["(no class counter)"]
        if "busy_cu_cycles_per_inst_x1024" in c:
            e["busy_cu_cycles_per_instruction"] = c["busy_cu_cycles_per_inst_x1024"]
# a select on the implicit vcc (VOP2) is slow only in a RUN of them: one among seven v_fma_f64 costs what they cost
mixed = table.get("7 x v_fma_f64 + 1 x v_cndmask_b32 vcc")
if mixed is not None:
    cycles["v_cndmask_b32 vcc"] = 4.0
    table["v_cndmask_b32 vcc"]["cycles_nominal_isolated"] = 4.0
    table["v_cndmask_b32 vcc"]["note"] = ("22.7 cycles each in a run of 128; one among seven v_fma_f64 leaves the loop at %.2f "
                                          "cycles per instruction, a compare + select pair at %.2f: priced at 4"
                                          % (mixed["cycles_measured"],
                                             table.get("v_cmp_lt_f64 vcc + v_cndmask_b32 vcc pairs", {}).get("cycles_measured", float("nan"))))
out = {
    "what": "VALU issue cost per wave64 instruction per SIMD on gfx950 (MI355X), experiments/ubench_issue.hip: "
            "128 copies of one instruction over 8 independent register chains x 1000 iterations, 1-8 waves per SIMD "
            "on all 1024 SIMDs; ns = launch duration (HIP events) / instructions per SIMD; cycles = ns x the shader "
            "clock of the same launch (s_memtime / s_memrealtime)",
    "device": device,
    "summary": {
        "two_cycle": sorted(n for n, e in table.items() if e["cycles_nominal"] == 2.0),
        "four_cycle": sorted(n for n, e in table.items() if e["cycles_nominal"] == 4.0),
        "slower": {n: e["cycles_nominal"] for n, e in sorted(table.items()) if e["cycles_nominal"] not in (2.0, 4.0)},
    },
    "cycles": cycles,
    "instructions": table,
}
json.dump(out, sys.stdout, indent=1)
print()
