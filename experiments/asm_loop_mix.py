#!/usr/bin/env python3
"""Static VALU instruction mix of the hot kernels' loops, priced with the measured issue cycles
(profiles/r05_ubench_issue.json) and split the way rocprofv3's SQ_INSTS_VALU_* counters split
them -> profiles/r06_valu_mix.json, which bench.py's roofline prices a kernel's counters with.

    python3 experiments/asm_loop_mix.py > profiles/r06_valu_mix.json        (compiles the kernels itself)

The hardware counts VALU instructions per CLASS (FP64 add / mul / fma, CVT, INT32, INT64, FP32,
transcendental; everything else in no class: SQ_INSTS_VALU minus the classes).  Within a class
the issue cost is one number for most (FP64, CVT, INT64: 4 cycles), but INT32 holds v_add_u32 /
v_sub_u32 (2 cycles) beside v_mad_u32_u24, v_lshl_add_u32 ... (4), and the unclassified rest holds
v_mov_b32 / v_and_b32 / v_or_b32 / v_lshrrev_b32 (2) beside v_perm_b32, compares, selects, DPP
moves, v_ldexp_f64 ... (4).  No counter tells those apart, so a class's price is the mean over
the kernel's instructions of that class that stand inside loops (static count; which operand a
v_mov_b32 reads decides its cost: 4 cycles from an SGPR, 2 otherwise).  The output carries the
bracket too: every instruction of a mixed class at 2 cycles / at 4.
"""
import collections
import hashlib
import json
import os
import re
import subprocess
import sys
import tempfile

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNELS = [
    ("ndt2d_match_lane.hip", "match_lane_compact_kernel", r"25match_lane_compact_kernelENS"),
    ("ndt2d_match_lane.hip", "match_lane_compact_parts_kernel", r"31match_lane_compact_parts_kernelENS"),
    ("ndt2d_match_small.hip", "match_small_kernel", r"18match_small_kernelILb1ELb1ELb1EEE"),
    ("ndt2d_poses_compact.hip", "score_poses_compact_kernel", r"26score_poses_compact_kernelILi256ELb1ELb1ELb0EEE"),
]


def kernel_lines(path, pattern):
    out, on = [], False
    for line in open(path):
        if not on:
            if re.match(r"^_Z\w*%s\w*:" % pattern, line):
                on = True
            continue
        out.append(line)
        if line.strip().startswith("s_endpgm"):
            break
    return out


def loop_depths(lines):
    label_at = {}
    for i, line in enumerate(lines):
        m = re.match(r"^(\.LBB\d+_\d+):", line)
        if m:
            label_at[m.group(1)] = i
    depth = [0] * len(lines)
    for i, line in enumerate(lines):
        m = re.match(r"^\s+s_c?branch\w*\s+(\.LBB\d+_\d+)", line)
        if m:
            j = label_at.get(m.group(1))
            if j is not None and j <= i:
                for k in range(j, i + 1):
                    depth[k] += 1
    return depth


def instruction_key(line):
    """(name as the ubench table has it, opcode) of a VALU instruction line, or None."""
    m = re.match(r"^\s+(v_[a-z0-9_]+)\s*(.*)$", line)
    if not m:
        return None
    op, operands = m.group(1), m.group(2)
    base = re.sub(r"_(e32|e64)$", "", op)
    if base == "v_mov_b32":
        src = operands.split(",")[-1].strip()
        return ("v_mov_b32 from sgpr" if re.match(r"^(s\d|s\[|vcc|exec|m0)", src) else "v_mov_b32"), op
    if base == "v_mov_b32_dpp":
        return "v_mov_b32_dpp row_shr", op
    if base == "v_cndmask_b32":
        return ("v_cndmask_b32 vcc" if operands.rstrip().endswith("vcc") and not op.endswith("e64") else "v_cndmask_b32 sgpr"), op
    if base.startswith("v_cmp"):
        if "_f64" in base:
            return "v_cmp_lt_f64 vcc", op
        if re.search(r"_[ui]16$", base):
            return "v_cmp_gt_u16 vcc", op
        if re.search(r"_[ui](32|64)$", base):
            return "v_cmp_le_u32 vcc", op
        return "v_cmp_lt_f64 vcc", op    # (f32 / class compares: a compare's cost, counted by no class)
    if base.endswith("_sdwa"):
        return "v_lshrrev_b32_sdwa", op
    return base, op


def counted_by(name, op, classes):
    """The SQ_INSTS_VALU_* class that counts this instruction (measured for the ubench's loops;
    by family for the others)."""
    c = classes.get(name)
    if c is not None:
        return c
    base = re.sub(r"_(e32|e64|sdwa|dpp)$", "", op)
    if re.match(r"v_(add|sub|subrev)_f64$", base):
        return "ADD_F64"
    if re.match(r"v_mul_f64$", base):
        return "MUL_F64"
    if re.match(r"v_(fma|fmac)_f64$", base):
        return "FMA_F64"
    if base.startswith("v_cvt_"):
        return "CVT"
    if re.match(r"v_(rcp|rsq|sqrt|exp|log|sin|cos)_\w*f32", base):
        return "TRANS_F32"
    if re.match(r"v_(rcp|rsq|sqrt)_\w*f64", base):
        return "TRANS_F64"
    if re.match(r"v_(mad_u64_u32|mad_i64_i32|lshl_add_u64|lshlrev_b64|lshrrev_b64|ashrrev_i64)", base):
        return "INT64"
    if re.match(r"v_(add|sub|subrev|addc|subb)_co_u32", base):
        return "INT32"
    if re.match(r"v_cmp_\w+_[ui](32|64)", base):
        return "INT32"
    if re.match(r"v_(add|sub|subrev|mul|mad|min|max|min3|max3|med3|add3|lshl_add|add_lshl|xad|mul_lo|mul_hi|mbcnt|sad|bfe)_?\w*[ui](16|24|32)", base) \
            or re.match(r"v_(mbcnt_lo|mbcnt_hi)_u32_b32", base):
        return "INT32"
    if re.match(r"v_(add|sub|mul|fma|fmac|mac|mad)_f32|v_pk_(add|mul|fma)_f32", base):
        return {"add": "ADD_F32", "sub": "ADD_F32", "mul": "MUL_F32"}.get(re.sub(r"^v_(pk_)?", "", base).split("_")[0], "FMA_F32")
    return "other"


def main():
    issue = json.load(open(os.path.join(R, "profiles", "r05_ubench_issue.json")))
    cycles = issue["cycles"]
    classes = {}
    for name, e in issue["instructions"].items():
        by = [c for c in e.get("counted_by", []) if not c.startswith("(")]
        classes[name] = by[0] if by else "other"
    out = {"what": "static VALU instruction mix inside loops of the hot kernels, per SQ_INSTS_VALU_* class, priced with "
                   "profiles/r05_ubench_issue.json (experiments/asm_loop_mix.py); bench.py prices a kernel's counters with "
                   "`mean_cycles` and reports the bracket `all_two` .. `all_four` for the classes that mix 2- and 4-cycle instructions",
           "kernels": {}}
    h = hashlib.sha256()
    csrc = os.path.join(R, "ndt_2d_amd", "csrc")
    with tempfile.TemporaryDirectory() as tmp:
        asm = {}
        for src, name, pattern in KERNELS:
            if src not in asm:
                path = os.path.join(tmp, src + ".s")
                subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off",
                                       "-I", os.path.join(R, "include"), "-I", csrc, "--cuda-device-only", "-S",
                                       os.path.join(csrc, src), "-o", path], stderr=subprocess.DEVNULL)
                asm[src] = path
            lines = kernel_lines(asm[src], pattern)
            if not lines:
                raise SystemExit("kernel %s not found" % name)
            depth = loop_depths(lines)
            per_class = collections.defaultdict(lambda: {"n": 0, "cycles": 0.0, "two_cycle": 0, "opcodes": collections.Counter()})
            unmeasured = collections.Counter()
            for line, d in zip(lines, depth):
                if d < 1:
                    continue
                key = instruction_key(line)
                if key is None:
                    continue
                name_k, op = key
                cyc = cycles.get(name_k)
                if cyc is None:
                    unmeasured[op] += 1
                    cyc = 4.0
                c = per_class[counted_by(name_k, op, classes)]
                c["n"] += 1
                c["cycles"] += cyc
                c["two_cycle"] += 1 if cyc == 2.0 else 0
                c["opcodes"][op] += 1
            rec = {}
            for cname, c in sorted(per_class.items()):
                rec[cname] = {"static_in_loops": c["n"], "two_cycle": c["two_cycle"],
                              "mean_cycles": round(c["cycles"] / c["n"], 4),
                              "opcodes": dict(c["opcodes"].most_common(12))}
            out["kernels"][name] = {"classes": rec, "unmeasured_opcodes_priced_at_4": dict(unmeasured)}
    for path in sorted(os.listdir(csrc)):
        if path.endswith((".hip", ".h")):
            h.update(path.encode() + b"\0")
            h.update(open(os.path.join(csrc, path), "rb").read())
    out["source_sha256"] = h.hexdigest()
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
