#!/usr/bin/env python3
"""Static VALU opcode mix of one kernel, by loop depth, from the compiler's assembly.

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Iinclude -Indt_2d_amd/csrc \\
          --cuda-device-only -S ndt_2d_amd/csrc/ndt2d_match_lane.hip -o /tmp/lane.s
    python3 experiments/asm_loop_mix.py /tmp/lane.s match_lane_compact_kernel [--issue profiles/r05_ubench_issue.json]

Loops are found from the labels and the backward branches; an instruction's depth is the number
of loops it lies in.  With --issue, every opcode is priced with the measured issue cycles of
profiles/r05_ubench_issue.json and the table prints, per PMC class (how rocprofv3's
SQ_INSTS_VALU_* counters split the VALU instructions), the static share of 2-cycle opcodes
inside loops -- the weight bench.py uses to price a class whose members issue at different rates.
"""
import collections
import json
import re
import sys


def kernel_lines(path, name):
    out, on = [], False
    for line in open(path):
        if not on:
            if re.match(r"^_Z\w*%s\w*:" % re.escape(name), line):
                on = True
            continue
        if line.strip().startswith("s_endpgm"):
            out.append(line)
            break
        out.append(line)
    return out


def loop_depths(lines):
    """depth[i] for every line: number of (label .. backward branch to it) ranges holding it."""
    label_at = {}
    for i, line in enumerate(lines):
        m = re.match(r"^(\.LBB\d+_\d+):", line)
        if m:
            label_at[m.group(1)] = i
    depth = [0] * len(lines)
    for i, line in enumerate(lines):
        m = re.match(r"^\s+s_cbranch_\w+\s+(\.LBB\d+_\d+)|^\s+s_branch\s+(\.LBB\d+_\d+)", line)
        if m:
            target = m.group(1) or m.group(2)
            j = label_at.get(target)
            if j is not None and j <= i:
                for k in range(j, i + 1):
                    depth[k] += 1
    return depth


# how SQ_INSTS_VALU_* classify an opcode (checked against the counters of experiments/ubench_issue.hip's loops)
def pmc_class(op):
    base = re.sub(r"_(e32|e64|sdwa|dpp)$", "", op)
    if re.match(r"v_(add|sub|subrev)_f64|v_fma_f64|v_fmac_f64|v_mul_f64", base):
        return "FP64"
    if re.match(r"v_(add|sub|subrev|mul|fma|fmac|mac|mad)_f32|v_pk_(fma|mul|add)_f32", base):
        return "FP32"
    if re.match(r"v_cvt_", base):
        return "CVT"
    if re.match(r"v_(rcp|rsq|sqrt|exp|log|sin|cos)_", base):
        return "TRANS"
    if re.match(r"v_(mad_u64_u32|mad_i64_i32|lshl_add_u64|lshlrev_b64|lshrrev_b64|ashrrev_i64|add_co_u32|addc_co_u32|sub_co_u32|subb_co_u32)", base):
        return "INT64"
    if re.match(r"v_(add|sub|subrev|mul|mad|min|max|min3|max3|med3|and|or|xor|not|lshl|lshr|ashr|bfe|bfi|add3|and_or|or3|xad|lshl_add|add_lshl|lshl_or|mul_lo|mul_hi|mbcnt|sad|alignbit|alignbyte|perm|cmp_\w+_[ui](16|32|64))", base) \
            and not re.search(r"_f(16|32|64)$", base):
        return "INT32"
    return "other"


def main():
    args = sys.argv[1:]
    issue = None
    if "--issue" in args:
        at = args.index("--issue")
        issue = json.load(open(args[at + 1]))["cycles"]
        del args[at:at + 2]
    path, name = args[0], args[1]
    lines = kernel_lines(path, name)
    if not lines:
        raise SystemExit("kernel %s not found in %s" % (name, path))
    depth = loop_depths(lines)
    by_depth = collections.defaultdict(collections.Counter)
    for line, d in zip(lines, depth):
        m = re.match(r"^\s+(v_[a-z0-9_]+)", line)
        if m:
            by_depth[min(d, 3)][m.group(1)] += 1
    total = collections.Counter()
    for d in sorted(by_depth):
        n = sum(by_depth[d].values())
        print("depth %d%s: %d VALU instructions" % (d, "+" if d == 3 else "", n))
        if d >= 1:
            total.update(by_depth[d])
    print("\nVALU opcodes inside loops (depth >= 1), static count%s:" % (", measured issue cycles" if issue else ""))
    classes = collections.defaultdict(lambda: [0, 0.0, 0])
    for op, n in total.most_common():
        cyc = None
        if issue:
            base = re.sub(r"_(e32|e64)$", "", op)
            cyc = issue.get(op, issue.get(base))
        c = pmc_class(op)
        classes[c][0] += n
        if cyc is not None:
            classes[c][1] += n * cyc
            classes[c][2] += n
        print("  %-26s %5d  %-6s %s" % (op, n, c, "" if cyc is None else "%.0f" % cyc))
    print("\nper PMC class: static instructions in loops, mean measured cycles (priced share)")
    for c, (n, cyc, priced) in sorted(classes.items()):
        print("  %-6s %5d  %s" % (c, n, "-" if priced == 0 else "%.2f (%d of %d priced)" % (cyc / priced, priced, n)))


if __name__ == "__main__":
    main()
