#!/usr/bin/env python3
"""Opcode-class histogram of one kernel from hipcc --save-temps assembly (round-3 verdict, next #2).
usage: isa_hist.py file.s kernel_name_substring [--loop]   (--loop: only the largest basic-block loop body, by label span)"""
import collections
import re
import sys


def kernel_lines(path, name):
    out, on = [], False
    for ln in open(path):
        if re.match(r"^_Z\w*%s\w*:" % re.escape(name), ln):
            on = True
            continue
        if on and ln.startswith("\t.section") or (on and ".end_amdhsa_kernel" in ln):
            break
        if on:
            out.append(ln.rstrip("\n"))
    return out


CLASSES = [
    ("v_mov / v_accvgpr", r"^v_(mov_b32|mov_b64|accvgpr)"),
    ("v_readlane / v_writelane / readfirstlane", r"^v_(readlane|writelane|readfirstlane)"),
    ("v_permlane / dpp moves", r"^v_(permlane|mov_b32_dpp)"),
    ("v_cndmask", r"^v_cndmask"),
    ("v_cmp*", r"^v_cmp"),
    ("fp32 fma/mul/add (scalar)", r"^v_(fma_f32|fmac_f32|mul_f32|add_f32|sub_f32|subrev_f32|mul_legacy|fmaak|fmamk|max_f32|min_f32|med3_f32|max3_f32|min3_f32)"),
    ("fp32 packed (v_pk_*)", r"^v_pk_"),
    ("division / sqrt helpers (div_scale, div_fmas, div_fixup, rcp, rsq, sqrt)", r"^v_(div_scale|div_fmas|div_fixup|rcp|rsq|sqrt)"),
    ("other transcendental / convert / round (cvt, ceil, floor, trunc, rndne, exp, log, ldexp, frexp)", r"^v_(cvt|ceil|floor|trunc|rndne|exp|log|ldexp|frexp)"),
    ("int add/sub/mul/mad", r"^v_(add_u32|sub_u32|subrev_u32|add_co|addc_co|sub_co|subb_co|add3_u32|mul_lo|mul_hi|mul_u32|mad_u32|mad_i32|mad_u64|mad_i64|lshl_add_u64|add_lshl|lshl_add_u32|sad_|add_nc|mul_i32|mad_i64_i32)"),
    ("int shift/logic/bfe/minmax", r"^v_(lshlrev|lshrrev|ashrrev|and_|or_|xor_|or3|and_or|lshl_or|bfe|bfi|not_|min_i32|max_i32|min_u32|max_u32|med3_i32|min3|max3|perm_b32|alignbit|bcnt|mbcnt|ffb|lshl_b64|lshr_b64)"),
    ("global/flat/scratch memory", r"^(global_|flat_|scratch_|buffer_)"),
    ("LDS (ds_*)", r"^ds_"),
    ("s_load / s_buffer_load", r"^s_(load|buffer_load)"),
    ("s_waitcnt / s_nop / s_barrier", r"^s_(waitcnt|nop|barrier|sleep)"),
    ("s_branch / s_cbranch", r"^s_(branch|cbranch)"),
    ("other SALU", r"^s_"),
    ("other VALU", r"^v_"),
]


def main():
    path, name = sys.argv[1], sys.argv[2]
    lines = kernel_lines(path, name)
    if "--range" in sys.argv:     # labels: from .LBBx_y to .LBBx_z
        a, b = sys.argv[sys.argv.index("--range") + 1].split(":")
        ia = next(i for i, l in enumerate(lines) if l.startswith(a + ":"))
        ib = next(i for i, l in enumerate(lines) if l.startswith(b + ":"))
        lines = lines[ia:ib]
    hist = collections.Counter()
    ops = collections.Counter()
    n = 0
    for ln in lines:
        s = ln.strip()
        if not s or s.startswith((";", ".", "//")) or s.endswith(":"):
            continue
        op = s.split()[0]
        if not re.match(r"^[a-z]", op):
            continue
        n += 1
        ops[op] += 1
        for cname, pat in CLASSES:
            if re.match(pat, op):
                hist[cname] += 1
                break
        else:
            hist["unclassified"] += 1
    valu = sum(c for k, c in ops.items() if k.startswith("v_"))
    print("| class | instructions |\n|---|---:|")
    for cname, _ in CLASSES + [("unclassified", "")]:
        if hist[cname]:
            print("| %s | %d |" % (cname, hist[cname]))
    print("| **total** | **%d** (vector ALU: %d) |" % (n, valu))
    if "--top" in sys.argv:
        print()
        for op, c in ops.most_common(40):
            print("%-28s %d" % (op, c))
    for ln in lines:
        if re.search(r"(sgpr_spill_count|vgpr_spill_count|NumVgprs|NumSgprs|ScratchSize|Occupancy|LDSByteSize|vgpr_count|sgpr_count|SGPRSpill|VGPRSpill)", ln, re.I):
            pass


if __name__ == "__main__":
    main()
