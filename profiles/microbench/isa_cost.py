#!/usr/bin/env python3
"""Issue-cost model of an ISA range, priced with the rates profiles/microbench/valu_rates measured on MI355X
(clk per wave-instruction per SIMD, 2.4 GHz nominal).  Usage: isa_cost.py file.s first_line last_line"""
import re, sys, collections
FAST = 2.7; HALF = 4.4; PK = 4.9; TRANS = 8.7
HALF_OPS = ("v_med3", "v_cvt", "v_floor", "v_fract", "v_rndne", "v_lshl_add", "v_add_lshl", "v_add3", "v_min", "v_max",
            "v_mul_u32_u24", "v_mad_u32_u24", "v_mul_lo", "v_mul_hi", "v_cmp", "v_cndmask", "v_mad_i64", "v_mad_u64",
            "v_lshl_or", "v_and_or", "v_or3", "v_bfe", "v_perm", "v_readfirstlane", "v_lshlrev_b64", "v_div_", "v_ldexp", "v_trunc", "v_ceil")
def cost(op, line):
    if op.startswith(("v_exp", "v_rcp", "v_log", "v_sqrt", "v_rsq", "v_sin", "v_cos")): return "trans", TRANS
    if op.startswith("v_pk_"): return "packed", PK
    if op.startswith(HALF_OPS): return "half", HALF
    if op.startswith("v_"):
        # an SGPR source makes v_fma half rate (measured); assume the same for the other fast ops
        if re.search(r"[ ,\-|]s\d+|s\[\d+:\d+\]", line.split(op, 1)[1]): return "fast+sgpr", HALF
        return "fast", FAST
    return None, 0.0
def main():
    f, a, b = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    tot = collections.Counter(); n = collections.Counter()
    for ln in open(f).read().splitlines()[a - 1:b]:
        m = re.match(r"^\t(\S+)", ln)
        if not m: continue
        k, c = cost(m.group(1).replace("_e32", "").replace("_e64", ""), ln)
        if k: tot[k] += c; n[k] += 1
    s = sum(tot.values())
    for k in tot: print(f"  {k:10s} n={n[k]:4d}  clk={tot[k]:7.1f}  {100*tot[k]/s:5.1f}%")
    print(f"  total VALU n={sum(n.values())} clk={s:.0f}")
main()
