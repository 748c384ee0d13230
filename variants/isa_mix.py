"""Static VALU mix of the product kernels priced with the per-class issue costs variants/micro/issue_mix.hip measured
(profiles/r4_micro_issue_mix.txt): what a wave64 vector instruction of each kernel costs its SIMD on average.

    hipcc <Makefile's HIPFLAGS> -S --cuda-device-only -o /tmp/asurf_api.s alphasurf_amd/csrc/asurf_api.hip
    python3 variants/isa_mix.py /tmp/asurf_api.s k_sweep_wave k_geom k_comp ...

The mix is the kernel's TEXT, not its execution (loops weigh as much as prologues), so the figure is an estimate: it
says whether "VALU issue at 2 cycles an instruction" (the fma rate bench.py prices with) is the right ceiling for a
kernel or an optimistic one."""
import collections, json, re, sys

# cycles a wave64 instruction occupies its SIMD, four waves a SIMD, 2.4 GHz nominal (profiles/r4_micro_issue_mix.txt)
COST = {"full": 2.8, "half": 4.4, "f64": 4.9, "trans": 8.4}
# full rate (measured: fma / add / mul f32, add / sub u32, and_b32, mov; assumed alike: the other two-operand adds,
# subtractions and bit-wise logic).  Measured HALF rate and therefore not here: v_min / v_max (f32 and i32), every
# three-operand integer form (v_lshl_add_u32, v_or3_b32, v_mad_u32_u24, ...), v_bcnt, v_readlane.
FULL = re.compile(r"v_(fma|fmac|mac|mad|add|sub|subrev|mul)_f32|v_(add|sub|subrev)_(u32|i32|co_u32)|v_(addc|subb|subbrev)_co_u32|"
                  r"v_(and|or|xor|not)_b32|v_mov_b32|v_fma_mix")
TRANS = re.compile(r"v_(rcp|rsq|sqrt|exp|log|sin|cos)_(f32|f16|f64)|v_rcp_iflag")
F64 = re.compile(r"v_\w+_f64")


def classify(op):
    if TRANS.match(op):
        return "trans"
    if F64.match(op):
        return "f64"
    if FULL.match(op):
        return "full"
    return "half"   # shifts, bit fields, compares, selects, integer multiplies, conversions, 64-bit shifts, packed, readlane...


def main():
    path, names = sys.argv[1], sys.argv[2:]
    cur, out = None, collections.OrderedDict()
    for line in open(path):
        m = re.match(r"^(_ZN5asurf\w+):", line)
        if m:
            cur = m.group(1)
            out[cur] = collections.Counter()
            continue
        t = line.strip()
        if cur is None or not t or t[0] in ";.":
            if t.startswith(".Lfunc_end"):
                cur = None
            continue
        op = t.split()[0]
        if op.startswith("v_"):
            out[cur][classify(op)] += 1
    res = {}
    for k, c in out.items():
        if not any(n in k for n in names):
            continue
        n = sum(c.values())
        if n < 200:
            continue
        cyc = sum(COST[cl] * v for cl, v in c.items()) / n
        res[k[:70]] = dict(valu=n, **{cl: round(v / n, 3) for cl, v in c.items()}, cycles_per_instruction=round(cyc, 2))
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
