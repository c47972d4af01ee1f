#!/usr/bin/env python3
"""Mix-weighted VALU pipe time of the render kernels: prices the dynamic instruction
classes of a launch with the per-class SIMD cycles tools/valu_microbench.hip measured on the MI355X.

  python tools/valu_ceiling.py profiles/r03_valu_mix.json c3=profiles/r03_c3_mix_unpacked.txt ...

Input: tools/pmc_summary.py summaries of the per-class SQ counters (tools/r03_mix.sh) taken on the UNPACKED
build of the library (since the round's end the product itself; before: tools/build_variant.sh unpacked): the
arithmetic of the ray pairs with every v_pk_*_f32 written as two plain instructions.  The counters put a packed op in the same class as a plain
one, so only that build's counts can be priced; a packed op occupies the SIMD like its two plain halves
(microbench: 4 cycles against 2 + 2) and the two builds run at the same speed (profiles/r03_ab_unpacked.txt),
so the pipe time is the product's too.

Prices (SIMD cycles per wave64 instruction at >= 3 waves per SIMD, profiles/r03_valu_microbench.txt):
  2  v_fma/add/mul/sub_f32, v_mov, v_add_u32, v_and/or/xor, shifts        ("simple")
  4  v_max/min/max3/min3/med3_f32, v_cmp_*, v_cndmask, v_readlane, v_cvt, v_mul_lo/hi_u32, v_bfe, v_or3,
     v_pk_*_f32, every f64 op                                              ("complex")
  8  v_rcp/sqrt/rsq_f32                                                    ("transcendental")
What the counters cannot split is priced both ways: `lo` takes the integer / logic class (SQ_INSTS_VALU_IOPS)
at 2 cycles and a fifth of the unclassified rest (v_mov's static share in the kernels) at 2, `hi` takes both at 4.
Simple instructions of one wave can also issue BESIDE a complex one of another wave (microbench: 1 v_fma : 1 v_max
costs 4.2 cycles a pair) -- but not once packed ops are in the stream, and the unpacked build is no faster, so
no credit is taken for that."""
import json
import sys

SIMPLE, COMPLEX, TRANS = 2.0, 4.0, 8.0


def parse(path):
    out, section = {}, None
    for line in open(path):
        f = line.split()
        if not f:
            continue
        if f[0] == "==":
            section = f[1]
            out[section] = {}
        elif section and "launch" in f:
            out[section][f[0]] = float(f[f.index("launch") + 1]) if f[-2] == "launch" else float(f[-1])
    return out


def main():
    dst = sys.argv[1]
    res = {"prices_cycles": {"simple": SIMPLE, "complex": COMPLEX, "transcendental": TRANS},
           "source": "tools/valu_microbench.hip (profiles/r03_valu_microbench.txt) x tools/r03_mix.sh counters of the unpacked build"}
    for arg in sys.argv[2:]:
        cfg, path = arg.split("=")
        res[cfg] = {}
        for kernel, c in parse(path).items():
            valu = c["SQ_INSTS_VALU"]
            fsimple = c["SQ_INSTS_VALU_FMA_F32"] + c["SQ_INSTS_VALU_MUL_F32"] + c["SQ_INSTS_VALU_ADD_F32"]
            iops = max(c["SQ_INSTS_VALU_IOPS"], c["SQ_INSTS_VALU_INT32"])
            trans = c["SQ_INSTS_VALU_TRANS_F32"]
            f64 = c.get("SQ_INSTS_VALU_ADD_F64", 0) + c.get("SQ_INSTS_VALU_MUL_F64", 0) + c.get("SQ_INSTS_VALU_FMA_F64", 0)
            cvt = c["SQ_INSTS_VALU_CVT"]
            rest = valu - fsimple - iops - trans - f64 - cvt
            fixed = SIMPLE * fsimple + TRANS * trans + COMPLEX * (f64 + cvt)
            lo = fixed + SIMPLE * iops + (0.2 * SIMPLE + 0.8 * COMPLEX) * rest
            hi = fixed + COMPLEX * iops + COMPLEX * rest
            res[cfg][kernel] = {
                "valu_insts_unpacked": valu, "float_simple": fsimple, "int_logic": iops, "transcendental": trans,
                "f64": f64, "cvt": cvt, "unclassified": rest, "salu": c.get("SQ_INSTS_SALU"), "smem": c.get("SQ_INSTS_SMEM"),
                "lds": c.get("SQ_INSTS_LDS"), "branch": c.get("SQ_INSTS_BRANCH"),
                "pipe_cycles_lo": lo, "pipe_cycles_hi": hi}
            print("%s %-8s VALU %.3g  simple f32 %.3g  int/logic %.3g  trans %.3g  rest %.3g  -> pipe cycles %.4g .. %.4g (%.2f .. %.2f per instruction)"
                  % (cfg, kernel, valu, fsimple, iops, trans, rest, lo, hi, lo / valu, hi / valu))
    json.dump(res, open(dst, "w"), indent=1)


if __name__ == "__main__":
    main()
