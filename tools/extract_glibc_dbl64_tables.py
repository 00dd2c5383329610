#!/usr/bin/env python3
"""The reference's unqualified exp() / pow() / atan2() calls are the host glibc's double functions:
since 2.28 the table-driven algorithms of sysdeps/ieee754/dbl-64/e_exp.c and e_pow.c (ARM optimized-routines;
error ~0.51 ulp, not correctly rounded).  This script reads their two constant tables out of libm.so.6
(__exp_data, __pow_log_data: hidden symbols, located by their leading entries) and writes them, numeric data
only, as dj_brdf_amd/csrc/djb_glibc_dbl64_tables.hpp (the one copy in the tree; oracle/Makefile derives the checker's C header from it).
atan2 is the IBM Accurate Mathematical Library routine (sysdeps/ieee754/dbl-64/e_atan2.c; since 2.34 without its
multi-precision fall-back): its 241 x 7 table cij (uatan.tbl) is read the same way, located by its first row; sin / cos
(s_sin.c) read the 440-entry __sincostab, tan (s_tan.c) the 186 x 4 xfg, acos (e_asin.c) asincos.tbl and root.tbl.
The operation order of the restatements (which multiply-adds are fused) was read off the disassembly of the
x86-64 FMA ifunc variants (__exp_fma, __pow_fma, __ieee754_atan2_fma, __sin_fma, __cos_fma, __tan_fma, __ieee754_acos_fma) of this image's glibc 2.35;
tests/test_oracle_golden.py::test_glibc_double_libm_restatement pins them against the host libm bit for bit."""
import os
import struct
import sys

LIBM = sys.argv[1] if len(sys.argv) > 1 else "/lib/x86_64-linux-gnu/libm.so.6"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
data = open(LIBM, "rb").read()


def find_all(b):
    out, i = [], data.find(b)
    while i >= 0:
        out.append(i); i = data.find(b, i + 1)
    return out


def dbl(off, n):
    return list(struct.unpack("<%dd" % n, data[off:off + 8 * n]))


def u64(off, n):
    return list(struct.unpack("<%dQ" % n, data[off:off + 8 * n]))


# struct exp_data { invln2N, shift, negln2hiN, negln2loN, poly[4], exp2_shift, exp2_poly[5], uint64 tab[2*128] }
exp_off = None
for off in find_all(struct.pack("<d", float.fromhex("0x1.71547652b82fep0") * 128)):
    if dbl(off + 8, 1)[0] == float.fromhex("0x1.8p52") and u64(off + 112, 2) == [0, 0x3ff0000000000000]:
        exp_off = off
# struct pow_log_data { ln2hi, ln2lo, poly[7], struct { invc, pad, logc, logctail } tab[128] }
pow_off = None
for off in find_all(struct.pack("<d", float.fromhex("0x1.62e42fefa3800p-1"))):
    if dbl(off + 16, 1)[0] == -0.5 and dbl(off + 72, 1)[0] == float.fromhex("0x1.6ap+0"):
        pow_off = off
assert exp_off and pow_off, (exp_off, pow_off)

# const number cij[241][7] (uatan.tbl): {x_i, atan(x_i), and the Taylor coefficients of atan about x_i}; row 0 = x 1/16
atan_off = None
for off in find_all(struct.pack("<2d", float.fromhex("0x1.0400665e0244ep-4"), float.fromhex("0x1.03a737b53dd20p-4"))):
    if abs(dbl(off + 56 * 240, 1)[0] - 0.9990241799010189) < 1e-15:
        atan_off = off
assert atan_off, "cij not found"
ATAN_CIJ = dbl(atan_off, 241 * 7)
# the polynomial and pi constants of e_atan2.c / atnat2.h, checked against the bytes of this libm
ATAN2_C = {"d3": "-0x1.5555555555555p-2", "d5": "0x1.99999999997fdp-3", "d7": "-0x1.24924923f7603p-3", "d9": "0x1.c71c6e5129a3bp-4",
           "d11": "-0x1.7458022b13c25p-4", "d13": "0x1.375f08b31cbcep-4", "hpi": "0x1.921fb54442d18p+0", "hpi1": "0x1.1a62633145c07p-54",
           "opi": "0x1.921fb54442d18p+1", "opi1": "0x1.1a62633145c07p-53", "qpi": "0x1.921fb54442d18p-1", "tqpi": "0x1.2d97c7f3321d2p+1"}
for name, hx in ATAN2_C.items():
    assert find_all(struct.pack("<d", float.fromhex(hx))), name

# __sincostab (sincostab.c, s_sin.c): 110 x {sin(x_k) hi, lo, cos(x_k) hi, lo}, x_k = k / 128
sincos_off = None
for off in find_all(struct.pack("<5d", 0.0, 0.0, 1.0, 0.0, float.fromhex("0x1.fffeaaaaeeeefp-8"))):
    if abs(dbl(off + 8 * 436, 1)[0] - 0.7523107112959804) < 1e-15:
        sincos_off = off
assert sincos_off, "__sincostab not found"
SINCOS_TAB = dbl(sincos_off, 440)
SINCOS_C = {"sn3": "-0x1.5555555555515p-3", "sn5": "0x1.11110e829872fp-7", "cs2": "0x1.0000000000000p-1", "cs4": "-0x1.5555555555535p-5",
            "cs6": "0x1.6c16bedd9e239p-10", "s1": "-0x1.5555555555555p-3", "s2": "0x1.1111111110ecep-7", "s3": "-0x1.a01a019db08b8p-13",
            "s4": "0x1.71de27b9a7ed9p-19", "s5": "-0x1.addffc2fcdf59p-26", "hpinv": "0x1.45f306dc9c883p-1", "mp1": "0x1.921fb58000000p+0",
            "mp2": "-0x1.dde973c000000p-27", "pp3": "-0x1.cb3b398000000p-55", "pp4": "-0x1.d747f23e32ed7p-83"}
for name, hx in SINCOS_C.items():
    assert find_all(struct.pack("<d", float.fromhex(hx))), name

# xfg[186][4] of s_tan.c (utan.tbl): {x_i, tan(x_i), cot(x_i), cot lo}, x_i ~ (i + 16) / 256
tan_off = None
for off in find_all(struct.pack("<2d", float.fromhex("0x1.000001e519d60p-4"), float.fromhex("0x1.0055796c4e240p-4"))):
    if dbl(off + 32 * 186, 1)[0] == -15.5:
        tan_off = off
assert tan_off, "xfg not found"
TAN_XFG = dbl(tan_off, 186 * 4)
TAN_C = {"d3": "0x1.5555555555555p-2", "d5": "0x1.11111111107c6p-3", "d7": "0x1.ba1ba1cdb8745p-5", "d9": "0x1.664ed49cfc666p-6",
         "d11": "0x1.2385a3cf2e4eap-7", "e0": "0x1.5555555554dbdp-2", "e1": "0x1.11112e0a6b45fp-3", "mp3": "-0x1.cb3b399d747f2p-55",
         "g2": "0x1.f212d00000000p-5", "g3": "0x1.92f1a00000000p-1"}
for name, hx in TAN_C.items():
    assert find_all(struct.pack("<d", float.fromhex(hx))), name

# asncs[2568] (asincos.tbl) and inroot[128] (root.tbl) of e_asin.c; powtwo[k] = 2^k is generated, not read
acos_off = None
for off in find_all(struct.pack("<2d", float.fromhex("0x1.0400000000000p-3"), float.fromhex("0x1.0216988994424p+0"))):
    if abs(dbl(off + 8 * 2567, 1)[0] - 0.006938468016094754) < 1e-17:
        acos_off = off
inroot_off = None
for off in find_all(struct.pack("<2d", float.fromhex("0x1.68a1f80d71820p+0"), float.fromhex("0x1.65de82af9631fp+0"))):
    if abs(dbl(off + 8 * 127, 1)[0] - 0.70849190843208) < 1e-13:
        inroot_off = off
assert acos_off and inroot_off, "asincos.tbl / root.tbl not found"
ASNCS = dbl(acos_off, 2568)
INROOT = dbl(inroot_off, 128)
ACOS_C = {"f1": "0x1.55555555554f9p-3", "f2": "0x1.333333336127dp-4", "f3": "0x1.6db6dae42c0e4p-5", "f4": "0x1.f1c7e04f4ad99p-6",
          "f5": "0x1.6e442c822d419p-6", "f6": "0x1.292d80f453c72p-6", "rt0": "0x1.fffffffecc1ddp-1", "rt1": "0x1.fffffff757304p-2",
          "rt2": "0x1.800496769c91ap-2", "rt3": "0x1.4006318d1dab9p-2"}
for name, hx in ACOS_C.items():
    assert find_all(struct.pack("<d", float.fromhex(hx))), name

EXP_C = dbl(exp_off, 8)                       # invln2N, shift, negln2hiN, negln2loN, C2, C3, C4, C5
EXP_TAB = u64(exp_off + 112, 256)             # {tail bits, scale bits} x 128
POW_C = dbl(pow_off, 9)                       # ln2hi, ln2lo, A[0..6]
raw = dbl(pow_off + 72, 512)
POW_TAB = [v for i in range(128) for v in (raw[4 * i], raw[4 * i + 2], raw[4 * i + 3])]   # invc, logc, logctail


def arr(vals, fmt=float.hex, per=4):
    return ",\n\t".join(", ".join(fmt(v) for v in vals[k:k + per]) for k in range(0, len(vals), per))


def emit(path, device):
    q = "__device__ const" if device else "static const"
    cq = "constexpr double" if device else "static const double"
    c0, c1 = ("// ", "") if device else ("/* ", " */")
    with open(path, "w") as f:
        for line in ("GENERATED by tools/extract_glibc_dbl64_tables.py from the host's libm.so.6 (GLIBC 2.35):",
                     "the constant tables of glibc's double exp / pow / atan2 / sin / cos / tan / acos (sysdeps/ieee754/dbl-64/e_exp.c, e_pow.c, e_atan2.c, s_sin.c, s_tan.c, e_asin.c).",
                     "Numeric data only; the algorithms are restated in djb_device.hpp / djb_oracle.c."):
            f.write(c0 + line + c1 + "\n")
        if device:
            f.write("#pragma once\n")
        f.write("\n%s__exp_data.tab: 128 x {tail bits, scale bits}%s\n%s unsigned long long DJB_GLIBC_EXP_TAB[256] = {\n\t" % (c0, c1, q)
                + arr(EXP_TAB, lambda v: "0x%016xull" % v) + "\n};\n")
        f.write("%sinvln2N, shift, negln2hiN, negln2loN, C2, C3, C4, C5%s\n%s DJB_GLIBC_EXP_C[8] = { %s };\n\n"
                % (c0, c1, cq, ", ".join(float.hex(v) for v in EXP_C)))
        f.write("%s__pow_log_data.tab: 128 x {invc, logc, logctail}%s\n%s double DJB_GLIBC_POW_LOG_TAB[384] = {\n\t" % (c0, c1, q)
                + arr(POW_TAB, per=3) + "\n};\n")
        f.write("%sln2hi, ln2lo, A[0..6]%s\n%s DJB_GLIBC_POW_C[9] = { %s };\n\n"
                % (c0, c1, cq, ", ".join(float.hex(v) for v in POW_C)))
        f.write("%scij[241][7] of e_atan2.c (uatan.tbl): x_i, atan(x_i), Taylor coefficients of atan about x_i%s\n%s double DJB_GLIBC_ATAN_CIJ[241 * 7] = {\n\t" % (c0, c1, q)
                + arr(ATAN_CIJ, per=7) + "\n};\n\n")
        f.write("%s__sincostab of s_sin.c: 110 x {sin hi, sin lo, cos hi, cos lo} at k / 128%s\n%s double DJB_GLIBC_SINCOS_TAB[440] = {\n\t" % (c0, c1, q)
                + arr(SINCOS_TAB, per=4) + "\n};\n\n")
        f.write("%sxfg[186][4] of s_tan.c (utan.tbl): x_i, tan(x_i), cot(x_i) hi, lo%s\n%s double DJB_GLIBC_TAN_XFG[186 * 4] = {\n\t" % (c0, c1, q)
                + arr(TAN_XFG, per=4) + "\n};\n\n")
        f.write("%sasncs[2568] of e_asin.c (asincos.tbl): per interval rows {x_i, Taylor coefficients of asin / acos about x_i, values at x_i}%s\n%s double DJB_GLIBC_ASNCS[2568] = {\n\t" % (c0, c1, q)
                + arr(ASNCS, per=6) + "\n};\n\n")
        f.write("%sinroot[128] of e_asin.c (root.tbl): 1 / sqrt seeds%s\n%s double DJB_GLIBC_INROOT[128] = {\n\t" % (c0, c1, q)
                + arr(INROOT, per=4) + "\n};\n")


emit(os.path.join(ROOT, "dj_brdf_amd", "csrc", "djb_glibc_dbl64_tables.hpp"), True)
print("wrote tables: __exp_data@%d __pow_log_data@%d cij@%d" % (exp_off, pow_off, atan_off))
