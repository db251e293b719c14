"""Diagnostic only: rewrites psk_fast_loop.h IN PLACE so that one section of the block loop is
skipped (results become wrong on purpose), to attribute instruction counts and time to sections.
Run it on a throw-away copy of the tree (the GPU box snapshot), never commit its output.

usage: python tools/ablate.py <section> [<section> ...]   sections: atan sincos fit scan argmax out rot pow
"""
import sys

path = "psk_soft_amd/csrc/psk_fast_loop.h"
s = open(path).read()


def swap(old, new):
    global s
    assert s.count(old) == 1, (old, s.count(old))
    s = s.replace(old, new)


for sec in sys.argv[1:]:
    if sec == "atan":
        swap("rawd[r] = (double)atan2f_wave(pw.im, pw.re, atab);", "rawd[r] = (double)(pw.im + pw.re);")
    elif sec == "sincos":
        swap("sincosf_wave(phaseCorrection, &sn, &cs, c);", "sn = phaseCorrection; cs = 1.0f - phaseCorrection;")
    elif sec == "pow":
        swap("cf32 pw = cpow_uint<false>(s[r], M);", "cf32 pw = s[r];")
    elif sec == "fit":
        swap(
            "        if (__builtin_expect(q0 >= n, 1)) {\n            pass = fit_block<false>(",
            "        if (true) { pass = 0;\n#pragma unroll\n for (int r = 0; r < kR; r++) { y[r] = (float)rawd[r]; est[r] = y[r]; "
            "ySum_l[r] = rawd[r]; xySum_l[r] = rawd[r]; }\n } else if (__builtin_expect(q0 >= n, 1)) {\n"
            "            pass = fit_block<false>(",
        )
    elif sec == "scan":
        swap("            wave_scan_f32_multi<S>(inc);\n", "")
    elif sec == "argmax":
        swap("                    m2[0] = med3_i32(m1[0], m2[0], p0);\n                    m2[1] = med3_i32(m1[1], m2[1], p1);\n", "")
    elif sec == "out":
        swap("        if (valid[1]) {\n            if (p.soft) {", "        if (false) {\n            if (p.soft) {")
    elif sec == "rot":
        swap("    return bperm_addr(p.src_addr[r], offered);", "    return offered;")
    else:
        raise SystemExit("unknown section " + sec)
open(path, "w").write(s)
print("ablated:", sys.argv[1:])
