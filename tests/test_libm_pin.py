"""The product's libm restatements (psk_soft_amd/csrc/psk_libm.h: atan2f / atanf / sinf / cosf of
glibc 2.35, and the known-divisor division) compiled for the HOST and compared bit-for-bit
with this machine's glibc -- the libm the oracle (like the reference) calls.  If this passes,
the device's transcendental results equal the oracle's whenever their arguments are equal."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_psk_libm_matches_glibc(tmp_path):
    exe = str(tmp_path / "libm_pin")
    subprocess.run(
        ["g++", "-O2", "-std=gnu++17", "-ffp-contract=off", "-mfma", "-I" + os.path.join(ROOT, "psk_soft_amd", "csrc"),
         "-o", exe, os.path.join(ROOT, "tests", "support", "libm_pin.cpp"), "-lm"],
        check=True,
    )
    r = subprocess.run([exe, "12000000"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "sinf_bad=0 cosf_bad=0 atan2f_bad=0 atanf_bad=0 div_bad=0" in r.stdout
