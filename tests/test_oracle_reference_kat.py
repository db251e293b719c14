"""Pins the CPU oracle against everything the reference itself offers for this path.

1. The six known-answer tests of the reference's component test
   (reference tests/test_psk_soft.py:160-238): 1000 symbols, 8 samples/baud,
   numAvg 100, M in {2,4,8}, differential on/off, sampleRate 100; soft symbols
   must be within 1e-3 of the transmitted ones.  The stimulus is regenerated with
   the reference's own seed and Python-2 ``random.choice`` semantics, in the
   order unittest runs the six tests (alphabetical) so each test sees the RNG
   state it sees in the reference run.
2. Behaviours of the verbatim reference recorded in SURVEY.md Appendix A.2
   (measured there on the compiled reference): output counts, all-zero QPSK
   bits (quirk Q1), pushSRI once per call (Q2), inf/NaN first differential
   output (Q10), latency numAvg-1 (Q16).
3. oracle/prim_check: every arithmetic primitive bit-for-bit against the
   libstdc++ / libgcc / glibc routine the reference calls, in gnu++98.
"""
import cmath
import math
import os
import random
import subprocess

import numpy as np
import pytest

from ref_stimulus import gen_psk

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# unittest runs test methods in alphabetical order; random.seed(100) happens once at import
REFERENCE_TEST_ORDER = [
    ("testDiffDecode8PSK", 8, True),
    ("testDiffDecodeBPSK", 2, True),
    ("testDiffDecodeQPSK", 4, True),
    ("testNonDiffDecode8PSK", 8, False),
    ("testNonDiffDecodeBPSK", 2, False),
    ("testNonDiffDecodeQPSK", 4, False),
]


def reference_stimuli():
    rng = random.Random(100)
    out = {}
    for name, M, diff in REFERENCE_TEST_ORDER:
        out[name] = (M, diff) + gen_psk(1000, samp_per_baud=8, num_syms=M, differential=diff, rng=rng)
    return out


def to_cx(v):
    return v[0::2].astype(np.float64) + 1j * v[1::2].astype(np.float64)


def run_reference_case(oracle_mod, M, diff, data):
    comp = oracle_mod.OracleComponent()
    comp.samplesPerBaud = 8
    comp.constelationSize = M
    comp.numAvg = 100
    comp.differentialDecoding = diff
    # sb.DataSource.push(..., sampleRate=100) -> SRI.xdelta = 0.01, mode = 1, one packet
    return oracle_mod.run_stream(comp, data, 1.0 / 100)


@pytest.mark.parametrize("name", [n for n, _, _ in REFERENCE_TEST_ORDER])
def test_reference_component_test(oracle_mod, name):
    M, diff, data, syms = reference_stimuli()[name]
    res = run_reference_case(oracle_mod, M, diff, data)
    out_cx = to_cx(res["soft"])
    assert len(out_cx) == 901  # 1000 - (numAvg - 1), SURVEY A.2 / Q16
    if diff:
        # tests/test_psk_soft.py:183-204
        if M == 4:
            rot = complex(math.cos(math.pi / 4), math.sin(math.pi / 4))
            expected = [rot * x for x in syms]
        else:
            expected = syms
        max_error = max(abs(x - y) for x, y in zip(out_cx[1:], expected[1:]))
    else:
        # tests/test_psk_soft.py:223-238: best of the M rotation ambiguities
        thetas = {
            2: [0, math.pi],
            4: [math.pi / 4, 3 * math.pi / 4, 5 * math.pi / 4, 7 * math.pi / 4],
            8: [k * math.pi / 4 for k in range(8)],
        }[M]
        max_error = 1e99
        for theta in thetas:
            sc = complex(math.cos(theta), math.sin(theta))
            max_error = min(max_error, max(abs(sc * x - y) for x, y in zip(out_cx[1:], syms[1:])))
    assert max_error < 1e-3, max_error


@pytest.mark.parametrize("M", [2, 4, 8])
def test_recorded_reference_behaviour(oracle_mod, M):
    """SURVEY.md Appendix A.2 (verbatim reference, 1000 symbols, S=8, A=100)."""
    rng = random.Random(7)
    data, _ = gen_psk(1000, samp_per_baud=8, num_syms=M, differential=False, rng=rng)
    bpb = {2: 1, 4: 2, 8: 3}[M]
    for packet, n_calls in ((None, 1), (1000, 8), (7, 1143)):
        comp = oracle_mod.OracleComponent()
        comp.samplesPerBaud = 8
        comp.constelationSize = M
        comp.numAvg = 100
        res = oracle_mod.run_stream(comp, data, 0.01, packet_complex=packet)
        assert res["soft"].size == 2 * 901
        assert res["phase"].size == 901
        assert res["index"].size == 901
        assert res["bits"].size == 901 * bpb
        assert res["n_sri"] == n_calls  # Q2: the SRI block runs on every call
        if M == 4:
            assert not res["bits"].any()  # Q1: QPSK bits are all zero
        else:
            assert 0 < res["bits"].sum() < res["bits"].size


def test_recorded_first_differential_output(oracle_mod):
    """Q10 / A.2: first differential output divides by last=(0,0):
    (inf,inf) for BPSK, (-nan,inf) for QPSK on the probe; here: non-finite."""
    for M in (2, 4):
        data, _ = gen_psk(200, samp_per_baud=8, num_syms=M, differential=True, rng=random.Random(3))
        comp = oracle_mod.OracleComponent()
        comp.samplesPerBaud = 8
        comp.constelationSize = M
        comp.numAvg = 100
        comp.differentialDecoding = True
        res = oracle_mod.run_stream(comp, data, 0.01)
        first = res["soft"][:2]
        assert not np.isfinite(first).all()
        assert np.isfinite(res["soft"][2:]).all()


def test_cold_start_latency_large(oracle_mod):
    """A.2 / Q16: 131072 symbols in -> 130973 out."""
    n_sym = 131072
    rng = np.random.default_rng(0)
    k = rng.integers(0, 4, n_sym)
    x = np.repeat(np.exp(2j * np.pi * k / 4 + 0.3j), 8) * np.tile(0.6 + 0.4 * np.sin(np.pi * (np.arange(8) + 0.5) / 8), n_sym)
    iq = np.empty(2 * x.size, np.float32)
    iq[0::2] = x.real
    iq[1::2] = x.imag
    comp = oracle_mod.OracleComponent()
    comp.samplesPerBaud = 8
    comp.numAvg = 100
    comp.constelationSize = 4
    res = oracle_mod.run_stream(comp, iq, 0.01)
    assert res["phase"].size == 130973
    assert not res["bits"].any()


def test_prim_check_binary(oracle_mod):
    """Every oracle primitive equals the toolchain routine the reference calls."""
    r = subprocess.run([os.path.join(ROOT, "oracle", "prim_check")], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "PRIM_CHECK OK" in r.stdout
    assert "__cplusplus = 199711L" in r.stdout
