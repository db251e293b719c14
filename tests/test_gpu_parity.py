"""Parity of the HIP path (through the C ABI) with the CPU oracle, on a real MI355X.

Bar (BASELINE.json north_star): bits and sampleIndex bit-exact; soft symbols and phase within 1e-5
relative.  What the tests assert is stricter: every float of the soft and phase streams has the oracle's
BITS (assert_parity).  Same packetisation on both sides (the reference's float outputs depend on it,
SURVEY.md quirk Q2)."""
import random

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TOL = 1e-5  # relative, on the float streams


def _handle(n=1, **kw):
    from psk_soft_amd import lib as pl

    return pl.Handle(n, device=0, **kw)


def run_gpu(h, ch, iq, xdelta, packet=None):
    n = iq.size // 2
    step = n if not packet else packet
    outs = {"soft": [], "bits": [], "phase": [], "index": []}
    pos, first = 0, True
    while True:
        cnt = min(step, n - pos)
        r = h.process_host(ch, [dict(data=iq[2 * pos : 2 * (pos + cnt)], xdelta=xdelta, sriChanged=first)])[0]
        first = False
        for k in outs:
            outs[k].append(r[k])
        pos += cnt
        if pos >= n:
            break
    return {k: np.concatenate(v) for k, v in outs.items()}


def assert_parity(got, ref, ctx=""):
    """bits and sampleIndex identical; soft and phase BIT-identical too (uint32 equality of every finite value,
    same non-finite pattern): the kernels restate the reference's arithmetic rounding for rounding, including
    the order of additions of LinearFit's running sums, so nothing is left for a tolerance to absorb."""
    assert got["bits"].size == ref["bits"].size and got["index"].size == ref["index"].size, ctx
    assert np.array_equal(got["bits"], ref["bits"]), ctx + " bits differ at %s" % np.nonzero(got["bits"] != ref["bits"])[0][:5]
    assert np.array_equal(got["index"], ref["index"]), ctx + " sampleIndex differs"
    for k in ("soft", "phase"):
        a, b = np.ascontiguousarray(got[k], np.float32), np.ascontiguousarray(ref[k], np.float32)
        assert a.size == b.size, ctx
        fin = np.isfinite(b)
        assert np.array_equal(np.isfinite(a), fin), ctx + " non-finite pattern differs on " + k
        assert np.array_equal(np.isnan(a), np.isnan(b)), ctx + " NaN pattern differs on " + k
        if fin.any():
            ua, ub = a.view(np.uint32)[fin], b.view(np.uint32)[fin]
            if not np.array_equal(ua, ub):
                d = np.nonzero(ua != ub)[0]
                err = np.abs(a[fin].astype(np.float64) - b[fin]).max() / max(np.abs(b[fin]).max(), 1e-30)
                raise AssertionError("%s %s: %d of %d values differ in their bits (first at %d: %.9g vs %.9g), rel err %g"
                                     % (ctx, k, d.size, ua.size, d[0], a[fin][d[0]], b[fin][d[0]], err))


def oracle_run(oracle_mod, iq, props, xdelta=0.01, packet=None):
    o = oracle_mod.OracleComponent()
    for k, v in props.items():
        setattr(o, k, v)
    return oracle_mod.run_stream(o, iq, xdelta, packet_complex=packet)


CASES = []
for M in (2, 4, 8):
    for S in (8, 10):
        for diff in (0, 1):
            CASES.append((M, S, diff, 1 << 14, None))
CASES += [(4, 8, 0, 1 << 15, 1000), (4, 8, 0, 5000, 7), (8, 10, 0, 1 << 15, 4096), (2, 4, 0, 1 << 14, 3000),
          (4, 16, 0, 1 << 14, None), (4, 5, 0, 1 << 14, None), (4, 2, 0, 1 << 14, None), (4, 8, 1, 1 << 14, 777)]


@pytest.mark.parametrize("force_seq", [0, 1])
@pytest.mark.parametrize("M,S,diff,N,packet", CASES)
def test_single_channel_parity(oracle_mod, M, S, diff, N, packet, force_seq):
    """BASELINE configs[1] family: one channel, wave phase-scan vs CPU (and the reference-order kernel)."""
    from psk_soft_amd.stimulus import synth_channel

    iq = synth_channel(17 * M + S, M, S, N)
    props = dict(samplesPerBaud=S, constelationSize=M, numAvg=100, differentialDecoding=diff)
    ref = oracle_run(oracle_mod, iq, props, packet=packet)
    h = _handle()
    h.set_force_sequential(force_seq)
    h.configure(0, [props])
    got = run_gpu(h, 0, iq, 0.01, packet)
    st = h.stats()
    assert (st["channels_sequential"] if force_seq else st["channels_fast"]) == 1, st
    assert_parity(got, ref, "M%d S%d diff%d N%d pkt%s seq%d" % (M, S, diff, N, packet, force_seq))
    h.close()


@pytest.mark.parametrize("name", ["testDiffDecode8PSK", "testDiffDecodeBPSK", "testDiffDecodeQPSK",
                                  "testNonDiffDecode8PSK", "testNonDiffDecodeBPSK", "testNonDiffDecodeQPSK"])
def test_reference_component_tests_on_gpu(oracle_mod, name):
    """The reference's own six tests (tests/test_psk_soft.py:160-238) against the HIP path."""
    import math

    from tests.test_oracle_reference_kat import reference_stimuli, to_cx

    M, diff, data, syms = reference_stimuli()[name]
    h = _handle()
    h.configure(0, [dict(samplesPerBaud=8, constelationSize=M, numAvg=100, differentialDecoding=int(diff))])
    got = run_gpu(h, 0, data, 1.0 / 100)
    out_cx = to_cx(got["soft"])
    assert len(out_cx) == 901
    if diff:
        rot = complex(math.cos(math.pi / 4), math.sin(math.pi / 4)) if M == 4 else 1
        max_error = max(abs(x - rot * y) for x, y in zip(out_cx[1:], syms[1:]))
    else:
        thetas = {2: [0, math.pi], 4: [math.pi / 4 + k * math.pi / 2 for k in range(4)], 8: [k * math.pi / 4 for k in range(8)]}[M]
        max_error = min(max(abs(complex(math.cos(t), math.sin(t)) * x - y) for x, y in zip(out_cx[1:], syms[1:])) for t in thetas)
    assert max_error < 1e-3
    # and the same stream against the oracle at the parity bar
    ref = oracle_run(oracle_mod, data, dict(samplesPerBaud=8, constelationSize=M, numAvg=100, differentialDecoding=int(diff)))
    assert_parity(got, ref, name)
    h.close()


def test_mixed_batch_parity(oracle_mod):
    """BASELINE configs[4]: mixed BPSK/QPSK/8-PSK batch with per-channel phaseAvg / numAvg, ragged packets."""
    from psk_soft_amd.stimulus import synth_channel

    rng = random.Random(5)
    n_ch = 48
    props, iqs = [], []
    for c in range(n_ch):
        M = (2, 4, 8)[c % 3]
        S = (8, 10, 4, 16)[c % 4]
        p = dict(samplesPerBaud=S, constelationSize=M, numAvg=(25, 100, 60)[c % 3], phaseAvg=(10, 50, 200)[(c // 3) % 3],
                 differentialDecoding=int(c % 7 == 0))
        props.append(p)
        iqs.append(synth_channel(1000 + c, M, S, 12000 + 37 * c))
    h = _handle(n_ch)
    h.configure(0, props)
    got = [dict(soft=[], bits=[], phase=[], index=[]) for _ in range(n_ch)]
    cuts = [[0] + sorted(rng.sample(range(1, iqs[c].size // 2), 3)) + [iqs[c].size // 2] for c in range(n_ch)]
    for k in range(4):
        pk = [dict(data=iqs[c][2 * cuts[c][k] : 2 * cuts[c][k + 1]], xdelta=0.01, sriChanged=(k == 0)) for c in range(n_ch)]
        res = h.process_host(0, pk)
        for c in range(n_ch):
            for key in got[c]:
                got[c][key].append(res[c][key])
    assert h.stats()["channels_fast"] == n_ch
    for c in range(n_ch):
        o = oracle_mod.OracleComponent()
        for kk, v in props[c].items():
            setattr(o, kk, v)
        ref = dict(soft=[], bits=[], phase=[], index=[])
        for k in range(4):
            r = o.service(iqs[c][2 * cuts[c][k] : 2 * cuts[c][k + 1]], 0.01, sriChanged=(k == 0))
            ref["soft"].append(r.soft); ref["bits"].append(r.bits); ref["phase"].append(r.phase); ref["index"].append(r.index)
        assert_parity({k: np.concatenate(v) for k, v in got[c].items()}, {k: np.concatenate(v) for k, v in ref.items()}, "ch%d" % c)
    h.close()


def test_edge_packets(oracle_mod):
    """Empty, sub-symbol and odd-length packets; a real-data packet; no packet at all."""
    from psk_soft_amd.stimulus import synth_channel

    iq = synth_channel(3, 4, 8, 6000)
    sizes = [0, 1, 3, 8, 790, 1, 0, 15, 2048, 5, 3000]
    h = _handle()
    h.configure(0, [dict(samplesPerBaud=8, constelationSize=4, numAvg=100)])
    o = oracle_mod.OracleComponent()
    o.samplesPerBaud = 8; o.constelationSize = 4; o.numAvg = 100
    pos = 0
    for i, n in enumerate(sizes):
        seg = iq[2 * pos : 2 * (pos + n)]
        if i == 4:
            seg = np.concatenate([seg, np.float32([0.5])])  # odd dataBuffer.size(): the last float is ignored
        mode = 0 if i == 7 else 1
        r = o.service(seg, 0.01, mode=mode, sriChanged=(i == 0))
        g = h.process_host(0, [dict(data=seg, xdelta=0.01, mode=mode, sriChanged=(i == 0))])[0]
        assert g["n_warn"] == r.n_warn and g["sri_pushed"] == r.sri_pushed
        assert_parity(g, dict(soft=r.soft, bits=r.bits, phase=r.phase, index=r.index), "packet %d" % i)
        if mode == 1:
            pos += n
    assert h.process_host(0, [None])[0]["ret"] == 0  # NOOP
    h.close()


def test_tie_heavy_signal_takes_the_exact_timing_path(oracle_mod):
    """Rectangular pulses (the reference test's own stimulus): every intra-symbol phase has the
    same energy up to noise, the float screening cannot vouch for the argmax.  The screened kernel
    settles such blocks itself, exactly: from its energy ring for numAvg <= 128, from the register
    history (or, numAvg > 512, from the window's samples read again) for larger windows -- round 3;
    before, those handed the whole call to the exact-timing kernel, which cost a noisy batch of numAvg
    400 80 %.  Either way the first-maximum tie rule has to match the reference."""
    import random as _random

    from ref_stimulus import gen_psk

    data, _ = gen_psk(3000, samp_per_baud=8, num_syms=4, differential=False, rng=_random.Random(11))
    for numAvg, in_kernel in ((100, True), (200, True), (400, True), (600, True)):
        props = dict(samplesPerBaud=8, constelationSize=4, numAvg=numAvg)
        ref = oracle_run(oracle_mod, data, props, packet=8192)
        h = _handle()
        h.configure(0, [props])
        got = run_gpu(h, 0, data, 0.01, 8192)
        st = h.stats()  # of the last call: steady state
        assert st["channels_sequential"] == 0 and st["timing_exact_blocks"] > 0, st
        assert st["channels_exact_timing"] == (0 if in_kernel else 1), st
        assert_parity(got, ref, "ties numAvg=%d" % numAvg)
        h.close()
    # exact ties: a constant-envelope signal with NO noise at all
    props = dict(samplesPerBaud=8, constelationSize=4, numAvg=100)
    k = np.random.default_rng(1).integers(0, 4, 6000)
    x = np.repeat(np.exp(2j * np.pi * k / 4 + 0.2j), 8)
    iq = np.empty(2 * x.size, np.float32)
    iq[0::2] = x.real
    iq[1::2] = x.imag
    ref = oracle_run(oracle_mod, iq, props, packet=16000)
    h2 = _handle()
    h2.configure(0, [props])
    got = run_gpu(h2, 0, iq, 0.01, 16000)
    st = h2.stats()
    assert st["channels_sequential"] == 0 and st["channels_exact_timing"] == 0 and st["timing_exact_blocks"] > 0, st
    assert_parity(got, ref, "exact ties")
    h2.close()


def test_host_sized_energy_ring(oracle_mod):
    """samplesPerBaud = 10 keeps its window energies in an LDS ring whose length the host picks per
    launch from the largest numAvg (and the phase ring from the largest phaseAvg) among the launch's
    channels: batches whose maxima differ, windows at both ends of the range, rectangular pulses
    (near-ties settled from the ring in the kernel) and ragged packets so that ring offsets start
    anywhere."""
    import random as _random

    from psk_soft_amd.stimulus import synth_channel
    from ref_stimulus import gen_psk

    rng = random.Random(77)
    for A_set, n_set in (((1, 2, 3, 28), (1, 10, 50)), ((76, 100, 5), (50, 128, 129)), ((127, 128, 64, 2), (384, 200, 1))):
        props, iqs, cuts = [], [], []
        for c in range(24):
            M = (2, 4, 8)[c % 3]
            p = dict(samplesPerBaud=10, constelationSize=M, numAvg=A_set[c % len(A_set)], phaseAvg=n_set[(c // 2) % len(n_set)],
                     differentialDecoding=int(c % 5 == 0))
            props.append(p)
            if c % 4 == 3:
                iq, _ = gen_psk(900 + 11 * c, samp_per_baud=10, num_syms=M, differential=False, rng=_random.Random(c))
                iq = np.asarray(iq, np.float32)
            else:
                iq = synth_channel(7000 + c, M, 10, 9000 + 110 * c, sigma=(0.01, 0.1)[c % 2])
            iqs.append(iq)
            N = iq.size // 2
            cuts.append([0] + sorted(rng.sample(range(1, N), 3)) + [N])
        h = _handle(len(props))
        h.configure(0, props)
        got = [dict(soft=[], bits=[], phase=[], index=[]) for _ in props]
        n_exact = 0
        for k in range(4):
            pk = [dict(data=iqs[c][2 * cuts[c][k] : 2 * cuts[c][k + 1]], xdelta=0.01, sriChanged=(k == 0)) for c in range(len(props))]
            res = h.process_host(0, pk)
            st = h.stats()
            assert st["channels_sequential"] == 0, st
            n_exact += st["timing_exact_blocks"]
            for c in range(len(props)):
                for key in got[c]:
                    got[c][key].append(res[c][key])
        assert n_exact > 0
        for c in range(len(props)):
            o = oracle_mod.OracleComponent()
            for kk, v in props[c].items():
                setattr(o, kk, v)
            ref = dict(soft=[], bits=[], phase=[], index=[])
            for k in range(4):
                r = o.service(iqs[c][2 * cuts[c][k] : 2 * cuts[c][k + 1]], 0.01, sriChanged=(k == 0))
                ref["soft"].append(r.soft); ref["bits"].append(r.bits); ref["phase"].append(r.phase); ref["index"].append(r.index)
            assert_parity({k: np.concatenate(v) for k, v in got[c].items()}, {k: np.concatenate(v) for k, v in ref.items()},
                          "A=%s n=%s ch%d %s" % (A_set, n_set, c, props[c]))
        h.close()


def test_every_samples_per_baud_2_to_16_on_the_wave_scan_kernel(oracle_mod):
    """One instantiation of the wave-scan kernel per samplesPerBaud 2 .. 32 and window class (numAvg
    <= 128 / <= 256 / <= 512, and <= 1024 up to samplesPerBaud 16): every one of them against the
    oracle, ragged packets, shaped and rectangular pulses; samplesPerBaud 33 and 40 have none and go
    through the time-tiled kernels' run-time front stage (round 1: the reference-order kernel)."""
    import random as _random

    from psk_soft_amd.stimulus import synth_channel
    from ref_stimulus import gen_psk

    rng = random.Random(31)
    props, iqs, cuts = [], [], []
    for S in list(range(2, 34)) + [40]:
        for A in (100, 200, 400) + ((800,) if S <= 16 else ()):
            M = rng.choice([2, 4, 8])
            props.append(dict(samplesPerBaud=S, constelationSize=M, numAvg=A, phaseAvg=rng.choice([10, 50, 200]),
                              differentialDecoding=int(rng.random() < 0.2)))
            N = S * (A + rng.choice([900, 1300]))
            if rng.random() < 0.25:
                iq, _ = gen_psk(N // S, samp_per_baud=S, num_syms=M, differential=False, rng=_random.Random(S * 1000 + A))
                iq = np.asarray(iq, np.float32)
            else:
                iq = synth_channel(9000 + 16 * S + A, M, S, N, sigma=rng.choice([0.01, 0.1]))
            iqs.append(iq)
            cuts.append([0] + sorted(rng.sample(range(1, iq.size // 2), 2)) + [iq.size // 2])
    n_ch = len(props)
    h = _handle(n_ch, max_window_samples=16 * 1024 + 64)
    h.configure(0, props)
    got = [dict(soft=[], bits=[], phase=[], index=[]) for _ in range(n_ch)]
    for k in range(3):
        pk = [dict(data=iqs[c][2 * cuts[c][k] : 2 * cuts[c][k + 1]], xdelta=0.01, sriChanged=(k == 0)) for c in range(n_ch)]
        res = h.process_host(0, pk)
        for c in range(n_ch):
            for key in got[c]:
                got[c][key].append(res[c][key])
    st = h.stats()
    assert st["channels_sequential"] - st["channels_guard"] == 0 and st["channels_tiled"] >= 6, st  # (33 and 40, three windows each)
    for c in range(n_ch):
        o = oracle_mod.OracleComponent()
        for kk, v in props[c].items():
            setattr(o, kk, v)
        ref = dict(soft=[], bits=[], phase=[], index=[])
        for k in range(3):
            r = o.service(iqs[c][2 * cuts[c][k] : 2 * cuts[c][k + 1]], 0.01, sriChanged=(k == 0))
            ref["soft"].append(r.soft); ref["bits"].append(r.bits); ref["phase"].append(r.phase); ref["index"].append(r.index)
        assert_parity({k: np.concatenate(v) for k, v in got[c].items()}, {k: np.concatenate(v) for k, v in ref.items()},
                      "ch%d %s" % (c, props[c]))
    h.close()


def test_deep_fit_windows(oracle_mod):
    """phaseAvg in the thousands: the phase ring takes up to 128 KiB of LDS (dynamic LDS beyond 64 KiB is asked for per
    kernel), the channel runs one wave per CU -- at the wave-scan kernels' speed, not the reference-order kernel's 4.7 us
    per symbol.  Three calls: filling window, steady state, a window larger than a call."""
    from psk_soft_amd.stimulus import synth_channel

    for S, M, n, N in ((8, 4, 5000, 3 * 40000), (10, 8, 30000, 3 * 120000), (4, 2, 2049, 3 * 9000)):
        props = dict(samplesPerBaud=S, constelationSize=M, numAvg=100, phaseAvg=n)
        iq = synth_channel(4300 + S, M, S, N)
        ref = oracle_run(oracle_mod, iq, props, packet=N // 3)
        h = _handle(1, max_phase_avg=32768)
        h.configure(0, [props])
        got = run_gpu(h, 0, iq, 0.01, N // 3)
        st = h.stats()
        assert st["channels_fast"] == 1 and st["channels_sequential"] == 0, st
        assert_parity(got, ref, "phaseAvg %d" % n)
        h.close()


def test_long_phase_averages_stay_on_the_wave_scan_kernel(oracle_mod):
    """The LDS ring of unwrapped phases is sized by the host per launch (a power of two >= phaseAvg + 128, up
    to 32768 floats): phaseAvg up to 32640 runs on the wave-scan kernels (round 1: 1920), channels beyond 1920 in
    launches of their own so that their ring does not take the residency of the others.  Mixed in one batch, ragged
    packets, a window that is still filling at first."""
    from psk_soft_amd.stimulus import synth_channel

    rng = random.Random(41)
    cfgs = [(8, 4, 385), (8, 2, 500), (10, 8, 1000), (8, 4, 1920), (4, 4, 700), (16, 2, 450), (10, 4, 50), (8, 4, 1921), (7, 8, 3000)]
    props, iqs, cuts = [], [], []
    for c, (S, M, n) in enumerate(cfgs):
        props.append(dict(samplesPerBaud=S, constelationSize=M, numAvg=100, phaseAvg=n))
        N = S * 6000
        iqs.append(synth_channel(9500 + c, M, S, N, sigma=(0.01, 0.1)[c % 2]))
        cuts.append([0] + sorted(rng.sample(range(1, N), 2)) + [N])
    h = _handle(len(cfgs), max_phase_avg=4096)
    h.configure(0, props)
    got = [dict(soft=[], bits=[], phase=[], index=[]) for _ in cfgs]
    for k in range(3):
        pk = [dict(data=iqs[c][2 * cuts[c][k] : 2 * cuts[c][k + 1]], xdelta=0.01, sriChanged=(k == 0)) for c in range(len(cfgs))]
        res = h.process_host(0, pk)
        for c in range(len(cfgs)):
            for key in got[c]:
                got[c][key].append(res[c][key])
    st = h.stats()
    assert st["channels_sequential"] - st["channels_guard"] == 0 and st["channels_fast"] == len(cfgs), st
    for c in range(len(cfgs)):
        o = oracle_mod.OracleComponent()
        for kk, v in props[c].items():
            setattr(o, kk, v)
        ref = dict(soft=[], bits=[], phase=[], index=[])
        for k in range(3):
            r = o.service(iqs[c][2 * cuts[c][k] : 2 * cuts[c][k + 1]], 0.01, sriChanged=(k == 0))
            ref["soft"].append(r.soft); ref["bits"].append(r.bits); ref["phase"].append(r.phase); ref["index"].append(r.index)
        assert_parity({k: np.concatenate(v) for k, v in got[c].items()}, {k: np.concatenate(v) for k, v in ref.items()},
                      "ch%d %s" % (c, props[c]))
    h.close()


@pytest.mark.parametrize("ties_in_place", [0, 1])
def test_partial_last_block_through_the_exact_timing_kernel_s11(oracle_mod, monkeypatch, ties_in_place):
    """A case the randomised comparison found (tools/fuzz_gpu.py seed 1002, round 5, channel 194; the
    signal is tests/golden/cases/s11_a257_tail.npy, made by that tool's generator): samplesPerBaud 11,
    numAvg 257, near-ties that send both calls through the exact-timing instantiation <11, 4, true>
    (304 VGPRs), the second call ending in a partial block of 84 symbols.  An experimental build (a loop
    added to fit_block that was never executed) wrote garbage soft symbols for exactly those 84; the
    committed code is right, and this keeps it so.  (Since round 3 the screened kernel settles such near-ties in place and the
    call no longer reaches the exact tier: ties_in_place = 0 -- PSK_SOFT_TIES_IN_PLACE=0 -- keeps the case on <11, 4, true>,
    ties_in_place = 1 runs it the way it runs now.)"""
    import os

    monkeypatch.setenv("PSK_SOFT_TIES_IN_PLACE", str(ties_in_place))
    sig = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "cases", "s11_a257_tail.npy"))
    props = dict(samplesPerBaud=11, constelationSize=4, numAvg=257, phaseAvg=385, differentialDecoding=0)
    cuts = [0, 35342, 132000]
    h = _handle(1, max_window_samples=11 * 257 + 64, max_phase_avg=512)
    h.configure(0, [props])
    o = oracle_mod.OracleComponent()
    for k, v in props.items():
        setattr(o, k, v)
    for k in range(2):
        d = sig[2 * cuts[k] : 2 * cuts[k + 1]]
        g = h.process_host(0, [dict(data=d, xdelta=0.01, sriChanged=(k == 0))])[0]
        st = h.stats()
        assert st["channels_exact_timing"] == 1 - ties_in_place and st["channels_sequential"] == 0, st
        r = o.service(d, 0.01, sriChanged=(k == 0))
        assert r.phase.size % 128 != 0
        assert_parity(g, dict(soft=r.soft, bits=r.bits, phase=r.phase, index=r.index), "call %d" % k)
        assert np.array_equal(g["soft"].view(np.uint32), r.soft.view(np.uint32)), "call %d: soft not bit-identical" % k
    h.close()


@pytest.mark.parametrize("ties_in_place", [0, 1])
def test_partial_last_block_wide_symbols_s30(oracle_mod, monkeypatch, ties_in_place):
    """Two cases of the randomised comparison (tools/fuzz_gpu.py seed 20261004, rounds 129 and 148: channels 154 and
    18; the fixtures are the tails of their streams): samplesPerBaud 30, numAvg 400, a rectangular pulse in noise -- the
    30 timing phases within 1e-4 of each other, so the exact-timing instantiation <30, 4, true> (512 registers, 1431
    spills) decides; the call ends in a partial block.  The build of the time picked phase 9 where 13 had the larger
    sum by 5e-4, at three positions of that block: one sample of a symbol read as zero.  Its load path let every lane
    choose its own branch (packet / carried samples / nothing wanted); load_block now has none of that
    (psk_fast_loop.h), and the wrong tail block of round 1 had the same signature (DESIGN.md section 4 names the mechanism).
    ties_in_place = 0 keeps the cases on <30, 4, true>; 1 is how they run since round 3: settled in place by <30, 4, false>."""
    import os

    monkeypatch.setenv("PSK_SOFT_TIES_IN_PLACE", str(ties_in_place))
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "cases")
    for name, M, n in (("s30_a400_tail_a.npy", 4, 1921), ("s30_a400_tail_b.npy", 8, 200)):
        sig = np.load(os.path.join(here, name))
        props = dict(samplesPerBaud=30, constelationSize=M, numAvg=400, phaseAvg=n)
        ref = oracle_run(oracle_mod, sig, props)
        assert ref["index"].size % 128 != 0
        h = _handle(1, max_window_samples=30 * 400 + 64, max_phase_avg=2048)
        h.configure(0, [props])
        got = run_gpu(h, 0, sig, 0.01)
        st = h.stats()
        assert st["channels_exact_timing"] == 1 - ties_in_place and st["channels_sequential"] == 0, st
        assert_parity(got, ref, name)
        h.close()


def test_random_configuration_sweep(oracle_mod):
    """256 channels with random (samplesPerBaud, numAvg, M, phaseAvg, diff), random noise level
    and ragged packetisation, three calls each, every stream against the oracle: a broad net for
    anything configuration- or lane-dependent (register allocation differs per instantiation)."""
    from psk_soft_amd.stimulus import synth_channel

    rng = random.Random(2024)
    n_ch = 256
    props, iqs, cuts = [], [], []
    for c in range(n_ch):
        S = rng.choice([2, 4, 5, 8, 8, 8, 10, 10, 16, 3, 7])
        A = rng.choice([1, 2, 17, 25, 64, 100, 100, 128, 129, 200, 256, 300, 400])
        M = rng.choice([2, 4, 8])
        n = rng.choice([1, 2, 10, 50, 50, 200, 384, 400])
        p = dict(samplesPerBaud=S, constelationSize=M, numAvg=A, phaseAvg=n, differentialDecoding=int(rng.random() < 0.2))
        props.append(p)
        N = S * rng.choice([300, 700, 1500])
        iqs.append(synth_channel(5000 + c, M, S, N, sigma=rng.choice([0.01, 0.01, 0.05, 0.2])))
        cuts.append([0] + sorted(rng.sample(range(1, N), 2)) + [N])
    h = _handle(n_ch, max_window_samples=16 * 400 + 64)
    h.configure(0, props)
    got = [dict(soft=[], bits=[], phase=[], index=[]) for _ in range(n_ch)]
    for k in range(3):
        pk = [dict(data=iqs[c][2 * cuts[c][k] : 2 * cuts[c][k + 1]], xdelta=0.01, sriChanged=(k == 0)) for c in range(n_ch)]
        res = h.process_host(0, pk)
        for c in range(n_ch):
            for key in got[c]:
                got[c][key].append(res[c][key])
    n_bitdiff = 0
    for c in range(n_ch):
        o = oracle_mod.OracleComponent()
        for kk, v in props[c].items():
            setattr(o, kk, v)
        ref = dict(soft=[], bits=[], phase=[], index=[])
        for k in range(3):
            r = o.service(iqs[c][2 * cuts[c][k] : 2 * cuts[c][k + 1]], 0.01, sriChanged=(k == 0))
            ref["soft"].append(r.soft); ref["bits"].append(r.bits); ref["phase"].append(r.phase); ref["index"].append(r.index)
        g = {k: np.concatenate(v) for k, v in got[c].items()}
        r = {k: np.concatenate(v) for k, v in ref.items()}
        assert g["bits"].size == r["bits"].size
        n_bitdiff += int((g["bits"] != r["bits"]).sum())
        assert_parity(g, r, "channel %d %s" % (c, props[c]))
    assert n_bitdiff == 0
    h.close()


def test_exactness_guard_hands_over(oracle_mod):
    """A burst 2^30 stronger than the noise around it breaks the exactness of the energy sums
    (quirk Q8): the wave-scan kernel must refuse and the reference-order kernel must match the
    oracle bit for bit on bits / sampleIndex."""
    from psk_soft_amd.stimulus import synth_channel

    iq = synth_channel(9, 4, 8, 1 << 14).copy()
    iq[: 2 * 4000] *= np.float32(1e-6)
    iq[2 * 9000 : 2 * 9100] *= np.float32(3e4)
    props = dict(samplesPerBaud=8, constelationSize=4, numAvg=100)
    ref = oracle_run(oracle_mod, iq, props)
    h = _handle()
    h.configure(0, [props])
    got = run_gpu(h, 0, iq, 0.01)
    st = h.stats()
    assert st["channels_guard"] == 1, st
    assert_parity(got, ref, "guard")
    h.close()


def test_non_finite_samples_stay_on_the_wave_scan_kernels(oracle_mod):
    """NaN / inf samples (the reference has no special case for them, cpp/psk_soft.cpp:445-452: it runs on at its
    normal speed and puts out what IEEE arithmetic makes of them).  The screened tier hands such a call to the
    exact-timing tier, whose window sums are updated symbol by symbol like the reference's (inf while the sample
    sits in the window, NaN for the rest of the call once it has left) and which carries libgcc's complex-multiply
    recovery and 64-bit unwrap counts ((long)NaN is LONG_MIN on x86): every call comes out with the oracle's bits,
    non-finite pattern included, and NONE goes to the reference-order kernel -- in round 1 one NaN sample sent its
    channel there for good, at 150 ms a call."""
    from psk_soft_amd.stimulus import synth_channel

    # (numAvg 200 ... 600: the window classes that settle near-ties in place from their history since round 3 -- a non-finite energy
    # there must still send the call to the exact tier: the reference's RUNNING sums are NaN from the moment an inf leaves the
    # window, a sum rebuilt from the history would be finite again.  Infinities at several timing phases, so that some are
    # never the picked sample and show in the energies only; calls long enough for them to leave the window inside the call.)
    # (phaseAvg 1: a fit window of one point returns the point itself, cpp/psk_soft.cpp:164-171, whatever LinearFit's sums have
    # become -- NaN once an infinite phase has passed through them; the steady-state formula returned NaN from there on until
    # round 3's randomised comparison with non-finite samples found it)
    for M, diff, S, A, n_ph in ((4, 0, 8, 100, 50), (2, 1, 8, 100, 50), (8, 0, 8, 100, 50), (4, 0, 10, 100, 50), (4, 0, 8, 300, 50),
                                (8, 1, 5, 40, 50), (4, 0, 8, 200, 50), (8, 0, 10, 400, 50), (4, 0, 8, 600, 50), (2, 0, 12, 520, 50),
                                (8, 0, 4, 100, 1), (4, 0, 8, 100, 1), (4, 1, 8, 400, 1), (4, 0, 8, 100, 2)):
        iq = synth_channel(21 + M, M, S, 1 << 14 if A <= 300 else 3 << 15).copy()
        iq[2 * 7000] = np.float32("nan")
        iq[2 * 9001 + 1] = np.float32("inf")
        iq[2 * 12000] = -np.float32("inf")
        if A > 100:
            for j in range(S):
                iq[2 * (13000 + 50 * S * j + j)] = np.float32("inf")
        props = dict(samplesPerBaud=S, constelationSize=M, numAvg=A, phaseAvg=n_ph, differentialDecoding=diff)
        ref = oracle_run(oracle_mod, iq, props, packet=4096)
        h = _handle()
        h.configure(0, [props])
        n = iq.size // 2
        outs = {"soft": [], "bits": [], "phase": [], "index": []}
        seq_calls = 0
        for k, pos in enumerate(range(0, n, 4096)):
            r = h.process_host(0, [dict(data=iq[2 * pos : 2 * min(pos + 4096, n)], xdelta=0.01, sriChanged=(k == 0))])[0]
            for key in outs:
                outs[key].append(r[key])
            seq_calls += h.stats()["channels_sequential"]
        got = {k: np.concatenate(v) for k, v in outs.items()}
        assert_parity(got, ref, "non-finite M%d diff%d S%d A%d n%d" % (M, diff, S, A, n_ph))
        assert seq_calls == 0, seq_calls
        assert not np.isfinite(ref["soft"]).all()  # (the case is what it claims to be)
        h.close()


def test_one_poisoned_channel_does_not_hold_up_the_batch(oracle_mod):
    """256 channels, one of them fed a NaN sample in the first call.  The reference's state never recovers from
    that (its phase estimate is NaN or astronomically large from then on), and neither does the channel here: every
    later call of it goes through the exact-timing tier.  What must not happen is what round 1 did -- the whole
    batch waiting 150 ms per call for that one channel on the reference-order kernel.  Compared: the same batch
    with and without the poison, timed over the same calls."""
    import time

    from psk_soft_amd.stimulus import synth_channel

    C, S, M, N, calls = 256, 8, 4, 32768, 6
    props = dict(samplesPerBaud=S, constelationSize=M, numAvg=100)
    base = [synth_channel(4000 + c, M, S, N * calls) for c in range(C)]
    sick = 77

    def run(poison):
        chans = [x for x in base]
        if poison:
            chans[sick] = base[sick].copy()
            chans[sick][2 * 5000] = np.float32("nan")
        h = _handle(C)
        h.configure(0, [props] * C)
        res_last, t_calls = None, []
        for k in range(calls):
            pk = [dict(data=x[2 * k * N : 2 * (k + 1) * N], xdelta=0.01, sriChanged=(k == 0)) for x in chans]
            t0 = time.perf_counter()
            res_last = h.process_host(0, pk)
            t_calls.append(time.perf_counter() - t0)
            st = h.stats()
            assert st["channels_sequential"] == 0, (k, st)
        h.close()
        return res_last, t_calls, chans

    clean, t_clean, _ = run(False)
    dirty, t_dirty, chans = run(True)
    # the healthy channels: same bits with and without the neighbour's poison (spot check against the oracle too)
    for c in (0, sick - 1, sick + 1, C - 1):
        for key in ("soft", "phase", "bits", "index"):
            assert np.array_equal(clean[c][key].view(np.uint8), dirty[c][key].view(np.uint8)), (c, key)
    o = oracle_mod.OracleComponent()
    o.samplesPerBaud, o.constelationSize, o.numAvg = S, M, 100
    r = None
    for k in range(calls):
        r = o.service(chans[sick][2 * k * N : 2 * (k + 1) * N], 0.01, sriChanged=(k == 0))
    assert_parity(dirty[sick], dict(soft=r.soft, bits=r.bits, phase=r.phase, index=r.index), "the poisoned channel, last call")
    # steady-state calls (the first one of each run warms things up): the poisoned batch within 25 % of the clean one
    # (host-buffer path: copies dominate; the reference-order kernel would add 150 ms to each)
    a, b = sorted(t_clean[1:])[len(t_clean[1:]) // 2], sorted(t_dirty[1:])[len(t_dirty[1:]) // 2]
    assert b < 1.25 * a + 0.004, (t_clean, t_dirty)


def test_noisy_unwrap_fixed_point(oracle_mod):
    """Low SNR: the speculated unwrap counts are wrong in many places; the
    fixed-point passes must converge to the reference's feedback unwrap."""
    from psk_soft_amd.stimulus import synth_channel

    for M, sigma in ((8, 0.15), (4, 0.3), (2, 0.6)):
        iq = synth_channel(21, M, 8, 1 << 14, sigma=sigma)
        props = dict(samplesPerBaud=8, constelationSize=M, numAvg=100)
        ref = oracle_run(oracle_mod, iq, props)
        h = _handle()
        h.configure(0, [props])
        got = run_gpu(h, 0, iq, 0.01)
        st = h.stats()
        assert st["channels_fast"] == 1 and st["unwrap_extra_passes"] > 0, st
        assert_parity(got, ref, "low SNR M=%d sigma=%g" % (M, sigma))
        h.close()


def test_state_roundtrip(oracle_mod):
    """export_state / import_state: a channel resumed on another handle continues identically."""
    from psk_soft_amd.stimulus import synth_channel

    iq = synth_channel(5, 8, 10, 20000)
    props = dict(samplesPerBaud=10, constelationSize=8, numAvg=100)
    a = _handle()
    a.configure(0, [props])
    run_gpu(a, 0, iq[: 2 * 9000], 0.01)
    blob = a.export_state(0)
    b = _handle()
    b.import_state(0, blob)
    ga = a.process_host(0, [dict(data=iq[2 * 9000 :], xdelta=0.01)])[0]
    gb = b.process_host(0, [dict(data=iq[2 * 9000 :], xdelta=0.01)])[0]
    for k in ("soft", "bits", "phase", "index"):
        assert np.array_equal(ga[k], gb[k]), k
    a.close(); b.close()


def _replay(oracle_mod, h, script):
    """script: list of ('set', name, value) / ('push', iq, kwargs).  Applies it to the oracle and
    to channel 0 of handle h, comparing every call's four streams and SRI side channel."""
    o = oracle_mod.OracleComponent()
    n_call = 0
    for step in script:
        if step[0] == "set":
            setattr(o, step[1], step[2])
            h.configure(0, [{step[1]: step[2]}])
            continue
        iq, kw = step[1], step[2]
        r = o.service(iq, kw.get("xdelta", 0.01), mode=kw.get("mode", 1), sriChanged=kw.get("sriChanged", False),
                      inputQueueFlushed=kw.get("inputQueueFlushed", False))
        g = h.process_host(0, [dict(data=iq, xdelta=kw.get("xdelta", 0.01), mode=kw.get("mode", 1),
                                    sriChanged=kw.get("sriChanged", False), inputQueueFlushed=kw.get("inputQueueFlushed", False))])[0]
        ctx = "call %d" % n_call
        assert g["sri_pushed"] == r.sri_pushed and g["n_warn"] == r.n_warn, ctx
        if r.sri_pushed:
            assert g["sri_soft_xdelta"] == r.sri_soft_xdelta, ctx
            assert g["sri_bits_xdelta"] == r.sri_bits_xdelta or (np.isnan(g["sri_bits_xdelta"]) and np.isnan(r.sri_bits_xdelta)), ctx
        assert_parity(g, dict(soft=r.soft, bits=r.bits, phase=r.phase, index=r.index), ctx)
        n_call += 1


def test_property_changes_mid_stream(oracle_mod):
    """samplesPerBaud / numAvg / constelationSize / phaseAvg / differentialDecoding /
    resetState changed between packets, queue flush, xdelta change, an unsupported constellation
    (no bits, one warning per symbol), numAvg shrinking below the filled window (the reference
    then stalls: nothing more comes out until the window can fill again)."""
    from psk_soft_amd.stimulus import synth_channel

    iq8 = synth_channel(31, 4, 8, 40000)
    iq10 = synth_channel(32, 8, 10, 30000)
    pos8 = [0]

    def take8(n):
        a = pos8[0]
        pos8[0] += n
        return iq8[2 * a : 2 * (a + n)]

    script = [
        ("set", "samplesPerBaud", 8), ("set", "constelationSize", 4), ("set", "numAvg", 100),
        ("push", take8(3000), dict(sriChanged=True)),
        ("set", "phaseAvg", 20), ("push", take8(2000), {}),
        ("set", "phaseAvg", 120), ("push", take8(2000), {}),
        ("set", "constelationSize", 8), ("push", take8(2500), {}),
        ("set", "constelationSize", 16), ("push", take8(1000), {}),      # unsupported: no bits, warnings
        ("set", "constelationSize", 2), ("set", "differentialDecoding", 1), ("push", take8(2000), {}),
        ("set", "differentialDecoding", 0), ("set", "numAvg", 40), ("push", take8(1500), {}),   # shrink -> stall
        ("push", take8(1000), {}),
        ("set", "numAvg", 150), ("push", take8(3000), {}),                 # window can fill again
        ("set", "resetState", 1), ("push", take8(2000), {}),
        ("push", take8(1500), dict(inputQueueFlushed=True)),
        ("push", take8(1500), dict(xdelta=0.5, sriChanged=True)),          # new sample rate: fit history cleared
        ("push", take8(1000), dict(xdelta=1.0, sriChanged=True)),          # xdelta == 1.0: the Q3 comparison
        ("push", take8(1000), dict(xdelta=1.0)),
        ("set", "samplesPerBaud", 10), ("set", "constelationSize", 8), ("set", "numAvg", 60),
        ("push", iq10[: 2 * 9000], {}),
        ("set", "samplesPerBaud", 7), ("push", iq10[2 * 9000 : 2 * 14000], {}),
        ("set", "samplesPerBaud", 1), ("push", iq10[2 * 14000 : 2 * 14500], {}),  # S == 1: nothing out (Q11)
        ("set", "numAvg", 0), ("push", iq10[2 * 14500 : 2 * 15000], {}),
        ("set", "resetState", 1), ("push", iq10[2 * 15000 : 2 * 15600], {}),      # S == 1, numAvg == 0: a symbol per sample
        ("push", iq10[2 * 15600 : 2 * 16000], {}),
        ("set", "samplesPerBaud", 10), ("set", "numAvg", 30), ("push", iq10[2 * 16000 : 2 * 22000], {}),
    ]
    h = _handle(max_window_samples=4096, max_phase_avg=256)
    _replay(oracle_mod, h, script)
    h.close()


@pytest.mark.parametrize("path", ["tiled", "reference_order", "handed_over"])
def test_samples_per_baud_1_leaves_the_carried_window_alone(oracle_mod, path):
    """samplesPerBaud == 1 EMITTING with numAvg > 0 (the window a resyncEnergy trimmed to numAvg samples, index 0): the
    reference pushes and pops nothing there (cpp/psk_soft.cpp:445, :468-469), so the samples carried from the earlier
    samplesPerBaud = 8 regime must still be the deque's content when a wider window is configured again -- its first
    timing picks are made from them.  Through the time-tiled kernels' run-time front stage, through the
    reference-order kernel, and through the hand-over from the one to the other (a NaN sample in the S == 1 call)."""
    from psk_soft_amd.stimulus import synth_channel

    iq = synth_channel(77, 4, 8, 30000)
    nanpkt = iq[2 * 9000 : 2 * 9600].copy()
    if path == "handed_over":
        nanpkt[2 * 300] = np.nan
    script = [
        ("set", "samplesPerBaud", 8), ("set", "constelationSize", 4), ("set", "numAvg", 100), ("set", "phaseAvg", 50),
        ("push", iq[: 2 * 9000], dict(sriChanged=True)),
        ("set", "samplesPerBaud", 1), ("set", "numAvg", 50),
        ("push", nanpkt, {}),                                   # emits 600 symbols, one per sample; the deque keeps its 50
        ("push", iq[2 * 9600 : 2 * 10000], {}),
        ("set", "samplesPerBaud", 8), ("set", "numAvg", 100),
        ("push", iq[2 * 10000 : 2 * 20000], {}),                # the window refills behind the 50 carried samples
        ("push", iq[2 * 20000 : 2 * 30000], {}),
    ]
    h = _handle(max_window_samples=4096, max_phase_avg=256)
    if path == "reference_order":
        h.set_force_sequential(1)
    _replay(oracle_mod, h, script)
    h.close()


def test_call_longer_than_resync_count(oracle_mod):
    """More than 1048576 symbols in ONE call: the reference resyncs symbolEnergy and the fit sums
    in mid-loop (cpp/psk_soft.cpp:51-52, 582-583).  The library cuts such a call at those boundaries and runs the
    pieces as continuations of one serviceFunction() call on the wave-scan / time-tiled kernels (round 2: the
    reference-order kernel, 4.7 us per symbol)."""
    n_sym = 1048576 + 3000
    S = 2
    rng = np.random.default_rng(9)
    k = rng.integers(0, 2, n_sym + 10)
    amp = np.tile(np.float32([1.0, 0.6]), n_sym + 10)
    ph = np.repeat(np.pi * k + 0.4, S) + 2e-5 * np.arange((n_sym + 10) * S)
    x = amp * np.exp(1j * ph) + 0.01 * (rng.standard_normal(ph.size) + 1j * rng.standard_normal(ph.size))
    iq = np.empty(2 * x.size, np.float32)
    iq[0::2] = x.real
    iq[1::2] = x.imag
    props = dict(samplesPerBaud=S, constelationSize=2, numAvg=10, phaseAvg=20)
    ref = oracle_run(oracle_mod, iq, props)
    assert ref["phase"].size > 1048576
    h = _handle(max_packet_complex=iq.size // 2)
    h.configure(0, [props])
    got = run_gpu(h, 0, iq, 0.01)
    st = h.stats()
    assert st["channels_sequential"] == 0 and st["channels_fast"] == 1, st
    assert_parity(got, ref, "long call")
    h.close()


@pytest.mark.parametrize("S,M,A,n_ph,n_sym", [(2, 4, 100, 50, (1 << 21) + 12345), (8, 8, 25, 200, (1 << 20) + 77)])
def test_long_calls_in_pieces_with_carried_state(oracle_mod, S, M, A, n_ph, n_sym):
    """Calls of more than 2^20 symbols, a short call in front (so that the long one starts from a filled window and a
    fit history) and one behind (the state the pieces leave is the state the reference is in): all four streams bit for
    bit, nothing on the reference-order kernel.  The 2^21-symbol call at samplesPerBaud 2 must take milliseconds."""
    import time

    from psk_soft_amd.stimulus import synth_channel

    n0, n2 = 5000, 7000
    iq = synth_channel(4242 + S, M, S, (n0 + n2) * S + n_sym * S)
    props = dict(samplesPerBaud=S, constelationSize=M, numAvg=A, phaseAvg=n_ph)
    cuts = [0, n0 * S, n0 * S + n_sym * S, iq.size // 2]
    o = oracle_mod.OracleComponent()
    for k, v in props.items():
        setattr(o, k, v)
    h = _handle(max_packet_complex=n_sym * S + 16, max_phase_avg=256)
    h.configure(0, [props])
    for k in range(3):
        x = iq[2 * cuts[k] : 2 * cuts[k + 1]]
        r = o.service(x, 0.01, sriChanged=(k == 0))
        t0 = time.perf_counter()
        g = h.process_host(0, [dict(data=x, xdelta=0.01, sriChanged=(k == 0))])[0]
        dt = time.perf_counter() - t0
        st = h.stats()
        assert st["channels_sequential"] == 0 and st["channels_fast"] == 1, (k, st)
        assert_parity(g, dict(soft=r.soft, bits=r.bits, phase=r.phase, index=r.index), "S=%d call %d" % (S, k))
        if k == 1:
            assert r.phase.size > (1 << 20)
            assert dt < 2.0, "a %d-symbol call took %.2f s" % (r.phase.size, dt)
    h.close()


def test_host_ingest_pipeline_chunks(oracle_mod, monkeypatch):
    """psk_soft_process_host cuts a batch into chunks that are packed, uploaded, processed and
    downloaded on rotating streams.  With a 1 MiB staging size the 48-channel batch below takes a
    dozen chunks per call (and a single packet larger than the staging size grows its slot);
    results must not depend on the chunking."""
    from psk_soft_amd import lib as pl
    from psk_soft_amd.stimulus import synth_channel

    monkeypatch.setenv("PSK_SOFT_STAGE_MB", "1")
    monkeypatch.setenv("PSK_SOFT_HOST_THREADS", "4")
    n_ch = 48
    rng = random.Random(5)
    props, iqs = [], []
    for c in range(n_ch):
        S = rng.choice([8, 8, 10, 4])
        M = rng.choice([2, 4, 8])
        props.append(dict(samplesPerBaud=S, constelationSize=M, numAvg=rng.choice([50, 100]), phaseAvg=rng.choice([20, 50])))
        N = rng.choice([20000, 33000, 50000]) if c != 7 else 200000  # channel 7: 1.6 MB > staging size
        iqs.append(synth_channel(7000 + c, M, S, N))
    h = pl.Handle(n_ch, device=0)
    h.configure(0, props)
    got = [dict(soft=[], bits=[], phase=[], index=[]) for _ in range(n_ch)]
    for k in range(2):
        pk = []
        for c in range(n_ch):
            n = iqs[c].size // 2
            a, b = (0, n // 2) if k == 0 else (n // 2, n)
            if c == 11 and k == 0:
                pk.append(None)  # getPacket() returned nothing for this channel
            else:
                pk.append(dict(data=iqs[c][2 * a : 2 * b], xdelta=0.01, sriChanged=(k == 0 or c == 11)))
        res = h.process_host(0, pk)
        for c in range(n_ch):
            for key in got[c]:
                got[c][key].append(res[c][key])
    for c in range(n_ch):
        o = oracle_mod.OracleComponent()
        for kk, v in props[c].items():
            setattr(o, kk, v)
        ref = dict(soft=[], bits=[], phase=[], index=[])
        n = iqs[c].size // 2
        for k in range(2):
            if c == 11 and k == 0:
                continue  # no packet for this channel in the first call
            a, b = (0, n // 2) if k == 0 else (n // 2, n)
            r = o.service(iqs[c][2 * a : 2 * b], 0.01, sriChanged=(k == 0 or c == 11))
            ref["soft"].append(r.soft); ref["bits"].append(r.bits); ref["phase"].append(r.phase); ref["index"].append(r.index)
        g = {k: np.concatenate(v) for k, v in got[c].items()}
        r = {k: np.concatenate(v) for k, v in ref.items()}
        assert_parity(g, r, "ingest ch%d" % c)
    h.close()


def test_zero_copy_from_pinned_host_memory(oracle_mod):
    """psk_soft_process_device with packets and result buffers in PINNED HOST memory
    (psk_soft_host_alloc): under HIP unified addressing the kernels read the packets and write the
    results straight over PCIe, no staging copy at all ('Many streams on one GPU')."""
    from psk_soft_amd import lib as pl
    from psk_soft_amd.stimulus import synth_channel

    n_ch, N, S, M = 8, 40000, 8, 4
    iq_np = [synth_channel(8100 + c, M, S, N) for c in range(n_ch)]
    cap = (N // S + 2 + 63) // 64 * 64
    iq = pl.host_alloc(n_ch * 2 * N, np.float32).reshape(n_ch, 2 * N)
    soft = pl.host_alloc(n_ch * 2 * cap, np.float32).reshape(n_ch, 2 * cap)
    phase = pl.host_alloc(n_ch * cap, np.float32).reshape(n_ch, cap)
    sidx = pl.host_alloc(n_ch * cap, np.int16).reshape(n_ch, cap)
    bits = pl.host_alloc(n_ch * 2 * cap, np.int16).reshape(n_ch, 2 * cap)
    for c in range(n_ch):
        iq[c] = iq_np[c]
    h = pl.Handle(n_ch, device=0)
    h.configure_all(samplesPerBaud=S, constelationSize=M)
    pk = (pl.Packet * n_ch)()
    out = (pl.Output * n_ch)()
    for c in range(n_ch):
        pk[c].data = iq[c].ctypes.data
        pk[c].n_floats = 2 * N
        pk[c].sri_xdelta = 0.01
        pk[c].sri_mode = 1
        pk[c].sriChanged = 1
        pk[c].present = 1
        out[c].soft = soft[c].ctypes.data
        out[c].bits = bits[c].ctypes.data
        out[c].phase = phase[c].ctypes.data
        out[c].sampleIndex = sidx[c].ctypes.data
        out[c].cap_symbols = cap
    h.process_device(0, pk, out)
    h.synchronize()
    for c in range(n_ch):
        n = int(out[c].n_symbols)
        ref = oracle_run(oracle_mod, iq_np[c], dict(samplesPerBaud=S, constelationSize=M))
        got = dict(soft=soft[c, : 2 * n].copy(), phase=phase[c, :n].copy(), bits=bits[c, : int(out[c].n_bits)].copy(),
                   index=sidx[c, : int(out[c].n_sampleIndex)].copy())
        assert_parity(got, ref, "pinned ch%d" % c)
    h.close()
    for a in (iq, soft, phase, sidx, bits):
        pl.host_free(a.reshape(-1))


def test_minimum_alignment_of_packets_and_outputs(oracle_mod):
    """The ABI asks for 8-byte aligned packets and soft rows and 4-byte aligned bits / phase /
    sampleIndex rows, no more.  The kernels fetch a lane's two symbols with 16-byte loads and write
    16 / 8 / 4-byte vectors: every buffer here sits at exactly its minimum alignment (8 resp. 4 bytes
    past a 16-byte boundary), in pinned host memory, for even and odd samplesPerBaud and two calls."""
    from psk_soft_amd import lib as pl
    from psk_soft_amd.stimulus import synth_channel

    cfgs = [(8, 4), (10, 8), (7, 2), (9, 4), (16, 4), (3, 8), (12, 2)]
    n_ch, N = len(cfgs), 30000
    iq_np = [synth_channel(8300 + c, M, S, N) for c, (S, M) in enumerate(cfgs)]
    cap = N // 2 + 8
    iq = pl.host_alloc(n_ch * (2 * N + 4), np.float32).reshape(n_ch, 2 * N + 4)
    soft = pl.host_alloc(n_ch * (2 * cap + 4), np.float32).reshape(n_ch, 2 * cap + 4)
    phase = pl.host_alloc(n_ch * (cap + 4), np.float32).reshape(n_ch, cap + 4)
    sidx = pl.host_alloc(n_ch * (cap + 8), np.int16).reshape(n_ch, cap + 8)
    bits = pl.host_alloc(n_ch * (3 * cap + 8), np.int16).reshape(n_ch, 3 * cap + 8)
    h = pl.Handle(n_ch, device=0)
    h.configure(0, [dict(samplesPerBaud=S, constelationSize=M) for S, M in cfgs])
    got = [dict(soft=[], bits=[], phase=[], index=[]) for _ in range(n_ch)]
    cuts = [0, 13001, N]
    for k in range(2):
        pk = (pl.Packet * n_ch)()
        out = (pl.Output * n_ch)()
        for c in range(n_ch):
            n_c = cuts[k + 1] - cuts[k]
            iq[c, 2 : 2 + 2 * n_c] = iq_np[c][2 * cuts[k] : 2 * cuts[k + 1]]
            pk[c].data = iq[c].ctypes.data + 8
            assert pk[c].data % 16 == 8
            pk[c].n_floats = 2 * n_c
            pk[c].sri_xdelta = 0.01
            pk[c].sri_mode = 1
            pk[c].sriChanged = int(k == 0)
            pk[c].present = 1
            out[c].soft = soft[c].ctypes.data + 8
            out[c].bits = bits[c].ctypes.data + 4
            out[c].phase = phase[c].ctypes.data + 4
            out[c].sampleIndex = sidx[c].ctypes.data + 4
            assert out[c].soft % 16 == 8 and out[c].bits % 8 == 4 and out[c].phase % 8 == 4 and out[c].sampleIndex % 8 == 4
            out[c].cap_symbols = cap
        h.process_device(0, pk, out)
        h.synchronize()
        for c in range(n_ch):
            n = int(out[c].n_symbols)
            got[c]["soft"].append(soft[c, 2 : 2 + 2 * n].copy())
            got[c]["phase"].append(phase[c, 1 : 1 + n].copy())
            got[c]["bits"].append(bits[c, 2 : 2 + int(out[c].n_bits)].copy())
            got[c]["index"].append(sidx[c, 2 : 2 + int(out[c].n_sampleIndex)].copy())
    assert h.stats()["channels_sequential"] == 0
    for c, (S, M) in enumerate(cfgs):
        o = oracle_mod.OracleComponent()
        o.samplesPerBaud = S
        o.constelationSize = M
        ref = dict(soft=[], bits=[], phase=[], index=[])
        for k in range(2):
            r = o.service(iq_np[c][2 * cuts[k] : 2 * cuts[k + 1]], 0.01, sriChanged=(k == 0))
            ref["soft"].append(r.soft); ref["bits"].append(r.bits); ref["phase"].append(r.phase); ref["index"].append(r.index)
        assert_parity({k: np.concatenate(v) for k, v in got[c].items()}, {k: np.concatenate(v) for k, v in ref.items()},
                      "S=%d M=%d" % (S, M))
    h.close()
    for a in (iq, soft, phase, sidx, bits):
        pl.host_free(a.reshape(-1))


@pytest.mark.parametrize("force_seq", [0, 1])
def test_opt_in_qpsk_sign_bitmap(oracle_mod, force_seq):
    """PSK_SOFT_OPT_QPSK_SIGN_BITMAP: QPSK bits follow the signs of the de-rotated symbol (the
    diagram at reference cpp/psk_soft.cpp:516-521) instead of being all zero (quirk Q1); everything
    else stays bit-identical to the default.  Checked against the soft symbols of the same run."""
    from psk_soft_amd.stimulus import synth_channel

    iq = synth_channel(77, 4, 8, 1 << 15)
    props = dict(samplesPerBaud=8, constelationSize=4)
    ref = oracle_run(oracle_mod, iq, props, packet=5000)
    h = _handle()
    h.set_force_sequential(force_seq)
    h.set_option(h.OPT_QPSK_SIGN_BITMAP, 1)
    h.configure(0, [props])
    got = run_gpu(h, 0, iq, 0.01, 5000)
    h.close()
    assert not ref["bits"].any()  # the reference's QPSK bits: all zero
    for k in ("soft", "phase", "index"):
        assert np.array_equal(got[k], ref[k]) or k != "index"
    re, im = got["soft"][0::2], got["soft"][1::2]
    want = np.empty(got["bits"].size, np.int16)
    want[0::2] = (re > 0) ^ (im > 0)
    want[1::2] = ~(im > 0)
    assert np.array_equal(got["bits"], want)
    assert 0.2 < got["bits"].mean() < 0.8


def test_large_carrier_offset_in_one_call(oracle_mod):
    """A carrier offset that runs the phase estimate far beyond 2*pi*M inside ONE call (the wrap only
    happens at call boundaries, cpp/psk_soft.cpp:592-603): sinf/cosf arguments reach hundreds of
    radians, i.e. the 192-bit argument reduction of glibc's sincosf, which the screened kernel does
    in line (stats: no hand-over to the other tiers)."""
    from psk_soft_amd.stimulus import synth_channel

    for M in (2, 4, 8):
        iq = synth_channel(900 + M, M, 8, 1 << 16, cfo_max=0.6, sigma=0.005)
        props = dict(samplesPerBaud=8, constelationSize=M)
        ref = oracle_run(oracle_mod, iq, props)
        assert np.abs(ref["phase"]).max() > 480.0  # |theta| = |est| / M well above 120
        h = _handle()
        h.configure(0, [props])
        got = run_gpu(h, 0, iq, 0.01)
        st = h.stats()
        assert st["channels_fast"] == 1 and st["channels_exact_timing"] == 0 and st["channels_sequential"] == 0, st
        assert_parity(got, ref, "large CFO M%d" % M)
        h.close()


def test_silent_streams_stay_on_the_fast_path(oracle_mod):
    """Idle channels are common in a multichannel system: all-zero packets, and streams with silent
    gaps, with and without differential decoding (where silence means 0/0 in __divsc3 on every
    symbol and NaN soft decisions in the reference).  Results as the oracle's, and no hand-over to
    the slower tiers (one refused channel costs the whole batch a second pass)."""
    from psk_soft_amd.stimulus import synth_channel

    for diff in (0, 1):
        for kind in ("zeros", "gap"):
            iq = synth_channel(950 + diff, 4, 8, 1 << 15).copy()
            if kind == "zeros":
                iq[:] = 0
            else:
                iq[2 * 9000 : 2 * 15000] = 0
            props = dict(samplesPerBaud=8, constelationSize=4, differentialDecoding=diff)
            ref = oracle_run(oracle_mod, iq, props, packet=8192)
            h = _handle()
            h.configure(0, [props])
            got = run_gpu(h, 0, iq, 0.01, 8192)
            st = h.stats()
            assert_parity(got, ref, "silence diff%d %s" % (diff, kind))
            assert st["channels_fast"] == 1 and st["channels_sequential"] == 0 and st["channels_exact_timing"] == 0, (st, diff, kind)
            h.close()


def test_randomised_streams(oracle_mod, monkeypatch):
    """Three rounds of tools/fuzz_gpu.py (fixed seeds): random properties -- including values only the
    reference-order kernel takes --, amplitudes over six decades, noise from none to -10 dB SNR,
    rectangular / triangular / shaped pulses, silent stretches, carrier offsets that wrap, ragged
    packets, property changes, resets and flushed queues between calls; every channel against the
    oracle.  (Longer runs of the same tool: DESIGN.md section 4.)"""
    import os
    import sys

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import fuzz_gpu

    monkeypatch.setattr(sys, "argv", ["fuzz_gpu.py", "3", "160", "7"])
    assert fuzz_gpu.main() == 0


@pytest.mark.parametrize("S,M,n", [(8, 4, 50), (10, 8, 50), (8, 2, 200), (8, 4, 10), (5, 4, 300), (12, 8, 50)])
def test_large_phase_estimate_is_bit_identical(oracle_mod, S, M, n):
    """LinearFit::xySum depends on the ORDER of its additions at the 2^-53 level (cpp/psk_soft.cpp:70-79:
    xdelta*ySum is a rounded product).  A kernel that sums the same terms in another order flips the float
    rounding of phaseEstimate now and then; one ulp of the estimate is 6e-5 rad beyond 512 rad, i.e. above 1e-5
    on the soft symbols -- the round-1 kernel's known deviation.  Here the estimate runs to +-700 ... 2400 rad
    inside single calls (the reference only wraps it at the end of a call, :592-603), several channels per
    case with carrier offsets of both signs, two calls each: every float must have the reference's bits."""
    from psk_soft_amd.stimulus import synth_channel

    nsym = 12000
    N = nsym * S
    chans = []
    for i, cfo in enumerate((0.06, -0.1, 0.2, -0.2, 0.13, -0.17)):
        iq = synth_channel(900 + 31 * i + S, M, S, 2 * N, cfo=cfo, sigma=0.02)
        chans.append(iq)
    props = dict(samplesPerBaud=S, constelationSize=M, numAvg=100, phaseAvg=n)
    h = _handle(len(chans))
    h.configure(0, [props] * len(chans))
    got = [dict(soft=[], bits=[], phase=[], index=[]) for _ in chans]
    for k in range(2):
        res = h.process_host(0, [dict(data=x[2 * k * N : 2 * (k + 1) * N], xdelta=0.01, sriChanged=(k == 0)) for x in chans])
        for c in range(len(chans)):
            for key in got[c]:
                got[c][key].append(res[c][key])
    st = h.stats()
    h.close()
    assert st["channels_fast"] == len(chans), st
    peak = 0.0
    for c, x in enumerate(chans):
        ref = oracle_run(oracle_mod, x, props, packet=N)
        g = {k: np.concatenate(v) for k, v in got[c].items()}
        assert_parity(g, ref, "large-estimate S=%d M=%d n=%d ch=%d" % (S, M, n, c))
        peak = max(peak, float(np.abs(ref["phase"]).max()))
    assert peak > 700.0, peak  # the regime this test is about was reached


def _device_batch(oracle_mod, M, S, A, n_ph, C, calls, check_channels, expect_tiled=None, phase0=False, min_chain_blocks=0):
    """C channels through psk_soft_process_device (device-resident packets and output rows, rows on 128-byte
    boundaries as bench.py lays them out), `calls` = list of samples per call; the channels in check_channels are
    replayed through the oracle and compared bit for bit on all four streams."""
    from concurrent.futures import ThreadPoolExecutor

    from psk_soft_amd import lib as pl
    from psk_soft_amd.stimulus import synth_channel

    total = sum(calls)
    with ThreadPoolExecutor(8) as ex:
        kw = dict(cfo=0.0, phi0=0.0) if phase0 else {}
        host = np.stack(list(ex.map(lambda c: synth_channel(70000 + 13 * M + S + c, M, S, total, **kw), range(C))))
    bpb = {2: 1, 4: 2, 8: 3}[M]
    cap = (max(calls) // S + 2 + 63) // 64 * 64
    h = pl.Handle(C, device=0)
    h.configure_all(samplesPerBaud=S, constelationSize=M, numAvg=A, phaseAvg=n_ph)
    got = {c: dict(soft=[], bits=[], phase=[], index=[]) for c in check_channels}
    row_in = (2 * max(calls) * 4 + 127) // 128 * 128  # bytes per packet row
    d_in = h.device_alloc(C * row_in)
    d_soft, d_phase = h.device_alloc(C * cap * 8), h.device_alloc(C * cap * 4)
    d_sidx, d_bits = h.device_alloc(C * cap * 2), h.device_alloc(C * cap * 2 * bpb)
    try:
        pos = 0
        for k, n in enumerate(calls):
            stage = np.zeros((C, row_in // 4), np.float32)
            stage[:, : 2 * n] = host[:, 2 * pos : 2 * (pos + n)]
            h.upload(d_in, stage)
            pk = (pl.Packet * C)()
            out = (pl.Output * C)()
            for c in range(C):
                pk[c].data = d_in + c * row_in
                pk[c].n_floats = 2 * n
                pk[c].sri_xdelta = 0.01
                pk[c].sri_mode = 1
                pk[c].sriChanged = int(k == 0)
                pk[c].present = 1
                out[c].soft = d_soft + c * cap * 8
                out[c].bits = d_bits + c * cap * 2 * bpb
                out[c].phase = d_phase + c * cap * 4
                out[c].sampleIndex = d_sidx + c * cap * 2
                out[c].cap_symbols = cap
            h.process_device(0, pk, out)
            h.synchronize()
            st = h.stats()
            assert st["channels_fast"] == C and st["channels_sequential"] == 0, st
            if k > 0:
                assert st["fit_chain_blocks"] >= min_chain_blocks, st
            if expect_tiled is not None:
                assert st["channels_tiled"] == expect_tiled, st
            soft = h.download(d_soft, (C, 2 * cap), np.float32)
            phase = h.download(d_phase, (C, cap), np.float32)
            sidx = h.download(d_sidx, (C, cap), np.int16)
            bits = h.download(d_bits, (C, bpb * cap), np.int16)
            for c in check_channels:
                ns = int(out[c].n_symbols)
                assert int(out[c].n_bits) == bpb * ns
                got[c]["soft"].append(soft[c, : 2 * ns].copy())
                got[c]["phase"].append(phase[c, :ns].copy())
                got[c]["index"].append(sidx[c, :ns].copy())
                got[c]["bits"].append(bits[c, : bpb * ns].copy())
            pos += n
    finally:
        for p in (d_in, d_soft, d_phase, d_sidx, d_bits):
            h.device_free(p)
        h.close()
    for c in check_channels:
        o = oracle_mod.OracleComponent()
        o.samplesPerBaud, o.constelationSize, o.numAvg, o.phaseAvg = S, M, A, n_ph
        ref = dict(soft=[], bits=[], phase=[], index=[])
        p2 = 0
        for k, n in enumerate(calls):
            r = o.service(host[c, 2 * p2 : 2 * (p2 + n)], 0.01, sriChanged=(k == 0))
            ref["soft"].append(r.soft); ref["bits"].append(r.bits); ref["phase"].append(r.phase); ref["index"].append(r.index)
            p2 += n
        assert_parity({k: np.concatenate(v) for k, v in got[c].items()}, {k: np.concatenate(v) for k, v in ref.items()},
                      "M=%d S=%d batch of %d, channel %d" % (M, S, C, c))


def test_machine_filling_batch_qpsk_s8(oracle_mod):
    """BASELINE configs[2] at the residency bench.py runs it at: 4096 channels in ONE launch = 16 single-wave
    workgroups on every CU, the 160 KiB of LDS of a CU exactly full.  Two calls (cold start, then carried state);
    first, last and every 32nd channel against the oracle, bit for bit."""
    C = 4096
    _device_batch(oracle_mod, 4, 8, 100, 50, C, [8192, 8192], sorted(set(range(0, C, 32)) | {C - 1}))


def test_machine_filling_batch_8psk_s10(oracle_mod):
    """The per-GPU shard of BASELINE configs[3]: 4096 channels of 8-PSK at 10 samples per baud, packets that are
    not a multiple of 10 samples (the symbol clock carries the leftovers from call to call)."""
    C = 4096
    _device_batch(oracle_mod, 8, 10, 100, 50, C, [8197, 8191], sorted(set(range(0, C, 32)) | {C - 1}))


def test_machine_filling_batch_all_channels_at_zero_phase(oracle_mod):
    """The data-dependent worst case of the wave-scan kernel, at the bench's residency: 4096 channels whose constellation sits
    at zero phase with no carrier offset -- the signal shape of the reference's own component test (reference
    tests/test_psk_soft.py:98-117).  The M-th-power phase is noise around zero, LinearFit's running sums hover around zero and
    cross binades from symbol to symbol: the wave-parallel candidates of the sums rarely verify and most blocks take the
    reference-order recurrence (fit_chain_blocks), in EVERY wave of the launch.  Bit for bit against the oracle."""
    C = 4096
    _device_batch(oracle_mod, 4, 8, 100, 50, C, [8192, 8192], sorted(set(range(0, C, 64)) | {C - 1}), phase0=True,
                  min_chain_blocks=C * 2)


def test_round1_one_ulp_case_is_bit_identical_now(oracle_mod):
    """The case that showed round 1's known deviation (tools/fuzz_gpu.py seed 702, round 0, channel 69; the signal is
    tests/golden/cases/s13_a64_large_estimate.npy, made by that tool's generator): samplesPerBaud 13, QPSK, numAvg 64,
    phaseAvg 128, ONE call of 156000 samples in which the phase estimate runs to -1342 rad.  Round 1's kernel, which
    summed LinearFit's xySum in tree order, put out one phase value one ulp off (symbol 10452: -1175.06396 against
    -1175.06384) and the soft symbol with it, 1.08e-5 relative -- above the 1e-5 bar.  Every float must have the
    oracle's bits."""
    import os

    sig = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "cases", "s13_a64_large_estimate.npy"))
    props = dict(samplesPerBaud=13, constelationSize=4, numAvg=64, phaseAvg=128, differentialDecoding=0)
    o = oracle_mod.OracleComponent()
    for k, v in props.items():
        setattr(o, k, v)
    r = o.service(sig, 0.01, sriChanged=True)
    assert r.phase.size == 11937 and float(np.abs(r.phase).max()) > 1300.0
    h = _handle(1)
    h.configure(0, [props])
    g = h.process_host(0, [dict(data=sig, xdelta=0.01, sriChanged=True)])[0]
    st = h.stats()
    h.close()
    assert st["channels_fast"] == 1 and st["channels_sequential"] == 0, st
    assert_parity(g, dict(soft=r.soft, bits=r.bits, phase=r.phase, index=r.index), "seed 702 / round 0 / channel 69")
    assert g["phase"].view(np.uint32)[10452] == r.phase.view(np.uint32)[10452]


def test_deferred_join_of_mixed_window_classes(oracle_mod):
    """PSK_SOFT_OPT_DEFERRED_JOIN: a batch that mixes window classes ends every class's calls on a stream of its own and joins
    them only when asked to (psk_soft_join / psk_soft_synchronize).  Eight calls issued back to back without a host wait in
    between, each into buffers of its own (an input or output touched out of order would show); in the middle a few channels
    change their window class (the library has to join before their next call: they move to another stream) and one call
    lacks a channel's packet (it runs the joined way).  Every channel of every call against the oracle, bit for bit."""
    from concurrent.futures import ThreadPoolExecutor

    from psk_soft_amd import lib as pl
    from psk_soft_amd.stimulus import synth_channel

    S, C, calls, n = 8, 384, 8, 6000
    Ms = [(2, 4, 8)[c % 3] for c in range(C)]
    props = [dict(samplesPerBaud=S, constelationSize=Ms[c], numAvg=(25, 100, 200, 400)[(c // 3) % 4], phaseAvg=(10, 50, 200)[(c // 12) % 3])
             for c in range(C)]
    with ThreadPoolExecutor(8) as ex:
        host = np.stack(list(ex.map(lambda c: synth_channel(9100 + c, Ms[c], S, calls * n), range(C))))
    cap = (n // S + 2 + 63) // 64 * 64
    h = pl.Handle(C, device=0)
    h.configure(0, props)
    h.set_option(pl.Handle.OPT_DEFERRED_JOIN, 1)
    row_in = 2 * n * 4
    d_in = h.device_alloc(calls * C * row_in)
    d_soft, d_phase = h.device_alloc(calls * C * cap * 8), h.device_alloc(calls * C * cap * 4)
    d_sidx, d_bits = h.device_alloc(calls * C * cap * 2), h.device_alloc(calls * C * cap * 6)
    moved = {7: 400, 100: 25, 205: 100}  # channel -> new numAvg, from call 4 on
    absent = (5, 33)                     # (call, channel) without a packet
    outs = []
    try:
        stage = np.ascontiguousarray(host.reshape(C, calls, 2 * n).transpose(1, 0, 2))
        h.upload(d_in, stage)
        h.synchronize()
        for k in range(calls):
            if k == 4:
                for c, A in moved.items():
                    props[c] = dict(props[c], numAvg=A)
                    h.configure(c, [props[c]])
            pk, out = (pl.Packet * C)(), (pl.Output * C)()
            for c in range(C):
                base = k * C + c
                pk[c].data, pk[c].n_floats, pk[c].sri_xdelta, pk[c].sri_mode = d_in + base * row_in, 2 * n, 0.01, 1
                pk[c].sriChanged, pk[c].present = int(k == 0), int((k, c) != absent)
                out[c].soft, out[c].bits = d_soft + base * cap * 8, d_bits + base * cap * 6
                out[c].phase, out[c].sampleIndex, out[c].cap_symbols = d_phase + base * cap * 4, d_sidx + base * cap * 2, cap
            h.process_device(0, pk, out)  # (no wait: the next call is issued while this one's classes are running)
            outs.append(out)
        h.join()
        h.synchronize()
        st = h.stats()
        assert st["channels_sequential"] == 0, st
        soft = h.download(d_soft, (calls, C, 2 * cap), np.float32)
        phase = h.download(d_phase, (calls, C, cap), np.float32)
        sidx = h.download(d_sidx, (calls, C, cap), np.int16)
        bits = h.download(d_bits, (calls, C, 3 * cap), np.int16)
    finally:
        for p in (d_in, d_soft, d_phase, d_sidx, d_bits):
            h.device_free(p)
        h.close()

    def check(c):
        o = oracle_mod.OracleComponent()
        first = dict(props[c], numAvg=(25, 100, 200, 400)[(c // 3) % 4])
        for kk, v in first.items():
            setattr(o, kk, v)
        b = {2: 1, 4: 2, 8: 3}[Ms[c]]
        for k in range(calls):
            if k == 4 and c in moved:
                o.numAvg = moved[c]
            if (k, c) == absent:
                assert int(outs[k][c].n_symbols) == 0
                continue
            r = o.service(host[c, 2 * n * k : 2 * n * (k + 1)], 0.01, sriChanged=(k == 0))
            ns = int(outs[k][c].n_symbols)
            assert ns == r.soft.size // 2, (c, k)
            assert_parity(dict(soft=soft[k, c, : 2 * ns], phase=phase[k, c, :ns], index=sidx[k, c, :ns], bits=bits[k, c, : b * ns]),
                          dict(soft=r.soft, phase=r.phase, index=r.index, bits=r.bits), "deferred join, channel %d call %d" % (c, k))

    for c in range(C):
        check(c)


@pytest.mark.parametrize("pieces", [2, 3, 5])
def test_mixed_batch_cut_in_time(oracle_mod, monkeypatch, pieces):
    """A batch that mixes window classes with calls of 128 blocks or more is cut in time inside the library (psk_capi.cpp:
    PSK_SOFT_SPLIT_CLASSES pieces, each a continuation of the ONE serviceFunction() call, the classes running through them on
    their own streams and joined at the end of the call).  Cut points fall in the middle of the call, where the reference
    neither rebuilds its energy sums nor resets its fit: everything -- timing picks, the unwrap across the cut, LinearFit's
    sums and count, differential decoding's `last`, the end-of-call wrap (once, at the end) -- has to come out as in the uncut
    call.  Two calls in a row, ragged lengths (a channel whose call is too short to cut, one that emits an odd number of
    symbols), every channel against the oracle."""
    from concurrent.futures import ThreadPoolExecutor

    from psk_soft_amd import lib as pl
    from psk_soft_amd.stimulus import synth_channel

    monkeypatch.setenv("PSK_SOFT_SPLIT_CLASSES", str(pieces))
    S, C, calls = 8, 30, 2
    Ms = [(2, 4, 8)[c % 3] for c in range(C)]
    props = [dict(samplesPerBaud=S, constelationSize=Ms[c], numAvg=(25, 100, 200, 400, 1000)[c % 5], phaseAvg=(10, 50, 200)[(c // 5) % 3],
                  differentialDecoding=int(c % 4 == 1)) for c in range(C)]
    lens = [140000 + 8 * (37 * c % 1000) + (8 if c % 2 else 0) for c in range(C)]
    lens[3], lens[4] = 9000, 70000  # (too short to cut; cut into fewer blocks than pieces x 128)
    with ThreadPoolExecutor(8) as ex:
        host = list(ex.map(lambda c: synth_channel(4200 + c, Ms[c], S, calls * lens[c], cfo=(0.02 if c % 7 == 0 else None)), range(C)))
    h = pl.Handle(C, device=0, max_window_samples=16384, max_phase_avg=512)
    h.configure(0, props)
    cap = [lens[c] // S + 2 for c in range(C)]
    d_in = [h.device_alloc(2 * lens[c] * 4) for c in range(C)]
    d_soft, d_phase = [h.device_alloc(cap[c] * 8) for c in range(C)], [h.device_alloc(cap[c] * 4) for c in range(C)]
    d_sidx, d_bits = [h.device_alloc(cap[c] * 2 + 4) for c in range(C)], [h.device_alloc(cap[c] * 6 + 4) for c in range(C)]
    got = [dict(soft=[], bits=[], phase=[], index=[]) for _ in range(C)]
    try:
        for k in range(calls):
            pk, out = (pl.Packet * C)(), (pl.Output * C)()
            for c in range(C):
                h.upload(d_in[c], host[c][2 * k * lens[c] : 2 * (k + 1) * lens[c]])
                pk[c].data, pk[c].n_floats, pk[c].sri_xdelta, pk[c].sri_mode, pk[c].sriChanged, pk[c].present = d_in[c], 2 * lens[c], 0.01, 1, int(k == 0), 1
                out[c].soft, out[c].bits, out[c].phase, out[c].sampleIndex, out[c].cap_symbols = d_soft[c], d_bits[c], d_phase[c], d_sidx[c], cap[c]
            h.process_device(0, pk, out)
            h.synchronize()
            st = h.stats()
            assert st["channels_fast"] == C and st["channels_sequential"] == 0, st
            for c in range(C):
                ns, b = int(out[c].n_symbols), {2: 1, 4: 2, 8: 3}[Ms[c]]
                assert int(out[c].n_bits) == b * ns and int(out[c].n_sampleIndex) == ns
                got[c]["soft"].append(h.download(d_soft[c], (2 * ns,), np.float32))
                got[c]["phase"].append(h.download(d_phase[c], (ns,), np.float32))
                got[c]["index"].append(h.download(d_sidx[c], (ns,), np.int16))
                got[c]["bits"].append(h.download(d_bits[c], (b * ns,), np.int16))
    finally:
        for lst in (d_in, d_soft, d_phase, d_sidx, d_bits):
            for q in lst:
                h.device_free(q)
        h.close()
    for c in range(C):
        ref = oracle_run(oracle_mod, host[c], props[c], packet=lens[c])
        assert_parity({k: np.concatenate(v) for k, v in got[c].items()}, ref, "cut in %d, channel %d (%s)" % (pieces, c, props[c]))


@pytest.mark.parametrize("scale", [1e-19, 1e-21, 3e-22, 2e-23])
def test_energies_among_the_denormals(oracle_mod, scale):
    """Samples of 1e-21: their energies, 1e-42, are denormal floats -- a fixed grid of 2^-149 on which the screening's RELATIVE
    error bounds mean nothing and the phase index it writes into the low bits of a window sum moves the sum by several grid
    steps.  The reference adds the same floats in double and picks the first maximum; the screened kernels have to notice that
    they cannot tell and settle the block exactly (an absolute floor on the acceptance threshold, psk_fast_loop.h).  Found by
    the randomised comparison at extreme amplitudes (tools/fuzz_gpu.py, PSK_FUZZ_EXTREME): wrong timing picks at 1e-21.  A
    rectangular pulse in noise (every timing phase within the noise of the others), short and long windows, a 14-samples-per-
    baud case like the one the comparison found; at 2e-23 most energies are 0 or 1 grid step."""
    import random as _random

    from psk_soft_amd.stimulus import synth_channel
    from ref_stimulus import gen_psk

    data, _ = gen_psk(3000, samp_per_baud=8, num_syms=4, differential=False, rng=_random.Random(12))
    cases = [(np.asarray(data, np.float64), dict(samplesPerBaud=8, constelationSize=4, numAvg=100), 8192),
             (np.asarray(data, np.float64), dict(samplesPerBaud=8, constelationSize=4, numAvg=300), 8192),
             (synth_channel(77, 4, 14, 60000, sigma=0.05).astype(np.float64), dict(samplesPerBaud=14, constelationSize=4, numAvg=100, phaseAvg=400), 25000)]
    for x, props, packet in cases:
        iq = (x * scale).astype(np.float32)
        ref = oracle_run(oracle_mod, iq, props, packet=packet)
        h = _handle(1, max_phase_avg=512)
        h.configure(0, [props])
        got = run_gpu(h, 0, iq, 0.01, packet)
        assert h.stats()["channels_sequential"] == 0
        assert_parity(got, ref, "scale %g %s" % (scale, props))
        h.close()
