"""The time-tiled kernels (psk_soft_amd/csrc/psk_tile_kernel.h): calls of few channels and many symbols are cut
along time -- timing recovery and raw phase per tile, the feedback unwrap and fit one wave per channel, de-rotation
and slicing per tile again.  Same bar as everywhere: all four streams carry the oracle's bits, whatever the
packetisation; a call the tiles cannot vouch for is redone by the wave-scan kernels."""
import os

import numpy as np
import pytest

from tests.test_gpu_parity import _handle, assert_parity, oracle_run, run_gpu

pytestmark = pytest.mark.gpu


def _tiled_handle(n=1, mode=2):
    from psk_soft_amd import lib as pl

    h = _handle(n)
    h.set_option(pl.Handle.OPT_TIME_TILED, mode)
    return h


CASES = [
    # M, S, diff, numAvg, phaseAvg, N, packet
    (4, 8, 0, 100, 50, 1 << 16, None),
    (4, 8, 0, 100, 50, 1 << 16, 20000),
    (4, 8, 1, 100, 50, 1 << 15, 9999),
    (2, 8, 0, 100, 50, 1 << 15, None),
    (2, 4, 1, 25, 10, 1 << 14, 5000),
    (8, 10, 0, 100, 50, (1 << 16) + 7, None),
    (8, 10, 1, 60, 200, 1 << 15, 12345),
    (8, 9, 0, 128, 50, 1 << 15, None),
    (4, 16, 0, 100, 300, 1 << 16, None),
    (4, 5, 0, 7, 3, 1 << 14, 3001),
    (4, 2, 0, 100, 50, 1 << 13, None),
    (4, 3, 0, 1, 1, 1 << 13, 1000),
    (4, 8, 0, 100, 50, 5000, 7),      # calls far smaller than a tile, most of them emitting nothing
    (4, 13, 0, 64, 50, 1 << 15, None),
]


# PSK_SOFT_PIPELINED=2 (environment) forces the pipelined mode of the time-tiled path on every call that can take it: the parity
# assertions below then test THAT mode; the statistics that count the parallel fit do not apply
PIPED = os.environ.get("PSK_SOFT_PIPELINED") == "2"


@pytest.mark.parametrize("M,S,diff,A,n,N,packet", CASES)
def test_tiled_single_channel(oracle_mod, M, S, diff, A, n, N, packet):
    from psk_soft_amd.stimulus import synth_channel

    iq = synth_channel(31 * M + S, M, S, N)
    props = dict(samplesPerBaud=S, constelationSize=M, numAvg=A, phaseAvg=n, differentialDecoding=diff)
    ref = oracle_run(oracle_mod, iq, props, packet=packet)
    h = _tiled_handle()
    h.configure(0, [props])
    got = run_gpu(h, 0, iq, 0.01, packet)
    st = h.stats()
    assert st["channels_fast"] == 1 and st["channels_sequential"] == 0, st
    if got["index"].size and (packet is None or packet > 8 * S):
        assert st["channels_tiled"] == 1, st
        # from the second call on the fit window is full: unwrap and fit go through the parallel path (psk_pfit.h)
        if packet is not None and n >= 2 and not PIPED:
            assert st["channels_parallel_fit"] == 1 and st["parallel_fit_refusals"] == 0, st
    assert_parity(got, ref, "tiled M%d S%d diff%d A%d n%d N%d pkt%s" % (M, S, diff, A, n, N, packet))
    h.close()


def test_tiled_matches_long_run(oracle_mod):
    """BASELINE configs[1]: one channel, 2^20 samples in one call -- the case the tiles are for -- and the automatic
    choice takes it there."""
    from psk_soft_amd.stimulus import synth_channel

    iq = synth_channel(7, 4, 8, 1 << 20)
    props = dict(samplesPerBaud=8, constelationSize=4, numAvg=100, phaseAvg=50)
    ref = oracle_run(oracle_mod, iq, props)
    h = _tiled_handle(mode=1)
    h.configure(0, [props])
    got = run_gpu(h, 0, iq, 0.01)
    st = h.stats()
    assert st["channels_tiled"] == 1 and st["channels_fast"] == 1, st
    assert_parity(got, ref, "tiled 2^20")
    h.close()


def test_tiled_largest_call(oracle_mod):
    """The largest call the wave-scan and time-tiled kernels take: 2^20 symbols out (LinearFit::count reaches 1048576 at the
    call's last symbol, cpp/psk_soft.cpp:51): 8192 blocks, 4096 tiles, a second call behind it on the carried state."""
    from psk_soft_amd.stimulus import synth_channel

    S, A = 4, 50
    n1 = ((1 << 20) + A - 1) * S
    iq = synth_channel(11, 4, S, n1 + 40000)
    props = dict(samplesPerBaud=S, constelationSize=4, numAvg=A, phaseAvg=50)
    o = oracle_mod.OracleComponent()
    for k, v in props.items():
        setattr(o, k, v)
    h = _handle(1, max_packet_complex=n1)
    h.configure(0, [props])
    for k, (lo, hi) in enumerate(((0, n1), (n1, n1 + 40000))):
        g = h.process_host(0, [dict(data=iq[2 * lo : 2 * hi], xdelta=0.01, sriChanged=(k == 0))])[0]
        r = o.service(iq[2 * lo : 2 * hi], 0.01, sriChanged=(k == 0))
        st = h.stats()
        assert st["channels_tiled"] == 1 and st["channels_sequential"] == 0, (k, st)
        if k == 0:
            assert r.phase.size == 1 << 20
        assert_parity(g, dict(soft=r.soft, bits=r.bits, phase=r.phase, index=r.index), "call %d" % k)
    h.close()


def test_tiled_mixed_batch_with_carried_state(oracle_mod):
    """Several window classes in one batch, three calls with ragged cuts: every call starts from the state the call
    before left (sample ring, LinearFit history and sums, differential `last`), tiled and not."""
    from psk_soft_amd.stimulus import synth_channel

    n_ch = 24
    props, iqs = [], []
    for c in range(n_ch):
        M = (2, 4, 8)[c % 3]
        S = (8, 10, 4, 16, 6)[c % 5]
        props.append(dict(samplesPerBaud=S, constelationSize=M, numAvg=(25, 100, 60)[c % 3], phaseAvg=(10, 50, 200)[(c // 3) % 3],
                          differentialDecoding=int(c % 4 == 0)))
        iqs.append(synth_channel(2000 + c, M, S, 30000 + 411 * c))
    for mode_seq in ((2, 2, 2), (2, 0, 2), (0, 2, 0)):
        h = _tiled_handle(n_ch)
        h.configure(0, props)
        got = [dict(soft=[], bits=[], phase=[], index=[]) for _ in range(n_ch)]
        cuts = [[0, 9000 + 17 * c, 9000 + 17 * c + 64 * (c % 5), iqs[c].size // 2] for c in range(n_ch)]
        from psk_soft_amd import lib as pl

        for k in range(3):
            h.set_option(pl.Handle.OPT_TIME_TILED, mode_seq[k])
            pk = [dict(data=iqs[c][2 * cuts[c][k] : 2 * cuts[c][k + 1]], xdelta=0.01, sriChanged=(k == 0)) for c in range(n_ch)]
            res = h.process_host(0, pk)
            st = h.stats()
            assert st["channels_fast"] == n_ch and st["channels_sequential"] == 0, st
            if mode_seq[k] == 2 and k != 1:
                assert st["channels_tiled"] == n_ch, st
            if mode_seq[k] == 0:
                assert st["channels_tiled"] == 0, st
            for c in range(n_ch):
                for key in got[c]:
                    got[c][key].append(res[c][key])
        for c in range(n_ch):
            o = oracle_mod.OracleComponent()
            for k2, v in props[c].items():
                setattr(o, k2, v)
            ref = dict(soft=[], bits=[], phase=[], index=[])
            for k in range(3):
                r = o.service(iqs[c][2 * cuts[c][k] : 2 * cuts[c][k + 1]], 0.01, sriChanged=(k == 0))
                ref["soft"].append(r.soft); ref["bits"].append(r.bits); ref["phase"].append(r.phase); ref["index"].append(r.index)
            ref = {key: np.concatenate(v) for key, v in ref.items()}
            assert_parity({key: np.concatenate(v) for key, v in got[c].items()}, ref, "modes %s ch %d" % (mode_seq, c))
        h.close()


def test_tiled_hands_over_what_it_cannot_carry(oracle_mod):
    """A rectangular pulse makes every timing phase tie (the reference's own test stimulus): the tiles re-decide
    exactly and stay; a NaN sample or a noisy unwrap is not theirs -- the call comes out of the wave-scan kernels
    with the same bits."""
    from psk_soft_amd.stimulus import synth_channel
    from tests.test_oracle_reference_kat import reference_stimuli

    M, diff, data, _ = reference_stimuli()["testNonDiffDecodeQPSK"]
    props = dict(samplesPerBaud=8, constelationSize=M, numAvg=100)
    ref = oracle_run(oracle_mod, data, props, xdelta=0.01)
    h = _tiled_handle()
    h.configure(0, [props])
    got = run_gpu(h, 0, data, 0.01)
    st = h.stats()
    assert st["channels_fast"] == 1 and st["channels_sequential"] == 0, st
    assert_parity(got, ref, "rectangular pulse")
    h.close()

    iq = synth_channel(5, 4, 8, 1 << 15).copy()
    iq[2 * 20000] = np.nan
    props = dict(samplesPerBaud=8, constelationSize=4, numAvg=100)
    ref = oracle_run(oracle_mod, iq, props)
    h = _tiled_handle()
    h.configure(0, [props])
    got = run_gpu(h, 0, iq, 0.01)
    st = h.stats()
    assert st["channels_tiled"] == 0 and st["channels_fast"] == 1 and st["channels_exact_timing"] == 1, st
    assert_parity(got, ref, "NaN sample")
    h.close()

    iq = synth_channel(6, 4, 8, 1 << 15, sigma=0.3)
    ref = oracle_run(oracle_mod, iq, props)
    h = _tiled_handle()
    h.configure(0, [props])
    got = run_gpu(h, 0, iq, 0.01)
    assert h.stats()["channels_sequential"] == 0
    assert_parity(got, ref, "10 dB")
    h.close()


@pytest.mark.skipif(PIPED, reason="counts the parallel fit, which the forced pipelined mode replaces")
def test_parallel_fit_verifies_or_steps_back(oracle_mod):
    """The parallel fit guesses the unwrap counts from consecutive raw phases: right on a clean signal (all calls but
    the first, whose fit window is still filling), wrong somewhere at 10 dB -- then the call is the block-by-block fit
    kernel's, and the statistics say why.  Same bits either way."""
    from psk_soft_amd.stimulus import synth_channel

    props = dict(samplesPerBaud=8, constelationSize=4, numAvg=100, phaseAvg=50)
    for sigma, expect in ((0.01, True), (0.3, False)):
        iq = synth_channel(77, 4, 8, 3 << 16, sigma=sigma)
        ref = oracle_run(oracle_mod, iq, props, packet=1 << 16)
        h = _tiled_handle()
        h.configure(0, [props])
        got = run_gpu(h, 0, iq, 0.01, 1 << 16)
        st = h.stats()
        assert st["channels_fast"] == 1 and st["channels_sequential"] == 0, st
        if expect:
            assert st["channels_parallel_fit"] == 1 and st["parallel_fit_refusals"] == 0, st
        else:  # (refused by the front tiles altogether, or the unwrap counts did not verify)
            assert st["channels_parallel_fit"] == 1 or st["channels_tiled"] == 0 or st["parallel_fit_refusals"] & 4, st
        assert_parity(got, ref, "sigma %g" % sigma)
        h.close()


def test_tiled_bursts_and_gaps(oracle_mod):
    """A tile does not know the largest window sum of its call, which the screening thresholds scale with: it starts from
    what the channel's last tiled call says the sums can reach (ChanState::emax_hint).  A stream that goes silent, comes
    back, or jumps by 60 dB inside a call stays on the tile path from its second call on; same bits."""
    from psk_soft_amd.stimulus import synth_channel

    for kind in ("gap", "burst", "fade", "zeros"):
        iq = synth_channel(960, 4, 8, 1 << 16).copy()
        if kind == "gap":
            iq[2 * 20000 : 2 * 30000] = 0
        elif kind == "burst":
            iq[: 2 * 40000] *= 1e-3
        elif kind == "fade":
            iq[2 * 40000 :] *= 1e-3
        else:
            iq[:] = 0
        props = dict(samplesPerBaud=8, constelationSize=4, numAvg=100, phaseAvg=50)
        ref = oracle_run(oracle_mod, iq, props, packet=1 << 14)
        h = _tiled_handle()
        h.configure(0, [props])
        n, tiled = 1 << 14, 0
        got = dict(soft=[], bits=[], phase=[], index=[])
        for k in range(4):
            r = h.process_host(0, [dict(data=iq[2 * n * k : 2 * n * (k + 1)], xdelta=0.01, sriChanged=(k == 0))])[0]
            st = h.stats()
            assert st["channels_fast"] == 1 and st["channels_sequential"] == 0, (kind, k, st)
            tiled += st["channels_tiled"]
            for key in got:
                got[key].append(r[key])
        assert tiled == 4, (kind, tiled)
        assert_parity({key: np.concatenate(v) for key, v in got.items()}, ref, kind)
        h.close()


@pytest.mark.skipif(PIPED, reason="counts the parallel fit, which the forced pipelined mode replaces")
def test_parallel_fit_second_round(oracle_mod):
    """16 dB: the first guess of the unwrap counts (towards a smoothed carrier trajectory) misses single symbols in
    some calls; the second round, on the counts the first round's estimates give, verifies.  With the round always
    enqueued (option 2), and on demand (the default: enqueued after a call reported a miss)."""
    from psk_soft_amd import lib as pl
    from psk_soft_amd.stimulus import synth_channel

    C, N, calls = 16, 1 << 15, 8
    props = dict(samplesPerBaud=8, constelationSize=4, numAvg=100, phaseAvg=50)
    iqs = [synth_channel(100 + c, 4, 8, calls * N, sigma=0.15) for c in range(C)]
    refs = [oracle_run(oracle_mod, iqs[c], props, packet=N) for c in range(C)]
    for mode in (2, 1):
        h = _tiled_handle(C)
        h.set_option(pl.Handle.OPT_PARALLEL_FIT, mode)
        h.configure(0, [props] * C)
        got = [dict(soft=[], bits=[], phase=[], index=[]) for _ in range(C)]
        second = par = 0
        for k in range(calls):
            res = h.process_host(0, [dict(data=iq[2 * k * N : 2 * (k + 1) * N], xdelta=0.01, sriChanged=(k == 0)) for iq in iqs])
            st = h.stats()
            assert st["channels_fast"] == C and st["channels_sequential"] == 0, st
            second += st["channels_parallel_fit_second_round"]
            par += st["channels_parallel_fit"]
            for c in range(C):
                for key in got[c]:
                    got[c][key].append(res[c][key])
        assert par >= (calls - 1) * C * 3 // 4, (mode, par)
        assert second >= 1, (mode, second, par)
        for c in range(C):
            assert_parity({key: np.concatenate(v) for key, v in got[c].items()}, refs[c], "mode %d ch %d" % (mode, c))
        h.close()


@pytest.mark.skipif(PIPED, reason="counts the parallel fit, which the forced pipelined mode replaces")
def test_parallel_fit_large_carrier_offset(oracle_mod):
    """|phaseEstimate| runs into the hundreds of radians inside the calls (and wraps at their ends): the sums cross
    binades, the walker's own blocks and the composed ones meet; 8-PSK, differential decoding, phaseAvg 200."""
    from psk_soft_amd.stimulus import synth_channel

    for M, S, n, diff, cfo in ((4, 8, 50, 0, 0.09), (8, 10, 200, 1, -0.05), (2, 4, 17, 0, 0.02)):
        iq = synth_channel(91 + M, M, S, 3 << 15, cfo=cfo)
        props = dict(samplesPerBaud=S, constelationSize=M, numAvg=100, phaseAvg=n, differentialDecoding=diff)
        ref = oracle_run(oracle_mod, iq, props, packet=1 << 15)
        h = _tiled_handle()
        h.configure(0, [props])
        got = run_gpu(h, 0, iq, 0.01, 1 << 15)
        st = h.stats()
        assert st["channels_parallel_fit"] == 1, st
        assert_parity(got, ref, "cfo M%d" % M)
        h.close()


def test_parallel_fit_walks_runs_of_blocks_itself(oracle_mod):
    """Channels without a carrier offset whose constellation sits a few hundredths of a radian off zero phase (channel 14 of the
    128-channel sweep point of tools/walker_hist.py: offset 0, M * phi0 = 0.077): LinearFit's sums hover where practically no
    block can be entered with a prepared map, and the walker (pf_xwalk, psk_pfit.h) walks runs of blocks itself -- one after the
    other without a scan in between, the next block's operands requested ahead, the grid candidates tried once per run.  Next to
    it exactly zero phase (the signal shape of the reference's own test, reference tests/test_psk_soft.py:98-117), 8-PSK with
    differential decoding, a call that ends in a partial block and one of more than 64 blocks (a run across the walker's rounds);
    three calls each (the runs start from carried sums).  Bit for bit, and on the parallel fit: a refusal would hide the path
    behind the block-by-block kernel."""
    from psk_soft_amd.stimulus import synth_channel

    walked = []
    for M, S, n, diff, mphi0, npk in ((4, 8, 50, 0, 0.0766, 1 << 16), (4, 8, 50, 0, 0.0, (1 << 16) - 40), (8, 10, 200, 1, 0.05, 70 * 1280 + 30),
                                      (2, 4, 17, 0, 0.03, 1 << 15), (4, 8, 50, 0, -0.02, 1 << 17)):
        iq = synth_channel(131 + M, M, S, 3 * npk, cfo=0.0, phi0=mphi0 / M)
        props = dict(samplesPerBaud=S, constelationSize=M, numAvg=100, phaseAvg=n, differentialDecoding=diff)
        ref = oracle_run(oracle_mod, iq, props, packet=npk)
        h = _tiled_handle()
        h.configure(0, [props])
        got = run_gpu(h, 0, iq, 0.01, npk)
        st = h.stats()
        assert st["channels_parallel_fit"] == 1 and st["parallel_fit_refusals"] == 0, st
        walked.append((st["fit_chain_blocks"], (npk // S + 127) // 128))
        assert_parity(got, ref, "hovering M%d M*phi0 %.3f" % (M, mphi0))
        h.close()
    assert any(w >= nb // 4 for w, nb in walked), walked  # (a quarter of some call's blocks walked by the wave itself: runs)


def test_tiled_device_batch_auto(oracle_mod):
    """64 channels x 2^16 samples through the device-pointer entry point: the automatic choice tiles them; spot
    channels against the oracle."""
    from tests.test_gpu_parity import _device_batch

    _device_batch(oracle_mod, 4, 8, 100, 50, 64, [1 << 16, (1 << 16) - 24], (0, 1, 31, 63), expect_tiled=64)


@pytest.mark.parametrize("name", [
    "test_mixed_batch_parity", "test_edge_packets", "test_tie_heavy_signal_takes_the_exact_timing_path",
    "test_host_sized_energy_ring", "test_every_samples_per_baud_2_to_16_on_the_wave_scan_kernel", "test_deep_fit_windows",
    "test_random_configuration_sweep", "test_exactness_guard_hands_over", "test_non_finite_samples_stay_on_the_wave_scan_kernels",
    "test_noisy_unwrap_fixed_point", "test_state_roundtrip", "test_property_changes_mid_stream",
    "test_minimum_alignment_of_packets_and_outputs", "test_large_carrier_offset_in_one_call",
    "test_silent_streams_stay_on_the_fast_path", "test_zero_copy_from_pinned_host_memory",
])
def test_parity_suite_through_the_tiles(oracle_mod, monkeypatch, name):
    """The edge cases of tests/test_gpu_parity.py once more with every call that has kernels for it cut along time and the
    parallel fit's second round always enqueued (the environment switches new handles read): property changes and resets
    between calls, ragged and empty packets, ties, non-finite samples, state export / import, deep fit windows."""
    import tests.test_gpu_parity as tp

    monkeypatch.setenv("PSK_SOFT_TIME_TILED", "2")
    monkeypatch.setenv("PSK_SOFT_PARALLEL_FIT", "2")
    getattr(tp, name)(oracle_mod)


ANY_CASES = [
    # M, S, diff, numAvg, phaseAvg, N, packet
    (4, 40, 0, 100, 50, 40 * 3000, None),
    (4, 64, 1, 100, 50, 64 * 2500, 64 * 1000 + 17),
    (8, 33, 0, 60, 200, 33 * 2600, 33 * 900),
    (2, 8, 0, 2000, 50, 8 * 9000, 8 * 3000 + 5),
    (4, 10, 0, 1025, 385, 10 * 6000, None),
    (4, 24, 0, 600, 50, 24 * 4000, 24 * 1500),
    (4, 100, 0, 30, 10, 100 * 1500, 100 * 500 + 3),
    (4, 1024, 0, 3, 5, 1024 * 300, None),
    (4, 200, 0, 100, 50, 200 * 1500, 200 * 700 + 11),  # four phases a lane, a window of several tiles' length
    (2, 300, 1, 17, 10, 300 * 900, None),             # sixteen phases a lane
    (4, 1, 0, 0, 50, 20000, 7001),    # samplesPerBaud 1 with numAvg 0: a symbol per sample, no timing recovery (quirk Q11)
    (8, 1, 1, 0, 200, 30000, None),
    (2, 1, 0, 0, 3, 5000, 64),
    (8, 1, 0, 0, 60, 600, None),
    (8, 1, 0, 0, 60, 20000, 3000),
]


@pytest.mark.parametrize("M,S,diff,A,n,N,packet", ANY_CASES)
def test_window_classes_without_an_instantiation(oracle_mod, M, S, diff, A, n, N, packet):
    """samplesPerBaud > 32, numAvg > 1024 (> 512 for samplesPerBaud > 16): no wave-scan instantiation; the time-tiled kernels
    behind the front stage that takes both at run time (timing phases across the lanes, symbols one after the other) carry
    them -- round 1: the reference-order kernel, 4.7 us per symbol."""
    from psk_soft_amd.stimulus import synth_channel

    iq = synth_channel(53 * M + S, M, S, N)
    props = dict(samplesPerBaud=S, constelationSize=M, numAvg=A, phaseAvg=n, differentialDecoding=diff)
    ref = oracle_run(oracle_mod, iq, props, packet=packet)
    h = _handle(1, max_window_samples=S * A + 64, max_phase_avg=512)
    h.configure(0, [props])
    got = run_gpu(h, 0, iq, 0.01, packet)
    st = h.stats()
    assert st["channels_fast"] == 1 and st["channels_sequential"] == 0 and st["channels_tiled"] == 1, st
    assert_parity(got, ref, "any front M%d S%d A%d n%d" % (M, S, A, n))
    h.close()


def test_run_time_front_hands_over(oracle_mod):
    """What the run-time front stage does not carry (a non-finite sample) comes out of the reference-order kernel with the
    reference's values -- samplesPerBaud 40 through its regular loop, samplesPerBaud 1 through its symbol-per-sample one."""
    from psk_soft_amd.stimulus import synth_channel

    for S, A in ((40, 100), (1, 0)):
        iq = synth_channel(71 + S, 4, max(S, 1), max(S, 1) * 3000).copy()
        iq[2 * (max(S, 1) * 2000 + 3)] = np.inf
        props = dict(samplesPerBaud=S, constelationSize=4, numAvg=A, phaseAvg=50)
        ref = oracle_run(oracle_mod, iq, props, packet=max(S, 1) * 1000)
        h = _handle(1, max_window_samples=max(S * A, 16) + 64)
        h.configure(0, [props])
        got = run_gpu(h, 0, iq, 0.01, max(S, 1) * 1000)
        st = h.stats()
        assert st["channels_sequential"] == 1 and st["channels_guard"] == 1, st
        assert_parity(got, ref, "S%d" % S)
        h.close()


@pytest.mark.parametrize("S,A", [(100, 20), (64, 30), (33, 100), (300, 5)])
def test_run_time_front_with_every_window_sum_nan(oracle_mod, S, A):
    """Samples of magnitude 3e19: every energy overflows float, the window sums go to inf as the symbols arrive and to
    inf - inf = NaN as they leave -- in EVERY phase, so no phase holds the maximum.  The run-time front stage refuses such a
    tile (non-finite energies), but it runs to its end and fetches the sample of its "pick": with more than one phase a lane
    that was the sentinel index of an empty comparison, 16 GiB behind the packet -- a GPU memory fault (found by the randomised
    comparison, seed 20270130 round 9 with samplesPerBaud up to 100 and PSK_FUZZ_EXTREME).  The pick is now a sample of the
    symbol whatever the sums are; the call is the reference-order kernel's and carries the oracle's values."""
    M = 4
    iq = np.empty(2 * S * 1500, np.float32)
    rng = np.random.default_rng(S)
    ph = 2 * np.pi * rng.integers(0, M, 1500) / M + 0.3
    x = np.repeat(np.exp(1j * ph), S) * 3.0e19  # (a rectangular pulse: no phase stays finite)
    iq[0::2] = x.real.astype(np.float32)
    iq[1::2] = x.imag.astype(np.float32)
    props = dict(samplesPerBaud=S, constelationSize=M, numAvg=A, phaseAvg=50)
    ref = oracle_run(oracle_mod, iq, props, packet=S * 700 + 13)
    h = _handle(1, max_window_samples=S * A + 64)
    h.configure(0, [props])
    got = run_gpu(h, 0, iq, 0.01, S * 700 + 13)
    st = h.stats()
    assert st["channels_sequential"] == 1 and st["channels_guard"] == 1, st
    assert_parity(got, ref, "S%d A%d" % (S, A))
    h.close()


def test_plans_pass_the_host_side_validation(oracle_mod, monkeypatch):
    """PSK_SOFT_VALIDATE=1: in front of the first launch of a call the host checks what the kernels take for granted about every
    plan (the samples a call reads exist, what it carries out fits the rings, its place in the scratch of the time-tiled kernels
    and of the parallel fit lies inside them) and refuses the call otherwise.  The batches below -- window classes side by side,
    tiled and not, ragged calls, the run-time front stage and its hand-over -- pass it, with the oracle's values.  (The randomised
    comparison runs under the switch too: DESIGN.md section 4.)"""
    monkeypatch.setenv("PSK_SOFT_VALIDATE", "1")
    test_tiled_mixed_batch_with_carried_state(oracle_mod)
    test_run_time_front_hands_over(oracle_mod)
    test_window_classes_without_an_instantiation(oracle_mod, 4, 100, 0, 30, 10, 100 * 1500, 100 * 500 + 3)


def test_more_channels_than_a_grid_dimension(oracle_mod):
    """A window class that always goes through the time-tiled kernels (samplesPerBaud 40: no wave-scan instantiation) with more
    channels than the y dimension of a grid holds (65535): the launches go out in slices of the class's channel list."""
    from psk_soft_amd import lib as pl
    from psk_soft_amd.stimulus import synth_channel

    C, S, M, A, N = 66000, 40, 4, 4, 2400
    base = [synth_channel(900 + c, M, S, N) for c in range(8)]
    h = pl.Handle(C, device=0, max_window_samples=256, max_phase_avg=64)
    h.configure_all(samplesPerBaud=S, constelationSize=M, numAvg=A, phaseAvg=20)
    pkts = [dict(data=base[c % 8], xdelta=0.01, sriChanged=True) for c in range(C)]
    res = h.process_host(0, pkts)
    st = h.stats()
    assert st["channels_fast"] == C and st["channels_tiled"] == C and st["channels_sequential"] == 0, st
    for c in (0, 1, 7, 65534, 65535, 65536, C - 1):
        ref = oracle_run(oracle_mod, base[c % 8], dict(samplesPerBaud=S, constelationSize=M, numAvg=A, phaseAvg=20))
        assert_parity(res[c], ref, "channel %d of %d" % (c, C))
    h.close()


@pytest.mark.parametrize("S,M,diff,n_ph", [(8, 4, 0, 50), (10, 8, 1, 200), (4, 2, 0, 10), (16, 4, 1, 300)])
def test_pipelined_ranges(oracle_mod, monkeypatch, S, M, diff, n_ph):
    """The pipelined mode of the time-tiled path (psk_tile.hip: psk_tile_fit_range_kernel): the call's tiles go out in ranges,
    front / fit / back of consecutive ranges on three streams, the fit state of a channel waiting in scratch between two
    launches.  Forced here on a small batch (PSK_SOFT_PIPELINED=2: ranges of three tiles): channels of ragged lengths (their
    calls end in different ranges, some in the first), partial last blocks, differential decoding (the back stage's `last`),
    several calls in a row (carried state), and one channel noisy enough to be handed over in the middle of a call (the later
    ranges must leave it alone and the wave-scan kernels redo it).  All four streams against the oracle, bit for bit."""
    from psk_soft_amd.stimulus import synth_channel

    monkeypatch.setenv("PSK_SOFT_PIPELINED", "2")
    C, calls = 7, 3
    lens = [40000, 40000, 1000 * S, 23456, 40000, 17 * 128 * S + 5 * S, 40000]
    props = dict(samplesPerBaud=S, constelationSize=M, numAvg=100, phaseAvg=n_ph, differentialDecoding=diff)
    iqs = [synth_channel(500 + 7 * S + c, M, S, calls * lens[c], sigma=(0.35 if c == 4 else 0.01)) for c in range(C)]
    h = _tiled_handle(C)
    h.configure(0, [props] * C)
    got = [dict(soft=[], bits=[], phase=[], index=[]) for _ in range(C)]
    tiled = 0
    for k in range(calls):
        res = h.process_host(0, [dict(data=iqs[c][2 * k * lens[c] : 2 * (k + 1) * lens[c]], xdelta=0.01, sriChanged=(k == 0)) for c in range(C)])
        st = h.stats()
        assert st["channels_fast"] == C and st["channels_sequential"] == 0 and st["channels_parallel_fit"] == 0, st
        tiled += st["channels_tiled"]
        for c in range(C):
            for key in got[c]:
                got[c][key].append(res[c][key])
    assert tiled >= calls * (C - 2), tiled  # (the noisy channel may be handed over; everything else stays on the tiled kernels)
    for c in range(C):
        o = oracle_mod.OracleComponent()
        for kk, v in props.items():
            setattr(o, kk, v)
        ref = dict(soft=[], bits=[], phase=[], index=[])
        for k in range(calls):
            r = o.service(iqs[c][2 * k * lens[c] : 2 * (k + 1) * lens[c]], 0.01, sriChanged=(k == 0))
            ref["soft"].append(r.soft); ref["bits"].append(r.bits); ref["phase"].append(r.phase); ref["index"].append(r.index)
        assert_parity({key: np.concatenate(v) for key, v in got[c].items()}, {key: np.concatenate(v) for key, v in ref.items()},
                      "pipelined S%d M%d ch%d" % (S, M, c))
    h.close()
