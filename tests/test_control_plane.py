"""Host logic of the C-ABI library (no GPU): the control plane that mirrors the non-data
state of psk_soft_i (psk_soft_amd/csrc/psk_ctl.h) must agree, call by call, with the
oracle's counters -- output counts, pushSRI decisions and SRI xdeltas, warning counts,
samples.size(), index and yvals.size() -- over random sequences of property changes,
resets, queue flushes, real-data packets and ragged packet sizes (reference
cpp/psk_soft.cpp:353-426, 454-457, 568-590, 619-651)."""
import math
import random

import numpy as np
import pytest

from psk_soft_amd import lib as pl


def _mk(oracle_mod):
    h = pl.Handle(1, device=pl.DEVICE_NONE, max_window_samples=1 << 16, max_phase_avg=4096)
    o = oracle_mod.OracleComponent()
    return h, o


def _set(h, o, name, value):
    h.configure(0, [{name: value}])
    setattr(o, name, value)


def _same(a, b):
    return a == b or (isinstance(a, float) and isinstance(b, float) and math.isnan(a) and math.isnan(b))


@pytest.mark.parametrize("seed", range(12))
def test_random_call_sequences(oracle_mod, seed):
    rng = random.Random(seed)
    nrng = np.random.default_rng(seed)
    h, o = _mk(oracle_mod)
    S_choices = [1, 2, 3, 4, 5, 8, 10, 16, 7]
    A_choices = [0, 1, 2, 5, 20, 100]
    M_choices = [2, 4, 8, 16, 3, 1]
    n_choices = [1, 2, 5, 50, 200]
    xd_choices = [0.01, 1.0, 0.5, 1e-6]
    for call in range(120):
        # property traffic between calls
        for _ in range(rng.choice([0, 0, 0, 1, 2])):
            what = rng.choice(["S", "A", "M", "n", "diff", "reset"])
            if what == "S":
                _set(h, o, "samplesPerBaud", rng.choice(S_choices))
            elif what == "A":
                _set(h, o, "numAvg", rng.choice(A_choices))
            elif what == "M":
                _set(h, o, "constelationSize", rng.choice(M_choices))
            elif what == "n":
                _set(h, o, "phaseAvg", rng.choice(n_choices))
            elif what == "diff":
                _set(h, o, "differentialDecoding", rng.choice([0, 1]))
            else:
                _set(h, o, "resetState", 1)
        n_complex = rng.choice([0, 1, 3, 7, 64, 100, 777, 1000, 2500])
        odd = rng.random() < 0.1
        n_floats = 2 * n_complex + (1 if odd else 0)
        data = nrng.standard_normal(n_floats).astype(np.float32)
        xdelta = rng.choice(xd_choices) if rng.random() < 0.2 or call == 0 else xdelta  # noqa: F821
        mode = 0 if rng.random() < 0.05 else 1
        sri = call == 0 or rng.random() < 0.1
        flushed = rng.random() < 0.05
        ro = o.service(data, xdelta, mode=mode, sriChanged=sri, inputQueueFlushed=flushed)
        rg = h.plan_only(0, [dict(n_floats=n_floats, xdelta=xdelta, mode=mode, sriChanged=sri, inputQueueFlushed=flushed)])[0]
        ctx = "seed %d call %d" % (seed, call)
        assert rg["ret"] == ro.ret, ctx
        assert rg["n_symbols"] == ro.phase.size, ctx
        assert 2 * rg["n_symbols"] == ro.soft.size, ctx
        assert rg["n_bits"] == ro.bits.size, ctx
        assert rg["n_sampleIndex"] == ro.index.size, ctx
        assert rg["sri_pushed"] == ro.sri_pushed, ctx
        if ro.sri_pushed:
            assert _same(rg["sri_soft_xdelta"], ro.sri_soft_xdelta), ctx
            assert _same(rg["sri_bits_xdelta"], ro.sri_bits_xdelta), ctx
        assert rg["n_warn"] == ro.n_warn, ctx
        pk = h.peek(0)
        assert pk["ring_len"] == o.ring_size, ctx
        assert pk["index"] == o.index, ctx
        assert pk["fit_len"] == o.fit_history().size, ctx


def test_large_batch_equals_channel_by_channel():
    """A whole-handle batch is planned on a scratch copy of the channel states that is swapped in on
    success (a partial range is copied back); the results, the committed state, the refusal reported
    and the roll-back of a refused call must be those of a one-channel-at-a-time pass."""
    rng = random.Random(99)
    n_ch = 3000
    big = pl.Handle(n_ch, device=pl.DEVICE_NONE, max_window_samples=1 << 14, max_phase_avg=512)
    one = pl.Handle(n_ch, device=pl.DEVICE_NONE, max_window_samples=1 << 14, max_phase_avg=512)
    props = [dict(samplesPerBaud=rng.choice([1, 2, 5, 8, 10, 16]), constelationSize=rng.choice([2, 4, 8, 3]),
                  numAvg=rng.choice([0, 1, 25, 100, 400]), phaseAvg=rng.choice([1, 50, 200, 500]),
                  differentialDecoding=rng.choice([0, 1])) for _ in range(n_ch)]
    big.configure(0, props)
    one.configure(0, props)
    for call in range(6):
        pk = [None if rng.random() < 0.05 else
              dict(n_floats=2 * rng.choice([0, 3, 64, 777, 4096]) + (rng.random() < 0.05), xdelta=rng.choice([0.01, 0.5]),
                   sriChanged=(call == 0), inputQueueFlushed=(rng.random() < 0.02), mode=(0 if rng.random() < 0.02 else 1))
              for _ in range(n_ch)]
        if call == 3:  # a refused call: one channel in the middle asks for an unsupported property
            bad = 1777
            big.configure(bad, [dict(phaseAvg=0)])
            before = [big.peek(c) for c in (0, bad - 1, bad, bad + 1, n_ch - 1)]
            with pytest.raises(pl.PskSoftError) as ei:
                big.plan_only(0, pk)
            assert "channel %d " % bad in str(ei.value)
            assert [big.peek(c) for c in (0, bad - 1, bad, bad + 1, n_ch - 1)] == before
            big.configure(bad, [dict(phaseAvg=props[bad]["phaseAvg"])])
            one.configure(bad, [dict(phaseAvg=0)])  # same listener traffic on the other side
            one.configure(bad, [dict(phaseAvg=props[bad]["phaseAvg"])])
        got = big.plan_only(0, pk)
        ref = [one.plan_only(c, [pk[c]])[0] for c in range(n_ch)]
        assert got == ref, call
        assert [big.peek(c) for c in range(0, n_ch, 7)] == [one.peek(c) for c in range(0, n_ch, 7)]
    # a partial range of a large handle (no swap of the state arrays)
    pk = [dict(n_floats=2 * 1000, xdelta=0.01) for _ in range(1500)]
    got = big.plan_only(700, pk)
    ref = [one.plan_only(700 + c, [pk[c]])[0] for c in range(1500)]
    assert got == ref
    assert [big.peek(c) for c in range(0, n_ch, 11)] == [one.peek(c) for c in range(0, n_ch, 11)]
    big.close()
    one.close()


def test_stamped_batch_equals_channel_by_channel():
    """Channels configured alike and fed equal packets are planned ONCE and the plan stamped into the other slots
    (psk_capi.cpp, "the stamped path"), their control state kept in one copy until somebody looks at a single channel.
    Results, committed state, statistics and refusals must be those of a one-channel-at-a-time pass -- through every way in
    and out of the stamped path: a packet that differs, a channel reconfigured and set back, a partial range, a peek in
    between, a refused call."""
    rng = random.Random(4242)
    n_ch = 600
    for S, A, n, M in ((8, 100, 50, 4), (10, 400, 200, 8), (1, 0, 50, 2), (3, 25, 1, 3)):
        big = pl.Handle(n_ch, device=pl.DEVICE_NONE, max_window_samples=1 << 14, max_phase_avg=512)
        one = pl.Handle(n_ch, device=pl.DEVICE_NONE, max_window_samples=1 << 14, max_phase_avg=512)
        prop = dict(samplesPerBaud=S, constelationSize=M, numAvg=A, phaseAvg=n)
        big.configure_all(**prop)
        one.configure_all(**prop)

        def both(ch0, pk, ctx):
            got = big.plan_only(ch0, pk)
            ref = [one.plan_only(ch0 + c, [pk[c]])[0] for c in range(len(pk))]
            assert got == ref, ctx
            return got

        for call in range(14):
            nf = 2 * rng.choice([0, 3, 64, 777, 4096, 20000]) + (rng.random() < 0.1)
            base = dict(n_floats=nf, xdelta=rng.choice([0.01, 0.01, 0.5]), sriChanged=(call == 0 or rng.random() < 0.1),
                        inputQueueFlushed=(rng.random() < 0.1), mode=(0 if rng.random() < 0.05 and call != 11 else 1))
            pk = [dict(base) for _ in range(n_ch)]
            ctx = (S, A, call)
            if call == 4:  # one packet differs: the whole batch is planned channel by channel
                pk[311]["n_floats"] = nf + 16
            if call == 6:  # a channel reconfigured and set back: its listeners have fired, it is no longer like the others
                for hh in (big, one):
                    hh.configure(17, [dict(phaseAvg=7)])
                    hh.configure(17, [dict(phaseAvg=n)])
            if call == 8:  # a missing packet
                pk[5] = None
            if call == 10:  # a partial range, twice, then the whole handle again
                both(100, pk[100:400], ctx)
                both(100, pk[100:400], ctx)
            if call == 11:  # a refused call leaves everything as it was
                for hh in (big, one):
                    hh.configure(333, [dict(phaseAvg=0)])
                before = [big.peek(c) for c in (0, 332, 333, 334, n_ch - 1)]
                with pytest.raises(pl.PskSoftError) as ei:
                    big.plan_only(0, pk)
                assert "channel 333 " in str(ei.value)
                assert [big.peek(c) for c in (0, 332, 333, 334, n_ch - 1)] == before
                for hh in (big, one):
                    hh.configure(333, [dict(phaseAvg=n)])
            if call in (12, 13):  # the handle fed in three slices (each keeps its own stamped run), then whole again
                both(0, pk[:200], ctx)
                both(200, pk[200:450], ctx)
                both(450, pk[450:], ctx)
                both(0, pk[:200], ctx)
                both(450, pk[450:], ctx)
                both(200, pk[200:450], ctx)
                if call == 13:
                    both(100, pk[100:300], ctx)  # (a slice across two runs)
            both(0, pk, ctx)
            if call % 3 == 2:  # (a look at single channels brings the per-channel mirror up to date)
                assert [big.peek(c) for c in range(0, n_ch, 37)] == [one.peek(c) for c in range(0, n_ch, 37)], ctx
            if call % 5 == 4:
                sb, so = big.stats(), one.stats()
                assert (sb["channels_fast"], sb["channels_sequential"]) == (so["channels_fast"], so["channels_sequential"]), ctx
        assert [big.peek(c) for c in range(n_ch)] == [one.peek(c) for c in range(n_ch)]
        as_tuple = lambda q: tuple(getattr(q, k) for k, _ in q._fields_)
        assert [as_tuple(big.query(c)) for c in range(0, n_ch, 50)] == [as_tuple(one.query(c)) for c in range(0, n_ch, 50)]
        big.close()
        one.close()


def test_which_kernel_a_configuration_is_planned_for():
    """The wave-scan kernels have instantiations for samplesPerBaud 2 .. 32, numAvg <= 1024 (<= 512 for samplesPerBaud
    > 16); window classes beyond that are planned for the time-tiled kernels behind their run-time front stage
    (samplesPerBaud up to 1024, any numAvg) -- both count as `channels_fast` --, phaseAvg up to 32640; what is left is
    planned for the reference-order kernel (154 ms per 32768-symbol call): a regression in that
    routing is a latency cliff, not a wrong result, so no parity test would see it.  (samplesPerBaud 1 emits nothing at
    all unless numAvg is 0, quirk Q11: not part of this table.)"""
    cfgs = [(S, A, n) for S in list(range(2, 35)) + [40, 64] for A in (1, 100, 128, 129, 256, 257, 512, 513, 1024, 1025)
            for n in (50,)] + [(8, 100, n) for n in (1, 384, 385, 1920, 1921, 4000, 32640, 32641)] + [(24, 300, 1000), (16, 1024, 1920)]
    h = pl.Handle(len(cfgs), device=pl.DEVICE_NONE, max_window_samples=64 * 1025 + 64, max_phase_avg=40000)
    h.configure(0, [dict(samplesPerBaud=S, numAvg=A, phaseAvg=n) for S, A, n in cfgs])
    # two calls: the first fills the window (nothing out, any configuration), the second emits
    for k in range(2):
        h.plan_only(0, [dict(n_floats=2 * S * (A + 300), xdelta=0.01, sriChanged=(k == 0)) for S, A, n in cfgs])
    for i, (S, A, n) in enumerate(cfgs):
        one = pl.Handle(1, device=pl.DEVICE_NONE, max_window_samples=64 * 1025 + 64, max_phase_avg=40000)
        one.configure(0, [dict(samplesPerBaud=S, numAvg=A, phaseAvg=n)])
        for k in range(2):
            one.plan_only(0, [dict(n_floats=2 * S * (A + 300), xdelta=0.01, sriChanged=(k == 0))])
        st = one.stats()
        fast = 2 <= S <= 1024 and n <= 32640
        assert (st["channels_fast"], st["channels_sequential"]) == ((1, 0) if fast else (0, 1)), (S, A, n, st)
        one.close()
    st = h.stats()
    n_fast = sum(1 for S, A, n in cfgs if 2 <= S <= 1024 and n <= 32640)
    assert st["channels_fast"] == n_fast and st["channels_sequential"] == len(cfgs) - n_fast, st
    h.close()


def test_no_packet_is_noop(oracle_mod):
    h, _ = _mk(oracle_mod)
    r = h.plan_only(0, [None])[0]
    assert r["ret"] == pl.NOOP and r["n_symbols"] == 0


def test_unsupported_and_limits():
    h = pl.Handle(1, device=pl.DEVICE_NONE, max_window_samples=1000, max_phase_avg=64)
    with pytest.raises(pl.PskSoftError) as e:
        h.configure(0, [{"samplesPerBaud": 8, "numAvg": 200}])
    assert e.value.status == 4  # PSK_SOFT_ERR_LIMIT
    with pytest.raises(pl.PskSoftError):
        h.configure(0, [{"phaseAvg": 100}])
    h.configure(0, [{"samplesPerBaud": 0}])
    with pytest.raises(pl.PskSoftError) as e:
        h.plan_only(0, [dict(n_floats=16, xdelta=0.01)])
    assert e.value.status == 5  # PSK_SOFT_ERR_UNSUPPORTED (undefined in the reference)
    # a refused call leaves the channel untouched
    h.configure(0, [{"samplesPerBaud": 8, "numAvg": 100}])
    assert h.peek(0)["ring_len"] == 0


def test_abi_exports_every_declared_symbol():
    L = pl.load()
    for name in pl.EXPORTS:
        assert hasattr(L, name), name
    assert L.psk_soft_abi_version() == 2
    import os
    import re

    hdr = open(os.path.join(os.path.dirname(pl._HERE), "include", "psk_soft_hip.h")).read()
    declared = set(re.findall(r"\b(psk_soft_[a-z_]+)\s*\(", hdr))
    assert declared == set(pl.EXPORTS), declared ^ set(pl.EXPORTS)


def test_create_without_gpu_fails_loudly():
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(pl.PskSoftError) as e:
        pl.Handle(1, device=0)
    assert e.value.status == 2  # PSK_SOFT_ERR_NO_DEVICE: no silent CPU path


def test_options_and_pinned_allocation_without_gpu():
    """psk_soft_set_option is control-plane state (works on a control-plane-only handle, rejects
    unknown options); psk_soft_host_alloc needs the HIP runtime and must fail loudly without it."""
    import torch

    h = pl.Handle(1, device=pl.DEVICE_NONE) if hasattr(pl, "DEVICE_NONE") else pl.Handle(1, device=-1)
    h.set_option(h.OPT_QPSK_SIGN_BITMAP, 1)
    h.set_option(h.OPT_QPSK_SIGN_BITMAP, 0)
    h.set_option(h.OPT_DEFERRED_JOIN, 1)  # (a control-plane-only handle has no streams: the option is kept, join is a no-op)
    h.join()
    h.set_option(h.OPT_DEFERRED_JOIN, 0)
    with pytest.raises(pl.PskSoftError) as e:
        h.set_option(12345, 1)
    assert e.value.status == 1  # PSK_SOFT_ERR_INVALID_ARG
    h.close()
    if not torch.cuda.is_available():
        with pytest.raises(MemoryError):
            pl.host_alloc(1024, "float32")


def test_state_blob_is_validated_on_import():
    """psk_soft_import_state takes a blob of exactly psk_soft_state_bytes with a header (magic, version, the limits it
    was written under) and refuses control state that the kernels would use as out-of-range indices; a refused
    import leaves the channel as it was."""
    import ctypes

    from psk_soft_amd import lib as pl

    h = pl.Handle(2, device=pl.DEVICE_NONE)
    h.configure(0, [dict(samplesPerBaud=8, numAvg=100)] * 2)
    h.plan_only(0, [dict(n_floats=2 * 5000, xdelta=0.01, sriChanged=True)] * 2)
    blob = h.export_state(0)
    before = h.peek(1)
    h.import_state(1, blob)  # a good blob goes in
    assert h.peek(1) == h.peek(0)
    bad = bytearray(blob)
    bad[0] ^= 0xFF  # magic
    with pytest.raises(pl.PskSoftError):
        h.import_state(1, bytes(bad))
    with pytest.raises(pl.PskSoftError):
        h.import_state(1, blob[:-4])  # short
    with pytest.raises(pl.PskSoftError):
        h.import_state(1, blob + b"\0" * 8)  # long
    other = pl.Handle(1, device=pl.DEVICE_NONE, max_phase_avg=100)  # other limits: other array sizes
    with pytest.raises(pl.PskSoftError):
        other.import_state(0, blob)
    # an index field out of range (ring_src lives in the ChanCtl right behind the 24-byte header; find it by value)
    ctl_off = 24
    ctl_len = len(blob) - 24 - 72 - 8 * 16384 - 4 * 513
    hit = 0
    for off in range(ctl_off, ctl_off + ctl_len - 3, 4):
        b2 = bytearray(blob)
        b2[off : off + 4] = (0x7FFFFFF0).to_bytes(4, "little")
        h.import_state(1, blob)
        try:
            h.import_state(1, bytes(b2))
        except pl.PskSoftError:
            hit += 1
            assert h.peek(1) == h.peek(0)  # untouched by the refused import
    assert hit >= 3  # lf_head, ring_src, lf_n / lf_len ... are all range-checked
    assert before is not None
