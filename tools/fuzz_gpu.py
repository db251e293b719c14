"""Randomised GPU-vs-oracle comparison (diagnostic; the fixed-seed cases that came out of it live in
tests/test_gpu_parity.py).  Every round draws a batch of channels with random properties, signal
shapes and packetisations, random property changes / resets between calls, runs it through the C ABI
and through the oracle, and reports every channel whose four output streams do not match
(bits / sampleIndex exactly; soft / phase exactly too -- PSK_FUZZ_STRICT=0 relaxes that to 1e-5 relative).

usage (GPU box): python tools/fuzz_gpu.py [rounds] [channels] [seed]
"""
import os
import random
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import pyoracle as po  # noqa: E402
from psk_soft_amd import lib as pl  # noqa: E402

NONFINITE = float(os.environ.get("PSK_FUZZ_NONFINITE", "0"))
EXTREME = float(os.environ.get("PSK_FUZZ_EXTREME", "0"))
# PSK_FUZZ_M: constellation sizes to draw from (default 2, 4, 4, 8; others -- 1, 3, 16 -- produce no bits, cpp/psk_soft.cpp:384-390);
# PSK_FUZZ_XD: SRI.xdelta values to draw from, one per channel (default 0.01 for all: LinearFit::xdelta = (float)(1 / sampleRate))
M_CHOICES = [int(v) for v in os.environ["PSK_FUZZ_M"].split(",")] if os.environ.get("PSK_FUZZ_M") else [2, 4, 4, 8]
XD_CHOICES = [float(v) for v in os.environ["PSK_FUZZ_XD"].split(",")] if os.environ.get("PSK_FUZZ_XD") else [0.01]
# PSK_FUZZ_MORE=1: two more kinds of events in the scripts -- samplesPerBaud changed between two calls, and a new SRI (another
# xdelta, sriChanged set) in the middle of a stream (off by default: the draws of the seeds quoted in DESIGN.md stay what they were)
MORE = os.environ.get("PSK_FUZZ_MORE", "0") != "0"
TOL = 1e-5
STRICT = os.environ.get("PSK_FUZZ_STRICT", "1") != "0"  # every float of soft / phase must equal the oracle's
XD = 0.01
# PSK_FUZZ_S / PSK_FUZZ_A / PSK_FUZZ_N: comma-separated lists that replace the default draws (to aim a run at some instantiations);
# PSK_FUZZ_WINDOW / PSK_FUZZ_PACKET: the handle's max_window_samples (samplesPerBaud * numAvg; the scripts set numAvg up to 300)
# and max_packet_complex (a stream is up to 12000 symbols long), for samplesPerBaud beyond 64
S_CHOICES = [int(v) for v in os.environ["PSK_FUZZ_S"].split(",")] if os.environ.get("PSK_FUZZ_S") else (
    [2, 4, 5, 8, 8, 8, 10, 10, 16, 3, 7, 1, 6, 9, 11, 12, 13, 14, 15, 33] + list(range(17, 33)))
A_CHOICES = [int(v) for v in os.environ["PSK_FUZZ_A"].split(",")] if os.environ.get("PSK_FUZZ_A") else (
    [1, 2, 3, 17, 25, 64, 100, 100, 127, 128, 129, 200, 256, 257, 400, 512, 520, 513, 1024])


N_CHOICES = [int(v) for v in os.environ["PSK_FUZZ_N"].split(",")] if os.environ.get("PSK_FUZZ_N") else (
    [1, 2, 3, 10, 50, 50, 128, 200, 384, 385, 400, 900, 1920, 1921])  # phaseAvg


def make_signal(rng, nrng, M, S, n):
    n_sym = n // S + 2
    k = nrng.integers(0, M, n_sym)
    kind = rng.choice(["shaped", "shaped", "rect", "tri"])
    j = np.arange(S)
    if kind == "shaped":
        pulse = 0.2 + 0.8 * np.sin(np.pi * (j + rng.uniform(0.1, 1.5)) / (S + rng.uniform(0.5, 2.0)))
    elif kind == "rect":
        pulse = np.ones(S)
    else:
        pulse = 1.0 - np.abs(j - rng.uniform(0, S - 1)) / S
    amp = 10.0 ** rng.uniform(-3.5, 2.5)
    if EXTREME and rng.random() < EXTREME:  # (PSK_FUZZ_EXTREME=p: energies that overflow or vanish in float)
        amp = 10.0 ** rng.choice([rng.uniform(17.0, 19.5), rng.uniform(-24.0, -18.0), rng.uniform(9.0, 17.0)])
    cfo = rng.choice([0.0, 1e-3, 1e-2, 0.2]) * rng.uniform(-1, 1) / M
    ph = 2 * np.pi * k / M + rng.uniform(0, 2 * np.pi)
    x = np.repeat(np.exp(1j * ph), S) * np.tile(pulse, n_sym)
    x = x[:n] * np.exp(1j * cfo * np.arange(n) / S) * amp
    sigma = rng.choice([0.0, 0.003, 0.03, 0.3]) * amp
    x = x + sigma * (nrng.standard_normal(n) + 1j * nrng.standard_normal(n))
    if rng.random() < 0.05:  # a silent stretch
        a = rng.randrange(0, n)
        x[a : a + rng.randrange(1, 4000)] = 0
    out = np.empty(2 * n, np.float32)
    out[0::2] = x.real
    out[1::2] = x.imag
    # PSK_FUZZ_NONFINITE=p: with probability p a stream gets a few NaN / +-inf components (off by default: the streams of the
    # seeds quoted in DESIGN.md stay what they were)
    if NONFINITE and rng.random() < NONFINITE:
        for _ in range(rng.randrange(1, 5)):
            out[rng.randrange(0, 2 * n)] = rng.choice([np.float32("nan"), np.float32("inf"), -np.float32("inf")])
    return out


def close(a, b):
    a = a.astype(np.float64)
    b = b.astype(np.float64)
    if a.size != b.size:
        return False, "size %d vs %d" % (a.size, b.size)
    fin = np.isfinite(b)
    if not np.array_equal(np.isfinite(a), fin):
        return False, "non-finite pattern"
    if fin.any():
        err = np.abs(a[fin] - b[fin]).max() / max(np.abs(b[fin]).max(), 1e-30)
        if err > TOL:
            return False, "rel err %g" % err
        if STRICT and not np.array_equal(a[fin], b[fin]):  # (float32 values widened: equal doubles = equal floats, +-0 aside)
            return False, "bits differ in %d values (rel err %g)" % (int((a[fin] != b[fin]).sum()), err)
    return True, ""


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    C = int(sys.argv[2]) if len(sys.argv) > 2 else 192
    seed = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    only = int(sys.argv[4]) if len(sys.argv) > 4 else None  # replay one round (each round draws from its own generator)
    bad_total = 0
    for rnd in range(rounds):
        if only is not None and rnd != only:
            continue
        rng = random.Random(seed * 1000 + rnd)
        nrng = np.random.default_rng(seed * 1000 + rnd)
        props, sigs, scripts = [], [], []
        for c in range(C):
            S = rng.choice(S_CHOICES)
            A = rng.choice(A_CHOICES)
            M = rng.choice(M_CHOICES)
            n = rng.choice(N_CHOICES)
            p = dict(samplesPerBaud=S, constelationSize=M, numAvg=A, phaseAvg=n, differentialDecoding=int(rng.random() < 0.25))
            N = max(S * rng.choice([50, 300, 1200, 3000, 12000]), 64)
            sig = make_signal(rng, nrng, M, max(S, 1), N)
            # script: a list of events; cuts with occasional tiny / empty packets, property changes, resets
            n_calls = rng.choice([1, 2, 3, 5])
            cuts = sorted(rng.sample(range(1, N), min(n_calls - 1, N - 1))) if n_calls > 1 else []
            ev, prev = [], 0
            for cut in cuts + [N]:
                if rng.random() < 0.15:
                    key = rng.choice(["phaseAvg", "numAvg", "constelationSize", "resetState", "differentialDecoding"])
                    val = {"phaseAvg": rng.choice([5, 50, 300]), "numAvg": rng.choice([10, 100, 300]),
                           "constelationSize": rng.choice([2, 4, 8]), "resetState": 1,
                           "differentialDecoding": rng.choice([0, 1])}[key]
                    ev.append(("set", key, val))
                new_xd = None
                if MORE:
                    if rng.random() < 0.08:
                        ev.append(("set", "samplesPerBaud", rng.choice([v for v in S_CHOICES if v * 300 <= 33 * 1024] or S_CHOICES)))
                    if rng.random() < 0.1:
                        new_xd = rng.choice([0.01, 0.02, 1e-3, 0.5, 2.5e-7])
                ev.append(("packet", prev, cut, rng.random() < 0.05, new_xd))
                prev = cut
            props.append(p)
            sigs.append(sig)
            scripts.append(ev)
            if os.environ.get("PSK_FUZZ_DUMP") and int(os.environ["PSK_FUZZ_DUMP"]) == c:  # with a single-round replay: keep this channel's case
                np.save(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "fuzz_case_sig.npy"), sig)
                print("DUMPED channel %d: props=%s script=%s" % (c, p, ev))
        h = pl.Handle(C, device=0, max_window_samples=int(os.environ.get("PSK_FUZZ_WINDOW", 33 * 1024 + 64)), max_phase_avg=max(2048, max(N_CHOICES)),
                      max_packet_complex=int(os.environ.get("PSK_FUZZ_PACKET", 1 << 20)))
        h.configure(0, props)
        oracles = []
        for c in range(C):
            o = po.OracleComponent()
            for k, v in props[c].items():
                setattr(o, k, v)
            oracles.append(o)
        got = [dict(soft=[], bits=[], phase=[], index=[]) for _ in range(C)]
        ref = [dict(soft=[], bits=[], phase=[], index=[]) for _ in range(C)]
        pos = [0] * C
        first = [True] * C
        xds = [XD_CHOICES[(rnd * 7 + c) % len(XD_CHOICES)] for c in range(C)]
        # walk the scripts in lock step: one batched call per "tick"
        while any(pos[c] < len(scripts[c]) for c in range(C)):
            pk = []
            for c in range(C):
                if pos[c] >= len(scripts[c]):
                    pk.append(None)
                    continue
                ev = scripts[c][pos[c]]
                while ev[0] == "set":
                    h.configure(c, [{ev[1]: ev[2]}])
                    setattr(oracles[c], ev[1], ev[2])
                    pos[c] += 1
                    ev = scripts[c][pos[c]]
                a, b, flushed = ev[1], ev[2], ev[3]
                data = sigs[c][2 * a : 2 * b]
                sri = first[c]
                if len(ev) > 4 and ev[4] is not None:  # (PSK_FUZZ_MORE: a new SRI in front of this packet)
                    xds[c] = ev[4]
                    sri = True
                pk.append(dict(data=data, xdelta=xds[c], sriChanged=sri, inputQueueFlushed=flushed))
                r = oracles[c].service(data, xds[c], sriChanged=sri, inputQueueFlushed=flushed)
                for k, v in (("soft", r.soft), ("bits", r.bits), ("phase", r.phase), ("index", r.index)):
                    ref[c][k].append(v)
                first[c] = False
                pos[c] += 1
            res = h.process_host(0, pk)
            for c in range(C):
                if pk[c] is not None:
                    for k in got[c]:
                        got[c][k].append(res[c][k])
        st = h.stats()
        h.close()
        bad = 0
        for c in range(C):
            g = {k: np.concatenate(v) if v else np.zeros(0) for k, v in got[c].items()}
            r = {k: np.concatenate(v) if v else np.zeros(0) for k, v in ref[c].items()}
            why = None
            if not np.array_equal(g["index"], r["index"]):
                why = "sampleIndex"
            elif not np.array_equal(g["bits"], r["bits"]):
                why = "bits (%d differ)" % int((g["bits"] != r["bits"]).sum()) if g["bits"].size == r["bits"].size else "bits size"
            else:
                for k in ("soft", "phase"):
                    ok, msg = close(g[k], r[k])
                    if not ok:
                        why = k + " " + msg
                        break
            if why:
                bad += 1
                print("MISMATCH round %d channel %d: %s  props=%s script=%s" % (rnd, c, why, props[c], scripts[c]))
                if g["phase"].size == r["phase"].size and g["soft"].size == r["soft"].size and g["phase"].size:
                    # where, and what the phase estimate is there: one ulp of a large estimate is the known case
                    dp = np.nonzero(g["phase"] != r["phase"])[0]
                    ds = np.nonzero(g["soft"] != r["soft"])[0]
                    print("   phase: %d of %d values differ%s; soft: %d floats differ; max |phase| %.1f"
                          % (dp.size, g["phase"].size,
                             "" if not dp.size else " (first at %d: %.9g vs %.9g, %d ulp)" % (
                                 dp[0], g["phase"][dp[0]], r["phase"][dp[0]],
                                 abs(int(g["phase"][dp[0]:dp[0] + 1].view(np.int32)[0]) - int(r["phase"][dp[0]:dp[0] + 1].view(np.int32)[0]))),
                             ds.size, float(np.abs(r["phase"][np.isfinite(r["phase"])]).max())))
                    if ds.size:
                        sym = np.unique(ds // 2)
                        print("   soft differs at symbols %s ... (%d symbols); got %s ref %s" % (
                            sym[:16].tolist(), sym.size, g["soft"][2 * sym[0] : 2 * sym[0] + 2], r["soft"][2 * sym[0] : 2 * sym[0] + 2]))
        bad_total += bad
        print("round %d: %d channels, %d mismatches, last-call stats %s" % (rnd, C, bad, st))
    print("TOTAL mismatches:", bad_total)
    return 1 if bad_total else 0


if __name__ == "__main__":
    sys.exit(main())
