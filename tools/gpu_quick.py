"""Debug driver (GPU box): runs parity cases of the HIP path against the oracle and prints a table."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import pyoracle as po  # noqa: E402  (checker only)
from psk_soft_amd import lib as pl  # noqa: E402
from psk_soft_amd.stimulus import synth_channel  # noqa: E402


def run_gpu(h, ch, iq, xdelta, packet):
    n = iq.size // 2
    step = n if not packet else packet
    outs = {"soft": [], "bits": [], "phase": [], "index": []}
    pos = 0
    first = True
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


def compare(a, b):
    res = {}
    for k in ("bits", "index"):
        res[k] = "n%d/%d" % (a[k].size, b[k].size) if a[k].size != b[k].size else int((a[k] != b[k]).sum())
    for k in ("soft", "phase"):
        if a[k].size != b[k].size:
            res[k] = "n%d/%d" % (a[k].size, b[k].size)
            continue
        fa, fb = a[k].astype(np.float64), b[k].astype(np.float64)
        fin = np.isfinite(fb)
        if not np.array_equal(np.isfinite(fa), fin):
            res[k] = "finite-mismatch"
            continue
        den = np.abs(fb[fin]).max() if fin.any() else 1.0
        res[k] = float(np.abs(fa[fin] - fb[fin]).max() / (den if den else 1.0)) if fin.any() else 0.0
        res[k + "_exact"] = float((a[k][fin] == b[k][fin]).mean()) if fin.any() else 1.0
    return res


def main():
    cases = []
    for M in (4, 2, 8):
        for S in (8, 10):
            for diff in (0, 1):
                cases.append(dict(M=M, S=S, diff=diff, N=1 << 15, packet=None))
    cases += [dict(M=4, S=8, diff=0, N=1 << 15, packet=1000), dict(M=4, S=8, diff=0, N=5000, packet=7),
              dict(M=8, S=10, diff=0, N=1 << 15, packet=4096), dict(M=4, S=8, diff=0, N=1 << 18, packet=None),
              dict(M=2, S=4, diff=0, N=1 << 14, packet=3000), dict(M=4, S=16, diff=0, N=1 << 14, packet=None),
              dict(M=4, S=5, diff=0, N=1 << 14, packet=None), dict(M=4, S=2, diff=0, N=1 << 14, packet=None)]
    h = pl.Handle(2, device=0)
    ok = True
    for force in (0, 1):
        for ci, c in enumerate(cases):
            if force and c["N"] > (1 << 15):
                continue
            iq = synth_channel(ci, c["M"], c["S"], c["N"])
            o = po.OracleComponent()
            o.samplesPerBaud = c["S"]; o.constelationSize = c["M"]; o.numAvg = 100; o.differentialDecoding = c["diff"]
            ref = po.run_stream(o, iq, 0.01, packet_complex=c["packet"])
            ch = 0
            hh = pl.Handle(1, device=0)
            hh.set_force_sequential(force)
            hh.configure(0, [dict(samplesPerBaud=c["S"], constelationSize=c["M"], numAvg=100, differentialDecoding=c["diff"])])
            t = time.time()
            got = run_gpu(hh, ch, iq, 0.01, c["packet"])
            dt = time.time() - t
            st = hh.stats()
            r = compare(got, ref)
            bad = (r["bits"] != 0) or (r["index"] != 0) or not isinstance(r["soft"], float) or r["soft"] > 1e-5 or not isinstance(r["phase"], float) or r["phase"] > 1e-5
            ok = ok and not bad
            print("force_seq=%d %s -> %s stats=%s %.3fs %s" % (force, c, r, st, dt, "FAIL" if bad else "ok"), flush=True)
            hh.close()
    print("ALL OK" if ok else "SOME FAILED")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
