"""Extreme amplitudes: at which scale does the HIP path part from the oracle?  (GPU box)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import pyoracle as po  # noqa: E402
from psk_soft_amd import lib as pl  # noqa: E402
from psk_soft_amd.stimulus import synth_channel  # noqa: E402

S, M, A, n = 8, 4, 100, 50
base = synth_channel(5, M, S, 40000).astype(np.float64)
for e in (-24, -23, -22, -21, -20, -19.5, -19, -18.5, -18, -15, 9, 15, 17, 18, 18.5, 19, 19.2, 19.4):
    iq = (base * 10.0 ** e).astype(np.float32)
    props = dict(samplesPerBaud=S, constelationSize=M, numAvg=A, phaseAvg=n)
    h = pl.Handle(1, device=0)
    h.configure(0, [props])
    o = po.OracleComponent()
    for k, v in props.items():
        setattr(o, k, v)
    g = h.process_host(0, [dict(data=iq, xdelta=0.01, sriChanged=True)])[0]
    r = o.service(iq, 0.01, sriChanged=True)
    st = h.stats()
    msg = []
    for key, ref in (("index", r.index), ("phase", r.phase), ("soft", r.soft), ("bits", r.bits)):
        a, b = np.asarray(g[key]), np.asarray(ref)
        if a.dtype == np.float32:
            av, bv = a.view(np.uint32), np.ascontiguousarray(b, np.float32).view(np.uint32)
        else:
            av, bv = a, b
        d = np.nonzero(av != bv)[0]
        if d.size:
            msg.append("%s %d/%d first %d (%s vs %s)" % (key, d.size, a.size, d[0], a[d[0]], b[d[0]]))
    print("amp 1e%s: tier fast %d exact %d seq %d exact-blocks %d  %s" % (e, st["channels_fast"], st["channels_exact_timing"], st["channels_sequential"], st["timing_exact_blocks"], "; ".join(msg) or "identical"))
    h.close()
