"""phaseAvg == 1 with a non-finite sample: where does the HIP path part from the oracle?  (GPU box)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import pyoracle as po  # noqa: E402
from psk_soft_amd import lib as pl  # noqa: E402
from psk_soft_amd.stimulus import synth_channel  # noqa: E402

for S, M, A, n, bad, val in ((4, 8, 100, 1, 700, "nan"), (4, 8, 100, 1, 700, "inf"), (8, 4, 100, 1, 3001, "nan"), (8, 4, 100, 2, 3001, "nan"), (8, 4, 100, 1, 3000, "-inf")):
    iq = synth_channel(5, M, S, 6000).copy()
    iq[bad] = np.float32(val)
    props = dict(samplesPerBaud=S, constelationSize=M, numAvg=A, phaseAvg=n)
    h = pl.Handle(1, device=0)
    h.configure(0, [props])
    o = po.OracleComponent()
    for k, v in props.items():
        setattr(o, k, v)
    g = h.process_host(0, [dict(data=iq, xdelta=0.01, sriChanged=True)])[0]
    r = o.service(iq, 0.01, sriChanged=True)
    st = h.stats()
    print("S%d M%d A%d n%d %s at float %d: tier fast %d exact %d seq %d" % (S, M, A, n, val, bad, st["channels_fast"], st["channels_exact_timing"], st["channels_sequential"]))
    for key, ref in (("index", r.index), ("phase", r.phase), ("soft", r.soft), ("bits", r.bits)):
        a, b = np.asarray(g[key]), np.asarray(ref)
        if a.dtype == np.float32:
            av, bv = a.view(np.uint32), np.ascontiguousarray(b, np.float32).view(np.uint32)
        else:
            av, bv = a, b
        d = np.nonzero(av != bv)[0]
        if d.size:
            i = d[0]
            print("   %s: %d of %d differ, first at %d: got %s ref %s (bits %s / %s); around: got %s ref %s" % (key, d.size, a.size, i, a[i], b[i], hex(int(av[i])), hex(int(bv[i])), a[max(0, i - 2): i + 3], b[max(0, i - 2): i + 3]))
    h.close()
