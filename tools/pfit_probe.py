"""One channel (or a few), one long call, through the time-tiled path: statistics of the call and parity with the oracle.
usage (GPU box): python tools/pfit_probe.py [channels] [nsamp] [sigma]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import pyoracle as po  # noqa: E402
from psk_soft_amd import lib as pl  # noqa: E402
from psk_soft_amd.stimulus import synth_channel  # noqa: E402

C = int(sys.argv[1]) if len(sys.argv) > 1 else 1
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 20
sigma = float(sys.argv[3]) if len(sys.argv) > 3 else 0.01
h = pl.Handle(C, device=0)
h.set_option(pl.Handle.OPT_TIME_TILED, 2)
props = dict(samplesPerBaud=8, constelationSize=4, numAvg=100, phaseAvg=50)
h.configure(0, [props] * C)
iqs = [synth_channel(100 + c, 4, 8, 2 * N, sigma=sigma) for c in range(C)]
for call in range(2):
    res = h.process_host(0, [dict(data=iq[2 * call * N : 2 * (call + 1) * N], xdelta=0.01, sriChanged=(call == 0)) for iq in iqs])
    print("call", call, h.stats())
    per = h.channel_stats(0, C)
    print("   fit_chain_blocks per channel (slow blocks of the walker where the parallel fit ran):", [p["fit_chain_blocks"] for p in per][:16])
bad = 0
for c in range(min(C, 4)):
    o = po.OracleComponent()
    for k, v in props.items():
        setattr(o, k, v)
    o.service(iqs[c][: 2 * N], 0.01, sriChanged=True)
    r = o.service(iqs[c][2 * N :], 0.01, sriChanged=False)
    g = res[c]
    for key, ref in (("soft", r.soft), ("phase", r.phase), ("bits", r.bits), ("index", r.index)):
        a, b = np.asarray(g[key]), np.asarray(ref)
        same = a.size == b.size and (np.array_equal(a.view(np.uint32), b.view(np.uint32)) if a.dtype == np.float32 else np.array_equal(a, b))
        if not same:
            bad += 1
            print("channel", c, key, "DIFFERS")
print("mismatching streams:", bad)
