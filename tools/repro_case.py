"""Replay one dumped fuzz case (PSK_FUZZ_DUMP) alone: tools/repro_case.py sig.npy S M A n "cut1,cut2,..." [tiled]
Prints where the four streams first differ from the oracle, per call, and the statistics of every call."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import pyoracle as po  # noqa: E402
from psk_soft_amd import lib as pl  # noqa: E402

sig = np.load(sys.argv[1])
S, M, A, n = (int(v) for v in sys.argv[2:6])
cuts = [int(v) for v in sys.argv[6].split(",")]
props = dict(samplesPerBaud=S, constelationSize=M, numAvg=A, phaseAvg=n)
h = pl.Handle(1, device=0, max_window_samples=33 * 1024 + 64, max_phase_avg=max(2048, n))
if len(sys.argv) > 7:
    h.set_option(pl.Handle.OPT_TIME_TILED, int(sys.argv[7]))
h.configure(0, [props])
o = po.OracleComponent()
for k, v in props.items():
    setattr(o, k, v)
base = 0
for k in range(len(cuts) - 1):
    x = sig[2 * cuts[k] : 2 * cuts[k + 1]]
    g = h.process_host(0, [dict(data=x, xdelta=0.01, sriChanged=False)])[0]
    r = o.service(x, 0.01, sriChanged=False)
    st = h.stats()
    line = "call %d: %d symbols (from %d)  tier: fast %d exact %d seq %d guard %d" % (
        k, r.index.size, base, st["channels_fast"], st["channels_exact_timing"], st["channels_sequential"], st["channels_guard"])
    for key, ref in (("index", r.index), ("phase", r.phase), ("soft", r.soft), ("bits", r.bits)):
        a, b = np.asarray(g[key]), np.asarray(ref)
        if a.dtype == np.float32:
            d = np.nonzero(a.view(np.uint32) != np.ascontiguousarray(b, np.float32).view(np.uint32))[0]
        else:
            d = np.nonzero(a != b)[0]
        if d.size:
            line += "  %s: %d differ, first at %d (%s vs %s)" % (key, d.size, d[0], a[d[0]], b[d[0]])
            if key == "index":
                line += " positions " + str(d[:24].tolist())
    print(line)
    base += r.index.size
