"""Debug: one channel configuration, packet by packet, against the oracle (GPU box)."""
import sys, os, random
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import pyoracle as po
from psk_soft_amd import lib as pl
from psk_soft_amd.stimulus import synth_channel

S, A, M, n, N = 16, 25, 2, 50, 12000 + 37 * 3
iq = synth_channel(1003, M, S, N)
rng = random.Random(5)
for c in range(4):
    cuts = [0] + sorted(rng.sample(range(1, (12000 + 37 * c)), 3)) + [12000 + 37 * c]
print("cuts", cuts)
for force in (0, 1):
    h = pl.Handle(1, device=0)
    h.set_force_sequential(force)
    props = dict(samplesPerBaud=S, constelationSize=M, numAvg=A, phaseAvg=n)
    h.configure(0, [props])
    o = po.OracleComponent()
    for k, v in props.items():
        setattr(o, k, v)
    for k in range(4):
        seg = iq[2 * cuts[k]: 2 * cuts[k + 1]]
        r = o.service(seg, 0.01, sriChanged=(k == 0))
        g = h.process_host(0, [dict(data=seg, xdelta=0.01, sriChanged=(k == 0))])[0]
        st = h.stats()
        bad = np.nonzero(g["index"] != r.index)[0]
        print("force", force, "call", k, "n", r.index.size, "idx mismatches", bad.size, bad[:5], "stats", {a: b for a, b in st.items() if b})
        if bad.size:
            print("   got", g["index"][bad[0] - 2: bad[0] + 6], "ref", r.index[bad[0] - 2: bad[0] + 6])
