import sys, os, random
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from psk_soft_amd import lib as pl
from psk_soft_amd.stimulus import synth_channel
S, A, M, n, N = 16, 25, 2, 50, 12000 + 37 * 3
iq = synth_channel(1003, M, S, N)
seg = iq[:2 * 850]
x = seg[0::2].astype(np.float32) + 1j * seg[1::2].astype(np.float32)
e = (seg[0::2] * seg[0::2] + seg[1::2] * seg[1::2]).astype(np.float32)
nsym = 850 // S
E = e[: nsym * S].reshape(nsym, S).astype(np.float64)
W = np.array([E[i:i + A].sum(0) for i in range(nsym - A + 1)])
print("n_out", W.shape[0], "true argmax", W.argmax(1)[:12])
srt = np.sort(W, 1)
print("true best", srt[:8, -1], "margin", (srt[:, -1] - srt[:, -2])[:8])
h = pl.Handle(1, device=0)
h.configure(0, [dict(samplesPerBaud=S, constelationSize=M, numAvg=A, phaseAvg=n)])
g = h.process_host(0, [dict(data=seg, xdelta=0.01, sriChanged=True)])[0]
print("gpu idx", g["index"][:12])
print("gpu best", g["soft"][0::2][:8], "margin", g["phase"][:8])
print("gpu e_old[8]", g["soft"][1::2][:10])
print("true e(i-1)[8]", np.concatenate([[0], E[:9, 8]]))
print("bits", g["bits"][:12])
