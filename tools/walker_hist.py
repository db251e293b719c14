"""Per-channel count of the blocks the parallel fit's walker ran itself (pf_xwalk, psk_pfit.h) in one time-tiled call of the
bench stimulus: the walker is one wave per channel, so its launch lasts as long as the channel with the most of them.
usage (GPU box): python tools/walker_hist.py [channels] [samples]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from psk_soft_amd import lib as pl  # noqa: E402
from psk_soft_amd.stimulus import synth_channels_torch  # noqa: E402

C = int(sys.argv[1]) if len(sys.argv) > 1 else 256
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 20
M, S = 4, 8
dev = torch.device("cuda:0")
iq = synth_channels_torch(C, M, S, N, dev, seed=0x5EED1000, periodic=True)
cap = (N // S + 2 + 63) // 64 * 64
soft = torch.empty((C, 2 * cap), dtype=torch.float32, device=dev)
phase = torch.empty((C, cap), dtype=torch.float32, device=dev)
sidx = torch.empty((C, cap), dtype=torch.int16, device=dev)
bits = torch.empty((C, 2 * cap), dtype=torch.int16, device=dev)
pk, out = (pl.Packet * C)(), (pl.Output * C)()
for c in range(C):
    pk[c].data, pk[c].n_floats, pk[c].sri_xdelta, pk[c].sri_mode, pk[c].present = iq[c].data_ptr(), 2 * N, 0.01, 1, 1
    out[c].soft, out[c].bits, out[c].phase, out[c].sampleIndex = soft[c].data_ptr(), bits[c].data_ptr(), phase[c].data_ptr(), sidx[c].data_ptr()
    out[c].cap_symbols = cap
torch.cuda.synchronize(dev)
h = pl.Handle(C, device=0)
h.set_option(pl.Handle.OPT_TIME_TILED, 1)

h.configure_all(samplesPerBaud=S, constelationSize=M, numAvg=100, phaseAvg=50)
for call in range(4):
    h.process_device(0, pk, out, stream=0)
    torch.cuda.synchronize(dev)
    cs = h.channel_stats()
    d = np.array([s["fit_chain_blocks"] for s in cs], np.int64)
    print("call %d: per-channel statistic  min %d  median %d  mean %.1f  p90 %d  max %d (channel %d)   stats %s" % (
        call, d.min(), np.median(d), d.mean(), np.percentile(d, 90), d.max(), int(d.argmax()),
        {k: h.stats()[k] for k in ("channels_parallel_fit", "parallel_fit_refusals")}))
order = np.argsort(d)
print("the ten largest (channel: value):", ", ".join("%d: %d" % (int(c), int(d[c])) for c in order[-10:]))
print("all:", d.tolist())
h.close()
