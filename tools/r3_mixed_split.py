"""Experiment: the two window classes of the mixed batch (configs[4]) as two handles on two streams, each fed its 2^18 samples
per step in P calls with no join between the classes -- how much would independent class streams with shorter work units buy?
(Timing only: P calls are P serviceFunction() calls, not one.)   usage (GPU box): python tools/r3_mixed_split.py
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from psk_soft_amd import lib as pl  # noqa: E402
from psk_soft_amd.stimulus import synth_channels_torch  # noqa: E402

dev = torch.device("cuda", 0)
S, N, C = 8, 1 << 18, 4096
props = [dict(samplesPerBaud=S, constelationSize=(2, 4, 8)[c % 3], phaseAvg=(10, 50, 200)[(c // 3) % 3], numAvg=(25, 100, 400)[(c // 9) % 3])
         for c in range(C)]
groups = [[c for c in range(C) if props[c]["numAvg"] == 400], [c for c in range(C) if props[c]["numAvg"] != 400]]
iq = torch.empty((C, 2 * N), dtype=torch.float32, device=dev)
for j, Mj in enumerate((2, 4, 8)):  # (every channel gets the constellation it is configured for, as in bench.py --mixed)
    idx = torch.arange(j, C, 3, device=dev)
    iq[idx] = synth_channels_torch(idx.numel(), Mj, S, N, dev, seed=1 + j, periodic=True)
cap = (N // S + 2 + 63) // 64 * 64
soft = torch.empty((C, 2 * cap), dtype=torch.float32, device=dev)
phase = torch.empty((C, cap), dtype=torch.float32, device=dev)
sidx = torch.empty((C, cap), dtype=torch.int16, device=dev)
bits = torch.empty((C, 3 * cap), dtype=torch.int16, device=dev)
torch.cuda.synchronize()


def build(chs, P):
    h = pl.Handle(len(chs), device=0, max_window_samples=16384, max_phase_avg=512)
    h.configure(0, [props[c] for c in chs])
    n = N // P
    calls = []
    for k in range(P):
        pk, out = (pl.Packet * len(chs))(), (pl.Output * len(chs))()
        for i, c in enumerate(chs):
            pk[i].data, pk[i].n_floats, pk[i].sri_xdelta, pk[i].sri_mode, pk[i].present = iq[c].data_ptr() + 8 * n * k, 2 * n, 0.01, 1, 1
            o = (n // S) * k
            out[i].soft, out[i].bits, out[i].phase, out[i].sampleIndex = soft[c].data_ptr() + 8 * o, bits[c].data_ptr() + 6 * o, phase[c].data_ptr() + 4 * o, sidx[c].data_ptr() + 2 * o
            out[i].cap_symbols = n // S + 2
        calls.append((pk, out))
    return h, calls


for P in (1, 2, 4, 8):
    lo, hi = torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else (0, -1)
    streams = [torch.cuda.Stream(device=dev, priority=-1), torch.cuda.Stream(device=dev, priority=0)]
    hs = [build(g, P) for g in groups]

    def step():
        for k in range(P):
            for (h, calls), st in zip(hs, streams):
                h.process_device(0, calls[k][0], calls[k][1], stream=st.cuda_stream)

    for _ in range(5):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 15
    for _ in range(reps):
        step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / reps * 1e3
    print("P = %d pieces per step: %.3f ms per step of %d channels x %d samples (%.1f %% of the read roofline)" % (P, ms, C, N, 8.0 * C * N / (ms * 1e-3) / 8e12 * 100))
    for h, _ in hs:
        st = h.stats()
        print("     ", {k: st[k] for k in ("channels_fast", "channels_exact_timing", "channels_sequential", "timing_exact_blocks", "fit_chain_blocks", "channels_tiled")})
        h.close()
