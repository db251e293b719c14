import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from psk_soft_amd import lib as pl
from psk_soft_amd.stimulus import synth_channels_torch
for C in (1, 4096):
    N, S, M = 1 << 18, 8, 4
    dev = torch.device("cuda", 0)
    iq = synth_channels_torch(C, M, S, N, dev, periodic=True)
    cap = (N // S + 2 + 63) // 64 * 64
    soft = torch.empty((C, 2 * cap), dtype=torch.float32, device=dev); phase = torch.empty((C, cap), dtype=torch.float32, device=dev)
    sidx = torch.empty((C, cap), dtype=torch.int16, device=dev); bits = torch.empty((C, 2 * cap), dtype=torch.int16, device=dev)
    pk = (pl.Packet * C)(); out = (pl.Output * C)()
    for c in range(C):
        pk[c].data = iq[c].data_ptr(); pk[c].n_floats = 2 * N; pk[c].sri_xdelta = 0.01; pk[c].sri_mode = 1; pk[c].present = 1
        out[c].soft = soft[c].data_ptr(); out[c].bits = bits[c].data_ptr(); out[c].phase = phase[c].data_ptr()
        out[c].sampleIndex = sidx[c].data_ptr(); out[c].cap_symbols = cap
    h = pl.Handle(C, device=0)
    h.configure_all(samplesPerBaud=S, constelationSize=M, numAvg=100, phaseAvg=50)
    h.set_force_sequential(1)
    h.process_device(0, pk, out); h.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        h.process_device(0, pk, out)
    h.synchronize()
    print("reference-order kernel, %d channel(s) x %d symbols: %.1f ms per call" % (C, N // S, (time.perf_counter() - t0) / 3 * 1e3))
    h.close()
