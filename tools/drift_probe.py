"""Does the headline step drift with what the GPU did before it?  One process, one set of buffers (4096 channels x 2^18 samples):
phase 1 measures 20 steps every 4 s with the GPU idle in between, phase 2 measures 20 steps back to back for 25 s, phase 3 is phase 1
again.  (GPU box) python tools/drift_probe.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from psk_soft_amd import lib as pl  # noqa: E402
from psk_soft_amd.stimulus import synth_channels_torch  # noqa: E402

C, N, M, S = 4096, 1 << 18, 4, 8
dev = torch.device("cuda:0")
t_start = time.time()
iq = synth_channels_torch(C, M, S, N, dev, seed=0x5EED0000, periodic=True)
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
stream = torch.cuda.Stream(device=dev)
stream.wait_stream(torch.cuda.current_stream(dev))
h = pl.Handle(C, device=0)
h.configure_all(samplesPerBaud=S, constelationSize=M, numAvg=100, phaseAvg=50)


def measure(steps=20):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(steps):
        h.process_device(0, pk, out, stream=stream.cuda_stream)
    e1.record(stream)
    torch.cuda.synchronize(dev)
    return e0.elapsed_time(e1) / steps


for _ in range(5):
    h.process_device(0, pk, out, stream=stream.cuda_stream)
torch.cuda.synchronize(dev)
if len(sys.argv) > 1 and sys.argv[1] == "gaps":
    # how long an idle gap it takes: a burst of 60 steps, a gap, then 5 untimed and 20 timed steps (the driver's bench.py call)
    for gap in (0.0, 0.01, 0.03, 0.1, 0.3, 1.0, 3.0, 0.0, 0.03, 0.3):
        measure(60)
        time.sleep(gap)
        measure(5)
        print("gap %5.2f s: the next 20 steps (behind 5 untimed ones) %.4f ms per step" % (gap, measure(20)), flush=True)
    h.close()
    sys.exit(0)
for name, reps, gap in (("idle between", 6, 4.0), ("back to back", 0, 0.0), ("idle between", 6, 4.0)):
    if reps:
        for _ in range(reps):
            print("t=%6.1f s  %-13s %.4f ms per step" % (time.time() - t_start, name, measure()), flush=True)
            time.sleep(gap)
    else:
        t0 = time.time()
        k = 0
        while time.time() - t0 < 25.0:
            ms = measure(40)
            if k % 25 == 0:
                print("t=%6.1f s  %-13s %.4f ms per step" % (time.time() - t_start, name, ms), flush=True)
            k += 1
h.close()
