"""Per-launch duration of the headline configuration over a long run of back-to-back steps (diagnostic:
how long the first-launches ramp of a fresh process lasts, and what it is).

usage (GPU box): python tools/launch_ramp.py [steps] [probe passes] [zero|other]
  probe passes: that many passes of the pure-read probe kernel right before the run (a busy GPU, another kernel)
  zero:         the output rows written once by another kernel before the run
  other:        ten launches of the same kernels on other channel state and other output rows before the run
Measured: the first ten launches run 10-12 % slower; neither a busy GPU nor touched output pages change that,
ten launches of the same kernels elsewhere remove it (2.60 -> 2.37 ms for steps 0-9): it is the GPU settling
on this instruction mix, i.e. warm-up proper."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from psk_soft_amd import lib as pl  # noqa: E402
from psk_soft_amd.stimulus import synth_channels_torch  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
C, N, S, M = 4096, 1 << 18, 8, 4
dev = torch.device("cuda", 0)
iq = synth_channels_torch(C, M, S, N, dev, periodic=True)
cap = (N // S + 2 + 63) // 64 * 64
soft = torch.empty((C, 2 * cap), dtype=torch.float32, device=dev)
phase = torch.empty((C, cap), dtype=torch.float32, device=dev)
sidx = torch.empty((C, cap), dtype=torch.int16, device=dev)
bits = torch.empty((C, 2 * cap), dtype=torch.int16, device=dev)
pk = (pl.Packet * C)()
out = (pl.Output * C)()
for c in range(C):
    pk[c].data = iq[c].data_ptr(); pk[c].n_floats = 2 * N; pk[c].sri_xdelta = 0.01; pk[c].sri_mode = 1; pk[c].present = 1
    out[c].soft = soft[c].data_ptr(); out[c].bits = bits[c].data_ptr(); out[c].phase = phase[c].data_ptr()
    out[c].sampleIndex = sidx[c].data_ptr(); out[c].cap_symbols = cap
h = pl.Handle(C, device=0)
h.configure_all(samplesPerBaud=S, constelationSize=M, numAvg=100, phaseAvg=50)
stream = torch.cuda.Stream(device=dev)
pre = int(sys.argv[2]) if len(sys.argv) > 2 else 0  # passes of the pure-read probe right before (a busy GPU, another kernel)
if pre:
    print("probe pass before the run: %.3f ms" % h.probe_read_ms(iq.data_ptr(), iq.numel() * 4, reps=pre))
mode = sys.argv[3] if len(sys.argv) > 3 else ""
if mode == "zero":  # the output rows written once by another kernel before the run
    for t in (soft, phase, sidx, bits):
        t.zero_()
if mode == "other":  # ten launches of the same kernels on OTHER channels' state and other output rows first
    h2 = pl.Handle(C, device=0)
    h2.configure_all(samplesPerBaud=S, constelationSize=M, numAvg=100, phaseAvg=50)
    soft2, phase2, sidx2, bits2 = (torch.empty_like(t) for t in (soft, phase, sidx, bits))
    out2 = (pl.Output * C)()
    for c in range(C):
        out2[c].soft = soft2[c].data_ptr(); out2[c].bits = bits2[c].data_ptr(); out2[c].phase = phase2[c].data_ptr()
        out2[c].sampleIndex = sidx2[c].data_ptr(); out2[c].cap_symbols = cap
    for _ in range(10):
        h2.process_device(0, pk, out2, stream=stream.cuda_stream)
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
ev[0].record(stream)
for k in range(steps):
    h.process_device(0, pk, out, stream=stream.cuda_stream)
    ev[k + 1].record(stream)
torch.cuda.synchronize()
ms = [ev[k].elapsed_time(ev[k + 1]) for k in range(steps)]
for a in range(0, steps, 10):
    print("steps %3d-%3d: mean %.3f ms" % (a, a + 9, sum(ms[a:a + 10]) / len(ms[a:a + 10])))
