"""Where the fixed cost of one psk_soft_process_device call goes (small packets).

usage (GPU box): python tools/call_cost.py [nsamp] [channels]
Prints, per call of `channels` channels x `nsamp` complex samples (QPSK, 8 samples/baud):
  plan only      -- the host control plane alone (a DEVICE_NONE handle: plan_call per channel, no HIP)
  issue          -- CPU time of the call with the GPU kept busy (enqueue rate, no sync)
  steady         -- wall time per call over a long run of back-to-back calls (= max(issue, device))
  device         -- HIP-event time around the calls on their stream
"""
import ctypes
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from psk_soft_amd import lib as pl  # noqa: E402
from psk_soft_amd.stimulus import synth_channels_torch  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
C = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
S, M = 8, 4
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
    pk[c].data = iq[c].data_ptr()
    pk[c].n_floats = 2 * N
    pk[c].sri_xdelta = 0.01
    pk[c].sri_mode = 1
    pk[c].present = 1
    out[c].soft = soft[c].data_ptr()
    out[c].bits = bits[c].data_ptr()
    out[c].phase = phase[c].data_ptr()
    out[c].sampleIndex = sidx[c].data_ptr()
    out[c].cap_symbols = cap


def run(h, reps, stream=None):
    t0 = time.perf_counter()
    for _ in range(reps):
        h.process_device(0, pk, out, stream=stream)
    return (time.perf_counter() - t0) / reps * 1e6


hd = pl.Handle(C, device=pl.DEVICE_NONE)
hd.configure_all(samplesPerBaud=S, constelationSize=M, numAvg=100, phaseAvg=50)
run(hd, 50)
print("plan only : %7.1f us per call" % run(hd, 500))

h = pl.Handle(C, device=0)
h.configure_all(samplesPerBaud=S, constelationSize=M, numAvg=100, phaseAvg=50)
stream = torch.cuda.Stream(device=dev)
run(h, 200, stream.cuda_stream)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
reps = 2000
e0.record(stream)
t0 = time.perf_counter()
issue = run(h, reps, stream.cuda_stream)
e1.record(stream)
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / reps * 1e6
print("issue     : %7.1f us per call (CPU, no sync)" % issue)
# (a call waits for the plan slot it is about to reuse -- four slots --, so a long run of calls is throttled to the device's pace:
# the CPU's own cost of a call shows in bursts shorter than that, each behind a synchronize)
burst = []
for _ in range(300):
    torch.cuda.synchronize()
    burst.append(run(h, 3, stream.cuda_stream))
burst.sort()
print("issue     : %7.1f us per call (CPU, bursts of three calls behind a synchronize: median; min %.1f)" % (burst[len(burst) // 2], burst[0]))
print("steady    : %7.1f us per call (wall, %d calls back to back)" % (wall, reps))
print("device    : %7.1f us per call (HIP events)" % (e0.elapsed_time(e1) / reps * 1e3))
# the same batch fed in G slices on G streams: the tail of one slice's launch (its slowest waves) overlaps the body of the next's
for G in (2, 4):
    streams = [torch.cuda.Stream(device=dev) for _ in range(G)]
    step = C // G
    views = [((pl.Packet * step).from_address(ctypes.addressof(pk) + g * step * ctypes.sizeof(pl.Packet)),
              (pl.Output * step).from_address(ctypes.addressof(out) + g * step * ctypes.sizeof(pl.Output))) for g in range(G)]

    def run_sliced(reps):
        t0 = time.perf_counter()
        for _ in range(reps):
            for g in range(G):
                h.process_device(g * step, views[g][0], views[g][1], stream=streams[g].cuda_stream)
        return (time.perf_counter() - t0) / reps * 1e6

    torch.cuda.synchronize()
    run_sliced(200)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    issue_g = run_sliced(reps)
    torch.cuda.synchronize()
    wall_g = (time.perf_counter() - t0) / reps * 1e6
    print("%d slices on %d streams: issue %7.1f us, steady %7.1f us per %d-channel batch (%.1f %% of the 8 TB/s read roofline)"
          % (G, G, issue_g, wall_g, C, 8.0 * C * N / (wall_g * 1e-6) / 8e12 * 100))
print("stream rate: %.1f Gsamples/s, %.1f %% of the 8 TB/s read roofline" % (C * N / wall / 1e3, 8.0 * C * N / (wall * 1e-6) / 8e12 * 100))
