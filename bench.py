#!/usr/bin/env python3
"""bench.py -- headline benchmark of the psk_soft hot path on MI355X.

Metric (BASELINE.json): complex IQ Msamples/s, QPSK, 8 samples/baud, 4096 channels per
GPU, inputs and outputs resident in HBM.  A "step" is one serviceFunction() body
(psk_soft_process_device) over one packet of every channel.  Weak scaling: every rank
(= GPU) owns its own 4096 channels -- channels are independent,
so there is no data-path collective; torch.distributed is used only for the barrier
and the max-over-ranks of the elapsed time.

    python bench.py --gpus N --steps K --warmup W

prints ONE JSON line on rank 0, with `roofline` (HBM read roofline of the wave-scan
kernel) and `cpu_baseline` (the CPU oracle timed on this node's host cores, N=1 only).

N > 1: one process per GPU.  Under a launcher (torchrun: WORLD_SIZE / RANK / LOCAL_RANK / MASTER_* in the
environment) this process is one rank.  Started plainly with --gpus N > 1 it is the launcher itself: before
anything touches a GPU it starts N fresh child processes of this script, one per device, with those variables
set, relays rank 0's JSON line and exits with the worst child status.
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E vendor peak, /opt/skills/guides/MI355X_MICROARCH.md


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=30,
                    help="untimed steps; the first ~30 launches of a process (75 ms) run up to 12 %% slower, "
                         "tools/launch_ramp.py and DESIGN.md section 3.1")
    ap.add_argument("--channels", type=int, default=4096, help="channels per GPU")
    ap.add_argument("--nsamp", type=int, default=1 << 18, help="complex samples per channel per step")
    ap.add_argument("--M", type=int, default=4)
    ap.add_argument("--S", type=int, default=8)
    ap.add_argument("--numAvg", type=int, default=100)
    ap.add_argument("--phaseAvg", type=int, default=50)
    ap.add_argument("--mixed", action="store_true",
                    help="BASELINE configs[4]: per-channel constelationSize {2,4,8}, phaseAvg {10,50,200}, numAvg {25,100,400}")
    ap.add_argument("--cfo", type=float, default=1e-3,
                    help="carrier offset bound of the stimulus: M*dphi per symbol uniform in [-cfo, cfo] rad (section 8(d): 1e-3)")
    ap.add_argument("--sigma", type=float, default=0.01, help="noise per component of the stimulus (section 8(d): 0.01)")
    ap.add_argument("--scale", type=float, default=1.0, help="amplitude factor applied to the whole stimulus")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=1.5, help="wall seconds of the CPU baseline sample")
    ap.add_argument("--check", dest="check", action="store_true", default=True,
                    help="replay two channels of the run through the oracle afterwards and compare all four streams (default)")
    ap.add_argument("--no-check", dest="check", action="store_false")
    ap.add_argument("--no-few", dest="no_few", action="store_true",
                    help="skip the one-channel measurement (BASELINE configs[1]) that the default run appends as `few_channels`")
    ap.add_argument("--no-extra", dest="no_extra", action="store_true",
                    help="skip the other configurations the default single-GPU run appends to its line (`configs3_per_gpu`, `mixed`, "
                         "`worst_case`, `few_channels_64` / `_512` / `_1024`, `end_to_end`), each measured like the headline with its own oracle check")
    ap.add_argument("--strong", type=int, default=0, metavar="TOTAL_CHANNELS",
                    help="strong scaling: TOTAL_CHANNELS channels shared by the ranks (BASELINE configs[3]: "
                         "--M 8 --S 10 --strong 32768) instead of --channels per rank")
    ap.add_argument("--serial-classes", action="store_true", help="--mixed: launch the window classes one after the other (A/B of PSK_SOFT_OPT_CONCURRENT_CLASSES)")
    ap.add_argument("--chain-hist", action="store_true", help="per-channel histogram of fit_chain_blocks of the last step (to stderr)")
    ap.add_argument("--phase0", action="store_true",
                    help="every channel at zero constellation phase and zero carrier offset (the signal shape of the reference's "
                         "own test, reference tests/test_psk_soft.py:98-117): LinearFit's sums hover around zero in EVERY channel, "
                         "the data-dependent worst case of the wave-scan kernel")
    ap.add_argument("--stamps", default="", metavar="FILE.npy",
                    help="(diagnostic builds, -DPSK_DIAG_STAMP) save every wave's start / end tick of the last step")
    return ap.parse_args()


def cpu_baseline(iq_host, M, S, A, n, wall_budget):
    """Time the CPU oracle (kind 'port': oracle/psk_soft_oracle.c, one psk_soft_i per
    channel) on all host cores over a bounded sample of the same workload."""
    from concurrent.futures import ThreadPoolExecutor

    from oracle import pyoracle as po

    po.build()
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = min(cores, int(os.environ.get("PSK_BENCH_CPU_THREADS", "16")))  # the 1-GPU box's CPU share is 16 cores
    n_ch, n_fl = iq_host.shape
    n_complex = n_fl // 2

    def work(tid):
        done = 0
        comps = {}
        t_end = time.perf_counter() + wall_budget
        c = tid
        while time.perf_counter() < t_end:
            ch = c % n_ch
            comp = comps.get(ch)
            if comp is None:
                comp = po.OracleComponent()
                comp.samplesPerBaud = S
                comp.constelationSize = M
                comp.numAvg = A
                comp.phaseAvg = n
                comps[ch] = comp
            comp.service(iq_host[ch], 0.01, sriChanged=(done == 0))
            done += n_complex
            c += cores
        return done

    # one untimed packet per thread to page everything in
    warm = po.OracleComponent()
    warm.samplesPerBaud = S
    warm.constelationSize = M
    warm.numAvg = A
    warm.service(iq_host[0], 0.01, sriChanged=True)
    t0 = time.perf_counter()
    with ThreadPoolExecutor(cores) as ex:
        total = sum(ex.map(work, range(cores)))
    dt = time.perf_counter() - t0
    # one thread, one channel, same packets, a fraction of the budget
    one = po.OracleComponent()
    one.samplesPerBaud = S
    one.constelationSize = M
    one.numAvg = A
    one.phaseAvg = n
    t1 = time.perf_counter()
    done1 = 0
    while time.perf_counter() - t1 < wall_budget / 3:
        one.service(iq_host[0], 0.01, sriChanged=(done1 == 0))
        done1 += n_complex
    dt1 = time.perf_counter() - t1
    # BASELINE configs[0] as specified: BPSK, 8 samples per baud, ONE channel, one
    # serviceFunction call over 2^20 complex samples, one thread (the oracle's serviceFunction-shaped component)
    from psk_soft_amd.stimulus import synth_channel

    x0 = synth_channel(0, 2, 8, 1 << 20)
    c0 = po.OracleComponent()
    c0.samplesPerBaud, c0.constelationSize, c0.numAvg, c0.phaseAvg = 8, 2, 100, 50
    c0.service(x0[: 2 * 65536], 0.01, sriChanged=True)  # (pages in, first-call resets done)
    t2 = time.perf_counter()
    r0 = c0.service(x0, 0.01, sriChanged=False)
    dt2 = time.perf_counter() - t2
    config0 = {"workload": "BPSK, samplesPerBaud=8, 1 channel, one call of 2^20 complex samples, one thread",
               "value": (1 << 20) / dt2 / 1e6, "unit": "complex IQ Msamples/s", "cores": 1, "kind": "port",
               "symbols_out": int(r0.phase.size)}
    return {
        "config0": config0,
        "single_thread_value": done1 / dt1 / 1e6,
        "value": total / dt / 1e6,
        "unit": "complex IQ Msamples/s",
        "cores": cores,
        "kind": "port",
        "sample": "%d-sample packets of %d distinct channels of the same workload, %d threads, %.1f s wall (%.0f Msamples)"
        % (n_complex, n_ch, cores, dt, total / 1e6),
    }


def dry_rank(a):
    """PSK_BENCH_DRY=1 (CPU rehearsal of the N > 1 path, tests/test_distributed_gloo.py): the same process group,
    barriers, sharding and reductions as a real run, control-plane-only handles (PSK_SOFT_DEVICE_NONE: every
    call is planned, nothing is computed), no GPU.  The line it prints is marked "dry": its rate means nothing."""
    import torch.distributed as dist

    from psk_soft_amd import lib as pl
    from psk_soft_amd.distributed import max_over_ranks, shard_channels, sum_over_ranks

    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    if world > 1:
        dist.init_process_group(os.environ.get("PSK_BENCH_BACKEND", "gloo"))
    C = shard_channels(a.strong, world, rank)[1] if a.strong else a.channels
    h = pl.Handle(C, device=pl.DEVICE_NONE)
    h.configure_all(samplesPerBaud=a.S, constelationSize=a.M, numAvg=a.numAvg, phaseAvg=a.phaseAvg)
    pk = [dict(n_floats=2 * a.nsamp, xdelta=0.01, sriChanged=False)] * C
    n_sym = 0
    for _ in range(a.warmup):
        h.plan_only(0, pk)
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        n_sym = sum(r["n_symbols"] for r in h.plan_only(0, pk))
    if world > 1:
        dist.barrier()
    elapsed = max_over_ranks(time.perf_counter() - t0, dist if world > 1 else None)
    total = int(sum_over_ranks(C * a.nsamp, dist if world > 1 else None))
    sym = int(sum_over_ranks(n_sym, dist if world > 1 else None))
    if rank == 0:
        print(json.dumps({"metric": "complex IQ Msamples/s, QPSK 8 sps, 4096 ch; % HBM roofline at 1/2/4/8 GPUs", "dry": True,
                          "value": total * a.steps / elapsed / 1e6, "unit": "complex IQ Msamples/s (planned only, not computed)",
                          "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "scaling": "strong" if a.strong else "weak",
                          "channels_total": total // a.nsamp, "symbols_per_step": sym}))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    h.close()


def few_channels(pl, torch, dev, dev_index, M, S, numAvg, phaseAvg, check, C=1, modes=((0, "one_wave_per_channel"), (1, "time_tiled"))):
    """BASELINE configs[1], one channel on one GPU, as one long call per step (2^20 complex samples): the calls the
    time-tiled kernels are for (psk_tile_kernel.h, psk_pfit.h), with the one-wave-per-channel kernels timed beside them
    (PSK_SOFT_OPT_TIME_TILED = 0) and, with --check, the last call compared with the oracle bit for bit."""
    from psk_soft_amd.stimulus import synth_channels_torch

    N = 1 << 20
    iq = synth_channels_torch(C, M, S, N, dev, seed=0x5EED1000, periodic=True)
    bpb = {2: 1, 4: 2, 8: 3}.get(M, 0)
    cap = (N // S + 2 + 63) // 64 * 64
    soft = torch.empty((C, 2 * cap), dtype=torch.float32, device=dev)
    phase = torch.empty((C, cap), dtype=torch.float32, device=dev)
    sidx = torch.empty((C, cap), dtype=torch.int16, device=dev)
    bits = torch.empty((C, max(bpb, 1) * cap), dtype=torch.int16, device=dev)
    pk, out = (pl.Packet * C)(), (pl.Output * C)()
    for c in range(C):
        pk[c].data, pk[c].n_floats, pk[c].sri_xdelta, pk[c].sri_mode, pk[c].present = iq[c].data_ptr(), 2 * N, 0.01, 1, 1
        out[c].soft, out[c].bits, out[c].phase, out[c].sampleIndex = soft[c].data_ptr(), bits[c].data_ptr(), phase[c].data_ptr(), sidx[c].data_ptr()
        out[c].cap_symbols = cap
    stream = torch.cuda.Stream(device=dev)
    stream.wait_stream(torch.cuda.current_stream(dev))  # (the stimulus is generated on torch's current stream)
    res, warm, steps = {}, 3, 10
    for mode, key in modes:
        h = pl.Handle(C, device=dev_index)
        h.set_option(pl.Handle.OPT_TIME_TILED, mode)
        h.configure_all(samplesPerBaud=S, constelationSize=M, numAvg=numAvg, phaseAvg=phaseAvg)
        for _ in range(warm):
            h.process_device(0, pk, out, stream=stream.cuda_stream)
        torch.cuda.synchronize(dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(steps):
            h.process_device(0, pk, out, stream=stream.cuda_stream)
        e1.record(stream)
        torch.cuda.synchronize(dev)
        ms = e0.elapsed_time(e1) / steps
        st = h.stats()
        res[key] = {"ms_per_call": ms, "Msamples_per_s": C * N / ms / 1e3, "frac_of_hbm_read_roofline": 8.0 * C * N / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                    "kernel_stats": {k: st[k] for k in ("channels_tiled", "channels_parallel_fit", "parallel_fit_refusals", "fit_chain_blocks")}}
        h.close()
    res["workload"] = "%s, samplesPerBaud=%d, %d channel%s x %d complex samples per call, numAvg=%d, phaseAvg=%d%s" % (
        {2: "BPSK", 4: "QPSK", 8: "8-PSK"}.get(M, "M=%d" % M), S, C, "" if C == 1 else "s", N, numAvg, phaseAvg, " (BASELINE configs[1])" if C == 1 else "")
    if "one_wave_per_channel" in res and "time_tiled" in res:
        res["speedup"] = res["one_wave_per_channel"]["ms_per_call"] / res["time_tiled"]["ms_per_call"]
    if check:
        import numpy as np

        from oracle import pyoracle as po

        comp = po.OracleComponent()
        comp.samplesPerBaud, comp.constelationSize, comp.numAvg, comp.phaseAvg = S, M, numAvg, phaseAvg
        x = iq[0].cpu().numpy()
        r = None
        for _ in range(warm + steps):
            r = comp.service(x, 0.01, sriChanged=False)
        n_out = int(out[0].n_symbols)
        same = all(np.array_equal(g.cpu().numpy().view(np.uint32 if g.dtype == torch.float32 else np.int16),
                                  np.ascontiguousarray(w, np.float32 if g.dtype == torch.float32 else np.int16).view(
                                      np.uint32 if g.dtype == torch.float32 else np.int16))
                   for g, w in ((soft[0, : 2 * n_out], r.soft), (phase[0, :n_out], r.phase), (bits[0, : bpb * n_out], r.bits),
                                (sidx[0, :n_out], r.index)))
        res["check"] = {"calls_replayed": warm + steps, "all_four_streams_bit_identical": bool(same)}
        assert same, "few_channels: the time-tiled path differs from the oracle"
    return res


def sub_bench(pl, torch, dev, dev_index, name, C=4096, N=1 << 18, M=4, S=8, numAvg=100, phaseAvg=50, mixed=False, phase0=False,
              steps=20, warmup=30, check=True, seed=0x5EED2000, then_headline=False):
    """One more configuration measured the way the headline is (inputs and outputs resident in HBM, HIP events on the
    launch stream, state carried across steps), with fewer steps, for the default line's `configs3_per_gpu`, `mixed`
    and `worst_case` entries; channels 0 and C-1 replayed through the oracle over all calls."""
    from psk_soft_amd.stimulus import synth_channels_torch

    if mixed:
        props = [dict(samplesPerBaud=S, constelationSize=(2, 4, 8)[c % 3], phaseAvg=(10, 50, 200)[(c // 3) % 3],
                      numAvg=(25, 100, 400)[(c // 9) % 3]) for c in range(C)]
        bpb_max = 3
    else:
        props = [dict(samplesPerBaud=S, constelationSize=M, numAvg=numAvg, phaseAvg=phaseAvg)] * C
        bpb_max = {2: 1, 4: 2, 8: 3}.get(M, 1)
    h = pl.Handle(C, device=dev_index, max_window_samples=max(16384, S * max(q["numAvg"] for q in props)),
                  max_phase_avg=max(512, max(q["phaseAvg"] for q in props)))
    if mixed:
        h.configure(0, props)
        iq = torch.empty((C, 2 * N), dtype=torch.float32, device=dev)
        for j, Mj in enumerate((2, 4, 8)):
            idx = torch.arange(j, C, 3, device=dev)
            iq[idx] = synth_channels_torch(idx.numel(), Mj, S, N, dev, seed=seed + j, periodic=True)
    else:
        h.configure_all(**props[0])
        iq = synth_channels_torch(C, M, S, N, dev, seed=seed, periodic=True, phase0=phase0)
    cap = (N // S + 2 + 63) // 64 * 64
    soft = torch.empty((C, 2 * cap), dtype=torch.float32, device=dev)
    phase = torch.empty((C, cap), dtype=torch.float32, device=dev)
    sidx = torch.empty((C, cap), dtype=torch.int16, device=dev)
    bits = torch.empty((C, bpb_max * cap), dtype=torch.int16, device=dev)
    pk, out = (pl.Packet * C)(), (pl.Output * C)()
    for c in range(C):
        pk[c].data, pk[c].n_floats, pk[c].sri_xdelta, pk[c].sri_mode, pk[c].present = iq[c].data_ptr(), 2 * N, 0.01, 1, 1
        out[c].soft, out[c].bits, out[c].phase, out[c].sampleIndex = soft[c].data_ptr(), bits[c].data_ptr(), phase[c].data_ptr(), sidx[c].data_ptr()
        out[c].cap_symbols = cap
    stream = torch.cuda.Stream(device=dev)
    stream.wait_stream(torch.cuda.current_stream(dev))  # (the stimulus is generated on torch's current stream)
    def timed(n_warm, n_steps):
        for _ in range(n_warm):
            h.process_device(0, pk, out, stream=stream.cuda_stream)
        torch.cuda.synchronize(dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(n_steps):
            h.process_device(0, pk, out, stream=stream.cuda_stream)
        h.join(stream.cuda_stream)  # (deferred join: the stream waits for the side streams of all the steps; else a no-op)
        e1.record(stream)
        torch.cuda.synchronize(dev)
        return e0.elapsed_time(e1) / n_steps

    joined_ms = None
    if mixed:
        # the mixed batch twice: every call joined into the caller's stream (the default), then with the join deferred to the
        # end of the timed steps (PSK_SOFT_OPT_DEFERRED_JOIN: the classes run ahead into the following calls on their own streams)
        joined_ms = timed(warmup, steps)
        h.set_option(pl.Handle.OPT_DEFERRED_JOIN, 1)
        ms = timed(5, steps)
        replay = warmup + steps + 5 + steps  # (calls the oracle has to replay below)
    else:
        ms = timed(warmup, steps)
        replay = warmup + steps
    st = h.stats()
    n_out = int(out[0].n_symbols)
    res = {"workload": name, "channels": C, "samples_per_channel_per_step": N, "ms_per_step": ms,
           "Msamples_per_s": C * N / ms / 1e3, "frac_of_hbm_read_roofline": 8.0 * C * N / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
           "steps": steps, "warmup": warmup,
           "kernel_stats": {k: st[k] for k in ("channels_fast", "channels_exact_timing", "channels_sequential", "unwrap_blocks",
                                               "unwrap_extra_passes", "timing_exact_blocks", "fit_chain_blocks")}}
    if joined_ms is not None:
        res["join"] = ("deferred (PSK_SOFT_OPT_DEFERRED_JOIN): the window classes end their calls on streams of their own and are "
                       "joined once, behind the last timed step, inside the timed region")
        res["ms_per_step_every_call_joined"] = joined_ms
        res["frac_of_hbm_read_roofline_every_call_joined"] = 8.0 * C * N / (joined_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
    if check:
        import numpy as np

        from oracle import pyoracle as po

        same = True
        for c in (0, C - 1):
            comp = po.OracleComponent()
            for kk, vv in props[c].items():
                setattr(comp, kk, vv)
            x = iq[c].cpu().numpy()
            r = None
            for _ in range(replay):
                r = comp.service(x, 0.01, sriChanged=False)
            b = {2: 1, 4: 2, 8: 3}.get(props[c]["constelationSize"], 0)
            for nm, g, w, dt in (("soft", soft[c, : 2 * n_out], r.soft, np.uint32), ("phase", phase[c, :n_out], r.phase, np.uint32),
                                 ("bits", bits[c, : b * n_out], r.bits, np.int16), ("sampleIndex", sidx[c, :n_out], r.index, np.int16)):
                ft = np.float32 if dt is np.uint32 else np.int16
                ga, wa = g.cpu().numpy().view(dt), np.ascontiguousarray(w, ft).view(dt)
                ok = ga.size == wa.size and np.array_equal(ga, wa)
                if not ok:  # (say where, before the assertion below ends the run)
                    d = np.nonzero(ga != wa)[0] if ga.size == wa.size else np.zeros(0, int)
                    sys.stderr.write("%s: channel %d, %s: %d of %d values differ (sizes %d / %d), first at %s\n"
                                     % (name[:40], c, nm, d.size, wa.size, ga.size, wa.size, d[:4].tolist()))
                same = same and ok
        res["check"] = {"channels": [0, C - 1], "calls_replayed": replay, "all_four_streams_bit_identical": bool(same)}
        assert same, "%s: the HIP path differs from the oracle" % name
    h.close()
    if then_headline:
        # the ordinary stimulus through the SAME input and output buffers (a fresh handle: fresh channel states): where the rows of
        # a run happen to fall in HBM moves the I/O floor of the geometry by several per cent, and a ratio
        # of two configurations should not carry that
        iq.copy_(synth_channels_torch(C, M, S, N, dev, seed=seed, periodic=True, phase0=False))
        torch.cuda.synchronize(dev)
        h = pl.Handle(C, device=dev_index, max_window_samples=max(16384, S * numAvg), max_phase_avg=max(512, phaseAvg))
        h.configure_all(**props[0])
        res["headline_ms_per_step_on_the_same_buffers"] = timed(warmup, steps)
        h.close()
    del iq, soft, phase, sidx, bits
    torch.cuda.empty_cache()
    return res


def end_to_end(pl, torch, dev, dev_index, C=4096, N=32768, M=4, S=8, calls=3):
    """"end-to-end incl. H2D/D2H separately" -- the same call with packets and results in HOST memory:
    psk_soft_process_host (pageable buffers, staged through pinned memory in chunks, three chunks in flight) and the
    zero-copy form (page-locked buffers from psk_soft_host_alloc handed to psk_soft_process_device: the kernels read and
    write them over PCIe).  Never `value`: these rates are the link's and the host's, not the kernel's."""
    import numpy as np

    from psk_soft_amd.stimulus import synth_channels_torch

    iq = synth_channels_torch(C, M, S, N, dev, seed=0x5EED3000, periodic=True).cpu().numpy()
    res = {"workload": "QPSK, samplesPerBaud=%d, %d channels x %d complex samples per call, packets and results in host memory" % (S, C, N)}
    h = pl.Handle(C, device=dev_index)
    h.configure_all(samplesPerBaud=S, constelationSize=M, numAvg=100, phaseAvg=50)
    pkts = [dict(data=iq[c], xdelta=0.01) for c in range(C)]
    h.process_host(0, pkts)
    t0 = time.perf_counter()
    for _ in range(calls):
        h.process_host(0, pkts)
    dt = (time.perf_counter() - t0) / calls
    h.close()
    in_b = 8.0 * C * N
    out_b = C * (N // S) * (8 + 4 + 2 + 4)
    res["process_host"] = {"ms_per_call": dt * 1e3, "Msamples_per_s": C * N / dt / 1e6, "pcie_in_GBps": in_b / dt / 1e9,
                           "pcie_out_GBps": out_b / dt / 1e9, "note": "pageable numpy buffers; includes the ctypes / numpy marshalling of this harness"}
    # zero copy: page-locked packets and result rows, device-pointer entry point
    cap = (N // S + 2 + 63) // 64 * 64
    h = pl.Handle(C, device=dev_index)
    h.configure_all(samplesPerBaud=S, constelationSize=M, numAvg=100, phaseAvg=50)
    hin = pl.host_alloc(C * 2 * N, np.float32)
    hsoft, hphase = pl.host_alloc(C * 2 * cap, np.float32), pl.host_alloc(C * cap, np.float32)
    hbits, hsidx = pl.host_alloc(C * 2 * cap, np.int16), pl.host_alloc(C * cap, np.int16)
    hin.reshape(C, 2 * N)[:] = iq
    pk, out = (pl.Packet * C)(), (pl.Output * C)()
    for c in range(C):
        pk[c].data, pk[c].n_floats, pk[c].sri_xdelta, pk[c].sri_mode, pk[c].present = hin.ctypes.data + 8 * N * c, 2 * N, 0.01, 1, 1
        out[c].soft, out[c].phase = hsoft.ctypes.data + 8 * cap * c, hphase.ctypes.data + 4 * cap * c
        out[c].bits, out[c].sampleIndex = hbits.ctypes.data + 4 * cap * c, hsidx.ctypes.data + 2 * cap * c
        out[c].cap_symbols = cap
    h.process_device(0, pk, out)
    h.synchronize()
    t0 = time.perf_counter()
    for _ in range(calls):
        h.process_device(0, pk, out)
    h.synchronize()
    dt = (time.perf_counter() - t0) / calls
    res["zero_copy"] = {"ms_per_call": dt * 1e3, "Msamples_per_s": C * N / dt / 1e6, "pcie_in_GBps": in_b / dt / 1e9,
                        "pcie_out_GBps": out_b / dt / 1e9, "note": "page-locked buffers (psk_soft_host_alloc) read and written by the kernels over PCIe"}
    h.close()
    for a in (hin, hsoft, hphase, hbits, hsidx):
        pl.host_free(a)
    return res


def cpu_model():
    try:
        for l in open("/proc/cpuinfo"):
            if l.startswith("model name"):
                return l.split(":", 1)[1].strip()
    except OSError:
        pass
    return None


def count_gpus_without_hip():
    """GPUs this process could use, WITHOUT initialising HIP in it (the launcher's children each open their own device; a
    parent that had opened the GPU and then forked would hand them a runtime in an undefined state).  The kernel driver's
    topology lists every agent; GPU nodes are those with SIMDs.  HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES restrict."""
    import glob

    n = 0
    for prop in glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties"):
        try:
            kv = dict(l.split(None, 1) for l in open(prop).read().splitlines() if " " in l)
        except OSError:
            continue
        if int(kv.get("simd_count", "0")) > 0:
            n += 1
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None and v.strip() != "":
            n = min(n, len([x for x in v.split(",") if x.strip() != ""])) if n else len([x for x in v.split(",") if x.strip() != ""])
    return n


def launch_ranks(a):
    """--gpus N > 1 without a launcher: start N ranks of this script, one per GPU.  This process never touches a GPU (it
    does not even import torch); it polls its children, and the first one that fails takes the others down with it."""
    import socket
    import subprocess

    n = a.gpus
    if os.environ.get("PSK_BENCH_DRY") != "1":
        visible = count_gpus_without_hip()
        if n > visible and os.environ.get("PSK_BENCH_SHARE_GPU") != "1":  # (rehearsal: several gloo ranks on one GPU)
            raise SystemExit("bench.py: --gpus %d but only %d device(s) visible" % (n, visible))
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, WORLD_SIZE=str(n), RANK=str(r), LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    # rank 0's stdout is read by a thread (a full pipe must not block it); the ranks are polled: the first failure, or the
    # overall limit, ends the others -- a rank stuck in a barrier behind a dead peer would otherwise wait out the backend's timeout
    import threading

    out0 = []
    reader = threading.Thread(target=lambda: out0.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    deadline = time.time() + float(os.environ.get("PSK_BENCH_LAUNCH_TIMEOUT", "1500"))
    rc = 0
    while True:
        codes = [p.poll() for p in procs]
        bad = [c for c in codes if c not in (None, 0)]
        if bad or time.time() > deadline:
            rc = bad[0] if bad else 124
            for p in procs:
                if p.poll() is None:
                    p.terminate()
            for p in procs:
                try:
                    p.wait(timeout=10)
                except subprocess.TimeoutExpired:
                    p.kill()
            break
        if all(c == 0 for c in codes):
            break
        time.sleep(0.05)
    reader.join(timeout=10)
    text = (out0[0] if out0 else b"").decode()
    # rank 0's JSON line and nothing else on stdout (what libraries print there -- gloo announces its peers -- goes to stderr)
    line = [l for l in text.splitlines() if l.startswith("{")]
    for l in text.splitlines():
        if not line or l is not line[-1]:
            print(l, file=sys.stderr)
    if line:
        sys.stdout.write(line[-1] + "\n")
    sys.stdout.flush()
    if rc:
        raise SystemExit("bench.py: a rank failed or the launch timed out (exit status %d)" % rc)
    if not line or json.loads(line[-1]).get("n_gpus") != n:
        raise SystemExit("bench.py: rank 0 did not report %d ranks" % n)


def main():
    a = parse()
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        return launch_ranks(a)
    if os.environ.get("PSK_BENCH_DRY") == "1":
        return dry_rank(a)
    import torch

    from psk_soft_amd import lib as pl
    from psk_soft_amd.stimulus import synth_channels_torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product path has no CPU fallback")
    # one process per GPU; PSK_BENCH_BACKEND=gloo lets several ranks share one GPU (rehearsal only)
    backend = os.environ.get("PSK_BENCH_BACKEND", "nccl")
    dev_index = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    red_dev = dev if backend == "nccl" else torch.device("cpu")
    use_dist = world > 1 or os.environ.get("PSK_BENCH_FORCE_DIST") == "1"  # (the switch: one rank through the RCCL calls)
    if use_dist:
        import torch.distributed as dist

        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    if world != a.gpus and "WORLD_SIZE" in os.environ and a.gpus != 1:
        raise SystemExit("bench.py: --gpus %d but the launcher started %d ranks" % (a.gpus, world))
    C, N, S, M = a.channels, a.nsamp, a.S, a.M
    if a.strong:  # a fixed total, shared: this rank's contiguous block (psk_soft_amd/distributed.py)
        from psk_soft_amd.distributed import shard_channels

        C = shard_channels(a.strong, world, rank)[1]
    bpb = {2: 1, 4: 2, 8: 3}.get(M, 0)
    if a.mixed:
        chan_props = [dict(samplesPerBaud=S, constelationSize=(2, 4, 8)[c % 3], phaseAvg=(10, 50, 200)[(c // 3) % 3],
                           numAvg=(25, 100, 400)[(c // 9) % 3]) for c in range(C)]
        bpb = 3
    else:
        chan_props = [dict(samplesPerBaud=S, constelationSize=M, numAvg=a.numAvg, phaseAvg=a.phaseAvg)] * C
    h = pl.Handle(C, device=dev_index, max_window_samples=max(16384, S * max(p["numAvg"] for p in chan_props)),
                  max_phase_avg=max(512, max(p["phaseAvg"] for p in chan_props)))
    if a.serial_classes:
        h.set_option(pl.Handle.OPT_CONCURRENT_CLASSES, 0)
    if a.mixed:
        h.configure(0, chan_props)
    else:
        h.configure_all(**chan_props[0])

    # synthetic section-8(d) workload, generated in HBM; every rank draws its own channels
    if a.mixed:
        iq = torch.empty((C, 2 * N), dtype=torch.float32, device=dev)
        for j, Mj in enumerate((2, 4, 8)):
            idx = torch.arange(j, C, 3, device=dev)
            iq[idx] = synth_channels_torch(idx.numel(), Mj, S, N, dev, seed=0x5EED0000 + rank * 3 + j, cfo_max=a.cfo, sigma=a.sigma, periodic=True)
    else:
        iq = synth_channels_torch(C, M, S, N, dev, seed=0x5EED0000 + rank, cfo_max=a.cfo, sigma=a.sigma, periodic=True, phase0=a.phase0)
    if a.scale != 1.0:
        iq *= a.scale
    # output rows start on 128-byte boundaries in all four streams (64 symbols of the narrowest one):
    # rows that straddle cache lines cost 8 % of the streaming ceiling (tools/micro/placement_probe.hip)
    cap = (N // S + 2 + 63) // 64 * 64
    if os.environ.get("PSK_BENCH_UNALIGNED_ROWS"):  # A/B switch for that measurement
        cap = N // S + 2
    soft = torch.empty((C, 2 * cap), dtype=torch.float32, device=dev)
    phase = torch.empty((C, cap), dtype=torch.float32, device=dev)
    sidx = torch.empty((C, cap), dtype=torch.int16, device=dev)
    bits = torch.empty((C, max(bpb, 1) * cap), dtype=torch.int16, device=dev)
    pk = (pl.Packet * C)()
    out = (pl.Output * C)()
    for c in range(C):
        pk[c].data = iq[c].data_ptr()
        pk[c].n_floats = 2 * N
        pk[c].sri_xdelta = 0.01
        pk[c].sri_mode = 1
        pk[c].sriChanged = 0
        pk[c].present = 1
        out[c].soft = soft[c].data_ptr()
        out[c].bits = bits[c].data_ptr()
        out[c].phase = phase[c].data_ptr()
        out[c].sampleIndex = sidx[c].data_ptr()
        out[c].cap_symbols = cap
    # a dedicated (non-null) stream: the library is handed its raw hipStream_t, and the HIP
    # events below are recorded on that same stream
    stream = torch.cuda.Stream(device=dev)
    stream.wait_stream(torch.cuda.current_stream(dev))

    def step():
        h.process_device(0, pk, out, stream=stream.cuda_stream)

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(a.warmup):
        step()
    barrier()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.steps)]
    t0 = time.perf_counter()
    for k in range(a.steps):
        ev[k][0].record(stream)
        step()
        ev[k][1].record(stream)
    barrier()
    elapsed = time.perf_counter() - t0
    if use_dist:
        from psk_soft_amd.distributed import max_over_ranks

        elapsed = max_over_ranks(elapsed, dist, red_dev)
    # device time of one launch set (plan upload + wave-scan kernel + reference-order kernel),
    # HIP events on the stream the kernels run on
    dev_ms = sorted(e0.elapsed_time(e1) for e0, e1 in ev)
    dev_ms_avg = sum(dev_ms) / len(dev_ms)
    # the box's empirical read ceiling over the same input buffer: 16-byte
    # loads and nothing else, so that "% of peak" and "% of achievable" can both be stated
    probe_ms = h.probe_read_ms(iq.data_ptr(), iq.numel() * 4, reps=5)
    st = h.stats()
    n_out = int(out[0].n_symbols)
    if a.chain_hist and rank == 0:
        import collections

        cs = [c["fit_chain_blocks"] for c in h.channel_stats()]
        hist = collections.Counter(min(v // 16, 16) for v in cs)
        sys.stderr.write("chain blocks per channel (bins of 16): %s; max %d; channels over 128: %s\n"
                         % (sorted(hist.items()), max(cs), [i for i, v in enumerate(cs) if v > 128][:20]))

    if a.stamps and rank == 0:
        import struct

        import numpy as np

        st_words = []
        for c in range(C):
            blob = h.export_state(c)
            ctl_bytes = struct.unpack("6I", blob[:24])[4]
            w = struct.unpack("20I", blob[24 + ctl_bytes: 24 + ctl_bytes + 80])
            st_words.append((w[19], w[18], w[16], w[17]))  # start tick (or HW_ID), end tick (100 MHz), chained blocks, (XCC_ID)
        np.save(a.stamps, np.array(st_words, dtype=np.int64))

    if use_dist:  # (strong scaling: ranks may own one channel more or less)
        from psk_soft_amd.distributed import sum_over_ranks

        samples_per_step = int(sum_over_ranks(C * N, dist, red_dev))
    else:
        samples_per_step = C * N
    value = samples_per_step * a.steps / elapsed / 1e6
    alg_read_bytes = 8.0 * C * N  # 8 B per complex input sample
    alg_write_bytes = C * n_out * (8 + 4 + 2 + 2 * bpb)
    achieved = alg_read_bytes / (dev_ms_avg * 1e-3) / 1e9
    traffic = None
    tf = os.path.join(ROOT, "profiles", "traffic_latest.json")
    # (the PMC traffic figure was collected on the headline configuration only)
    headline = (M, S, C, N, a.numAvg, a.phaseAvg, a.mixed) == (4, 8, 4096, 1 << 18, 100, 50, False)
    if headline and os.path.exists(tf):
        try:
            traffic = json.load(open(tf)).get("hbm_bytes_per_launch")
        except Exception:
            traffic = None
    res = {
        "metric": "complex IQ Msamples/s, QPSK 8 sps, 4096 ch; % HBM roofline at 1/2/4/8 GPUs",
        "value": value,
        "unit": "complex IQ Msamples/s",
        "n_gpus": world,
        "steps": a.steps,
        "warmup": a.warmup,
        "ms_per_step": elapsed / a.steps * 1e3,
        "higher_is_better": True,
        "scaling": "strong" if a.strong else "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": ("mixed BPSK/QPSK/8-PSK, samplesPerBaud=%d, %d channels per GPU x %d complex samples per step, "
                         "per-channel phaseAvg {10,50,200} and numAvg {25,100,400}, inputs/outputs resident in HBM" % (S, C, N))
            if a.mixed else
            "%s, samplesPerBaud=%d, %d channels per GPU x %d complex samples per step, numAvg=%d, phaseAvg=%d, "
            "inputs/outputs resident in HBM%s" % ({2: "BPSK", 4: "QPSK", 8: "8-PSK"}.get(M, "M=%d" % M), S, C, N, a.numAvg, a.phaseAvg,
                                                 ", every channel at zero constellation phase and zero carrier offset" if a.phase0 else ""),
            "channels_per_gpu": C,
            "samples_per_channel_per_step": N,
            "symbols_out_per_channel_per_step": n_out,
            "parallelism": "channels sharded, %d per GPU, no collective" % C,
        },
        "roofline": {
            "bound": "hbm",
            "achieved": achieved,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS,
            "traffic": traffic,
            "traffic_source": "profiles/traffic_latest.json: PMC counters (2*FETCH_SIZE + WRITE_SIZE) of a profiled run of this "
                              "configuration, a constant of the build, not a measurement of this run" if traffic else None,
            "kernel": "psk_fast_kernel<%d>" % S,
            "algorithmic_read_bytes_per_launch": alg_read_bytes,
            "algorithmic_write_bytes_per_launch": alg_write_bytes,
            "launch_ms_avg": dev_ms_avg,
            "launch_ms_min": dev_ms[0],
            "empirical_read_ceiling": alg_read_bytes / (probe_ms * 1e-3) / 1e9,
            "frac_of_empirical_read_ceiling": probe_ms / dev_ms_avg,
        },
        "kernel_stats": st,
    }

    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        n_host = min(C, 64)
        iq_host = iq[:n_host].cpu().numpy()
        res["cpu_baseline"] = cpu_baseline(iq_host, M, S, a.numAvg, a.phaseAvg, a.cpu_seconds)
        res["cpu_baseline"]["gpu_over_cpu"] = value / res["cpu_baseline"]["value"]
        res["cpu_baseline"]["cpu_model"] = cpu_model()

    if a.check and rank == 0:
        from oracle import pyoracle as po

        # replay channel 0 and C-1 through the oracle with the same packetisation
        import numpy as np

        worst, same = 0.0, True
        for c in (0, C - 1):
            comp = po.OracleComponent()
            for kk, vv in chan_props[c].items():
                setattr(comp, kk, vv)
            x = iq[c].cpu().numpy()
            r = None
            for k in range(a.warmup + a.steps):
                r = comp.service(x, 0.01, sriChanged=False)
            gs = soft[c, : 2 * n_out].cpu().numpy()
            gp = phase[c, :n_out].cpu().numpy()
            gb = bits[c, : {2: 1, 4: 2, 8: 3}.get(chan_props[c]["constelationSize"], 0) * n_out].cpu().numpy()
            gi = sidx[c, :n_out].cpu().numpy()
            assert np.array_equal(gb, r.bits) and np.array_equal(gi, r.index), "bit/index mismatch on channel %d" % c
            worst = max(worst, float(np.abs(gs - r.soft).max() / np.abs(r.soft).max()))
            same = same and np.array_equal(gs.view(np.uint32), np.ascontiguousarray(r.soft, np.float32).view(np.uint32)) \
                and np.array_equal(gp.view(np.uint32), np.ascontiguousarray(r.phase, np.float32).view(np.uint32))
        res["check"] = {"channels": [0, C - 1], "calls_replayed": a.warmup + a.steps, "bits_index_exact": True,
                        "soft_max_rel_err": worst, "soft_phase_bit_identical": bool(same)}

    if rank == 0 and world == 1 and not a.mixed and not a.no_few:
        res["few_channels"] = few_channels(pl, torch, dev, dev_index, M, S, a.numAvg, a.phaseAvg, a.check)

    if rank == 0 and world == 1 and headline and not a.phase0 and not a.no_extra:
        # the other BASELINE configurations and the two ends of SURVEY section 8(d), measured beside the headline in the same run
        h.close()
        del iq, soft, phase, sidx, bits
        torch.cuda.empty_cache()
        res["configs3_per_gpu"] = sub_bench(pl, torch, dev, dev_index, "8-PSK, samplesPerBaud=10, 4096 channels x 262144 complex samples per step "
                                            "(BASELINE configs[3]: one GPU's shard of the 32768 channels)", M=8, S=10, check=a.check)
        res["mixed"] = sub_bench(pl, torch, dev, dev_index, "mixed BPSK/QPSK/8-PSK, samplesPerBaud=8, 4096 channels x 262144 complex samples per step, "
                                 "per-channel phaseAvg {10,50,200} and numAvg {25,100,400} (BASELINE configs[4])", mixed=True, check=a.check)
        wc = sub_bench(pl, torch, dev, dev_index, "QPSK, samplesPerBaud=8, 4096 channels x 262144 complex samples per step, EVERY channel at zero "
                       "constellation phase and zero carrier offset (the signal shape of the reference's own test): LinearFit's sums hover "
                       "around zero in every channel, every block takes the reference-order chain", phase0=True, check=a.check, then_headline=True)
        # (against the headline configuration measured again right behind it, the same way and on the box as warm as it is by
        # now: the headline of this line ran first, on the cold box, and the boxes drift by several per cent as they warm up)
        again = sub_bench(pl, torch, dev, dev_index, "headline configuration, again", check=False)
        wc["headline_ms_per_step_measured_alongside"] = again["ms_per_step"]
        wc["slowdown_vs_headline"] = wc["ms_per_step"] / again["ms_per_step"]
        wc["slowdown_vs_headline_of_this_line"] = wc["ms_per_step"] / dev_ms_avg
        wc["slowdown_vs_headline_on_the_same_buffers"] = wc["ms_per_step"] / wc["headline_ms_per_step_on_the_same_buffers"]
        res["worst_case"] = wc
        if not a.no_few:
            for cf in (64, 512, 1024):
                res["few_channels_%d" % cf] = few_channels(pl, torch, dev, dev_index, M, S, a.numAvg, a.phaseAvg, a.check, C=cf,
                                                           modes=((1, "time_tiled"),))
        res["end_to_end"] = end_to_end(pl, torch, dev, dev_index)
        h = None

    if rank == 0:
        print(json.dumps(res))
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    if h is not None:
        h.close()


if __name__ == "__main__":
    main()
