"""N>1 path on the CPU (and, marked gpu, two ranks sharing the one GPU of the test box): two ranks (gloo, 127.0.0.1) shard a 10-channel mixed batch, each rank
plans its own channels on its own (control-plane-only) handle, and the aggregate the bench
reports -- total symbols, max elapsed -- is the same as a single rank handling all channels."""
import os
import socket
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist

    from psk_soft_amd import lib as pl
    from psk_soft_amd.distributed import max_over_ranks, shard_channels, sum_over_ranks

    dist.init_process_group("gloo", rank=rank, world_size=world)
    total = 10
    first, count = shard_channels(total, world, rank)
    h = pl.Handle(count, device=pl.DEVICE_NONE)
    props = [dict(samplesPerBaud=(8, 10)[c % 2], constelationSize=(2, 4, 8)[c % 3], numAvg=100) for c in range(first, first + count)]
    h.configure(0, props)
    n_sym = 0
    for call in range(3):
        res = h.plan_only(0, [dict(n_floats=2 * (5000 + 100 * c), xdelta=0.01, sriChanged=(call == 0)) for c in range(first, first + count)])
        n_sym += sum(r["n_symbols"] for r in res)
    dist.barrier()
    elapsed = max_over_ranks(0.5 + rank, dist)
    total_sym = sum_over_ranks(n_sym, dist)
    q.put((rank, first, count, elapsed, total_sym))
    dist.destroy_process_group()


def test_two_ranks_shard_channels():
    import torch.multiprocessing as mp

    from psk_soft_amd import lib as pl
    from psk_soft_amd.distributed import shard_channels

    assert [shard_channels(10, 4, r) for r in range(4)] == [(0, 3), (3, 3), (6, 2), (8, 2)]
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [(o[1], o[2]) for o in out] == [(0, 5), (5, 5)]
    assert all(o[3] == 1.5 for o in out)  # max over ranks
    # single-rank reference of the same batch
    h = pl.Handle(10, device=pl.DEVICE_NONE)
    h.configure(0, [dict(samplesPerBaud=(8, 10)[c % 2], constelationSize=(2, 4, 8)[c % 3], numAvg=100) for c in range(10)])
    n_sym = 0
    for call in range(3):
        n_sym += sum(r["n_symbols"] for r in h.plan_only(0, [dict(n_floats=2 * (5000 + 100 * c), xdelta=0.01, sriChanged=(call == 0)) for c in range(10)]))
    assert all(o[4] == n_sym for o in out) and n_sym > 0


def _run_bench(args, env_extra, timeout=300, keep_world=False):
    import json
    import subprocess

    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    env.update(env_extra)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, timeout=timeout)
    assert out.returncode == 0, out.stderr.decode()[-2000:]
    lines = [l for l in out.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout.decode()
    return json.loads(lines[0])


def test_bench_launcher_starts_the_ranks_it_is_asked_for():
    """`python bench.py --gpus N` with no launcher around it starts N ranks itself (round 1 ran one rank and
    printed n_gpus 1 whatever --gpus said).  Here: the same launcher, barriers, sharding and reductions on the
    CPU (gloo, control-plane-only handles)."""
    dry = {"PSK_BENCH_DRY": "1", "PSK_BENCH_BACKEND": "gloo"}
    r = _run_bench(["--gpus", "2", "--steps", "2", "--warmup", "1", "--channels", "48", "--nsamp", "8192"], dry)
    assert r["n_gpus"] == 2 and r["scaling"] == "weak" and r["channels_total"] == 96
    assert r["symbols_per_step"] == 96 * (8192 // 8)  # steady state: every complete symbol comes out
    # strong scaling of BASELINE configs[3] (8-PSK, 10 samples per baud): a fixed total shared by three ranks
    r = _run_bench(["--gpus", "3", "--steps", "2", "--warmup", "1", "--strong", "100", "--nsamp", "8190", "--M", "8", "--S", "10"], dry)
    assert r["n_gpus"] == 3 and r["scaling"] == "strong" and r["channels_total"] == 100
    assert r["symbols_per_step"] == 100 * 819


def test_bench_launcher_refuses_more_ranks_than_devices():
    import subprocess

    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "64", "--steps", "1"], env=env,
                         capture_output=True, timeout=300)
    assert out.returncode != 0 and b"device(s) visible" in out.stderr


@pytest.mark.gpu
def test_bench_two_ranks_share_the_gpu():
    """The real rank path (HIP handles, kernels, events, the oracle check of --check) under the launcher, two
    ranks on the one GPU of the test box (gloo for the barrier and the reductions: RCCL wants a GPU per rank)."""
    r = _run_bench(["--gpus", "2", "--steps", "3", "--warmup", "2", "--channels", "256", "--nsamp", "16384", "--check",
                    "--no-cpu-baseline"], {"PSK_BENCH_BACKEND": "gloo", "PSK_BENCH_SHARE_GPU": "1"}, timeout=600)
    assert r["n_gpus"] == 2 and r["check"]["bits_index_exact"] and r["check"]["soft_max_rel_err"] == 0.0
    assert r["config"]["channels_per_gpu"] == 256


@pytest.mark.gpu
def test_bench_rank_through_rccl():
    """One rank through the calls an N-GPU run makes on RCCL (`init_process_group("nccl")`, barrier, the max and sum
    reductions on device tensors): the test box has one GPU, so the multi-GPU launch itself is the driver's to run."""
    import socket

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    r = _run_bench(["--steps", "3", "--warmup", "2", "--channels", "256", "--nsamp", "16384", "--no-cpu-baseline"],
                   {"PSK_BENCH_FORCE_DIST": "1", "WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0", "MASTER_ADDR": "127.0.0.1",
                    "MASTER_PORT": str(port), "HSA_ENABLE_IPC_MODE_LEGACY": "0"}, timeout=600, keep_world=True)
    assert r["n_gpus"] == 1 and r["check"]["soft_phase_bit_identical"]
