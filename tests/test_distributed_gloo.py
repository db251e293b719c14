"""N>1 path on the CPU: two ranks (gloo, 127.0.0.1) shard a 10-channel mixed batch, each rank
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
