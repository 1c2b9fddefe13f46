"""world_size-2 gloo test (CPU) of the multi-process path: image sharding + the final code-length gather."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import PKG, ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_items, q):
    import sys
    for p in (ROOT, PKG):
        if p not in sys.path:
            sys.path.insert(0, p)
    from irec import sharding
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        mine = sharding.shard_indices(n_items, rank, world)
        local = torch.tensor([1000.0 + 3.0 * i for i in mine], dtype=torch.float64)   # "bits of image i"
        full = sharding.gather_per_item(local, n_items, rank, world, dist)
        t = torch.tensor([float(rank + 1)], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)          # the bench's max-over-ranks timing reduction
        q.put((rank, full.tolist(), float(t.item()), mine.tolist()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_items", [7, 8, 300])
def test_two_rank_shard_and_gather(n_items):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_items, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    expected = [1000.0 + 3.0 * i for i in range(n_items)]
    seen = []
    for rank, full, tmax, mine in results:
        assert full == expected           # every rank holds every image's value, in image order
        assert tmax == float(world)
        seen += mine
    assert sorted(seen) == list(range(n_items))   # every image coded exactly once


def test_shard_sizes_config3():
    from irec import sharding
    assert sharding.shard_sizes(300, 8) == [38, 38, 38, 38, 37, 37, 37, 37]
    assert sum(sharding.shard_sizes(300, 8)) == 300
    assert sharding.shard_indices(10, 1, 4).tolist() == [1, 5, 9]


def test_code_nats_per_tensor():
    from irec import sharding
    from irec.engine import BlockLayout
    lay = BlockLayout(torch.device("cpu"), 3, 8192, 1000, 42)
    K_nat = np.arange(27).reshape(3, 9) % 5 + 1           # K of (tensor, block)
    K_rows = torch.from_numpy(K_nat.reshape(-1)[lay.order].astype(np.int32))
    got = sharding.code_nats_per_tensor(K_rows, lay, 36)
    assert np.allclose(got.numpy(), K_nat.sum(axis=1) * np.log(36))
