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


# ---- SURVEY.md §8e, optional mode: the blocks of ONE call spread over the ranks + the one all_gather ------------------
def _oracle_rows(sub, max_K, stats, seed, omega, S, B):
    """What irec_beam_encode returns for the rows of `sub`, computed block by block with the CPU oracle."""
    from oracle import oracle as O
    flat = [np.ascontiguousarray(a.reshape(-1)) for a in stats]
    bs = sub.block_size
    K = np.zeros(sub.n_blocks, dtype=np.int32)
    idx = np.zeros((sub.n_blocks, max(max_K, 1)), dtype=np.int32)
    sample = np.full(flat[0].shape, np.nan, dtype=np.float32)          # untouched elements must never reach a result
    at = sub.element_index(np.arange(sub.n_blocks), bs)
    for r in range(sub.n_blocks):
        e = at[r][at[r] >= 0]
        ix, smp = O.encode_block(flat[0][e], flat[1][e], flat[2][e], flat[3][e], seed, omega, S, B)
        K[r] = len(ix)
        if len(ix) <= max_K:
            idx[r, :len(ix)] = ix
            sample[e] = smp
    return torch.from_numpy(K), torch.from_numpy(idx), torch.from_numpy(sample).reshape(stats[0].shape)


def _block_worker(rank, world, port, q):
    import sys
    for p in (ROOT, PKG):
        if p not in sys.path:
            sys.path.insert(0, p)
    import irec
    from irec import sharding
    from oracle import oracle as O
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n_t, n, bs, seed, omega, B = 2, 2500, 700, 42, 3.0, 20
        S = O.n_samples(omega, 1.2)
        lat = [O.synthetic_latent(600 + i, n) for i in range(n_t)]
        stats = [np.stack([l[k] for l in lat]) for k in range(4)]
        coder = irec.BeamSearchCoder(kl_per_partition=omega, n_beams=B, extra_samples=1.2, block_size=bs)
        coder._max_K_hint = 3          # too few index slots at first: every rank must come back for more, together
        idx, sample = sharding.encode_block_sharded(
            coder, *[torch.from_numpy(a) for a in stats], seed, rank, world, dist,
            encode_rows=lambda sub, max_K: _oracle_rows(sub, max_K, stats, seed, omega, S, B))
        q.put((rank, idx, sample.numpy()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_block_sharded_call_reassembles_to_the_unsharded_result(world, oracle):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_block_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    n_t, n, bs = 2, 2500, 700
    S = oracle.n_samples(3.0, 1.2)
    for t in range(n_t):
        mq, sq, mp_, sp = oracle.synthetic_latent(600 + t, n)
        ridx, rs = oracle.encode_tensor(mq, sq, mp_, sp, 42, 3.0, S, 20, block_size=bs)
        for rank, idx, sample in results:
            assert idx[t] == ridx, (rank, t)
            assert np.array_equal(sample[t], rs), (rank, t)


def test_block_shard_rows_cover_every_block_once():
    from irec import sharding
    rows = np.concatenate([sharding.block_shard_rows(302, r, 8) for r in range(8)])
    assert sorted(rows.tolist()) == list(range(302))
    assert [len(sharding.block_shard_rows(302, r, 8)) for r in range(8)] == [38, 38, 38, 38, 38, 38, 37, 37]


@pytest.mark.gpu
def test_block_sharded_call_on_the_device_two_ranks(tmp_path):
    """scripts/block_shard_check.py under torch.distributed.run: two ranks (gloo, sharing the test box's one GPU) code the
    odd and the even rows of a 302-block call with the HIP encoder, exchange, and must hold exactly what the whole call gives."""
    import json
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env.update(IREC_DIST_BACKEND="gloo", TMPDIR=str(tmp_path))
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29631", os.path.join(ROOT, "scripts", "block_shard_check.py")],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    res = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert res["sharded_equals_whole"] and res["n_indices"] > 0 and res["backend"] == "gloo"


def test_layout_subset_and_element_index():
    """BlockLayout.subset keeps the parent's tensors / permutation and lists only the chosen rows; element_index walks a block
    in the shuffled order of Coder.split (coder.py:62-83): together every element of every tensor is named exactly once."""
    from irec.engine import BlockLayout
    lay = BlockLayout(torch.device("cpu"), 2, 2500, 700, 42)
    assert lay.n_blocks == 8 and sorted(lay.block_dim.tolist(), reverse=True) == lay.block_dim.tolist()   # largest first
    rows = np.array([1, 4, 6])
    sub = lay.subset(rows)
    assert sub.n_blocks == 3 and sub.max_dim == lay.max_dim and sub.perm is lay.perm
    assert sub.block_dim.tolist() == lay.block_dim[rows].tolist() and sub.block_base.tolist() == lay.block_base[rows].tolist()
    at = lay.element_index(np.arange(lay.n_blocks), 700)
    named = at[at >= 0]
    assert named.size == 2 * 2500 and np.array_equal(np.sort(named), np.arange(2 * 2500))
    assert ((at >= 0).sum(axis=1) == lay.block_dim.numpy()).all()
    # a block's elements are perm[pos : pos + dim] of its tensor
    r = 5
    base, pos, dim = int(lay.block_base[r]), int(lay.block_pos[r]), int(lay.block_dim[r])
    assert np.array_equal(at[r][:dim], base + lay.perm_host[pos:pos + dim])
    # without block_size: one block per tensor, natural order
    flat = BlockLayout(torch.device("cpu"), 3, 100, None, 7)
    assert np.array_equal(flat.element_index(np.arange(3), 100), np.arange(300).reshape(3, 100))
