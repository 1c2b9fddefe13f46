"""Sharding of independent latent tensors (images) over one process per GPU, and the path's only exchange step.

The reference is single-process; its compress loop walks images one by one
(examples/lossless/compression_performance.py:305).  Images are independent, so image i goes to rank i mod G with
no data-path collective; the code lengths (bits per image) are gathered once at the end -- over RCCL/xGMI on the GPU
box (`backend="nccl"`), over gloo in the CPU tests.

Optional second mode ("intra-image sharding"): ONE coder call whose blocks are spread over the ranks --
the 302 blocks of a Kodak image's first latent level (large_2_level_vae.py:394-408) -- followed by the one exchange
step that mode has: an all_gather of every rank's index rows and of its blocks' share of the sample (1.2 MB for the
301 056-dim level), so that every rank holds the merged sample the synthesis transform needs.
"""
import numpy as np
import torch


def shard_indices(n_items, rank, world):
    """Items handled by `rank`: i with i % world == rank (config 3: 300 images -> 38/38/38/38/37/37/37/37)."""
    return np.arange(rank, n_items, world, dtype=np.int64)


def shard_sizes(n_items, world):
    return [len(range(r, n_items, world)) for r in range(world)]


def gather_per_item(local_values, n_items, rank, world, dist=None):
    """All ranks get the [n_items] vector whose entry i was produced by rank i % world.
    `local_values`: 1-D tensor, entry k = value of item rank + k*world.  One all_gather of a padded vector."""
    if world == 1 or dist is None:
        return local_values.clone()
    per_rank = (n_items + world - 1) // world
    padded = torch.zeros(per_rank, dtype=local_values.dtype, device=local_values.device)
    padded[:local_values.numel()] = local_values
    out = torch.empty(world * per_rank, dtype=local_values.dtype, device=local_values.device)
    dist.all_gather_into_tensor(out, padded)
    out = out.reshape(world, per_rank).t().reshape(-1)  # item i = out[i // world ... ] -> interleave ranks
    return out[:n_items].contiguous()


def code_nats_per_tensor(K, layout, n_samples):
    """sum over the blocks of a tensor of K_block * ln(S)  (BeamSearchCoder.get_codelength, beam_search_coder.py:150-151).
    K: [n_blocks] int tensor in layout row order."""
    owner = torch.as_tensor(layout.order // layout.blocks_per_tensor, device=K.device)
    out = torch.zeros(layout.n_tensors, dtype=torch.float64, device=K.device)
    out.index_add_(0, owner, K.to(torch.float64) * float(np.log(n_samples)))
    return out


def block_shard_rows(n_blocks, rank, world):
    """Rows (in layout order: largest blocks first, so the ranks' loads match) of the call's blocks that `rank` codes."""
    return np.arange(rank, n_blocks, world, dtype=np.int64)


def encode_block_sharded(coder, q_loc, q_scale, p_loc, p_scale, seed, rank, world, dist, max_K=None, encode_rows=None):
    """`coder.encode(..., batched=True)` with the BLOCKS of the call spread over `world` ranks (block row r of the layout
    -> rank r mod world) and one all_gather.  Every rank passes the same tensors ([N, ...], float32) and gets the same
    result: (indices[tensor][block], sample) -- bit for bit what the unsharded call returns, because a block's code
    depends on nothing but its own statistics and the shared seed (coder.py:435-452).

    `encode_rows(sub_layout, max_K) -> (K [rows], idx [rows, max_K], sample [like q_loc])` replaces the device encoder
    in the CPU tests of the exchange (tests/test_sharding_gloo.py)."""
    from .coding.beam_search_coder import MorePartitionsNeeded
    while True:   # every rank reads the same gathered K, so all of them come back for more index slots together
        try:
            return _encode_block_sharded_once(coder, q_loc, q_scale, p_loc, p_scale, seed, rank, world, dist, max_K, encode_rows)
        except MorePartitionsNeeded as e:
            max_K = e.need


def _encode_block_sharded_once(coder, q_loc, q_scale, p_loc, p_scale, seed, rank, world, dist, max_K, encode_rows):
    from .coding.beam_search_coder import PendingCode
    from .engine import BlockLayout, get_engine
    src = torch.as_tensor(q_loc)
    n_tensors, n = src.shape[0], src[0].numel()
    max_K = coder._max_K_hint if max_K is None else int(max_K)
    bs = n if coder.block_size is None else int(coder.block_size)
    if encode_rows is None:
        eng = get_engine(src.device if src.device.type == "cuda" else None)
        lay = eng.layout(n_tensors, n, coder.block_size, seed)
    else:
        eng, lay = None, BlockLayout(torch.device("cpu"), n_tensors, n, coder.block_size, seed)
    mine = block_shard_rows(lay.n_blocks, rank, world)
    sub = lay.subset(mine)
    if encode_rows is None:
        params = coder._params()
        dev = [coder._dev(t, eng.device) for t in (q_loc, q_scale, p_loc, p_scale)]
        K, idx, sample = eng.encode_blocks(params, sub, *dev, seed, max_K)
    else:
        K, idx, sample = encode_rows(sub, max_K)
    # ---- the exchange: per block row [K | idx[max_K] | the block's sample elements in shuffled order (as int32 bits)] ----
    per_rank = (lay.n_blocks + world - 1) // world
    width = 1 + max(max_K, 1) + bs
    cache = lay.__dict__.setdefault("_shard_maps", {})         # index maps of this (layout, rank, world): built once
    key = (rank, world, str(sample.device))
    if key not in cache:
        r = np.arange(lay.n_blocks)
        put = torch.as_tensor(lay.element_index(r, bs), device=sample.device)
        cache[key] = (torch.as_tensor(lay.element_index(mine, bs), device=sample.device).clamp(min=0),
                      torch.as_tensor((r % world) * ((lay.n_blocks + world - 1) // world) + r // world, device=sample.device),
                      put[put >= 0], put >= 0)
    take, slot, put_at, ok = cache[key]                        # take: [rows, bs] (padding reads element 0, never used)
    vals = sample.reshape(-1)[take].contiguous().view(torch.int32)
    packed = torch.zeros((per_rank, width), dtype=torch.int32, device=sample.device)
    packed[:len(mine), 0] = K
    packed[:len(mine), 1:1 + idx.shape[1]] = idx
    packed[:len(mine), 1 + max(max_K, 1):] = vals
    if world > 1:
        coll = packed if dist.get_backend() == "nccl" else packed.cpu()
        everyone = torch.empty((world * per_rank, width), dtype=torch.int32, device=coll.device)
        dist.all_gather_into_tensor(everyone, coll)
        everyone = everyone.to(sample.device)
    else:
        everyone = packed
    # ---- reassembly: rank rr's j-th row is layout row rr + j * world ----
    rows_all = everyone[slot]                                                            # layout row order
    K_all = rows_all[:, 0].contiguous()
    idx_all = rows_all[:, 1:1 + max(max_K, 1)].contiguous()
    full = torch.empty(n_tensors * n, dtype=torch.float32, device=sample.device)
    full[put_at] = rows_all[:, 1 + max(max_K, 1):].contiguous().view(torch.float32)[ok]
    pending = PendingCode(coder, lay, K_all, idx_all, full.reshape(src.shape).to(src.device), max_K)   # (sample on the inputs' device)
    return pending.to_lists(), pending.sample
