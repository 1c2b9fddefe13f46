"""Sharding of independent latent tensors (images) over one process per GPU, and the path's only exchange step.

The reference is single-process; its compress loop walks images one by one
(examples/lossless/compression_performance.py:305).  Images are independent, so image i goes to rank i mod G with
no data-path collective; the code lengths (bits per image) are gathered once at the end -- over RCCL/xGMI on the GPU
box (`backend="nccl"`), over gloo in the CPU tests.
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
