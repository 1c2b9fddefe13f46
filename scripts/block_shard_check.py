#!/usr/bin/env python3
"""SURVEY.md §8e, optional mode, on the device: one coder call whose blocks are spread over the ranks
(irec.sharding.encode_block_sharded: layout row r -> rank r mod G, one all_gather of index rows + sample shares) against
the same call coded whole by every rank.  Default shape: the first latent level of a Kodak image in the two-level lossy
model, 301 056 dims = 302 blocks, B = 10, Omega = 3, S = 20 (BASELINE.json configs[3]).  Prints one JSON line on rank 0.
Run under torch.distributed.run (backend nccl = RCCL, one rank per GPU; IREC_DIST_BACKEND=gloo lets several ranks share
the one GPU of a test box)."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "relative-entropy-coding_amd")]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dims", type=int, default=301056)
    ap.add_argument("--beams", type=int, default=10)
    ap.add_argument("--eps1", type=float, default=1.0)
    ap.add_argument("--kl-scale", type=float, default=1.0, help="scales q's offset from p: more partitions per block")
    args = ap.parse_args()
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    device = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")) % torch.cuda.device_count())
    torch.cuda.set_device(device)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group(backend=os.environ.get("IREC_DIST_BACKEND", "nccl"))
    import irec
    from irec import sharding
    rng = np.random.default_rng(4321)                 # the same statistics on every rank (SURVEY §8d recipe)
    n = args.dims
    mp = rng.standard_normal(n).astype(np.float32)
    sp = np.exp(0.25 * rng.standard_normal(n)).astype(np.float32)
    mq = (mp + sp * (0.2 * args.kl_scale * rng.standard_normal(n))).astype(np.float32)
    sq = (sp * np.exp(-np.abs(0.05 * rng.standard_normal(n)))).astype(np.float32)
    t = [torch.from_numpy(a)[None].to(device) for a in (mq, sq, mp, sp)]
    coder = irec.BeamSearchCoder(kl_per_partition=3.0, n_beams=args.beams, extra_samples=args.eps1, block_size=1000)

    def whole():
        return coder.encode_tensors(*t, 42, 1000)

    def sharded():
        return sharding.encode_block_sharded(coder, *t, 42, rank, world, dist)

    idx_w, smp_w = whole()
    idx_s, smp_s = sharded()
    same = idx_w == idx_s and bool(torch.equal(smp_w, smp_s))
    times = {}
    for name, fn in (("whole", whole), ("sharded", sharded)):
        ts = []
        for _ in range(5):
            if dist is not None:
                dist.barrier()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        times[name] = sorted(ts)[2]
    if rank == 0:
        print(json.dumps({"config": f"{n} dims = {len(idx_w[0])} blocks, B={args.beams}, S={coder.n_samples}, {world} rank(s)",
                          "backend": dist.get_backend() if dist is not None else None, "sharded_equals_whole": same,
                          "n_indices": sum(len(b) for b in idx_w[0]), "ms_whole": 1e3 * times["whole"],
                          "ms_sharded_incl_exchange": 1e3 * times["sharded"]}), flush=True)
    if dist is not None:
        ok = torch.tensor([1 if same else 0], dtype=torch.int32, device=device if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        dist.destroy_process_group()
        if int(ok.item()) != 1:
            sys.exit(3)
    elif not same:
        sys.exit(3)


if __name__ == "__main__":
    main()
