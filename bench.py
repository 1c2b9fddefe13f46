#!/usr/bin/env python3
"""bench.py -- encoded latents/s of the iREC beam-search encoder on MI355X (BASELINE.json metric).

One "step" = one pass of the hot path (irec_beam_encode: one persistent kernel launch) over one batch of
--latents synthetic RVAE latent tensors [16,16,32] (8192 dims -> 8 blocks of 1000 + 1 of 192 dims each) that are
already resident in HBM, with B=20, Omega=3, 1+eps=1.2 (S=36): BASELINE.json configs[1].  Multi-GPU: one process per
GPU, every rank codes its own batch (weak scaling, no data-path collective); the only collective is the final RCCL
all_gather of the per-latent code lengths.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "relative-entropy-coding_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
N_CU, LANES = 256, 64
TENSOR_SHAPE = (16, 16, 32)
N_DIMS = 8192
BLOCK_SIZE = 1000
OMEGA, EPS1, BEAMS, SEED = 3.0, 1.2, 20, 42


def synthetic_batch(n_latents, device, rank):
    """SURVEY.md §8d statistics, drawn on the device (torch generator seeded per rank); values differ from the numpy
    fixtures, the distribution does not."""
    g = torch.Generator(device=device)
    g.manual_seed(1234 + rank)
    shape = (n_latents, N_DIMS)
    mp = torch.randn(shape, generator=g, device=device)
    lsp = 0.25 * torch.randn(shape, generator=g, device=device)
    sp = torch.exp(lsp)
    mq = mp + sp * 0.2 * torch.randn(shape, generator=g, device=device)
    sq = torch.exp(lsp - (0.05 * torch.randn(shape, generator=g, device=device)).abs())
    return tuple(t.float().contiguous() for t in (mq, sq, mp, sp))


def host_cores():
    """Cores this process may actually run on (cgroup/affinity aware), capped: eager torch on small tensors does not
    scale past a few dozen threads and oversubscription makes it far slower."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(n, 32))


def log(*a):
    print("[bench]", *a, file=sys.stderr, flush=True)


def cpu_baselines(q, n_ref, n_opt):
    """Timed on the host cores of this box, rank 0 only.  (1) reference-shaped torch-eager port of the TF path
    (oracle/ref_shaped_torch.py, all cores), (2) the C oracle (1 core)."""
    from oracle import oracle as O
    from oracle import ref_shaped_torch as R
    S = O.n_samples(OMEGA, EPS1)
    host = [t[:max(n_ref, n_opt)].cpu().numpy() for t in q]
    torch.set_num_threads(host_cores())
    log(f"cpu baseline on {host_cores()} threads (affinity {len(os.sched_getaffinity(0))}, cpu_count {os.cpu_count()})")
    R.encode_tensor(*(h[0] for h in host), SEED, OMEGA, S, BEAMS, BLOCK_SIZE)  # warm-up
    log("cpu warm-up done")
    t0 = time.perf_counter()
    for i in range(n_ref):
        R.encode_tensor(*(h[i] for h in host), SEED, OMEGA, S, BEAMS, BLOCK_SIZE)
    t_ref = time.perf_counter() - t0
    t0 = time.perf_counter()
    ref_out = [O.encode_tensor(*(h[i] for h in host), SEED, OMEGA, S, BEAMS, block_size=BLOCK_SIZE) for i in range(n_opt)]
    t_opt = time.perf_counter() - t0
    return n_ref / t_ref, n_opt / t_opt, ref_out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--latents", type=int, default=2048, help="latent tensors per step per GPU")
    ap.add_argument("--cpu-ref-latents", type=int, default=48)
    ap.add_argument("--cpu-opt-latents", type=int, default=96)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device; there is no CPU fallback")
    n_dev = torch.cuda.device_count()
    dev_index = local_rank % n_dev          # (test rigs may run several ranks on one GPU: IREC_DIST_BACKEND=gloo)
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    dist = None
    backend = os.environ.get("IREC_DIST_BACKEND", "nccl")   # "nccl" IS RCCL on ROCm
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=device)
        else:
            dist.init_process_group(backend=backend)

    import irec
    eng = irec.get_engine(device)
    S = int(np.exp(OMEGA * EPS1))
    params = eng.params(OMEGA, S, BEAMS)
    L = args.latents
    q = synthetic_batch(L, device, rank)
    lay = eng.layout(L, N_DIMS, BLOCK_SIZE, SEED)
    max_K = 32
    out = (torch.empty(lay.n_blocks, dtype=torch.int32, device=device),
           torch.empty((lay.n_blocks, max_K), dtype=torch.int32, device=device), torch.empty_like(q[0]))
    eng.workspace(params, lay.max_dim, max_K)  # allocate scratch outside the timed region

    coll_dev = device if backend == "nccl" else torch.device("cpu")   # gloo rigs exchange through host tensors

    def barrier():
        torch.cuda.synchronize(device)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(device)

    from irec import sharding

    def exchange(K):
        # the path's only exchange step: per-latent code length (nats), all ranks <- all ranks (RCCL over xGMI)
        return sharding.gather_per_item(sharding.code_nats_per_tensor(K, lay, S).to(coll_dev), world * L, rank, world, dist)

    log(f"rank {rank}/{world}: {L} latents, {lay.n_blocks} blocks per step")
    for _ in range(max(args.warmup, 1)):
        eng.encode_blocks(params, lay, *q, SEED, max_K, out=out)
        exchange(out[0])
    barrier()
    log("warm-up done")
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for a, b in ev:
        a.record()
        eng.encode_blocks(params, lay, *q, SEED, max_K, out=out)   # same stream as the events (torch current stream)
        b.record()
    K = out[0]
    gathered = exchange(K)
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    kernel_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
    log(f"timed region {elapsed:.3f} s, kernel {kernel_ms:.2f} ms per step")
    Kh = K.cpu().numpy().astype(np.int64)
    dims = lay.block_dim.cpu().numpy().astype(np.int64)
    assert Kh.min() >= 0 and Kh.max() <= max_K, "a block needed more than max_K partitions"
    algo_bytes = int((24 * dims + 4 * Kh).sum())                        # 24 D + 4 K per block
    evals = int((S * dims * (1 + np.maximum(Kh - 1, 0) * BEAMS) * (Kh > 0)).sum())
    clk_ghz = 2.4
    # HBM-side traffic per launch from the committed PMC profile (FETCH_SIZE / WRITE_SIZE passes, corrected as
    # MI355X_MICROARCH.md prescribes: KiB units, FETCH_SIZE x2 for 16 B/lane streams); scaled by the latent count.
    traffic = None
    tj = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tj):
        tr = json.load(open(tj))
        traffic = tr["hbm_bytes_per_latent"] * L

    result = {
        "metric": "encoded latents/sec", "value": world * L * args.steps / elapsed, "unit": "latents/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "RVAE Cifar10-shape latents [16,16,32], beam_search B=20 Omega=3 eps=0.2 (S=36), "
                               "block_size=1000 (configs[1])",
                   "latents_per_step_per_gpu": L, "blocks_per_step_per_gpu": int(lay.n_blocks),
                   "parallelism": f"latents sharded over {world} GPU(s), no data-path collective"},
        "roofline": {"bound": "hbm", "achieved": algo_bytes / (kernel_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": algo_bytes / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": traffic,
                     "kernel": "encode_team_kernel<20,2,1>", "kernel_ms": kernel_ms, "algorithmic_bytes": algo_bytes},
        "secondary": {"proposal_evals_per_s": evals / (kernel_ms * 1e-3),
                      "evals_per_clk_per_cu": evals / (kernel_ms * 1e-3) / (N_CU * clk_ghz * 1e9),
                      # the ceiling that actually binds: random 4-byte LDS look-ups per clock per CU, measured by
                      # scripts/microbench/gather_rates.hip (profiles/r01j): 8.9 for one table copy, 13.4 with the
                      # 2-choice bank assignment over three copies that the encoder uses (8 waves per CU)
                      "lds_gather_roofline": {"achieved": evals / (kernel_ms * 1e-3) / (N_CU * clk_ghz * 1e9), "peak": 13.4,
                                              "unit": "look-ups/clk/CU at 2.4 GHz",
                                              "frac": evals / (kernel_ms * 1e-3) / (N_CU * clk_ghz * 1e9) / 13.4},
                      "mean_K": float(Kh.mean()), "code_nats_per_latent": float(gathered.mean().item())},
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        ref_lps, opt_lps, ref_out = cpu_baselines(q, args.cpu_ref_latents, args.cpu_opt_latents)
        # parity of the timed outputs against the oracle on the sampled latents
        idx_h = out[1].cpu().numpy()
        samp_h = out[2].cpu().numpy()
        for i, (ridx, rs) in enumerate(ref_out):
            for j in range(lay.blocks_per_tensor):
                row = lay.natural[i * lay.blocks_per_tensor + j]
                assert idx_h[row, :Kh[row]].tolist() == ridx[j], f"parity: latent {i} block {j}"
            assert np.array_equal(samp_h[i], rs), f"parity: latent {i} sample"
        result["cpu_baseline"] = {"value": ref_lps, "unit": "latents/s", "cores": host_cores(), "kind": "port",
                                  "sample": f"{args.cpu_ref_latents} latents of the same batch, reference-shaped "
                                            "torch-eager restatement of the TF path (oracle/ref_shaped_torch.py)"}
        result["cpu_baseline_c_oracle"] = {"value": opt_lps, "unit": "latents/s", "cores": 1, "kind": "port",
                                           "sample": f"{args.cpu_opt_latents} latents, oracle/irec_oracle.c canonical mode"}
        result["parity_checked_latents"] = len(ref_out)
    if rank == 0:
        print(json.dumps(result), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
