#!/usr/bin/env python3
"""bench.py -- encoded latents/s of the iREC beam-search encoder on MI355X (BASELINE.json metric).

One "step" = one pass of the hot path (irec_beam_encode: one persistent kernel launch) over one batch of
--latents (default 65536) synthetic RVAE latent tensors [16,16,32] (8192 dims -> 8 blocks of 1000 + 1 of 192 dims each) that are
already resident in HBM, with B=20, Omega=3, 1+eps=1.2 (S=36): BASELINE.json configs[1].  Multi-GPU: one process per
GPU, every rank codes its own batch (weak scaling, no data-path collective); the only collective is the final RCCL
all_gather of the per-latent code lengths.

Launching.  `python bench.py --gpus N` with no WORLD_SIZE in the environment starts the N ranks itself: the parent
touches no GPU, spawns N children of this file with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set,
relays rank 0's JSON line and exits non-zero if any child fails.  Under `python -m torch.distributed.run
--nproc-per-node N bench.py --gpus N` the environment is already there; `--gpus` must equal WORLD_SIZE.
IREC_DIST_BACKEND=gloo (test rigs with fewer GPUs than ranks) shares devices round-robin and exchanges on the host.

Prints ONE JSON line on rank 0.  Every figure in it is measured by this run or read from `profiles/` of the same
source hash (roofline.traffic; null when the committed PMC profile is from other kernel sources).
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "relative-entropy-coding_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
LDS_HW_LOOKUPS = 32.0     # ds_read_b32: 128 B/clk/CU = 32 four-byte look-ups per clock per CU (MI355X_MICROARCH.md §LDS)
LDS_STREAM_LOOKUPS = 13.68   # the same mixture with the kernel's own instruction stream (profiles/r05b/bank_limits.log)
LDS_2CHOICE_LOOKUPS = 13.70  # measured ceiling of random look-ups with the 2-choice bank assignment at the 12 waves per CU
                             # the default kernel runs (profiles/r01j/gather_rates.log; 13.4 at 8 waves, 14.0 at 16; round 5, with the
                             # kernel's own instruction stream: 13.68, profiles/r05b/bank_limits.log -- DESIGN.md §4 "The gather ceiling")
TENSOR_SHAPE = (16, 16, 32)
N_DIMS = 8192
BLOCK_SIZE = 1000
OMEGA, EPS1, BEAMS, SEED = 3.0, 1.2, 20, 42
KERNEL_SOURCES = ("irec_team.hip", "irec_ten.hip", "irec_lone.hip", "irec_kernels.hip", "irec_fast_common.h", "irec_team_common.h",
                  "irec_device.h", "irec_kernels.h", "irec_host.cpp")


def kernel_source_hash():
    """sha256 (16 hex) over the kernel sources: profiles/traffic.json carries the hash it was measured on."""
    h = hashlib.sha256()
    for name in KERNEL_SOURCES:
        with open(os.path.join(PKG, "csrc", name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def log(*a):
    print("[bench]", *a, file=sys.stderr, flush=True)


# ---------------------------------------------------------------------------------------------------------------------
#  launcher: `python bench.py --gpus N` -> N rank processes (the parent never initialises a GPU)
# ---------------------------------------------------------------------------------------------------------------------
def launch_ranks(args, argv):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    import tempfile
    procs = []
    with tempfile.TemporaryFile() as out0_f:
        for r in range(args.gpus):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                       MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                          stdout=out0_f if r == 0 else subprocess.DEVNULL))
        # a rank that dies must not leave the others waiting in a collective: poll, and end the job on the first failure
        failed = False
        while any(p.poll() is None for p in procs):
            if any(p.poll() not in (None, 0) for p in procs):
                failed = True
                for p in procs:
                    if p.poll() is None:
                        p.kill()      # exactly the PIDs this launcher started
                break
            time.sleep(0.05)
        rcs = [p.wait() for p in procs]
        out0_f.seek(0)
        out0 = out0_f.read().decode()
    if failed or any(rcs):
        sys.stderr.write(out0)
        raise SystemExit(f"bench.py: rank exit codes {rcs}")
    line = [ln for ln in out0.splitlines() if ln.startswith("{")]
    if not line:
        raise SystemExit("bench.py: rank 0 printed no JSON line")
    res = json.loads(line[-1])
    if res.get("n_gpus") != args.gpus or res.get("world_size") != args.gpus:
        raise SystemExit(f"bench.py: asked for {args.gpus} ranks, the job reports n_gpus={res.get('n_gpus')}")
    print(line[-1], flush=True)


# ---------------------------------------------------------------------------------------------------------------------
#  one rank
# ---------------------------------------------------------------------------------------------------------------------
def synthetic_batch(n_latents, device, rank, n_dims=N_DIMS):
    """SURVEY.md §8d statistics, drawn on the device (torch generator seeded per rank); values differ from the numpy
    fixtures, the distribution does not."""
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(1234 + rank)
    shape = (n_latents, n_dims)
    mp = torch.randn(shape, generator=g, device=device)
    lsp = 0.25 * torch.randn(shape, generator=g, device=device)
    sp = torch.exp(lsp)
    mq = mp + sp * 0.2 * torch.randn(shape, generator=g, device=device)
    sq = torch.exp(lsp - (0.05 * torch.randn(shape, generator=g, device=device)).abs())
    return tuple(t.float().contiguous() for t in (mq, sq, mp, sp))


def skewed_batch(n_latents, device, rank, n_dims=N_DIMS, sigma=1.0, clip=3.0):
    """SURVEY.md §8d statistics with a per-TENSOR log-normal scale on delta (sigma = 1, clipped at `clip`): real posteriors
    differ by orders of magnitude in KL from one residual block / image to the next (resnet_vae.py:462-476), the plain
    synthetic batch has K = 7.3 +- 1.  KL per 1000-dim block ~ 1.7 + 20 scale^2 nats: K from 1 to ~60 within one call."""
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(4321 + rank)
    shape = (n_latents, n_dims)
    mp = torch.randn(shape, generator=g, device=device)
    lsp = 0.25 * torch.randn(shape, generator=g, device=device)
    sp = torch.exp(lsp)
    scale = torch.exp(sigma * torch.randn((n_latents, 1), generator=g, device=device)).clamp_(max=clip)
    mq = mp + sp * 0.2 * scale * torch.randn(shape, generator=g, device=device)
    sq = torch.exp(lsp - (0.05 * torch.randn(shape, generator=g, device=device)).abs())
    return tuple(t.float().contiguous() for t in (mq, sq, mp, sp))


def skewed_K_leg(eng, device, n_tensors, reps, check):
    """The block hand-out under heavy-tailed K (beam_search_coder.py:57-59: K = ceil(KL / Omega) per block): the headline
    settings on `skewed_batch`, look-ups per clock per CU next to the plain batch's, `check` tensors against the oracle."""
    import torch
    from oracle import oracle as O
    S = int(np.exp(OMEGA * EPS1))
    max_K = 128
    params = eng.params(OMEGA, S, BEAMS, table_steps=max_K)
    q = skewed_batch(n_tensors, device, 0)
    lay = eng.layout(n_tensors, N_DIMS, BLOCK_SIZE, SEED)
    out = (torch.empty(lay.n_blocks, dtype=torch.int32, device=device),
           torch.empty((lay.n_blocks, max_K), dtype=torch.int32, device=device), torch.empty_like(q[0]))
    plan = eng.plan(params, lay, max_K)
    res = {}
    for name, kw in (("as_listed", {}), ("longest_first", {"order_by_K": True})):
        for _ in range(2):
            eng.encode_blocks(params, lay, *q, SEED, max_K, out=out, **kw)
        torch.cuda.synchronize(device)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
        ev[0].record()
        for r in range(reps):
            eng.encode_blocks(params, lay, *q, SEED, max_K, out=out, **kw)
            ev[r + 1].record()
        torch.cuda.synchronize(device)
        ms = float(np.median([ev[r].elapsed_time(ev[r + 1]) for r in range(reps)]))
        Kh = out[0].cpu().numpy().astype(np.int64)
        assert Kh.min() >= 0 and Kh.max() <= max_K, ("skewed K", int(Kh.min()), int(Kh.max()))
        dims = lay.block_dim.cpu().numpy().astype(np.int64)
        evals = float((S * dims * (1 + np.maximum(Kh - 1, 0) * BEAMS) * (Kh > 0)).sum())
        res[name] = {"ms_per_call": ms, "lookups_per_clk_per_cu": evals / (ms * 1e-3) / (plan["n_cu"] * plan["clock_mhz"] * 1e6)}
        if check:
            c = min(check, n_tensors)
            hb = [t[:c].cpu().numpy() for t in q]
            ridx, rsamp, _ = O.encode_tensors_omp(*hb, SEED, OMEGA, S, BEAMS, BLOCK_SIZE, max_K=max_K, n_threads=host_cores())
            ih, sh = out[1].cpu().numpy(), out[2][:c].cpu().numpy()
            bpt = lay.blocks_per_tensor
            for i in range(c):
                for j in range(bpt):
                    row = lay.natural[i * bpt + j]
                    assert ih[row, :Kh[row]].tolist() == ridx[i][j], f"skewed K ({name}): parity, tensor {i} block {j}"
                assert np.array_equal(sh[i], rsamp[i]), f"skewed K ({name}): parity, tensor {i} sample"
    big = Kh[dims == dims.max()]
    out_ = {"workload": "headline settings, delta scaled per tensor by a log-normal (sigma 1, clipped at 3)", "tensors_per_call": n_tensors,
            "kernel": plan["kernel"], "K_min": int(Kh.min()), "K_max": int(Kh.max()), "K_mean": float(Kh.mean()),
            "K_percentiles_1000_dim_blocks": {str(p_): float(np.percentile(big, p_)) for p_ in (1, 10, 50, 90, 99)},
            "oracle_checked_tensors": min(check, n_tensors), **res}
    log(f"secondary skewed K: K {out_['K_min']}..{out_['K_max']} (mean {out_['K_mean']:.1f}); as listed {res['as_listed']['ms_per_call']:.2f} ms = "
        f"{res['as_listed']['lookups_per_clk_per_cu']:.2f} look-ups/clk/CU; longest first {res['longest_first']['ms_per_call']:.2f} ms = "
        f"{res['longest_first']['lookups_per_clk_per_cu']:.2f}")
    del q, out
    return out_


def margins_leg(eng, device, params, lay, q, out, S, max_K):
    """Encoder-side exposure of index parity, from the device, over EVERY block of a step (round 4's review, "Next" #1): the call of
    the timed region once more through irec_beam_encode_ex (outside the timed region) -- same K / indices / sample, asserted -- plus
    four floats per block: the smallest gap between the last candidate kept and the best one rejected over the block's steps, and
    the winner's lead at the last step (beam_search_coder.py:85-89,118-122).  An index can differ under another float32 summation
    order (TensorFlow's reduce_sum) only where such a comparison is closer than the orders disagree; the noise figures
    are the literal-vs-canonical score differences scripts/margins.py measured for these statistics (profiles/margins.json), the
    measured flip rate next to them is scripts/margins_flips.py's (profiles/margins_flips.json, >= 10^5 blocks on the CPU).
    K = ceil(KL / Omega) (:57-59): exposed where KL / Omega lies within the float32-sum noise of an integer (scripts/k_margins.py)."""
    import torch
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    eng.encode_blocks_margins(params, lay, *q, SEED, max_K)           # (scratch of the flag's plan allocated, tables built)
    ev[0].record()
    K2, idx2, samp2, mg = eng.encode_blocks_margins(params, lay, *q, SEED, max_K)
    ev[1].record()
    torch.cuda.synchronize(device)
    plan = eng.plan(params, lay, max_K, margins=True)
    assert torch.equal(K2, out[0]) and torch.equal(samp2, out[2]), "margins call: K / sample differ from the timed call's"
    live = torch.arange(max_K, device=device)[None, :] < out[0][:, None]
    assert bool(((idx2 == out[1]) | ~live).all()), "margins call: indices differ from the timed call's"
    m = mg.cpu().numpy().astype(np.float64)
    gap, at, top, top_at = m[:, 0], m[:, 1], m[:, 2], m[:, 3]
    small = np.minimum(gap, top)                                       # the closest comparison that decides an index of the block
    fin = np.isfinite(small)
    edges = [0.0, 1e-6, 3e-6, 1e-5, 3e-5, 1e-4, 3e-4, 1e-3, 1e-2, 1e-1, 1.0, np.inf]
    noise_p99, noise_max = 4.6e-5, 6.5e-5                              # profiles/margins.json, by_regime "SURVEY 8d statistics"
    rel = gap[np.isfinite(gap)] / np.maximum(at[np.isfinite(gap)], 1e-30)
    kl, Kk = eng.block_kl(params, lay, *q)
    x = kl.cpu().numpy().astype(np.float64) / float(np.float32(OMEGA))
    dK = np.abs(x - np.rint(x))
    k_noise = 2.0e-5                                                   # float32-sum noise of KL / Omega, profiles/margins.json k_margin (max 1.7e-5)
    res = {"what": "per block: min over its steps of score(last kept) - score(best rejected), and the winner's lead at the last step "
                   "(irec_beam_encode_ex; include/irec.h)", "blocks": int(m.shape[0]), "kernel": plan["kernel"],
           "ms_per_call": float(ev[0].elapsed_time(ev[1])), "same_outputs_as_the_timed_call": True,
           "set_gap_quantiles": {str(qq): float(np.quantile(gap[np.isfinite(gap)], qq)) for qq in (0.0, 1e-5, 1e-4, 1e-3, 1e-2, 0.1, 0.5)},
           "winner_lead_quantiles": {str(qq): float(np.quantile(top[np.isfinite(top)], qq)) for qq in (0.0, 1e-5, 1e-4, 1e-3, 1e-2, 0.1, 0.5)},
           "set_gap_over_score_quantiles": {str(qq): float(np.quantile(rel, qq)) for qq in (0.0, 1e-4, 1e-2, 0.5)},
           "closest_comparison_histogram": {"edges": [str(e) for e in edges], "blocks": np.histogram(small[fin], bins=edges)[0].tolist()},
           "exact_ties_at_a_deciding_comparison": int((small == 0).sum()),
           "noise": {"score_noise_p99": noise_p99, "score_noise_max": noise_max, "source": "profiles/margins.json (literal vs canonical, these statistics)"},
           "blocks_closer_than_noise_p99": float((small < noise_p99).mean()), "blocks_closer_than_noise_max": float((small < noise_max).mean()),
           "K_exposure": {"what": "|KL / Omega - nearest integer| per block (irec_block_kl)", "blocks_within_float32_sum_noise": float((dK < k_noise).mean()),
                          "noise": k_noise, "margin_quantiles": {str(qq): float(np.quantile(dK, qq)) for qq in (0.0, 1e-4, 1e-2, 0.5)}}}
    fj = os.path.join(ROOT, "profiles", "margins_flips.json")
    if os.path.exists(fj):
        fl = json.load(open(fj))
        res["measured_on_cpu"] = {"source": "profiles/margins_flips.json (scripts/margins_flips.py: canonical vs literal restatement, same statistics)",
                                  "blocks": fl.get("blocks"), "index_flips": fl.get("index_flips_literal_vs_canonical"),
                                  "flip_rate_per_block": fl.get("measured_flip_rate_per_block"), "K_differs": fl.get("K_differs_literal_vs_canonical"),
                                  "predicted_exposure_below_noise_max": (fl.get("predicted_exposure") or {}).get("blocks_below_noise_max")}
    log(f"secondary margins: {plan['kernel']} {res['ms_per_call']:.1f} ms; closest comparison below noise p99 / max: "
        f"{100 * res['blocks_closer_than_noise_p99']:.3f} % / {100 * res['blocks_closer_than_noise_max']:.3f} % of {res['blocks']} blocks, "
        f"{res['exact_ties_at_a_deciding_comparison']} exact ties; K exposed: {100 * res['K_exposure']['blocks_within_float32_sum_noise']:.4f} %")
    del K2, idx2, samp2, mg
    return res


def host_cores():
    """Cores this process may run on (cgroup / affinity aware)."""
    try:
        return max(1, len(os.sched_getaffinity(0)))
    except AttributeError:
        return max(1, os.cpu_count() or 1)


def cpu_baselines(q, n_ref, n_opt_budget_s):
    """BASELINE.md §3, timed on the host cores of this box, rank 0 at N = 1 only.
    CPU-ref: reference-shaped torch-eager restatement of the TF path (oracle/ref_shaped_torch.py): 3 warm-ups, then 5
             repeats of n_ref latents, median.  Thread count: 16, 32 and 64 (capped at the core count) are probed on two
             latents each and the FASTEST setting is used -- eager torch on [S,B,1,D] tensors stops scaling at a few dozen
             threads (the protocol's torch.set_num_threads(cpu_count) = 256 on the GPU box runs at 0.005 latents/s), and
             the baseline gets its best case.
    CPU-opt: the C oracle, OpenMP over blocks, all cores, for about n_opt_budget_s seconds."""
    import torch
    from oracle import oracle as O
    from oracle import ref_shaped_torch as R
    S = O.n_samples(OMEGA, EPS1)
    cores = host_cores()
    n_host = max(n_ref, 64)
    host = [t[:n_host].cpu().numpy() for t in q]

    def ref(i):
        return R.encode_tensor(*(h[i % n_host] for h in host), SEED, OMEGA, S, BEAMS, BLOCK_SIZE)

    probe = {}
    # (measured once on the 256-thread GPU box, profiles/archive/r02a/bench_default.err: 256 threads -> 0.005 latents/s against 3.9
    # at 32 -- every eager op then pays a 256-way fork/join; the probe stops at 64 so that the bench finishes in minutes)
    for nt in sorted({min(cores, 16), min(cores, 32), min(cores, 64)}):
        torch.set_num_threads(nt)
        ref(0)
        t0 = time.perf_counter()
        ref(1); ref(2)
        probe[nt] = 2 / (time.perf_counter() - t0)
    threads = max(probe, key=probe.get)
    torch.set_num_threads(threads)
    log(f"cpu-ref thread probe (latents/s): {probe} -> {threads} threads; affinity {cores}, cpu_count {os.cpu_count()}")
    for i in range(3):
        ref(i)
    reps = []
    for rep in range(5):
        t0 = time.perf_counter()
        for i in range(n_ref):
            ref(rep * n_ref + i)
        reps.append(n_ref / (time.perf_counter() - t0))
    ref_lps = float(np.median(reps))
    log(f"cpu-ref repeats (latents/s): {[round(r, 2) for r in reps]}")

    # CPU-opt: batches of 2 latents per core until the budget is spent; the first batch is also the parity sample
    batch = min(max(16, 2 * cores), int(q[0].shape[0]))     # (at least one batch also when --latents is small)
    done, t_opt, first, used = 0, 0.0, None, cores
    while (t_opt < n_opt_budget_s or done == 0) and done + batch <= q[0].shape[0]:
        hb = [t[done:done + batch].cpu().numpy() for t in q]
        t0 = time.perf_counter()
        idx, samp, used = O.encode_tensors_omp(*hb, SEED, OMEGA, S, BEAMS, BLOCK_SIZE, n_threads=cores)
        t_opt += time.perf_counter() - t0
        if first is None:
            first = (idx, samp)
        done += batch
    return {"ref_lps": ref_lps, "ref_threads": threads, "ref_probe": probe, "ref_reps": reps, "n_ref": n_ref,
            "opt_lps": done / max(t_opt, 1e-9), "opt_threads": used, "opt_latents": done, "opt_first": first}


def secondary_config(eng, device, name, omega, eps1, beams, n_tensors, n_dims, reps, check, ref_line, block_size=BLOCK_SIZE, max_K=32):
    """One further BASELINE configuration, outside the timed headline: `reps` calls of irec_beam_encode on `n_tensors`
    synthetic tensors of `n_dims` dims (blocks of 1000) by HIP events; kernel from irec_encode_plan; look-ups per clock per
    CU; the first `check` tensors compared with the oracle (indices and sample, bit for bit).
    max_K = 32 as in the headline and as irec.BeamSearchCoder issues its calls (its first hint; until r04x 48, which made every
    call launch the -- empty -- second pass behind the 32-step table window)."""
    import torch
    from oracle import oracle as O
    S = int(np.exp(omega * eps1))
    params = eng.params(omega, S, beams, table_steps=max_K if max_K > 32 else 0)
    q = synthetic_batch(n_tensors, device, 77, n_dims)
    lay = eng.layout(n_tensors, n_dims, block_size, SEED)
    out = (torch.empty(lay.n_blocks, dtype=torch.int32, device=device),
           torch.empty((lay.n_blocks, max_K), dtype=torch.int32, device=device), torch.empty_like(q[0]))
    plan = eng.plan(params, lay, max_K)
    for _ in range(2):
        eng.encode_blocks(params, lay, *q, SEED, max_K, out=out)
    torch.cuda.synchronize(device)
    # calls of a fraction of a millisecond: twenty of them are over before the clocks have settled after the host-side check of the
    # configuration before (round 6: 302 blocks read 0.195 ms in a burst of 20, 0.185 ms from the 30th call on, profiles/r06end/).  Such calls
    # are issued for 20 ms before the timed ones, and at least 20 ms of them are timed.
    t_est = time.perf_counter()
    eng.encode_blocks(params, lay, *q, SEED, max_K, out=out)
    torch.cuda.synchronize(device)
    t_est = max(time.perf_counter() - t_est, 1e-5)
    n_settle = int(min(400, max(0, np.ceil(0.02 / t_est))))
    reps = int(min(400, max(reps, np.ceil(0.02 / t_est))))
    for _ in range(n_settle):
        eng.encode_blocks(params, lay, *q, SEED, max_K, out=out)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    ev[0].record()
    for r in range(reps):
        eng.encode_blocks(params, lay, *q, SEED, max_K, out=out)
        ev[r + 1].record()
    torch.cuda.synchronize(device)
    ms = float(np.median([ev[r].elapsed_time(ev[r + 1]) for r in range(reps)]))
    # the same call with the proposal tables kept across calls (IREC_FLAG_REUSE_TABLES, the Python coder's default: the tables
    # depend on (seed, S, D) only, so the 24 residual blocks of an image, or the images of a run, build them once)
    import irec
    keep = eng.params(omega, S, beams, irec._lib.IREC_FLAG_REUSE_TABLES, table_steps=params.table_steps)
    for _ in range(2):
        eng.encode_blocks(keep, lay, *q, SEED, max_K, out=out)
    ev[0].record()
    for r in range(reps):
        eng.encode_blocks(keep, lay, *q, SEED, max_K, out=out)
        ev[r + 1].record()
    torch.cuda.synchronize(device)
    ms_keep = float(np.median([ev[r].elapsed_time(ev[r + 1]) for r in range(reps)]))
    Kh = out[0].cpu().numpy().astype(np.int64)
    assert Kh.min() >= 0 and Kh.max() <= max_K, (name, int(Kh.min()), int(Kh.max()))
    dims = lay.block_dim.cpu().numpy().astype(np.int64)
    evals = float((S * dims * (1 + np.maximum(Kh - 1, 0) * beams) * (Kh > 0)).sum())
    lookups = evals / (ms * 1e-3) / (plan["n_cu"] * plan["clock_mhz"] * 1e6)
    checked = 0
    if check:
        c = min(check, n_tensors)
        hb = [t[:c].cpu().numpy() for t in q]
        if block_size is None:   # (no shuffle without a block size, coder.py:415-419: the per-tensor entry of the checker)
            per = [O.encode_tensor(*(h[i] for h in hb), SEED, omega, S, beams, block_size=None) for i in range(c)]
            ridx, rsamp = [[p_[0]] for p_ in per], np.stack([p_[1].reshape(-1) for p_ in per])
        else:
            ridx, rsamp, _ = O.encode_tensors_omp(*hb, SEED, omega, S, beams, block_size, max_K=max_K, n_threads=host_cores())
        ih, sh = out[1].cpu().numpy(), out[2][:c].cpu().numpy()
        bpt = lay.blocks_per_tensor
        for i in range(c):
            for j in range(bpt):
                row = lay.natural[i * bpt + j]
                assert ih[row, :Kh[row]].tolist() == ridx[i][j], f"{name}: parity, tensor {i} block {j}"
            assert np.array_equal(sh[i], rsamp[i]), f"{name}: parity, tensor {i} sample"
        checked = c
    # how close this configuration's selections are (irec_beam_encode_ex on the same batch, outside the timing; same outputs asserted)
    margins = None
    if n_tensors >= 64:
        K2, idx2, samp2, mg = eng.encode_blocks_margins(params, lay, *q, SEED, max_K)
        torch.cuda.synchronize(device)
        assert torch.equal(K2, out[0]) and torch.equal(samp2, out[2]), f"{name}: margins call differs from the timed call"
        m = mg.cpu().numpy().astype(np.float64)
        small = np.minimum(m[:, 0], m[:, 2])
        margins = {"kernel": eng.plan(params, lay, max_K, margins=True)["kernel"], "blocks": int(m.shape[0]),
                   "blocks_closer_than_1e-5": float((small < 1e-5).mean()), "blocks_closer_than_6.5e-5": float((small < 6.5e-5).mean()),
                   "exact_ties": int((small == 0).sum()), "median_closest_comparison": float(np.median(small[np.isfinite(small)]))}
        del K2, idx2, samp2, mg
    res = {"name": name, "reference": ref_line, "omega": omega, "extra_samples": eps1, "n_beams": beams, "n_samples": S,
           "tensors_per_call": n_tensors, "dims_per_tensor": n_dims, "block_size": block_size, "blocks_per_call": int(lay.n_blocks), "kernel": plan["kernel"],
           "grid": plan["grid"], "teams_sharing_a_row": plan["split"], "ms_per_call": ms, "ms_per_call_tables_kept": ms_keep,
           "tensors_per_s": n_tensors / (ms * 1e-3),
           "lookups_per_clk_per_cu": lookups, "mean_K": float(Kh.mean()), "oracle_checked_tensors": checked, "margins": margins}
    log(f"secondary {name}: {plan['kernel']} {ms:.3f} ms/call ({ms_keep:.3f} with the tables kept), {res['tensors_per_s']:.0f} tensors/s, {lookups:.2f} look-ups/clk/CU, "
        f"oracle-checked {checked}")
    del q, out
    return res


class PowerSampler:
    """Socket power of the busiest card over the timed region, from the amdgpu hwmon files (microwatts; no GPU call, no subprocess): both batch
    kernels run the card at its power cap (profiles/r06end/power_probe.log), which is what the look-ups per NOMINAL clock and the
    5 % between the pool's boxes are to be read against.  None where the files are not there."""

    def __init__(self, period=0.25, pci=None):
        """pci: "dddd:bb:dd" of the card to read (torch's device properties); without a match, the card that drew most"""
        import glob
        self.files = sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_average") or
                            glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_input"))
        if pci:
            mine = [f for f in self.files if pci.lower() in os.path.realpath(f.split("/hwmon/")[0]).lower()]
            self.files = mine or self.files
        self.period, self.samples, self._stop, self._thread = period, [], None, None

    @staticmethod
    def _read(path):
        try:
            with open(path) as f:
                return float(f.read().strip()) * 1e-6
        except (OSError, ValueError):
            return None

    def start(self):
        if not self.files:
            return self
        import threading
        self._stop = threading.Event()

        def loop():
            while not self._stop.wait(self.period):
                self.samples.append([self._read(f) for f in self.files])
        self._thread = threading.Thread(target=loop, daemon=True)
        self._thread.start()
        return self

    def stop(self):
        """{"power_w": mean over the region of the card that drew most, "power_cap_w": that card's cap, "samples": n} or None"""
        if self._thread is None:
            return None
        self._stop.set()
        self._thread.join(timeout=2.0)
        if not self.samples:
            return None
        means = []
        for i in range(len(self.files)):
            v = [row[i] for row in self.samples if row[i] is not None]
            means.append(sum(v) / len(v) if v else -1.0)
        i = int(np.argmax(means))
        if means[i] <= 0:
            return None
        cap = self._read(self.files[i].rsplit("/", 1)[0] + "/power1_cap")
        return {"power_w": means[i], "power_cap_w": cap, "samples": len(self.samples), "card": self.files[i].split("/")[4]}


def run_rank_launch_only(args):
    """IREC_BENCH_LAUNCH_ONLY=1 (CPU test rigs, tests/test_bench_launcher.py): everything of the N-rank job except the GPU
    work -- rendezvous, the code-length exchange of irec/sharding.py over gloo on made-up K, max-over-ranks timing, the
    JSON line.  Never a measurement: "value" is null."""
    import torch
    import torch.distributed as dist
    from irec import sharding
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        dist.init_process_group(backend="gloo")
    L = 4
    local = torch.arange(L, dtype=torch.float64) * world + rank          # item i = rank + k * world  ->  value i
    t0 = time.perf_counter()
    got = sharding.gather_per_item(local, world * L, rank, world, dist if world > 1 else None)
    mine = time.perf_counter() - t0
    assert got.tolist() == list(range(world * L)), got
    times = [mine]
    if world > 1:
        tall = [torch.zeros(1, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(tall, torch.tensor([mine], dtype=torch.float64))
        times = [float(t.item()) for t in tall]
    if rank == 0:
        # the keys the driver reads from the real line, so that a rehearsal of its command checks the contract end to end;
        # per-rank "rates" are the made-up items over the exchange time (never a measurement: value stays null), and neither
        # the secondary configurations nor the CPU baselines run -- as in the real job at N > 1
        print(json.dumps({"metric": "encoded latents/sec", "value": None, "unit": "latents/s", "launch_only": True, "n_gpus": world,
                          "world_size": dist.get_world_size() if world > 1 else 1, "backend": "gloo",
                          "steps": args.steps, "warmup": args.warmup, "ranks_timed": len(times), "higher_is_better": True,
                          "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                          "per_rank_latents_per_s": [L * args.steps / max(t, 1e-9) for t in times],
                          "config": {"workload": "launch-only rehearsal (no GPU work)", "parallelism":
                                     f"latents sharded over {world} rank(s), no data-path collective"}}), flush=True)
    if world > 1:
        dist.destroy_process_group()


def run_rank(args):
    if os.environ.get("IREC_BENCH_LAUNCH_ONLY") == "1":
        return run_rank_launch_only(args)
    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device; there is no CPU fallback")
    n_dev = torch.cuda.device_count()
    backend = os.environ.get("IREC_DIST_BACKEND", "nccl")   # "nccl" IS RCCL on ROCm
    if backend == "nccl" and world > n_dev:
        raise SystemExit(f"bench.py: {world} ranks but {n_dev} GPU(s) visible (IREC_DIST_BACKEND=gloo shares devices on test rigs)")
    dev_index = local_rank % n_dev
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    dist = None
    # (IREC_BENCH_FORCE_DIST=1: test rigs run the collective code path -- process group, barrier, all_gathers over RCCL --
    #  with a world of ONE rank on the one GPU they have; tests/test_bench_launcher.py)
    if world > 1 or os.environ.get("IREC_BENCH_FORCE_DIST") == "1":
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=device)
        else:
            dist.init_process_group(backend=backend)
        if dist.get_world_size() != args.gpus:
            raise SystemExit(f"bench.py: process group has {dist.get_world_size()} ranks, --gpus {args.gpus}")

    import irec
    from irec import sharding
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
    plan = eng.plan(params, lay, max_K)        # the kernels irec_beam_encode launches for exactly this call

    coll_dev = device if backend == "nccl" else torch.device("cpu")   # gloo rigs exchange through host tensors

    def barrier():
        torch.cuda.synchronize(device)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(device)

    def exchange(K):
        # the path's only exchange step: per-latent code length (nats), all ranks <- all ranks (RCCL over xGMI)
        return sharding.gather_per_item(sharding.code_nats_per_tensor(K, lay, S).to(coll_dev), world * L, rank, world, dist)

    log(f"rank {rank}/{world}: {L} latents, {lay.n_blocks} blocks per step, kernel {plan['kernel']} grid {plan['grid']}")
    for _ in range(max(args.warmup, 1)):
        eng.encode_blocks(params, lay, *q, SEED, max_K, out=out)
        exchange(out[0])
    barrier()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    power = None
    if world == 1 and rank == 0:
        try:
            pr = torch.cuda.get_device_properties(device)
            pci = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}"
        except Exception:
            pci = None
        power = PowerSampler(pci=pci).start()
    t0 = time.perf_counter()
    for a, b in ev:
        a.record()
        eng.encode_blocks(params, lay, *q, SEED, max_K, out=out)   # same stream as the events (torch current stream)
        b.record()
        gathered = exchange(out[0])                                # every step ends with the path's exchange (no host sync over RCCL)
    K = out[0]
    barrier()
    my_elapsed = time.perf_counter() - t0
    power = power.stop() if power is not None else None
    elapsed = my_elapsed
    per_rank = [L * args.steps / my_elapsed]
    if dist is not None:
        tall = [torch.zeros(1, dtype=torch.float64, device=coll_dev) for _ in range(world)]
        dist.all_gather(tall, torch.tensor([my_elapsed], dtype=torch.float64, device=coll_dev))
        times = [float(t.item()) for t in tall]
        elapsed = max(times)                                  # MAX over ranks
        per_rank = [L * args.steps / t for t in times]

    kernel_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))   # table kernels + block kernel of one call
    log(f"rank {rank}: timed region {my_elapsed:.3f} s, {kernel_ms:.2f} ms per call (HIP events)")
    Kh = K.cpu().numpy().astype(np.int64)
    dims = lay.block_dim.cpu().numpy().astype(np.int64)
    assert Kh.min() >= 0 and Kh.max() <= max_K, "a block needed more than max_K partitions"
    algo_bytes = int((24 * dims + 4 * Kh).sum())                        # 24 D + 4 K per block
    evals = int((S * dims * (1 + np.maximum(Kh - 1, 0) * BEAMS) * (Kh > 0)).sum())
    n_cu, clk_ghz = plan["n_cu"], plan["clock_mhz"] / 1e3
    lookups = evals / (kernel_ms * 1e-3) / (n_cu * clk_ghz * 1e9)       # per clock per CU at the device's max clock
    # HBM-side traffic per launch: the committed PMC profile (FETCH_SIZE / WRITE_SIZE passes, corrected as
    # MI355X_MICROARCH.md prescribes) counts only if it was taken on THESE kernel sources
    src_hash = kernel_source_hash()
    traffic, traffic_note = None, "no profiles/traffic.json"
    pmc = {"lds_conflict_frac": None, "lds_busy": None, "valu_busy": None}   # (PMC passes of the same sources, or null)
    tj = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tj):
        tr = json.load(open(tj))
        # (the profiler prints a template's bool arguments, irec_encode_plan names only those that are set)
        def norm(name):   # "irec::encode_team_kernel<20, 3, 1, false, ...": no namespace, spaces, unset bool arguments or (truncated) tail
            return name.split("::")[-1].replace(" ", "").replace(",false", "").rstrip(",>")
        if tr.get("source_sha16") == src_hash and norm(tr.get("kernel", "")) == norm(plan["kernel"]):
            traffic = tr["hbm_bytes_per_latent"] * L
            traffic_note = tr.get("source", "")
            pmc = {k: tr.get(k) for k in ("lds_conflict_frac", "lds_busy", "valu_busy")}
        else:
            traffic_note = (f"profiles/traffic.json is for sources {tr.get('source_sha16')} / {tr.get('kernel')}; "
                            f"this build is {src_hash} / {plan['kernel']}: not reported")

    # decoder leg (outside the timed region): the batch just coded is decoded from its indices -- at the full size the
    # round trip must be exact (decode == the encoder's sample, bit for bit), and the decoder's own rate is reported
    samp_dec = eng.decode_blocks(params, lay, q[2], q[3], SEED, K, out[1])
    torch.cuda.synchronize()
    round_trip_exact = bool(torch.equal(samp_dec, out[2]))
    # per-call events, median: the first call after `samp_dec` is held may pay the allocator's hipMalloc of a fresh output
    dev_ = [torch.cuda.Event(enable_timing=True) for _ in range(6)]
    dev_[0].record()
    for i in range(5):
        eng.decode_blocks(params, lay, q[2], q[3], SEED, K, out[1])
        dev_[i + 1].record()
    torch.cuda.synchronize()
    decode_ms = sorted(dev_[i].elapsed_time(dev_[i + 1]) for i in range(5))[2]
    assert round_trip_exact, "decode(encode) differs from the encoder's sample"

    result = {
        "metric": "encoded latents/sec", "value": world * L * args.steps / elapsed, "unit": "latents/s",
        "n_gpus": world, "world_size": dist.get_world_size() if dist is not None else 1,
        "backend": backend if dist is not None else None, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "RVAE Cifar10-shape latents [16,16,32], beam_search B=20 Omega=3 eps=0.2 (S=36), "
                               "block_size=1000 (configs[1])",
                   "latents_per_step_per_gpu": L, "blocks_per_step_per_gpu": int(lay.n_blocks),
                   "parallelism": f"latents sharded over {world} GPU(s), no data-path collective"},
        "per_rank_latents_per_s": per_rank,
        "roofline": {"bound": "hbm", "achieved": algo_bytes / (kernel_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": algo_bytes / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": traffic,
                     "traffic_note": traffic_note, "kernel": plan["kernel"], "table_kernel": plan["table_kernel"],
                     "grid": plan["grid"], "waves_per_wg": plan["waves_per_wg"], "lds_bytes": plan["lds_bytes"],
                     "kernel_ms": kernel_ms, "algorithmic_bytes": algo_bytes, "source_sha16": src_hash,
                     # "report all three fractions" -- the contractual one above is a property of the algorithm
                     # (~2e3 op/B); what BINDS the kernel is the LDS gather pipe (look-ups per clock per CU at the device's maximum clock)
                     "binding": {"pipe": "lds_gather", "unit": "look-ups/clk/CU", "achieved": lookups,
                                 "hw_peak": LDS_HW_LOOKUPS, "frac_hw": lookups / LDS_HW_LOOKUPS,
                                 "microbench_peak": LDS_STREAM_LOOKUPS, "frac_microbench": lookups / LDS_STREAM_LOOKUPS,
                                 "lds_conflict_frac": pmc["lds_conflict_frac"], "lds_busy": pmc["lds_busy"], "valu_busy": pmc["valu_busy"],
                                 "power_w": power["power_w"] if power else None, "power_cap_w": power["power_cap_w"] if power else None,
                                 "note": "microbench_peak: the kernel's own address mixture and instruction stream at 12 waves per CU "
                                         "(scripts/microbench/bank_limits.hip, profiles/r05b); PMC fractions from profiles/traffic.json "
                                         "when it was taken on these sources, else null; power_w: socket power over the timed region "
                                         "(amdgpu hwmon) -- the kernel runs the card at its cap, so 'achieved' per NOMINAL clock understates "
                                         "the rate per delivered clock (DESIGN.md 4)"},
                     "configs": []},
        "secondary": {"proposal_evals_per_s": evals / (kernel_ms * 1e-3), "n_cu": n_cu, "clock_ghz": clk_ghz,
                      "lookups_per_clk_per_cu": lookups,
                      # hardware rate of the instruction the look-ups use (conflict-free ds_read_b32)
                      "lds_hw": {"achieved": lookups, "peak": LDS_HW_LOOKUPS, "unit": "look-ups/clk/CU",
                                 "frac": lookups / LDS_HW_LOOKUPS},
                      # what random addresses allow: scripts/microbench/gather_rates.hip (profiles/r01j), 2-choice bank
                      # assignment over three table copies, 8 waves per CU
                      "lds_gather_roofline": {"achieved": lookups, "peak": LDS_2CHOICE_LOOKUPS,
                                              "unit": "look-ups/clk/CU", "frac": lookups / LDS_2CHOICE_LOOKUPS},
                      "mean_K": float(Kh.mean()), "code_nats_per_latent": float(gathered.mean().item()),
                      "decode_round_trip_exact": round_trip_exact, "decode_ms": decode_ms,
                      "decoded_latents_per_s": L / (decode_ms * 1e-3)},
    }
    # decoder: its own roofline entry (12 D + 4 K algorithmic bytes per block)
    dec_bytes = int((12 * dims + 4 * Kh).sum())
    result["secondary"]["decode_roofline"] = {"bound": "hbm", "achieved": dec_bytes / (decode_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                                              "unit": "GB/s", "frac": dec_bytes / (decode_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                              "algorithmic_bytes": dec_bytes, "kernel_ms": decode_ms,
                                              "kernel": "decode_tensor_kernel<true,3> (whole tensors staged in LDS) + 2 x alpha_table_kernel",
                                              "note": "VALU-bound on the IEEE chain a = rho (var_p - c), sqrt, c += a of "
                                                      "beam_search_coder.py:131-147 (DESIGN.md §4)"}
    if rank == 0 and world == 1 and not args.no_secondary:
        # BASELINE configs[3] and configs[4] (and the lossy settings on a latent batch), outside the timed region
        del samp_dec
        torch.cuda.empty_cache()
        lossy = "examples/lossy/compress_with_lossy_model.py:120-124,222-227 (B = 10, Omega = 3, S = 20)"
        stress = "BASELINE.json configs[4] (ImageNet32 RVAE stress: B = 30, Omega = 5)"
        sec = []
        sec.append(secondary_config(eng, device, "configs[3] Kodak level 2 (one image: 12 288 dims = 13 blocks)", 3.0, 1.0, 10, 1, 12288, 20, 1, lossy))
        sec.append(secondary_config(eng, device, "configs[3] Kodak level 1 (one image: 301 056 dims = 302 blocks)", 3.0, 1.0, 10, 1, 301056, 20, 1, lossy))
        sec.append(secondary_config(eng, device, "configs[3] settings, 1024 latents of 8192 dims", 3.0, 1.0, 10, 1024, N_DIMS, 10, 16, lossy))
        sec.append(secondary_config(eng, device, "configs[4] S = 148, 1024 latents of 8192 dims", 5.0, 1.0, 30, 1024, N_DIMS, 5, 16, stress))
        sec.append(secondary_config(eng, device, "configs[4] S = 403 (eps = 0.2), 1024 latents of 8192 dims", 5.0, 1.2, 30, 1024, N_DIMS, 3, 16, stress))
        result["secondary"]["configs"] = sec
        # calls of fewer blocks than team slots: the literal per-call shapes of configs[2] (one image's residual block: 9 blocks; one
        # GPU's share of 300 images: 38 latents = 342 blocks per call), compression_performance.py:305-347, resnet_vae.py:821-826
        rvae = "BASELINE.json configs[1]/[2] settings (B = 20, Omega = 3, S = 36)"
        result["secondary"]["midsize"] = [
            secondary_config(eng, device, "one GPU's share of config 3: 38 latents = 342 blocks per call", OMEGA, EPS1, BEAMS, 38, N_DIMS, 30, 2, rvae),
            secondary_config(eng, device, "14 latents = 126 blocks per call (about half a block per CU: every row shared between teams)", OMEGA, EPS1, BEAMS, 14, N_DIMS, 30, 2, rvae),
            secondary_config(eng, device, "one image's residual block: 1 latent = 9 blocks per call", OMEGA, EPS1, BEAMS, 1, N_DIMS, 30, 1, rvae)]
        # blocks of more than 1024 dims (coder.py:29-36,415-419: block_size is the caller's, None = the whole tensor as one block)
        big = "Coder.__init__(block_size=...) beyond 1024 dims (rec/coding/coder.py:29-36,415-419), headline settings"
        result["secondary"]["large_blocks"] = [
            secondary_config(eng, device, "block_size = 2048, 512 latents of 8192 dims", OMEGA, EPS1, BEAMS, 512, N_DIMS, 5, 2, big, block_size=2048, max_K=128),
            # (one 8192-dim block holds a team for ~40 ms: four blocks per team slot of the three-team build, 768 slots)
            secondary_config(eng, device, "block_size = None (one 8192-dim block per latent), 3072 latents", OMEGA, EPS1, BEAMS, 3072, N_DIMS, 3, 2, big, block_size=None, max_K=128),
            # (round 5: calls of few such blocks are coded by gangs of teams -- chunk owners x sample stripes over the CUs)
            secondary_config(eng, device, "block_size = None, ONE latent per call (1 block of 8192 dims: a gang of 72 teams)", OMEGA, EPS1, BEAMS, 1, N_DIMS, 10, 1, big, block_size=None, max_K=128),
            secondary_config(eng, device, "block_size = None, one image's 24 latents per call (24 blocks: gangs of 8 teams)", OMEGA, EPS1, BEAMS, 24, N_DIMS, 5, 1, big, block_size=None, max_K=128)]
        # the driver keeps `roofline`: the other BASELINE configurations and the mid-size calls, compactly (full entries: secondary.*)
        result["roofline"]["configs"] = [{"name": c["name"], "kernel": c["kernel"], "ms_per_call": c["ms_per_call"],
                                          "lookups_per_clk_per_cu": c["lookups_per_clk_per_cu"], "tensors_per_s": c["tensors_per_s"]}
                                         for c in sec + result["secondary"]["midsize"]]
        result["secondary"]["skewed_K"] = skewed_K_leg(eng, device, 8192, 5, 16)
        result["secondary"]["margins"] = margins_leg(eng, device, params, lay, q, out, S, max_K)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import oracle as O
        cb = cpu_baselines(q, args.cpu_ref_latents, args.cpu_opt_seconds)
        # parity of the timed GPU outputs against the oracle on the first CPU-opt batch
        idx_h = out[1].cpu().numpy()
        samp_h = out[2].cpu().numpy()
        ridx, rsamp = cb["opt_first"]
        for i in range(len(ridx)):
            for j in range(lay.blocks_per_tensor):
                row = lay.natural[i * lay.blocks_per_tensor + j]
                assert idx_h[row, :Kh[row]].tolist() == ridx[i][j], f"parity: latent {i} block {j}"
            assert np.array_equal(samp_h[i], rsamp[i]), f"parity: latent {i} sample"
        result["cpu_baseline"] = {"value": cb["ref_lps"], "unit": "latents/s", "cores": cb["ref_threads"], "kind": "port",
                                  "sample": f"median of 5 repeats x {cb['n_ref']} latents of the same batch after 3 warm-ups, "
                                            "reference-shaped torch-eager restatement of the TF path "
                                            "(oracle/ref_shaped_torch.py, BASELINE.md §3 CPU-ref)",
                                  "thread_probe_latents_per_s": {str(k): v for k, v in cb["ref_probe"].items()},
                                  "repeats": cb["ref_reps"], "host_cores": host_cores()}
        result["cpu_baseline_opt"] = {"value": cb["opt_lps"], "unit": "latents/s", "cores": cb["opt_threads"], "kind": "port",
                                      "sample": f"{cb['opt_latents']} latents, oracle/irec_oracle.c canonical mode, OpenMP over "
                                                "blocks (BASELINE.md §3 CPU-opt)"}
        result["parity_checked_latents"] = len(ridx)
    if rank == 0:
        print(json.dumps(result), flush=True)
    if dist is not None:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--latents", type=int, default=65536, help="latent tensors per step per GPU (65 536: 589 824 blocks, "
                    "~378 ms per step -- a timed region of 7.6 s at the driver's 20 steps, so that its 5-second device "
                    "samples cannot miss it; the rate per latent is the same from 8192 latents up, profiles/archive/r03g/batch_size.log)")
    ap.add_argument("--cpu-ref-latents", type=int, default=20, help="latents per CPU-ref repeat (5 repeats, median)")
    ap.add_argument("--cpu-opt-seconds", type=float, default=8.0, help="time budget of the CPU-opt (OpenMP oracle) leg")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the configs[3] / configs[4] runs after the timed region")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ:
        if args.gpus > 1:
            return launch_ranks(args, sys.argv[1:])     # before anything touches a GPU
    elif int(os.environ["WORLD_SIZE"]) != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={os.environ['WORLD_SIZE']}")
    run_rank(args)


if __name__ == "__main__":
    main()
