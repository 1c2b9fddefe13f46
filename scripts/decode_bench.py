"""Decoder timing on the bench workload (BASELINE configs[1]: 8192-dim latents, B = 20, S = 36): the three decode paths of
irec_decode.hip by HIP events, each checked against the encoder's sample; optional gather ablation (blocks in natural order:
perm = None) to price the shuffled 4-byte gathers.  Usage: python scripts/decode_bench.py [--latents 8192] [--reps 20]"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "relative-entropy-coding_amd")]
import irec  # noqa: E402


def synth(n_t, n, seed=5000):
    g = torch.Generator(device="cuda"); g.manual_seed(seed)
    mp = torch.randn((n_t, n), generator=g, device="cuda")
    lsp = 0.25 * torch.randn((n_t, n), generator=g, device="cuda")
    sp = torch.exp(lsp)
    mq = mp + sp * 0.2 * torch.randn((n_t, n), generator=g, device="cuda")
    sq = torch.exp(lsp - (0.05 * torch.randn((n_t, n), generator=g, device="cuda")).abs())
    return mq, sq, mp, sp


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--latents", type=int, default=8192)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--omega", type=float, default=3.0)
    ap.add_argument("--eps1", type=float, default=1.2)
    ap.add_argument("--beams", type=int, default=20)
    a = ap.parse_args()
    eng = irec.get_engine()
    n, bs = 8192, 1000
    S = int(np.exp(a.omega * a.eps1))
    mq, sq, mp, sp = synth(a.latents, n)
    lay = eng.layout(a.latents, n, bs, 42)
    params = eng.params(a.omega, S, a.beams)
    K, idx, sample = eng.encode_blocks(params, lay, mq, sq, mp, sp, 42, 16)
    torch.cuda.synchronize()
    Ksum = int(K.sum()); dims = a.latents * n
    alg = 12 * dims + 4 * Ksum
    print(f"latents {a.latents}, blocks {lay.n_blocks}, sum K {Ksum}, algorithmic bytes {alg / 1e6:.1f} MB", flush=True)
    for mode in ("tensors", "tensors_fused", "tables", "fused", "legacy"):
        rec = eng.decode_blocks(params, lay, mp, sp, 42, K, idx, mode=mode)
        ok = bool(torch.equal(rec, sample))
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(a.reps + 1)]
        ev[0].record()
        for r in range(a.reps):
            eng.decode_blocks(params, lay, mp, sp, 42, K, idx, mode=mode)
            ev[r + 1].record()
        torch.cuda.synchronize()
        ts = sorted(ev[r].elapsed_time(ev[r + 1]) for r in range(a.reps))
        med = ts[len(ts) // 2]
        print(f"decode[{mode:7s}] exact={ok} median {med:.3f} ms min {ts[0]:.3f} ms -> {a.latents / med / 1e3:.2f} M latents/s, "
              f"{alg / med / 1e6:.0f} GB/s algorithmic = {alg / med / 1e6 / 8000:.3f} of 8 TB/s", flush=True)
    # gather ablation: same blocks in natural order (no shuffle): coalesced reads and writes, same arithmetic
    lay0 = eng.layout(a.latents, n, bs, 42)
    import copy
    nat = copy.copy(lay0); nat.perm = None
    for mode in ("tables", "fused"):
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        eng.decode_blocks(params, nat, mp, sp, 42, K, idx, mode=mode)
        ev0.record()
        for r in range(a.reps):
            eng.decode_blocks(params, nat, mp, sp, 42, K, idx, mode=mode)
        ev1.record(); torch.cuda.synchronize()
        print(f"decode[{mode}] with perm = None (coalesced, wrong samples by design): {ev0.elapsed_time(ev1) / a.reps:.3f} ms", flush=True)


if __name__ == "__main__":
    main()
