#!/usr/bin/env python3
"""BASELINE.json configs[3] end to end on ONE GPU: the two-level lossy shim on Kodak-size (768 x 512) images, B = 10, Omega = 3,
eps = 0 (S = 20), block_size 1000, max_index 20 -- compress_with_lossy_model.py:36-37,54 -- compress -> .rec -> decompress.

Level 2 is [1, 8, 12, 128] (12 288 dims, 13 blocks), level 1 [1, 32, 48, 196] (301 056 dims, 302 blocks); the two coder calls
are sequential (level 1's prior is computed from the level-2 sample).  Weights are random-init (no checkpoint exists offline):
the numbers are throughput / round-trip evidence, not rate-distortion results.  Prints one JSON line.
"""
import argparse
import json
import os
import sys
import tempfile
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "relative-entropy-coding_amd")]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--images", type=int, default=6)
    ap.add_argument("--height", type=int, default=512)
    ap.add_argument("--width", type=int, default=768)
    ap.add_argument("--scale", type=float, default=0.2,
                    help="multiplier on the random-init head weights: 0.2 keeps q near p (K = 1 per block); ~1 gives tens to "
                         "hundreds of partitions per block, the regime of a trained model at 0.2-1 bpp")
    args = ap.parse_args()
    import irec
    from irec.models import Large2LevelVAE
    torch.manual_seed(3)
    m = Large2LevelVAE().cuda().eval()
    with torch.no_grad():   # keep the random-init posteriors near the priors (K of a few per block, as a trained model has)
        for mod in (m.analysis_transform[-1], m.hyper_analysis_transform[-1], m.hyper_synthesis_transform[-1],
                    m._prior_loc_head, m._prior_log_scale_head, m._level_1_posterior_loc_combiner,
                    m._level_1_posterior_log_scale_combiner):
            mod.weight.mul_(args.scale)
    sampler = irec.BeamSearchCoder(kl_per_partition=3., n_beams=10, extra_samples=1., block_size=1000)
    g = torch.Generator().manual_seed(11)
    out_dir = tempfile.mkdtemp(prefix="irec_cfg4_")
    rows = []
    calls = []
    orig = sampler.encode

    def spy(target, coder, seed, **kw):   # device time of each coder call (events on the current stream)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = orig(target, coder, seed, **kw)
        e1.record()
        calls.append((tuple(target.loc.shape), e0, e1))
        return out
    sampler.encode = spy
    dcalls = []
    orig_dec = sampler.decode

    def dspy(coder, indices, seed, **kw):   # wall time of each decode call (host list -> tensor conversion included), device time
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); t0 = time.perf_counter(); e0.record()
        out = orig_dec(coder, indices=indices, seed=seed, **kw)
        e1.record(); torch.cuda.synchronize()
        dcalls.append((1e3 * (time.perf_counter() - t0), e0.elapsed_time(e1)))
        return out
    sampler.decode = dspy
    for i in range(args.images + 1):      # image 0 is the warm-up (MIOpen, tables, workspaces)
        image = torch.rand(args.height, args.width, 3, generator=g).cuda() - 0.5
        path = os.path.join(out_dir, f"kodak_{i}.rec")
        calls.clear()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        recon = m.compress(path, image, seed=42, sampler=sampler, block_size=1000, max_index=20)
        torch.cuda.synchronize(); t_c = time.perf_counter() - t0
        coder_ms = [(shape, e0.elapsed_time(e1)) for shape, e0, e1 in calls]
        dcalls.clear()
        t0 = time.perf_counter()
        recon2 = m.decompress(path, sampler)
        torch.cuda.synchronize(); t_d = time.perf_counter() - t0
        seed, shape, bs, block_indices = irec.io.read_compressed_code(path)
        if i:
            rows.append({"compress_ms": 1e3 * t_c, "decompress_ms": 1e3 * t_d, "coder_ms": [round(ms, 3) for _, ms in coder_ms],
                         "decode_wall_ms": [round(w, 3) for w, _ in dcalls], "decode_dev_ms": [round(dv, 3) for _, dv in dcalls],
                         "coder_shapes": [list(s) for s, _ in coder_ms], "round_trip": bool(torch.equal(recon, recon2)),
                         "file_bytes": os.path.getsize(path), "blocks": [len(b) for b in block_indices],
                         "K_max": max(len(ix) for b in block_indices for ix in b),
                         "indices": sum(len(ix) for b in block_indices for ix in b)})
    med = lambda k: sorted(r[k] for r in rows)[len(rows) // 2]   # noqa: E731
    pix = args.height * args.width
    print(json.dumps({
        "config": f"{args.images} images {args.width}x{args.height}, two-level lossy shim, B=10 Omega=3 eps=0 (S=20), block_size 1000 (configs[3])",
        "coder_calls_per_image": rows[0]["coder_shapes"], "blocks_per_call": rows[0]["blocks"],
        "compress_ms_median": med("compress_ms"), "decompress_ms_median": med("decompress_ms"),
        "coder_ms_per_call_median": [sorted(r["coder_ms"][k] for r in rows)[len(rows) // 2] for k in range(len(rows[0]["coder_ms"]))],
        "decode_wall_ms_per_call": rows[-1]["decode_wall_ms"], "decode_device_ms_per_call": rows[-1]["decode_dev_ms"],
        "all_round_trips_exact": all(r["round_trip"] for r in rows),
        "file_bytes_median": med("file_bytes"), "bpp_median": 8.0 * med("file_bytes") / pix,
        "indices_per_image_median": med("indices"), "K_max": max(r["K_max"] for r in rows), "head_weight_scale": args.scale, "images_per_s_compress": 1e3 / med("compress_ms")}), flush=True)


if __name__ == "__main__":
    main()
