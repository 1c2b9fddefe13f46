"""Per-image compress latency of the RVAE-shaped shim (24 residual blocks, 160/32 filters, random-init weights):
24 strictly sequential coder.encode calls of 9 blocks each -- the reference's per-image `comp_time` situation
(examples/lossless/compression_performance.py:345-378).  Diagnostic, single image, batch 1."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "relative-entropy-coding_amd")]
from irec.models import BidirectionalResNetVAE

torch.manual_seed(0)
m = BidirectionalResNetVAE(num_res_blocks=24, sampler="beam_search", sampler_args={"n_beams": 20, "extra_samples": 1.2},
                           coder_args={"block_size": 1000}, deterministic_filters=160, stochastic_filters=32,
                           kl_per_partition=3.)
with torch.no_grad():
    for b in m.residual_blocks:
        for head in (b.gen_posterior_loc_head, b.gen_posterior_log_scale_head, b.infer_posterior_loc_head,
                     b.infer_posterior_log_scale_head, b.prior_loc_head, b.prior_log_scale_head):
            head.weight.mul_(0.05)
m = m.cuda().eval()
img = torch.rand(1, 3, 32, 32, device="cuda") - 0.5
for it in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    idx, rec = m.compress(img, seed=42)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    nidx = sum(len(ix) for bi in idx for ix in bi)
    print(f"compress: {dt * 1e3:.1f} ms per 32x32 image, {nidx} indices ({nidx * 3.58 / 0.693 / 3072:.2f} bits/dim at ln36 nats each)", flush=True)
torch.cuda.synchronize(); t0 = time.perf_counter()
rec2 = m.decompress(idx, seed=42, image_shape=img.shape)
torch.cuda.synchronize(); print(f"decompress: {(time.perf_counter() - t0) * 1e3:.1f} ms; identical reconstruction: {torch.equal(rec, rec2)}")
