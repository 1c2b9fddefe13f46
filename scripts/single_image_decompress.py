"""Diagnostic: latency of ONE 32x32 image through the 24-block RVAE shim, compress and decompress (eager, N = 1), and whether
the decompressed reconstruction equals the compress pass's.  Same model as scripts/config3_harness.py.
Usage: python scripts/single_image_decompress.py [--block-size-none] [--no-split]"""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "relative-entropy-coding_amd"), os.path.join(ROOT, "scripts")]
from config3_harness import build_model
BS = None if "--block-size-none" in sys.argv else 1000      # (--block-size-none: the reference's default Coder block_size -- one block per latent)
m = build_model(torch.device("cuda"), block_size=BS)
if "--no-split" in sys.argv:                                 # (every block on one team: what the gangs of round 5 replace)
    for b in m.residual_blocks:
        b.coder.no_split = True
g = torch.Generator().manual_seed(7)
images = (torch.rand(10, 3, 32, 32, generator=g) - 0.5).cuda()
tc, td, same = [], [], True
for i in range(10):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    idx, rec = m.compress(images[i:i + 1], seed=42)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    out = m.decompress(idx, 42, images[i:i + 1].shape)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    tc.append(t1 - t0); td.append(t2 - t1)
    same = same and torch.equal(out, rec)
med = lambda v: sorted(v)[len(v) // 2]
print(f"single image, block_size {BS}{', no_split' if '--no-split' in sys.argv else ''}, eager: compress {1e3 * med(tc[2:]):.2f} ms, decompress {1e3 * med(td[2:]):.2f} ms (median of 8), reconstruction identical: {same}")
