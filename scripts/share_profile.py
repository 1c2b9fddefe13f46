"""Diagnostic: kernel timeline of one GPU's share of config 3 (38 images through the 24-block RVAE shim) -- run under
rocprofv3 --kernel-trace.  LANES = sub-batches replayed side by side (GraphedCompress), 0 = the eager pass."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "relative-entropy-coding_amd"), os.path.join(ROOT, "scripts")]
from config3_harness import build_model
from irec.models import GraphedCompress
N = int(os.environ.get("IMAGES", "38"))
lanes = int(os.environ.get("LANES", "2"))
m = build_model(torch.device("cuda"))
g = torch.Generator().manual_seed(7)
images = (torch.rand(N, 3, 32, 32, generator=g) - 0.5).cuda()
run = (lambda: m.compress(images, seed=42)) if lanes == 0 else GraphedCompress(m, tuple(images.shape), seed=42, lanes=lanes)
fn = run if lanes == 0 else (lambda: run(images))
for i in range(4):
    fn()
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for i in range(5):
    fn()
torch.cuda.synchronize()
print(f"lanes {lanes}: {(time.perf_counter() - t0) / 5 * 1e3:.2f} ms per {N}-image pass")
