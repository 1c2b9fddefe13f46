"""Diagnostic: the per-kernel cost of ONE 32x32 image through the 24-block RVAE shim (eager, N = 1) -- run under
rocprofv3 --kernel-trace --stats.  Same model as scripts/config3_harness.py."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "relative-entropy-coding_amd"), os.path.join(ROOT, "scripts")]
from config3_harness import build_model
m = build_model(torch.device("cuda"))
g = torch.Generator().manual_seed(7)
images = (torch.rand(12, 3, 32, 32, generator=g) - 0.5).cuda()
for i in range(12):
    idx, rec = m.compress(images[i:i + 1], seed=42)
torch.cuda.synchronize()
print("K per block of the last image:", [len(ix) for bi in idx[:3] for ix in bi], "...")
