"""Kernel-trace subject: the 342-block call (38 latents, B = 20, S = 36) 50 times with the tables kept, MODE=listed|cost (GPU box)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "relative-entropy-coding_amd")]
import irec
from oracle import oracle as O
eng = irec.get_engine()
L, n, bs, B, omega, eps1 = int(os.environ.get("LATENTS", "38")), int(os.environ.get("N_DIMS", "8192")), 1000, int(os.environ.get("BEAMS", "20")), 3.0, float(os.environ.get("EPS1", "1.2"))
S = int(np.exp(omega * eps1))
st = [O.synthetic_latent(1234 + i, n) for i in range(L)]
q = [torch.from_numpy(np.stack([s[k] for s in st])).cuda().contiguous() for k in range(4)]
lay = eng.layout(L, n, bs, 42)
fl = (0 if os.environ.get("AS_ISSUED") else irec._lib.IREC_FLAG_REUSE_TABLES) | (irec._lib.IREC_FLAG_LISTED_ORDER if os.environ.get("MODE", "cost") == "listed" else 0)
params = eng.params(omega, S, B, fl)
for _ in range(50):
    eng.encode_blocks(params, lay, *q, 42, int(os.environ.get("MAXK", "48")))
torch.cuda.synchronize()
print("done", eng.plan(params, lay, 48)["kernel"])
