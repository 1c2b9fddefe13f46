#!/usr/bin/env python3
"""Diagnostic: the model shim's convolutions (16x16 latent grid, 160 -> 288 / 160 -> 224 / 192 -> 160 channels, 3x3) in NCHW
and channels_last, eager and inside a HIP graph, with the deterministic algorithms the compress / decompress passes pin."""
import time
import torch
import torch.nn.functional as F
torch.backends.cudnn.deterministic, torch.backends.cudnn.benchmark = True, False
dev = torch.device("cuda")
for n in (1, 38):
    for cin, cout in ((160, 288), (160, 224), (192, 160), (160, 160)):
        for fmt_name, fmt in (("nchw", torch.contiguous_format), ("nhwc", torch.channels_last)):
            x = torch.randn(n, cin, 16, 16, device=dev).contiguous(memory_format=fmt)
            w = torch.randn(cout, cin, 3, 3, device=dev).contiguous(memory_format=fmt)
            b = torch.randn(cout, device=dev)
            for _ in range(5):
                y = F.conv2d(x, w, b, padding=1)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                for _ in range(20):
                    y = F.conv2d(x, w, b, padding=1)
            g.replay(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                g.replay()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 200
            print(f"n={n:3d} {cin}->{cout} {fmt_name}: {dt * 1e6:7.1f} us per conv (graph), out strides {tuple(y.stride())}", flush=True)
