"""Host cost of ONE layer operator call (development tool): 2,000 forward calls of train_ops.linear on a small float16 batch under
cProfile (the GPU work is negligible: what is measured is Python + ctypes + launch)."""
import cProfile
import os
import pstats
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from moldiff_amd import train_ops as T  # noqa: E402

dev = 'cuda:0'
x = torch.randn(2048, 64, device=dev).half().requires_grad_(True)
w = torch.randn(256, 64, device=dev, requires_grad=True)
b = torch.randn(256, device=dev, requires_grad=True)
with T.precision('fp16'):
    for _ in range(100):
        y = T.linear(x, w, b)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(2000):
        y = T.linear(x, w, b)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    print(f'linear forward: {(t1 - t0) / 2000 * 1e6:.1f} us per call (host)')
    with torch.no_grad():
        t0 = time.perf_counter()
        for _ in range(2000):
            y = T.sgemm_nt(x.detach(), w.detach(), b.detach())
        t1 = time.perf_counter()
        torch.cuda.synchronize()
    print(f'sgemm_nt alone: {(t1 - t0) / 2000 * 1e6:.1f} us per call (host)')
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(2000):
        y = T.linear(x, w, b)
    pr.disable()
    torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats('tottime').print_stats(14)
