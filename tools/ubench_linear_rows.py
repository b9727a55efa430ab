"""Row-owner Linear (csrc/mdx_linear_rows.hip) against the LDS-tile SGEMM (sgemm_nt_kernel) per layer shape, M = 154,666 rows.
    python tools/ubench_linear_rows.py        # on the GPU box"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from moldiff_amd import train_ops as T

dev = 'cuda:0'
M = 154666
print('%5s %5s %6s | %9s %9s | %9s %9s   (us; TFLOP/s)' % ('K', 'N', 'trans', 'rows', 'TF', 'tile', 'TF'))
for K, N in [(256, 256), (64, 256), (256, 64), (64, 64), (80, 64), (64, 128), (128, 128), (128, 64), (64, 32), (32, 64)]:
    for trans in (False, True):
        x = torch.randn(M, K, device=dev)
        w = torch.randn(K, N, device=dev) if trans else torch.randn(N, K, device=dev)
        def run(fn, n=20):
            for _ in range(3): fn()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(n): fn()
            torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
        a = run(lambda: T.linear_rows(x, w, trans))
        b = run(lambda: T.sgemm_nt(x, T.transpose(w) if trans else w))
        fl = 2.0 * M * K * N / 1e12
        print('%5d %5d %6s | %9.1f %9.1f | %9.1f %9.1f' % (K, N, trans, a, fl / (a * 1e-6), b, fl / (b * 1e-6)))
