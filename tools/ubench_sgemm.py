"""Stand-alone timing of the training GEMM kernels on the layer shapes of one NodeEdgeNet block (development tool)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from moldiff_amd import train_ops as T  # noqa: E402

E = 154666
dev = 'cuda:0'
prec = sys.argv[1] if len(sys.argv) > 1 else 'f32'
T.precision(prec).__enter__()
print('GEMM operand precision:', prec)
for (M, K, N) in [(E, 256, 256), (E, 64, 256), (E, 256, 64), (E, 64, 64), (E, 128, 128), (E, 80, 64), (6279, 256, 960)]:
    a = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev); g = torch.randn(M, N, device=dev)
    for name, fn, flops, byts in (('nt  fwd ', lambda: T.sgemm_nt(a, w), 2.0 * M * K * N, 4.0 * (M * K + M * N)),
                                  ('tn wgrad', lambda: T.sgemm_tn(g, a, T._splits_for(M, N, K)), 2.0 * M * K * N, 4.0 * (M * K + M * N))):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 20
        print(f'{name} M={M} K={K} N={N}: {dt * 1e6:8.1f} us  {flops / dt / 1e12:6.1f} TFLOP/s  {byts / dt / 1e12:5.2f} TB/s (min traffic)')
