"""Accuracy and speed of an fp32 GEMM emulated on the bf16 matrix pipe with three-way operand splits (experiment).

    make -C moldiff_amd/csrc clean && make -C moldiff_amd/csrc -j8 EXTRA=-DMDX_EXPERIMENTAL OUT=../libmoldiff_hip_exp.so
    make -C moldiff_amd/csrc clean && make -C moldiff_amd/csrc -j8
    python tools/ubench_bf16x3.py
"""
import ctypes
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import moldiff_amd._lib as _lib  # noqa: E402

_lib.LIB_PATH = os.path.join(ROOT, 'moldiff_amd', 'libmoldiff_hip_exp.so')
from moldiff_amd import train_ops as T  # noqa: E402

L = _lib.lib()
L.mdx_debug_hgemm3_nt.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64,
                                  ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p]


def x3(a, w):
    out = torch.empty(a.shape[0], w.shape[0], device=a.device)
    _lib.check(L.mdx_debug_hgemm3_nt(_lib.ptr(a), a.stride(0), _lib.ptr(w), w.stride(0), _lib.ptr(out), out.stride(0), a.shape[0],
                                     w.shape[0], a.shape[1], _lib.stream()))
    return out


def timeit(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 20


torch.manual_seed(0)
for (M, K, N) in [(154666, 256, 256), (154666, 64, 256), (20000, 256, 64)]:
    a = torch.randn(M, K, device='cuda') * torch.exp(torch.randn(M, 1, device='cuda'))       # rows of very different scale
    w = torch.randn(N, K, device='cuda') / K ** 0.5
    ref = (a[:4096].double() @ w.double().T)
    err = lambda y: float(((y[:4096].double() - ref).abs().max(dim=1).values / ref.abs().max(dim=1).values).max())
    y32, y3 = T.sgemm_nt(a, w), x3(a, w)
    with T.precision('bf16'):
        y16 = T.sgemm_nt(a, w)
    fl = 2.0 * M * K * N
    t32, t3 = timeit(lambda: T.sgemm_nt(a, w)), timeit(lambda: x3(a, w))
    print(f'M={M} K={K} N={N}: max row-relative error  fp32 MFMA {err(y32):.2e} | bf16x3 {err(y3):.2e} | bf16 {err(y16):.2e};  '
          f'time fp32 {t32 * 1e6:.0f} us ({fl / t32 / 1e12:.0f} TF) | bf16x3 {t3 * 1e6:.0f} us ({fl / t3 / 1e12:.0f} TF-equivalent)')
