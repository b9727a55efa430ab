"""Fused Linear + LayerNorm launch against the two launches, per shape (development tool; GPU box).  MDX_LIB = alternative build."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import moldiff_amd._lib as _lib  # noqa: E402
if os.environ.get('MDX_LIB'):
    _lib.LIB_PATH = os.path.abspath(os.environ['MDX_LIB'])
from moldiff_amd import train_ops as T  # noqa: E402


def timed(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    dev = torch.device('cuda:0')
    M = 154666
    print(f'{"K":>4s} {"N":>4s} {"fused us":>9s} {"linear us":>10s} {"ln us":>7s}')
    for K, N in ((64, 256), (256, 256), (128, 128), (64, 32), (64, 64), (256, 64)):
        x = torch.randn(M, K, device=dev).half()
        w = torch.randn(N, K, device=dev) * K ** -0.5
        b, g, be = torch.randn(N, device=dev), torch.rand(N, device=dev) + 0.5, torch.randn(N, device=dev)
        with torch.no_grad(), T.precision('fp16'):
            tf = timed(lambda: T._LinearLnRelu.apply(x, w, b, None, g, be))
            pre = T.linear(x, w, b)
            tl = timed(lambda: T.linear(x, w, b))
            tn = timed(lambda: T.ln_relu(pre, g, be, True))
        print(f'{K:4d} {N:4d} {tf:9.1f} {tl:10.1f} {tn:7.1f}', flush=True)


if __name__ == '__main__':
    main()
