"""cProfile of the Python body of the two most frequent training operators (development tool; GPU box)."""
import cProfile
import os
import pstats
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from moldiff_amd import train_ops as T  # noqa: E402


def main():
    dev = torch.device('cuda:0')
    x = torch.randn(4096, 64, device=dev, dtype=torch.float16, requires_grad=True)
    w = torch.randn(64, 64, device=dev, requires_grad=True)
    b = torch.randn(64, device=dev, requires_grad=True)
    g = torch.randn(64, device=dev, requires_grad=True)
    with T.precision('fp16'):
        for name, fn in (('linear fwd', lambda: T.linear(x, w, b)), ('ln_relu fwd', lambda: T.ln_relu(x, g, b))):
            for _ in range(100):
                fn()
            torch.cuda.synchronize()
            pr = cProfile.Profile()
            pr.enable()
            for _ in range(3000):
                fn()
            pr.disable()
            torch.cuda.synchronize()
            print('=====', name, '(3000 calls)')
            pstats.Stats(pr).sort_stats('tottime').print_stats(18)

        def fb():
            out = T.linear(x, w, b)
            out.backward(out.detach())
        for _ in range(50):
            fb()
        torch.cuda.synchronize()
        pr = cProfile.Profile()
        pr.enable()
        for _ in range(1000):
            fb()
        pr.disable()
        print('===== linear fwd + bwd (1000 calls; the backward body runs on the autograd thread and is not in this profile)')
        pstats.Stats(pr).sort_stats('tottime').print_stats(12)


if __name__ == '__main__':
    main()
