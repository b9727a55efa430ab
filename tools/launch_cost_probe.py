"""Host cost of the pieces of one operator call on the training path (development tool; GPU box)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from moldiff_amd import _lib  # noqa: E402
from moldiff_amd import train_ops as T  # noqa: E402


def per_call(fn, n=20000):
    for _ in range(200):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    t = (time.perf_counter() - t0) / n * 1e6
    torch.cuda.synchronize()
    return t


def main():
    dev = torch.device('cuda:0')
    L = _lib.lib()
    a = torch.randn(64, device=dev)
    b = torch.randn(64, device=dev)
    o = torch.empty(64, device=dev)
    pa, pb, po = a.data_ptr(), b.data_ptr(), o.data_ptr()
    st = _lib.stream()
    print('raw C-ABI call, tiny element-wise launch (hipLaunchKernel + ctypes): %.2f us' % per_call(lambda: L.mdx_op_ew_fwd(0, pa, pb, po, 64, st)))
    print('  + stream() + check():                                             %.2f us' % per_call(lambda: _lib.check(L.mdx_op_ew_fwd(0, pa, pb, po, 64, _lib.stream()))))
    print('torch.empty(64):                                                     %.2f us' % per_call(lambda: torch.empty(64, device=dev)))
    print('torch.empty(154666, 256, half):                                      %.2f us' % per_call(lambda: torch.empty(154666, 256, dtype=torch.float16, device=dev), 5000))
    print('torch add (a + b), tiny:                                             %.2f us' % per_call(lambda: a + b))
    x = torch.randn(2048, 64, device=dev, requires_grad=True)
    y = torch.randn(2048, 64, device=dev, requires_grad=True)
    print('T.add forward (autograd node, tiny rows):                            %.2f us' % per_call(lambda: T.add(x, y), 5000))
    w = torch.randn(64, 64, device=dev, requires_grad=True)
    bb = torch.randn(64, device=dev, requires_grad=True)
    print('T.linear forward fp32 (autograd node, 2048 x 64 -> 64):              %.2f us' % per_call(lambda: T.linear(x, w, bb), 5000))

    def fb():
        out = T.linear(x, w, bb)
        out.backward(out.detach())
        x.grad = y.grad = w.grad = bb.grad = None
    print('T.linear forward + backward (no sink):                               %.2f us' % per_call(fb, 3000))


if __name__ == '__main__':
    main()
