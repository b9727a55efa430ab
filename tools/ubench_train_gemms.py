"""Per-shape timing of the training GEMMs in the float16 mode (development tool): forward / grad_input (sgemm_nt) and weight gradient
(sgemm_tn) at the edge-row shapes of one block, against the time their operand traffic needs at 5 TB/s.

    python tools/ubench_train_gemms.py            # on the GPU box
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from moldiff_amd import train_ops as T  # noqa: E402

DEV = 'cuda:0'


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3   # us


def main():
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 154666
    print(f'rows {M}; us per call (traffic floor at 5 TB/s)')
    with T.precision('fp16'):
        for K, N in [(64, 64), (64, 128), (128, 128), (128, 64), (64, 256), (256, 256), (256, 64), (80, 64), (64, 32), (32, 64)]:
            x = torch.randn(M, K, device=DEV).half()
            w = torch.randn(N, K, device=DEV)
            b = torch.randn(N, device=DEV)
            gy = torch.randn(M, N, device=DEV).half()
            wt = w.t().contiguous()
            floor = M * (K + N) * 2 / 5e12 * 1e6
            t_f = timeit(lambda: T.sgemm_nt(x, w, b))
            t_d = timeit(lambda: T.sgemm_nt(gy, wt, out_dtype=torch.float16))
            sp = T._splits_for(M, N, K, True)
            t_w = timeit(lambda: T.sgemm_tn(gy, x, sp, want_bias=True))
            print(f'K {K:4d} N {N:4d}: forward {t_f:7.1f}  grad_input {t_d:7.1f}  weight_grad {t_w:7.1f} (splits {sp})   floor {floor:6.1f}')


if __name__ == '__main__':
    main()
