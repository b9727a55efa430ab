"""Host-side profile of the fp16 training step (development tool, GPU box): cProfile over the forward (get_loss) and, separately,
wall time of forward / backward / optimizer per step -- where the ~22 ms of Python per step go.  Usage: python tools/profile_train_host.py"""
import cProfile
import os
import pstats
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import moldiff_amd as M  # noqa: E402
from moldiff_amd import train_ops  # noqa: E402
from moldiff_amd.harness import default_config, GEOM_DRUGS_MEAN_ATOMS, GEOM_DRUGS_STD_ATOMS  # noqa: E402
from moldiff_amd.trainer import Trainer  # noqa: E402


def main():
    dev = torch.device('cuda:0')
    np.random.seed(2920)
    sizes = np.maximum(np.random.normal(GEOM_DRUGS_MEAN_ATOMS, GEOM_DRUGS_STD_ATOMS, size=256).astype('int64'), 2)
    model = M.MolDiff(default_config('MolDiff'), 8, 6)
    model.load_state_dict(M.recipe_state_dict(model, 20230807), strict=True)
    model = model.to(dev).train()
    tr = Trainer(model, lr=1e-4, betas=(0.99, 0.999), weight_decay=1e-8, max_grad_norm=50.0, precision='fp16')
    batch = bench.clean_batch([int(s) for s in sizes], 100, dev)
    for _ in range(6):
        tr.step(*batch)
    torch.cuda.synchronize()
    # phases, host time only (no sync inside)
    from moldiff_amd.diffusion import deferred_class_checks
    acc = {'fwd': 0.0, 'bwd': 0.0, 'opt': 0.0}
    n = 10
    for _ in range(n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        tr.zero_grad(); tr.check_deferred(); tr.wt.refresh()
        with train_ops.grad_sink(tr.flat), train_ops.transposed_params(tr.wt):
            with train_ops.precision(tr.precision), deferred_class_checks() as chk:
                out = model.get_loss(*batch)
            tr._deferred = chk.finish(group=None)
            t1 = time.perf_counter()
            (out['loss'] * tr.state[0]).backward()
            t2 = time.perf_counter()
            train_ops.flush_grad_sink()
            from moldiff_amd import _lib
            f = tr.flat
            _lib.check(_lib.lib().mdx_op_amp_adamw(_lib.ptr(f.data), _lib.ptr(f.grad), _lib.ptr(tr.m), _lib.ptr(tr.v), f.numel, tr.lr, 0.99, 0.999, 1e-8, 1e-8,
                                                    50.0, _lib.ptr(tr.state), tr.growth[0], tr.growth[1], tr.growth[2], _lib.ptr(tr.ws), _lib.stream()))
            tr._stale()
        t3 = time.perf_counter()
        acc['fwd'] += t1 - t0; acc['bwd'] += t2 - t1; acc['opt'] += t3 - t2
    print('host ms per step: forward %.2f  backward %.2f  flush+optimizer %.2f' % tuple(1e3 * acc[k] / n for k in ('fwd', 'bwd', 'opt')))
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(3):
        tr.step(*batch)
    pr.disable()
    torch.cuda.synchronize()
    st = pstats.Stats(pr, stream=sys.stdout)
    st.sort_stats('tottime').print_stats(45)
    st.sort_stats('cumulative').print_stats(40)


if __name__ == '__main__':
    main()
