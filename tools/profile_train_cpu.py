"""Is the training step bound by the GPU or by the host issuing it?  (development tool)

    python tools/profile_train_cpu.py [precision]      # on the GPU box

Prints the wall time of a step, the time the host needs to ISSUE one (no synchronisation: if it is close to the wall time the GPU
waits for the host), and a cProfile of three steps by own time.
"""
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
from moldiff_amd.harness import default_config, GEOM_DRUGS_MEAN_ATOMS, GEOM_DRUGS_STD_ATOMS  # noqa: E402
from moldiff_amd.trainer import Trainer  # noqa: E402


def main():
    prec = sys.argv[1] if len(sys.argv) > 1 else 'fp16'
    dev = torch.device('cuda:0')
    np.random.seed(2920)
    sizes = np.maximum(np.random.normal(GEOM_DRUGS_MEAN_ATOMS, GEOM_DRUGS_STD_ATOMS, size=256).astype('int64'), 2)
    model = M.MolDiff(default_config('MolDiff'), 8, 6)
    model.load_state_dict(M.recipe_state_dict(model, 20230807), strict=True)
    model = model.to(dev).train()
    tr = Trainer(model, precision=prec)
    batch = bench.clean_batch([int(s) for s in sizes], 100, dev)
    for _ in range(3):
        tr.step(*batch)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        tr.step(*batch)
    t_issue = (time.perf_counter() - t0) / 5
    torch.cuda.synchronize()
    t_wall = (time.perf_counter() - t0) / 5
    print(f'{prec}: wall {t_wall * 1e3:.2f} ms per step, host issue {t_issue * 1e3:.2f} ms per step')
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(3):
        tr.step(*batch)
    torch.cuda.synchronize()
    pr.disable()
    st = pstats.Stats(pr)
    st.sort_stats('tottime').print_stats(35)


if __name__ == '__main__':
    main()
