"""Which torch (aten) operators does one fp16 training step still run, and from where?  (development tool; GPU box)"""
import os
import sys

import numpy as np
import torch
from torch.profiler import profile, ProfilerActivity

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import moldiff_amd as M  # noqa: E402
from moldiff_amd.harness import default_config, GEOM_DRUGS_MEAN_ATOMS, GEOM_DRUGS_STD_ATOMS  # noqa: E402
from moldiff_amd.trainer import Trainer  # noqa: E402


def main():
    dev = torch.device('cuda:0')
    np.random.seed(2920)
    sizes = np.maximum(np.random.normal(GEOM_DRUGS_MEAN_ATOMS, GEOM_DRUGS_STD_ATOMS, size=256).astype('int64'), 2)
    model = M.MolDiff(default_config('MolDiff'), 8, 6)
    model.load_state_dict(M.recipe_state_dict(model, 20230807), strict=True)
    model = model.to(dev).train()
    tr = Trainer(model, precision='fp16')
    batch = bench.clean_batch([int(s) for s in sizes], 100, dev)
    for _ in range(4):
        tr.step(*batch)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU], with_stack=True) as prof:
        tr.step(*batch)
        torch.cuda.synchronize()
    print(prof.key_averages().table(sort_by='self_cpu_time_total', row_limit=40, max_name_column_width=60))
    print(prof.key_averages(group_by_stack_n=6).table(sort_by='self_cpu_time_total', row_limit=45, max_name_column_width=50, max_src_column_width=110))


if __name__ == '__main__':
    main()
