"""Host time of the phases of one training step (development tool): forward issue, backward issue, optimizer; no synchronisation in
between, so each number is how long the HOST needs to issue that phase.   python tools/profile_train_phases.py [precision]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import moldiff_amd as M  # noqa: E402
from moldiff_amd import train_ops  # noqa: E402
from moldiff_amd.harness import default_config  # noqa: E402
from moldiff_amd.trainer import Trainer  # noqa: E402

prec = sys.argv[1] if len(sys.argv) > 1 else 'fp16'
dev = torch.device('cuda:0')
np.random.seed(2920)
sizes = np.maximum(np.random.normal(24.923464980477522, 5.516291901819105, size=256).astype('int64'), 2)
model = M.MolDiff(default_config('MolDiff'), 8, 6)
model.load_state_dict(M.recipe_state_dict(model, 20230807), strict=True)
model = model.to(dev).train()
tr = Trainer(model, precision=prec)
batch = bench.clean_batch([int(s) for s in sizes], 100, dev)
for _ in range(3):
    tr.step(*batch)
torch.cuda.synchronize()
acc = np.zeros(4)
n = 5
for _ in range(n):
    t0 = time.perf_counter()
    tr.zero_grad()
    tr.wt.refresh()
    with train_ops.grad_sink(tr.flat), train_ops.transposed_params(tr.wt):
        with train_ops.precision(prec):
            out = model.get_loss(*batch)
        t1 = time.perf_counter()
        (out['loss'] * tr.state[0]).backward()
        t2 = time.perf_counter()
        train_ops.flush_grad_sink()
    tr.zero_grad()
    t3 = time.perf_counter()
    torch.cuda.synchronize()
    t4 = time.perf_counter()
    acc += [t1 - t0, t2 - t1, t3 - t2, t4 - t3]
print(f'{prec}: host ms per step -- forward issue {acc[0] / n * 1e3:.2f}, backward issue {acc[1] / n * 1e3:.2f}, sink flush {acc[2] / n * 1e3:.2f}, '
      f'GPU still busy after the host is done {acc[3] / n * 1e3:.2f}')
