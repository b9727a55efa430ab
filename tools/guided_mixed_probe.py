"""Guided step with the denoiser on the exact fp32 path and the guidance bond predictor on the split float16 path (development tool)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def run(model_path, bp_path, steps=100):
    dev = torch.device('cuda:0')
    model, ph, sizes = bench.build_workload(256, 0, dev, 'MolDiff')
    model = model.to(dev)
    bp = bench.build_bond_predictor().to(dev)
    model.matrix_path, bp.matrix_path = model_path, bp_path
    sm = model.sampler(256, ph['batch_node'], ph['halfedge_index'], ph['batch_halfedge'], seed=2023, return_traj=False,
                       bond_predictor=bp, guidance=['uncertainty', 1e-4])
    sm.init()
    el, _ = bench.run_chain(sm, steps, 5, torch.cuda.synchronize, prof=0)
    st = sm.state()
    return el / steps * 1e3, st['pos'].clone(), st['h_node'].argmax(-1).clone(), st['h_halfedge'].argmax(-1).clone()


def main():
    res = {}
    for name, (a, b) in {'exact': ('exact_f32', 'exact_f32'), 'mixed': ('exact_f32', 'split_f16'), 'split': ('split_f16', 'split_f16')}.items():
        res[name] = run(a, b)
        print(f'{name:6s} {res[name][0]:7.3f} ms/step', flush=True)
    e = res['exact']
    for name in ('mixed', 'split'):
        r = res[name]
        print(f'{name} vs exact after 105 steps: max |dpos| {float((r[1] - e[1]).abs().max()):.3e}, atom classes differ '
              f'{int((r[2] != e[2]).sum())} of {e[2].numel()}, bond classes differ {int((r[3] != e[3]).sum())} of {e[3].numel()}')


if __name__ == '__main__':
    main()
