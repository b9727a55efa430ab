"""node_kernel duration against the number of 16-node tiles it is launched on (development tool; GPU box).

    python tools/ubench_node_tiles.py            ->  table on stdout (kept under profiles/)

The node kernel's unit is a 16-node tile on a 4-wave workgroup, two workgroups per CU (mdx_node.hip).  A tile costs 15.5 MFLOP =
25 us of one CU's four f32 matrix pipes at the peak; the launch takes as long as its busiest CU.  This measures that directly:
batches of 16-atom molecules give exactly `tiles` full tiles (no partial tile, no other change), and the kernel is timed by
hipEvents inside a sampling chain (mdx_profile_*).  If the time steps up from <= 256 tiles (one per CU) to 257..512 (two on some
CUs) and stays flat in between, the launch time at the bench workload (393 tiles) is set by tile granularity, not by anything
inside a tile; the per-tile efficiency is the 256-tile figure against 25 us.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import moldiff_amd._lib as _lib  # noqa: E402
if os.environ.get('MDX_LIB'):
    _lib.LIB_PATH = os.path.abspath(os.environ['MDX_LIB'])   # an alternative build (tools/build_variant.sh)
import bench  # noqa: E402
import moldiff_amd as M  # noqa: E402
from moldiff_amd.harness import default_config, placeholder_from_sizes  # noqa: E402


def main():
    dev = torch.device('cuda:0')
    model = M.MolDiff(default_config('MolDiff_simple'), 8, 6).eval()
    model.load_state_dict(M.recipe_state_dict(model, 20230807), strict=True)
    model = model.to(dev)
    print(f'{"tiles":>6s} {"nodes":>7s} {"edges":>8s} {"node_kernel us":>15s} {"per tile-round us":>18s}   (7 launches per step)')
    for tiles in (64, 128, 192, 256, 257, 320, 393, 448, 512, 513, 640, 768):
        sizes = np.full(tiles, 16, dtype=np.int64)           # 16-atom molecules: one tile each
        ph = placeholder_from_sizes(sizes, dev)
        sm = model.sampler(len(sizes), ph['batch_node'], ph['halfedge_index'], ph['batch_halfedge'], seed=1, return_traj=False)
        sm.init()
        _, prof = bench.run_chain(sm, 20, 3, torch.cuda.synchronize, prof=2 << 2)   # events on the node kernel only
        n, ms = prof['node']
        us = ms / n * 1e3
        rounds = -(-tiles // 512) if tiles > 256 else 1
        per = us / (2 if 256 < tiles <= 512 else rounds * (2 if tiles > 512 else 1))
        print(f'{tiles:6d} {int(sizes.sum()):7d} {int((sizes * (sizes - 1)).sum()):8d} {us:15.1f} {per:18.1f}')
        del sm
    # the bench workload itself
    m2, ph, sizes = bench.build_workload(256, 0, dev)
    sm = m2.to(dev).sampler(256, ph['batch_node'], ph['halfedge_index'], ph['batch_halfedge'], seed=1, return_traj=False)
    sm.init()
    _, prof = bench.run_chain(sm, 20, 3, torch.cuda.synchronize, prof=2 << 2)
    n, ms = prof['node']
    N = int(sizes.sum())
    print(f'bench workload: {N} nodes = {-(-N // 16)} tiles: {ms / n * 1e3:.1f} us per launch')


if __name__ == '__main__':
    main()
