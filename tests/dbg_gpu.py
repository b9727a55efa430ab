import sys, traceback
sys.path.insert(0, '.')
import numpy as np, torch
from tests import util as U
from oracle import moldiff_oracle as O
DEV='cuda:0'
g = U.gold('blocks_full.npz')
bn, hei, bh, ei, be = U.graph_from_sizes([5,7])
x, ea, pos, tg = U.t32(g['x']), U.t32(g['edge_attr']), U.t32(g['pos']), torch.from_numpy(g['t'])
m = U.moldiff('MolDiff', DEV)
nt = (tg[bn].unsqueeze(-1)/1000).float(); et = (tg[be].unsqueeze(-1)/1000).float()
for i in (0,3):
    try:
        out = m.denoiser.node_blocks_with_edge[i](x.to(DEV), ei.to(DEV), ea.to(DEV), nt.to(DEV)); torch.cuda.synchronize()
        print('nodeblock', i, U.maxdiff(out, g[f'nodeblock{i}_out']), float(np.abs(g[f'nodeblock{i}_out']).max()))
        out = m.denoiser.edge_blocks[i](ea.to(DEV), ei.to(DEV), x.to(DEV), et.to(DEV)); torch.cuda.synchronize()
        print('edgeblock', i, U.maxdiff(out, g[f'edgeblock{i}_out']), float(np.abs(g[f'edgeblock{i}_out']).max()))
        rel = pos[ei[0]]-pos[ei[1]]; dist = torch.norm(rel, dim=-1)
        out = m.denoiser.pos_blocks[i](x.to(DEV), ea.to(DEV), ei.to(DEV), rel.to(DEV), dist.to(DEV), et.to(DEV)); torch.cuda.synchronize()
        print('posupdate', i, U.maxdiff(out, g[f'posupdate{i}_out']), float(np.abs(g[f'posupdate{i}_out']).max()))
    except Exception as e:
        traceback.print_exc()
