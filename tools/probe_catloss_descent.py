"""GPU probe (round 6): one fp16 step's flat gradient with the fused categorical loss tail on / off, and the 12-step loss
sequences of tests/test_gpu_trainer.py::test_fp16_autocast_training_steps_reduce_the_loss for both."""
import copy, sys
import torch
sys.path.insert(0, '.')
from tests import util as U
from tests.test_gpu_trainer import _tiny_batch, DEV
from moldiff_amd.trainer import Trainer
import moldiff_amd.model as MM

base = U.moldiff('MolDiff_simple', DEV)
batch = _tiny_batch(7)
t = torch.tensor([120, 480, 700, 930], device=DEV)
g = U.rng(8)
N, Eh = batch[1].shape[0], batch[3].shape[0]
noise = dict(eps_pos=U.t32(g.standard_normal((N, 3))).to(DEV), u_node=U.t32(g.random((N, 8))).to(DEV), u_halfedge=U.t32(g.random((Eh, 6))).to(DEV))
grads, seqs = {}, {}
for fused in (False, True):
    MM._FUSED_LOSS = fused
    for lr in (0.0, 2e-4, 1e-4):
        m = copy.deepcopy(base)
        for mod in m.modules():
            if hasattr(mod, '_eng'):
                mod._eng, mod._eng_sig = None, None
        tr = Trainer(m, lr=lr, max_grad_norm=50.0, precision='fp16', init_scale=1024.0)
        ls = []
        for it in range(12 if lr else 1):
            out = tr.step(*batch, time_step=t, noise=noise)
            ls.append(round(float(out['loss']), 4))
        if lr == 0.0:
            grads[fused] = tr.flat.grad.clone()
            print('fused', fused, 'grad norm', float(out['grad_norm']))
        else:
            print('fused', fused, 'lr', lr, ls)
a, b = grads[False], grads[True]
print('grad rel L2', float((a - b).norm() / a.norm()), 'max', float((a - b).abs().max()), 'scale', float(a.abs().max()))
