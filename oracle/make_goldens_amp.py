"""Golden vectors for the reference's MIXED-PRECISION training step (scripts/train_drug3d.py:86-109, use_amp: True in every
train config): `MolDiff.get_loss` under ``torch.autocast`` + a scaled backward, run with the REAL reference on CPU.
BUILD-CONTAINER ONLY (needs /root/reference).

The reference trains with ``torch.autocast(device_type='cuda', dtype=torch.float16)`` and ``GradScaler``.  There is no CUDA here, so
the same model code is run under CPU autocast, once with float16 and once with bfloat16 (the autocast op lists are the same: Linear
in the low-precision type with fp32 accumulation, LayerNorm / softmax / losses in fp32).  Inputs and pinned draws are those of
tests/golden/loss.npz ('full' and 'simple' cases of oracle/make_goldens_loss.py); the loss is multiplied by a fixed scale before
``backward()`` and the gradients divided by it afterwards, as GradScaler.scale / unscale_ do.

Saved (tests/golden/loss_amp.npz): the four loss values and, per parameter, the L2 norm of its gradient (plus the full tensor where
it has <= 256 elements) for each low-precision type.  The product's `precision='fp16' | 'bf16'` training mode is compared against
these with the tolerance stated in tests/test_loss.py.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle import moldiff_oracle as O  # noqa: E402
from oracle import ref_shim  # noqa: E402
from oracle.make_goldens_loss import pinned_randomness, SEED_MOLDIFF  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), 'tests', 'golden')
SCALE = 1024.0


def main():
    torch.set_num_threads(8)
    MolDiff, BondPredictor, G, TR, DF, CM = ref_shim.load()
    gold = np.load(os.path.join(OUT, 'loss.npz'))
    out = {'scale': np.float32(SCALE)}
    for nm, yml in (('full', 'configs/train/train_MolDiff.yml'), ('simple', 'configs/train/train_MolDiff_simple.yml')):
        cfg = ref_shim.load_yaml_cfg(yml)
        m = MolDiff(cfg.model, 8, 6).eval()
        sd = m.state_dict()
        shapes = {k: tuple(v.shape) for k, v in sd.items() if not O.is_frozen_key(k)}
        sd.update(O.recipe_state_dict(shapes, SEED_MOLDIFF))
        m.load_state_dict(sd, strict=True)
        sizes = [int(s) for s in gold[f'{nm}_sizes']]
        bn, hei, bh, off = [], [], [], 0
        for i, n in enumerate(sizes):
            bn += [i] * n
            tri = torch.triu_indices(n, n, 1) + off
            hei.append(tri)
            bh += [i] * tri.shape[1]
            off += n
        bn, hei, bh = torch.tensor(bn), torch.cat(hei, 1), torch.tensor(bh)
        B = len(sizes)
        node_type, node_pos = torch.from_numpy(gold[f'{nm}_node_type']), torch.from_numpy(gold[f'{nm}_node_pos'])
        half_type = torch.from_numpy(gold[f'{nm}_halfedge_type'])
        t_half = torch.from_numpy(gold[f'{nm}_t_half'])
        eps_pos, u_node, u_half = (torch.from_numpy(gold[f'{nm}_{k}']) for k in ('eps_pos', 'u_node', 'u_halfedge'))
        for tag, dt in (('fp16', torch.float16), ('bf16', torch.bfloat16)):
            m.zero_grad(set_to_none=True)
            with pinned_randomness(t_half, eps_pos, [u_node, u_half]):
                with torch.autocast(device_type='cpu', dtype=dt):
                    ref = m.get_loss(node_type, node_pos, bn, half_type, hei, bh, B)
            (ref['loss'] * SCALE).backward()
            grads = {k: (v.grad.detach().float() / SCALE) for k, v in m.named_parameters() if v.grad is not None}
            finite = all(bool(torch.isfinite(g).all()) for g in grads.values())
            print(nm, tag, {k: float(v) for k, v in ref.items()}, 'finite grads:', finite, 'fp32 loss', float(gold[f'{nm}_loss']))
            for k in ('loss', 'loss_pos', 'loss_node', 'loss_edge'):
                out[f'{nm}/{tag}/{k}'] = np.float32(float(ref[k]))
            for k, g in grads.items():
                out[f'{nm}/{tag}/norm/{k}'] = np.float64(g.double().norm())
                if g.numel() <= 256:
                    out[f'{nm}/{tag}/full/{k}'] = g.numpy()
    np.savez_compressed(os.path.join(OUT, 'loss_amp.npz'), **out)


if __name__ == '__main__':
    main()
