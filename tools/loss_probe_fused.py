import sys; sys.path.insert(0,'/root/repo')
import numpy as np, torch
from tests import util as U
from tests.test_loss import _case, KEYS
from moldiff_amd import train_ops
z = U.gold('loss_amp.npz')
for nm, kind in (('full','MolDiff'),('simple','MolDiff_simple')):
    args, t, noise, _ = _case(nm, 'cuda')
    m = U.moldiff(kind, 'cuda')
    for label, flags in (('per-op', (False, False, False, False)), ('bondffn', (True, False, False, False)), ('+tail', (True, True, False, False)), ('+pos', (True, True, True, False)), ('+node', (True, True, True, True))):
        train_ops._FUSED, train_ops._FUSED_TAIL, train_ops._FUSED_POS, train_ops._FUSED_NODE = flags[0], flags[1], flags[2], flags[3]
        train_ops.FUSED_MIN_ROWS = 1
        with train_ops.precision('fp16'), torch.no_grad():
            pass
        m.zero_grad(set_to_none=True)
        with train_ops.precision('fp16'):
            got = m.get_loss(*args, time_step=t, noise=noise)
        print(nm, label, {k: round(float(got[k]),5) for k in KEYS}, 'ref', {k: round(float(z[f'{nm}/fp16/{k}']),5) for k in KEYS})
    with train_ops.precision('f32'):
        got = m.get_loss(*args, time_step=t, noise=noise)
    print(nm, 'f32', {k: round(float(got[k]),5) for k in KEYS})
