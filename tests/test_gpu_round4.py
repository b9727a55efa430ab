"""Round-4 GPU tests: the opt-in split-precision matrix path under the UNCHANGED parity tests, RCCL on device tensors,
and the stand-alone forwards of the small modules against goldens written by the real reference."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from tests import util as U

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ---------------------------------------------------------------------------------------------------------------------------
# split float16 matrix path (csrc/mdx_split.h, mdx_edge2s.hip, mdx_edge2bs.hip): the parity tests of the exact path, unchanged
# ---------------------------------------------------------------------------------------------------------------------------
@pytest.fixture
def split_path():
    from moldiff_amd import _lib
    with _lib.default_matrix_path('split_f16'):
        yield


def test_split_path_is_selected_and_actually_differs_from_the_exact_path():
    """The switch reaches the kernels: same inputs, results equal to fp32 rounding but not bit-identical; and back."""
    from moldiff_amd import _lib
    g = U.gold('forward.npz')
    bn, hei, bh, ei, be = U.graph_from_sizes(g['sizes'], DEV)
    import torch.nn.functional as F
    xn = F.one_hot(torch.from_numpy(g['node_type']), 8).float().to(DEV)
    xh = F.one_hot(torch.from_numpy(g['halfedge_type']), 6).float().to(DEV)
    pos = U.t32(g['pos']).to(DEV)
    t = torch.full((8,), 500, dtype=torch.long, device=DEV)
    m = U.moldiff('MolDiff', DEV)

    def run():
        with torch.no_grad():
            return m(xn, pos, bn, torch.cat([xh, xh]), ei, be, t)
    exact = run()
    with _lib.default_matrix_path('split_f16'):
        split = run()
        assert m._engine()._path == 'split_f16'
    again = run()
    assert m._engine()._path == 'exact_f32'
    for k in exact:
        assert torch.equal(exact[k], again[k])
        d = U.maxdiff(split[k], exact[k])
        assert 0 < d < (1e-4 if k == 'pred_pos' else 2e-5), (k, d)
    with pytest.raises(ValueError):
        _lib.default_matrix_path('fp8')


# (The exact path's parity tests used to be re-run here one by one under the split path.  Since round 5 every sampling-path test is
# parametrized over BOTH matrix paths where it is defined -- tests/conftest.py `matrix_path`, `@U.both_paths` -- so the wrappers are
# gone; what stays here is what is specific to the split path.)


def test_split_path_refuses_weights_outside_float16_range():
    """mdx_model_set_matrix_path(split) on a model holding a weight of magnitude >= 65504: MDX_ERR_UNSUPPORTED, the exact path keeps working"""
    import moldiff_amd as M
    from moldiff_amd.harness import default_config
    m = M.MolDiff(default_config('MolDiff_simple'), 8, 6).eval()
    sd = M.recipe_state_dict(m, U.KEYS['seeds']['MolDiff'])
    sd['denoiser.edge_blocks.0.self_ffn.weight'] = sd['denoiser.edge_blocks.0.self_ffn.weight'].clone()
    sd['denoiser.edge_blocks.0.self_ffn.weight'][0, 0] = 7.0e4
    m.load_state_dict(sd, strict=True)
    m = m.to(DEV)
    m.matrix_path = 'split_f16'
    with pytest.raises(RuntimeError, match="float16's range"):
        m._engine()
    m.matrix_path = 'exact_f32'
    assert m._engine()._path == 'exact_f32'


def test_split_molecule_result_does_not_depend_on_its_batch(split_path):
    """The in-kernel segment sums are cut per graph in the split kernels too: a molecule alone == the molecule inside a batch."""
    m = U.moldiff('MolDiff_simple', DEV)
    sizes = [9, 23, 17, 30, 12]
    res = {}
    for tag, sz, ids in (('batch', sizes, np.arange(5)), ('alone', [sizes[3]], np.array([3]))):
        bn, hei, bh, ei, be = U.graph_from_sizes(sz, DEV)
        sm = m.sampler(len(sz), bn, hei, bh, seed=7, mol_ids=ids.astype(np.int64))
        sm.init()
        for i in range(3):
            sm.step(i)
        st = sm.state()
        res[tag] = (st['pos'].clone(), bn.clone())
    pos_b, bn_b = res['batch']
    assert torch.equal(pos_b[bn_b == 3], res['alone'][0])


# ---------------------------------------------------------------------------------------------------------------------------
# RCCL once: a real world-size-1 `nccl` group on cuda:0 through every collective the package issues
# ---------------------------------------------------------------------------------------------------------------------------
_RCCL_SCRIPT = r'''
import os, socket, sys
import torch, torch.distributed as dist
sys.path.insert(0, %(root)r)
s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
dev = torch.device('cuda', 0)
torch.cuda.set_device(dev)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
assert dist.get_backend() == 'nccl'
from moldiff_amd import distributed as D
from moldiff_amd.trainer import allreduce_mean_, broadcast_replicas_
g = torch.Generator(device='cpu').manual_seed(3)
pred = [torch.randn(101, 8, generator=g).to(dev), torch.randn(101, 3, generator=g).to(dev), torch.randn(1260, 6, generator=g).to(dev)]
got = D.gather_pred(pred, dst=0)                      # device path: .item() on device counts + padded device all_gather
assert all(torch.equal(a, b) for a, b in zip(got, pred)) and got[0].is_cuda
empty = D.gather_variable(torch.zeros(0, 8, device=dev))
assert empty[0].shape == (0, 8)
counts = torch.tensor([5, 2], dtype=torch.int64, device=dev)   # the entry point's loop-condition all-reduce (sample_drug3d.py)
dist.all_reduce(counts)
assert counts.tolist() == [5, 2]
flat = torch.randn(5_500_000, generator=g).to(dev)             # the trainer's 22 MB flat gradient buffer
ref = flat.clone()
dist.all_reduce(flat, op=dist.ReduceOp.SUM)                    # what allreduce_mean_ issues for world > 1
assert torch.equal(flat, ref)
assert allreduce_mean_(flat) == 1 and broadcast_replicas_([flat]) == 1
dist.broadcast(flat, 0)
mx = torch.tensor([1.25], dtype=torch.float64, device=dev)     # bench.py's max-over-ranks of the elapsed time
dist.all_reduce(mx, op=dist.ReduceOp.MAX)
assert float(mx) == 1.25
dist.barrier()
torch.cuda.synchronize()
dist.destroy_process_group()
print('RCCL_OK', torch.cuda.get_device_name(0))
'''


def test_rccl_world_size_one_group_runs_every_collective_of_the_package():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_PORT'):
        env.pop(k, None)
    r = subprocess.run([sys.executable, '-c', _RCCL_SCRIPT % {'root': ROOT}], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and 'RCCL_OK' in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_bench_headline_line_comes_from_a_real_rccl_group():
    """`python bench.py --gpus 1` joins a one-rank RCCL group (the SCALE run's N = 1 line exercises the N = 8 code)."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_PORT', 'MDX_BENCH_BACKEND'):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--steps', '5', '--warmup', '2', '--batch', '32',
                        '--headline-only', '--no-cpu-baseline'], env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-4000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line['ranks_seen'] == 1 and line['backend'].startswith('rccl'), line.get('backend')
    assert line['gather_rows'][0] > 0 and line['gather_ms'] > 0


# ---------------------------------------------------------------------------------------------------------------------------
# stand-alone forwards of the small modules vs the real reference
# ---------------------------------------------------------------------------------------------------------------------------
def test_standalone_gaussian_smearing_forward_vs_reference_golden():
    """models/common.py:233-237 -- distances incl. 0 and beyond the cutoff, integer time steps (smearing.npz: the reference's outputs)."""
    g = U.gold('smearing.npz')
    m, bp = U.moldiff('MolDiff', DEV), U.bondpred(DEV)
    d = U.t32(g['dist']).to(DEV)
    tid = torch.from_numpy(g['tid']).to(DEV)
    assert U.maxdiff(m.denoiser.distance_expansion(d), g['d15_out']) < 2e-6
    assert U.maxdiff(bp.encoder.distance_expansion(d), g['d20_out']) < 2e-6
    assert U.maxdiff(m.time_emb[0](tid), g['t10_out']) < 2e-6
    assert U.maxdiff(bp.time_emb(tid), g['t20_out']) < 2e-6
    out = m.denoiser.distance_expansion(d.view(8, 8))
    assert out.shape == (64, 16)


def test_standalone_mlp_forward_vs_reference_golden():
    """models/common.py:200-201: a 2-layer MLP (node_net) and the bond predictor's 3-layer decoder; differentiable."""
    g, b = U.gold('dropin.npz'), U.gold('blocks_full.npz')
    m, bp = U.moldiff('MolDiff', DEV), U.bondpred(DEV)
    x = U.t32(b['x']).to(DEV)
    with torch.no_grad():
        assert U.maxdiff(m.denoiser.node_blocks_with_edge[0].node_net(x), g['mlp_node_net0_out']) < 2e-5
        assert U.maxdiff(bp.edge_decoder(U.t32(g['x320']).to(DEV)), g['mlp_bond_decoder_out']) < 2e-5
        y3 = m.denoiser.node_blocks_with_edge[0].node_net(x.view(3, 4, 256))
        assert y3.shape == (3, 4, 256) and U.maxdiff(y3.reshape(12, 256), g['mlp_node_net0_out']) < 2e-5
    xg = x.clone().requires_grad_(True)
    net = m.denoiser.node_blocks_with_edge[0].node_net
    net(xg).square().sum().backward()
    P = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in net.state_dict().items()}
    xc = x.cpu().clone().requires_grad_(True)
    h = torch.nn.functional.linear(xc, P['net.0.weight'], P['net.0.bias'])
    h = torch.relu(torch.nn.functional.layer_norm(h, (256,), P['net.1.weight'], P['net.1.bias']))
    torch.nn.functional.linear(h, P['net.3.weight'], P['net.3.bias']).square().sum().backward()
    assert U.maxdiff(xg.grad, xc.grad) < 1e-4 * float(xc.grad.abs().max())
    for p in net.parameters():
        p.grad = None


def test_standalone_edge_lin_bondffn_vs_reference_golden():
    """models/graph.py:133-141 as PosUpdate.edge_lin (:389): BondFFN with out_dim = 1 called on its own."""
    g, b = U.gold('dropin.npz'), U.gold('blocks_full.npz')
    m = U.moldiff('MolDiff', DEV)
    el = m.denoiser.pos_blocks[0].edge_lin
    with torch.no_grad():
        out = el(U.t32(b['edge_attr']).to(DEV), U.t32(g['edge_lin_node']).to(DEV), U.t32(g['edge_lin_time']).to(DEV))
    assert out.shape == (62, 1) and U.maxdiff(out, g['edge_lin0_out']) < 2e-5


# ---------------------------------------------------------------------------------------------------------------------------
# two deferred gradient records for the same parameter slot must both arrive
# ---------------------------------------------------------------------------------------------------------------------------
def test_deferred_gradient_sink_handles_a_layer_applied_twice():
    """A Linear used twice in one graph and a second backward() inside one grad_sink: the flat gradient equals autograd's."""
    from moldiff_amd import train_ops as T
    from moldiff_amd.trainer import FlatParams
    torch.manual_seed(0)
    lin = torch.nn.Linear(64, 64).to(DEV)
    ln = torch.nn.LayerNorm(64).to(DEV)
    mod = torch.nn.ModuleList([lin, ln])
    x = torch.randn(4096, 64, device=DEV)

    def graph(linear, lnrelu):
        h = linear(x, lin.weight, lin.bias)
        h = lnrelu(h, ln.weight, ln.bias)
        h = linear(h, lin.weight, lin.bias)          # the same layer again
        h = lnrelu(h, ln.weight, ln.bias)
        return h.square().mean()

    ref = graph(torch.nn.functional.linear,
                lambda h, g, b: torch.relu(torch.nn.functional.layer_norm(h, (64,), g, b)))
    gref = torch.autograd.grad(ref, list(mod.parameters()))
    flat = FlatParams(mod)
    flat.zero_grad()
    with T.grad_sink(flat):
        loss = graph(T.linear, lambda h, g, b: T.ln_relu(h, g, b, True))
        loss.backward()
        T.flush_grad_sink()
        one = flat.grad.clone()
        loss2 = graph(T.linear, lambda h, g, b: T.ln_relu(h, g, b, True))
        loss2.backward()                               # a second backward inside the same sink accumulates
    want = torch.cat([g.reshape(-1) for g in gref])
    scale = float(want.abs().max())
    assert U.maxdiff(one, want) < 2e-5 * scale
    assert U.maxdiff(flat.grad, 2 * want) < 4e-5 * scale


# ---------------------------------------------------------------------------------------------------------------------------
# BASELINE config #5 at its own size: one 256-molecule training step against the oracle's autograd
# ---------------------------------------------------------------------------------------------------------------------------
def test_full_size_training_step_loss_and_every_gradient_match_oracle_autograd():
    """train_MolDiff.yml's batch (batch_size 256; sizes = the reference's recipe, numpy seed 2920: N = 6,279 atoms, E = 154,666
    directed edges), full model, fp32: `get_loss` + `backward` through the HIP layer operators (scripts/train_drug3d.py:88-109,
    models/model.py:128-201) against autograd through the CPU oracle with the same time steps and noise.  Tolerances as at
    fixture size (tests/test_loss.py): loss 2e-5, every one of the 566 parameter gradients 1e-4 relative to its norm (floored at
    1e-3 of the largest)."""
    from oracle import moldiff_oracle as O
    np.random.seed(2920)
    sizes = np.maximum(np.random.normal(24.923464980477522, 5.516291901819105, size=256).astype('int64'), 2)
    bn, hei, bh, _, _ = U.graph_from_sizes(sizes)
    N, Eh, B = len(bn), len(bh), len(sizes)
    assert (N, Eh) == (6279, 77333)
    g = U.rng(2024)
    node_type = torch.from_numpy(g.integers(0, 7, N))
    node_pos = U.t32(g.standard_normal((N, 3)) * 2)
    half_type = torch.from_numpy((g.random(Eh) < 0.25) * g.integers(1, 5, Eh))
    t = torch.from_numpy(g.integers(0, 1000, B))
    t[0], t[1] = 0, 999
    noise = dict(eps_pos=U.t32(g.standard_normal((N, 3))), u_node=U.t32(g.random((N, 8))), u_halfedge=U.t32(g.random((Eh, 6))))
    m = U.moldiff('MolDiff', DEV)
    m.zero_grad(set_to_none=True)
    c = lambda x: x.to(DEV)
    got = m.get_loss(c(node_type), c(node_pos), c(bn), c(half_type), c(hei), c(bh), B, time_step=c(t),
                     noise={k: c(v) for k, v in noise.items()})
    got['loss'].backward()
    torch.cuda.synchronize()
    import os as _os
    torch.set_num_threads(min(64, _os.cpu_count() or 1))
    P = U.params(U.moldiff('MolDiff'))
    names = [k for k, p in m.named_parameters() if p.requires_grad]
    assert len(names) == 566
    Pg = {k: (v.clone().requires_grad_(True) if k in names else v) for k, v in P.items()}
    want = O.moldiff_loss(Pg, U.CFG, U.tables(Pg), node_type, node_pos, bn, half_type, hei, bh, B, t, noise)
    want['loss'].backward()
    for k in ('loss', 'loss_pos', 'loss_node', 'loss_edge'):
        assert abs(float(got[k]) - float(want[k])) <= 2e-5 * max(1.0, abs(float(want[k]))), (k, float(got[k]), float(want[k]))
    gmax = max(float(Pg[k].grad.norm()) for k in names)
    worst = ('', 0.0)
    for k, p in m.named_parameters():
        if p.requires_grad:
            w = Pg[k].grad
            err = float((p.grad.cpu() - w).double().norm()) / max(float(w.double().norm()), 1e-3 * gmax)
            if err > worst[1]:
                worst = (k, err)
    print(f'\n    256-molecule training step: loss {float(got["loss"]):.6f} (oracle {float(want["loss"]):.6f}), worst gradient {worst[0]} {worst[1]:.2e}')
    assert worst[1] <= 1e-4, worst
    m.zero_grad(set_to_none=True)
