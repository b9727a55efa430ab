"""Harness consumers of the path's outputs: seperate_outputs / decode_output (host form, CPU) and the on-device
decode_batch (GPU) against goldens produced by the reference's own functions (oracle/make_goldens_decode.py)."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from moldiff_amd.harness import placeholder_from_sizes
from moldiff_amd.postprocess import FeaturizeMol, seperate_outputs
from moldiff_amd.sample_drug3d import is_connected, mol_block
from oracle import moldiff_oracle as O
from tests import util as U

FEAT = FeaturizeMol([6, 7, 8, 9, 15, 16, 17], [1, 2, 3, 4], use_mask_node=True, use_mask_edge=True)
KEYS = ('element', 'atom_pos', 'bond_type', 'bond_index', 'atom_prob', 'bond_prob')


def _check(info, g, i):
    for k in KEYS:
        ref = g[f'mol{i}_{k}']
        got = np.asarray(info[k])
        assert got.shape == ref.shape, (i, k, got.shape, ref.shape)
        if ref.dtype.kind == 'f':
            assert np.abs(got - ref).max(initial=0) < 1e-6, (i, k)
        else:
            assert np.array_equal(got, ref), (i, k)


def test_featurizer_attributes():
    assert (FEAT.num_element, FEAT.num_bond_types, FEAT.num_node_types, FEAT.num_edge_types) == (7, 4, 8, 6)
    from moldiff_amd.data import synthetic_records
    rec = synthetic_records(1, seed=0)[0]
    out = FEAT(rec)                                   # FeaturizeMol.__call__ of the training path (moldiff_amd/data.py)
    assert out['node_type'].shape[0] == rec['num_atoms'] and int((out['halfedge_type'] > 0).sum()) == rec['num_bonds']


def test_host_separate_and_decode_vs_reference_golden():
    g = U.gold('decode.npz')
    ph = placeholder_from_sizes(g['sizes'])
    bn, hei, bh = (ph[k].numpy() for k in ('batch_node', 'halfedge_index', 'batch_halfedge'))
    outputs = {'pred': [g['pred_node'], g['pred_pos'], g['pred_halfedge']],
               'traj': [g['pred_node'][None], g['pred_pos'][None], g['pred_halfedge'][None]]}
    sep = seperate_outputs(outputs, len(g['sizes']), bn, hei, bh)
    osep = O.separate_outputs(outputs['pred'], len(g['sizes']), bn, hei, bh)
    for i, (a, b) in enumerate(zip(sep, osep)):
        assert np.array_equal(a['halfedge_index'], g[f'mol{i}_halfedge_index'])
        assert np.array_equal(a['halfedge_index'], b['halfedge_index'])
        assert a['traj'][2].shape == (1,) + a['pred'][2].shape
        _check(FEAT.decode_output(a['pred'][0], a['pred'][1], a['pred'][2], a['halfedge_index']), g, i)
        _check(O.decode_output(b['pred'][0], b['pred'][1], b['pred'][2], b['halfedge_index']), g, i)


def test_single_atom_molecule_raises_like_reference():
    ph = placeholder_from_sizes([3, 1])
    bn, hei, bh = (ph[k].numpy() for k in ('batch_node', 'halfedge_index', 'batch_halfedge'))
    outputs = {'pred': [np.zeros((4, 8), np.float32), np.zeros((4, 3), np.float32), np.zeros((3, 6), np.float32)]}
    outputs['traj'] = [x[None] for x in outputs['pred']]
    with pytest.raises(ValueError):
        seperate_outputs(outputs, 2, bn, hei, bh)


def test_connectivity_and_mol_block():
    bi = np.array([[0, 1, 1, 2], [1, 0, 2, 1]])
    assert is_connected(3, bi) and not is_connected(4, bi) and not is_connected(0, bi[:, :0])
    info = {'element': np.array([6, 8, 7]), 'atom_pos': np.zeros((3, 3), np.float32), 'bond_type': np.array([1, 4, 1, 4]),
            'bond_index': np.array([[0, 1, 1, 2], [1, 2, 0, 1]])}
    blk = mol_block(info).splitlines()
    assert blk[3].startswith('  3  2') and blk[-1] == 'M  END' and blk[4].split()[3] == 'C' and blk[8].split() == ['2', '3', '4', '0']


@pytest.mark.gpu
def test_device_decode_batch_vs_reference_golden():
    g = U.gold('decode.npz')
    ph = placeholder_from_sizes(g['sizes'], 'cuda:0')
    pred = [U.t32(g[k]).to('cuda:0') for k in ('pred_node', 'pred_pos', 'pred_halfedge')]
    mols = FEAT.decode_batch(pred, ph['batch_node'], ph['halfedge_index'], ph['batch_halfedge'], len(g['sizes']))
    assert len(mols) == len(g['sizes'])
    for i, info in enumerate(mols):
        _check(info, g, i)


@pytest.mark.gpu
def test_device_decode_matches_host_decode_on_a_large_random_batch():
    r = U.rng(77)
    sizes = r.integers(2, 40, 64)
    ph = placeholder_from_sizes(sizes)
    bn, hei, bh = (ph[k].numpy() for k in ('batch_node', 'halfedge_index', 'batch_halfedge'))
    N, Eh = len(bn), len(bh)
    pred = [(r.standard_normal((N, 8)) * 3).astype(np.float32), r.standard_normal((N, 3)).astype(np.float32),
            (r.standard_normal((Eh, 6)) * 3).astype(np.float32)]
    dev = [torch.from_numpy(p).to('cuda:0') for p in pred]
    phd = {k: v.to('cuda:0') for k, v in ph.items()}
    got = FEAT.decode_batch(dev, phd['batch_node'], phd['halfedge_index'], phd['batch_halfedge'], len(sizes))
    sep = O.separate_outputs(pred, len(sizes), bn, hei, bh)
    for info, s in zip(got, sep):
        ref = O.decode_output(s['pred'][0], s['pred'][1], s['pred'][2], s['halfedge_index'])
        for k in KEYS:
            a, b = np.asarray(info[k]), np.asarray(ref[k])
            assert a.shape == b.shape, k
            assert (np.abs(a - b).max(initial=0) < 1e-6) if b.dtype.kind == 'f' else np.array_equal(a, b), k


@pytest.mark.gpu
def test_sample_drug3d_entry_point_end_to_end(tmp_path):
    """The drop-in entry point with synthetic weights: 4 molecules, batch 6, full T=1000 chain."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, '-m', 'moldiff_amd.sample_drug3d', '--config', os.path.join(root, 'configs', 'sample_MolDiff_simple.yml'),
           '--outdir', str(tmp_path / 'outputs'), '--device', 'cuda:0', '--batch_size', '6', '--num_mols', '2', '--recipe-weights']
    env = dict(os.environ, PYTHONPATH=root)
    res = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    assert '[Pool] Finished' in res.stdout
    runs = [d for d in os.listdir(tmp_path / 'outputs') if not d.endswith('_SDF')]
    pool = torch.load(tmp_path / 'outputs' / runs[0] / 'samples_all.pt', weights_only=False)
    assert set(pool) == {'finished', 'failed'} and len(pool['finished']) + len(pool['failed']) > 0
    for info in pool['finished'] + pool['failed']:
        assert set(info) >= set(KEYS) and info['atom_pos'].shape == (len(info['element']), 3)


@pytest.mark.gpu
def test_entry_point_writes_trajectories_of_the_drawn_molecules(tmp_path):
    """sample.save_traj_prob (scripts/sample_drug3d.py:155-190): with probability 1 every finished molecule gets a
    traj_mol<id>.sdf with T+1 frames decoded from the compact on-device trajectory; the last frame's atoms are those of the
    molecule's own mol block only up to the final arg-max over logits (frames hold SAMPLED types, the block the predicted)."""
    import yaml
    from moldiff_amd import sample_drug3d
    cfg = yaml.safe_load(open('configs/sample_MolDiff_simple.yml'))
    cfg['sample'].update(num_mols=2, batch_size=4, save_traj_prob=1.0)
    p = tmp_path / 'sample_MolDiff_simple.yml'
    p.write_text(yaml.safe_dump(cfg))
    import sys
    argv = sys.argv
    try:
        log_dir = sample_drug3d.main(['--config', str(p), '--outdir', str(tmp_path / 'out'), '--device', 'cuda:0', '--recipe-weights'])
    finally:
        sys.argv = argv
    pool = torch.load(os.path.join(log_dir, 'samples_all.pt'), weights_only=False)
    trajs = [f for f in os.listdir(log_dir + '_SDF') if f.startswith('traj_mol')]
    assert len(trajs) >= len(pool['finished'])     # drawn from every connected molecule of every batch
    for info in pool['finished']:
        assert info['traj_file'] == 'traj_mol%d.sdf' % info['mol_id']
        txt = open(os.path.join(log_dir + '_SDF', info['traj_file'])).read()
        assert txt.count('$$$$') == 1001 and txt.count('M  END') == 1001


@pytest.mark.gpu
def test_two_rank_sampling_run_equals_the_single_rank_run(tmp_path):
    """The sharded entry point (2 ranks, here sharing the test box's GPU over gloo) produces exactly the molecules of the
    1-rank run: sizes come from the same numpy stream, noise is keyed by global molecule id, rank 0 gathers."""
    import socket
    import subprocess
    import sys
    import yaml
    cfg = yaml.safe_load(open('configs/sample_MolDiff_simple.yml'))
    cfg['sample'].update(num_mols=6, batch_size=8, save_traj_prob=1.0)
    p = tmp_path / 'sample_MolDiff_simple.yml'
    p.write_text(yaml.safe_dump(cfg))

    def run(world, outdir):
        s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
        procs = []
        for r in range(world):
            env = dict(os.environ, WORLD_SIZE=str(world), RANK=str(r), LOCAL_RANK=str(r), MASTER_ADDR='127.0.0.1',
                       MASTER_PORT=str(port), MDX_DIST_BACKEND='gloo')
            if world == 1:
                for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK'):
                    env.pop(k)
            procs.append(subprocess.Popen([sys.executable, '-m', 'moldiff_amd.sample_drug3d', '--config', str(p), '--outdir', outdir,
                                           '--device', 'cuda:0', '--recipe-weights'], env=env, stdout=subprocess.PIPE,
                                          stderr=subprocess.STDOUT, text=True))
        outs = [q.communicate(timeout=600)[0] for q in procs]
        assert all(q.returncode == 0 for q in procs), outs
        run_dir = [d for d in os.listdir(outdir) if not d.endswith('_SDF')][0]
        pool = torch.load(os.path.join(outdir, run_dir, 'samples_all.pt'), weights_only=False)
        # every selected molecule carries its trajectory, whichever rank sampled (and wrote) it
        for info in pool['finished']:
            assert info['traj_file'] == 'traj_mol%d.sdf' % info['mol_id']
            assert os.path.exists(os.path.join(outdir, run_dir + '_SDF', info['traj_file'])), (world, info['mol_id'])
        return pool

    # the seed depends on the characters of --outdir (scripts/sample_drug3d.py:47): same string length/sum for both runs
    one = run(1, str(tmp_path / 'oa'))
    two = run(2, str(tmp_path / 'ao'))
    for key in ('finished', 'failed'):
        assert len(one[key]) == len(two[key])
        for a, b in zip(one[key], two[key]):
            assert np.array_equal(a['element'], b['element']) and np.array_equal(a['bond_index'], b['bond_index'])
            assert np.array_equal(a['atom_pos'], b['atom_pos'])
