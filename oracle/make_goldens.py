"""Generate tests/golden/*.npz by running the REAL reference on CPU.  BUILD-CONTAINER ONLY.

    python oracle/make_goldens.py            # needs /root/reference (read-only) + oracle/ref_shim.py

What is committed are the outputs (inputs/expected outputs as small arrays), never reference
source.  The same run pins oracle/moldiff_oracle.py: every oracle function is compared with the
reference module it restates and the max abs difference is recorded in tests/golden/PINNING.json
(bit-exact == 0.0 at the thread count used here; see SURVEY.md section 4 for why thread count matters).

Weights are *recipe weights* (oracle.recipe_state_dict, PCG64 seed below): regenerated anywhere
from the seed + the key/shape list in tests/golden/state_dict_keys.json.
"""
import hashlib
import json
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle import moldiff_oracle as O  # noqa: E402
from oracle import ref_shim  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), 'tests', 'golden')
SEED_MOLDIFF = 20230807
SEED_BONDPRED = 20230808
PIN = {}


def pin(name, a, b):
    d = float((a - b).abs().max()) if a.numel() else 0.0
    PIN[name] = max(PIN.get(name, 0.0), d)
    return d


def rng_inputs(seed):
    return np.random.Generator(np.random.PCG64(seed))


def graph(B, max_size=None, seed=2920):
    np.random.seed(seed)
    ph = O.placeholder(B, max_size)
    bn, hei, bh = ph['batch_node'], ph['halfedge_index'], ph['batch_halfedge']
    ei = torch.cat([hei, hei.flip(0)], 1)
    be = torch.cat([bh, bh])
    return ph, bn, hei, bh, ei, be


def two_mol_graph():
    """n = (5, 7): N = 12, E = 62."""
    sizes = [5, 7]
    bn, hei, bh, off = [], [], [], 0
    for i, n in enumerate(sizes):
        bn += [i] * n
        tri = torch.triu_indices(n, n, 1) + off
        hei.append(tri)
        bh += [i] * tri.shape[1]
        off += n
    bn, hei, bh = torch.tensor(bn), torch.cat(hei, 1), torch.tensor(bh)
    return bn, hei, bh, torch.cat([hei, hei.flip(0)], 1), torch.cat([bh, bh])


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)
    MolDiff, BondPredictor, G, TR, DF, CM = ref_shim.load()
    cfg_full = ref_shim.load_yaml_cfg('configs/train/train_MolDiff.yml')
    cfg_simple = ref_shim.load_yaml_cfg('configs/train/train_MolDiff_simple.yml')
    cfg_bond = ref_shim.load_yaml_cfg('configs/train/train_bondpred.yml')

    def build(cls, cfg, kn, ke, seed):
        m = cls(cfg.model, kn, ke).eval()
        sd = m.state_dict()
        shapes = {k: tuple(v.shape) for k, v in sd.items() if not O.is_frozen_key(k)}
        sd2 = dict(sd)
        sd2.update(O.recipe_state_dict(shapes, seed))
        m.load_state_dict(sd2, strict=True)
        return m, {k: v.detach().clone() for k, v in m.state_dict().items()}

    m_full, P_full = build(MolDiff, cfg_full, 8, 6, SEED_MOLDIFF)
    m_simple, P_simple = build(MolDiff, cfg_simple, 8, 6, SEED_MOLDIFF)
    m_bond, P_bond = build(BondPredictor, cfg_bond, 8, 5, SEED_BONDPRED)
    CFG = dict(num_timesteps=1000, num_blocks=6, cutoff=15)
    CFGB = dict(num_timesteps=1000, num_blocks=8, cutoff=20)

    # ---- state_dict key grammar (the strict-load contract) ---------------------------------
    keys = {'MolDiff': {k: list(v.shape) for k, v in P_full.items()},
            'BondPredictor': {k: list(v.shape) for k, v in P_bond.items()}}
    h = hashlib.sha256()
    for k in sorted(P_full):
        if not O.is_frozen_key(k):
            h.update(P_full[k].numpy().tobytes())
    keys['recipe_sha256_MolDiff'] = h.hexdigest()
    h = hashlib.sha256()
    for k in sorted(P_bond):
        if not O.is_frozen_key(k):
            h.update(P_bond[k].numpy().tobytes())
    keys['recipe_sha256_BondPredictor'] = h.hexdigest()
    keys['seeds'] = {'MolDiff': SEED_MOLDIFF, 'BondPredictor': SEED_BONDPRED}
    with open(os.path.join(OUT, 'state_dict_keys.json'), 'w') as f:
        json.dump(keys, f, indent=0, sort_keys=True)

    # ---- schedules / tables -----------------------------------------------------------------
    probe = np.array([0, 1, 499, 599, 600, 998, 999])
    sch = {}
    for nm, P, cfg in (('full', P_full, cfg_full), ('simple', P_simple, cfg_simple)):
        d = cfg.model.diff
        for part, key in (('pos', 'diff_pos'), ('node', 'diff_atom'), ('edge', 'diff_bond')):
            betas_ref = DF.get_beta_schedule(num_timesteps=1000, **d[key])
            betas = O.beta_schedule(dict(d[key]), 1000)
            pin('betas', torch.from_numpy(betas), torch.from_numpy(betas_ref))
            sch[f'{nm}_{part}_betas'] = betas_ref
        pt = O.pos_tables(sch[f'{nm}_pos_betas'])
        for k in ('coef_x0', 'coef_xt', 'std', 'alphas_bar'):
            pin('pos_tables', pt[k], P['pos_transition.' + k])
            sch[f'{nm}_pos_{k}'] = P['pos_transition.' + k].numpy()
        for part, K, kind in (('node', 8, 'tomask'), ('edge', 6, 'absorb')):
            ct = O.cat_tables(sch[f'{nm}_{part}_betas'], K, kind)
            for k in ('q_mats', 'transpopse_q_onestep_mats'):
                pin('cat_tables', ct[k], P[f'{part}_transition.{k}'])
                sch[f'{nm}_{part}_{k}_probe'] = P[f'{part}_transition.{k}'][probe].numpy()
            sch[f'{nm}_{part}_init_prob'] = ct['init_prob']
    sch['probe_t'] = probe
    np.savez_compressed(os.path.join(OUT, 'schedules.npz'), **sch)

    # ---- smearing ---------------------------------------------------------------------------
    g = rng_inputs(11)
    dist = np.concatenate([[0.0, 1e-3, 15.0, 25.0, 20.0, 19.999], g.uniform(0, 22, 58)]).astype(np.float32)
    tid = np.concatenate([[0, 1, 499, 999, 1000], g.integers(0, 1000, 59)]).astype(np.int64)
    sm = {'dist': dist, 'tid': tid}
    for nm, (start, stop, n, kind) in {'d15': (0.0, 15, 16, 'exp'), 'd20': (0.0, 20, 16, 'exp'),
                                        't10': (0.0, 1000, 10, 'linear'), 't20': (0.0, 1000, 20, 'linear')}.items():
        mod = CM.GaussianSmearing(start=start, stop=stop, num_gaussians=n, type_=kind)
        x = torch.from_numpy(dist) if nm[0] == 'd' else torch.from_numpy(tid)
        ref = mod(x)
        off, co = O.smearing_table(start, stop, n, kind)
        pin('smearing', O.smear(x, off, co, start, stop), ref)
        pin('smearing_tab', off, mod.offset)
        pin('smearing_tab', co, mod.coeff)
        sm[nm + '_out'] = ref.numpy()
        sm[nm + '_offset'] = mod.offset.numpy()
        sm[nm + '_coeff'] = mod.coeff.numpy()
    np.savez_compressed(os.path.join(OUT, 'smearing.npz'), **sm)

    # ---- single blocks at full dims, 2-molecule graph ----------------------------------------
    bn, hei, bh, ei, be = two_mol_graph()
    N, E = len(bn), ei.shape[1]
    g = rng_inputs(12)
    x = torch.from_numpy(g.standard_normal((N, 256), dtype=np.float32))
    ea = torch.from_numpy(g.standard_normal((E, 64), dtype=np.float32))
    pos = torch.from_numpy(g.standard_normal((N, 3), dtype=np.float32) * 2)
    tg = torch.tensor([700, 30])
    nt = (tg[bn].unsqueeze(-1) / 1000)
    et = (tg[be].unsqueeze(-1) / 1000)
    blk = {'batch_node': bn.numpy(), 'halfedge_index': hei.numpy(), 'batch_halfedge': bh.numpy(),
           'x': x.numpy(), 'edge_attr': ea.numpy(), 'pos': pos.numpy(), 't': tg.numpy()}
    den = m_full.denoiser
    with torch.no_grad():
        for i in (0, 3):
            r = den.node_blocks_with_edge[i](x, ei, ea, nt)
            pin('node_block', O.node_block(P_full, f'denoiser.node_blocks_with_edge.{i}', x, ei, ea, nt), r)
            blk[f'nodeblock{i}_out'] = r.numpy()
            r = den.edge_blocks[i](ea, ei, x, et)
            pin('edge_block', O.edge_block(P_full, f'denoiser.edge_blocks.{i}', ea, ei, x, et), r)
            blk[f'edgeblock{i}_out'] = r.numpy()
            rel = pos[ei[0]] - pos[ei[1]]
            dist_ = torch.norm(rel, dim=-1)
            r = den.pos_blocks[i](x, ea, ei, rel, dist_, et)
            pin('pos_update', O.pos_update(P_full, f'denoiser.pos_blocks.{i}', x, ea, ei, rel, dist_, et), r)
            blk[f'posupdate{i}_out'] = r.numpy()
            r = den.edge_blocks[i].bond_ffn_left(ea, x[ei[0]], et)
            pin('bond_ffn', O.bond_ffn(P_full, f'denoiser.edge_blocks.{i}.bond_ffn_left', ea, x[ei[0]], et), r)
            blk[f'bondffn_left{i}_out'] = r.numpy()
    np.savez_compressed(os.path.join(OUT, 'blocks_full.npz'), **blk)

    # ---- NodeEdgeNet 6-block (MolDiff) and 8-block no-pos (BondPredictor) ---------------------
    net = {}
    for tag, (bn_, hei_, bh_, ei_, be_) in {'n12': two_mol_graph(), 'n204': graph(8)[1:]}.items():
        N, E = len(bn_), ei_.shape[1]
        g = rng_inputs(13)
        hn = torch.from_numpy(g.standard_normal((N, 256), dtype=np.float32))
        he = torch.from_numpy(g.standard_normal((E, 64), dtype=np.float32))
        pos = torch.from_numpy(g.standard_normal((N, 3), dtype=np.float32) * 2)
        B = int(bn_.max()) + 1
        tg = torch.from_numpy(g.integers(0, 1000, B))
        nt, et = tg[bn_].unsqueeze(-1) / 1000, tg[be_].unsqueeze(-1) / 1000
        with torch.no_grad():
            r = m_full.denoiser(hn, pos, he, ei_, nt, et)
            o = O.node_edge_net(P_full, 'denoiser', hn, pos, he, ei_, nt, et, num_blocks=6, cutoff=15)
            for a, b in zip(o, r):
                pin('node_edge_net6', a, b)
            rb = m_bond.encoder(hn, pos, he, ei_, nt, et)
            ob = O.node_edge_net(P_bond, 'encoder', hn, pos, he, ei_, nt, et, num_blocks=8, cutoff=20, update_pos=False)
            for a, b in zip(ob, rb):
                pin('node_edge_net8', a, b)
        net[f'{tag}_t'] = tg.numpy()
        net[f'{tag}_sizes'] = np.bincount(bn_.numpy())
        net[f'{tag}_6_h_node'], net[f'{tag}_6_pos'] = r[0].numpy(), r[1].numpy()
        net[f'{tag}_8_h_node'] = rb[0].numpy()
        stride = 1 if tag == 'n12' else 16
        net[f'{tag}_6_h_edge_s{stride}'] = r[2][::stride].numpy()
        net[f'{tag}_8_h_edge_s{stride}'] = rb[2][::stride].numpy()
    net['input_seed'] = np.array(13)
    np.savez_compressed(os.path.join(OUT, 'nodeedgenet.npz'), **net)

    # ---- MolDiff.forward at t in {999, 500, 0}; BondPredictor.forward ----------------------------
    ph, bn, hei, bh, ei, be = graph(8)
    N, Eh = len(bn), len(bh)
    g = rng_inputs(14)
    tn = torch.from_numpy(g.integers(0, 8, N))
    th = torch.from_numpy(g.integers(0, 6, Eh))
    pos = torch.from_numpy(g.standard_normal((N, 3), dtype=np.float32) * 2)
    xn, xh = F.one_hot(tn, 8).float(), F.one_hot(th, 6).float()
    fw = {'sizes': ph['n_nodes_list'], 'node_type': tn.numpy(), 'halfedge_type': th.numpy(), 'pos': pos.numpy()}
    with torch.no_grad():
        for tval in (999, 500, 0):
            t = torch.full((8,), tval, dtype=torch.long)
            r = m_full(xn, pos, bn, torch.cat([xh, xh]), ei, be, t)
            o = O.moldiff_forward(P_full, CFG, xn, pos, bn, torch.cat([xh, xh]), ei, be, t)
            for k in r:
                pin('moldiff_forward', o[k], r[k])
                fw[f't{tval}_{k}'] = r[k].numpy()
        tmix = torch.from_numpy(g.integers(0, 1000, 8))
        r = m_full(xn, pos, bn, torch.cat([xh, xh]), ei, be, tmix)
        o = O.moldiff_forward(P_full, CFG, xn, pos, bn, torch.cat([xh, xh]), ei, be, tmix)
        fw['tmix'] = tmix.numpy()
        for k in r:
            pin('moldiff_forward', o[k], r[k])
            fw[f'tmix_{k}'] = r[k].numpy()
        rb = m_bond(xn, pos, bn, ei, be, tmix)
        pin('bondpred_forward', O.bondpred_forward(P_bond, CFGB, xn, pos, bn, ei, be, tmix), rb)
        fw['tmix_bond_logits'] = rb.numpy()
    np.savez_compressed(os.path.join(OUT, 'forward.npz'), **fw)

    # ---- guidance delta -----------------------------------------------------------------------------
    gd = {}
    for tag, gr in {'n12': two_mol_graph(), 'n101': graph(4)[1:]}.items():
        bn_, hei_, bh_, ei_, be_ = gr
        N = len(bn_)
        B = int(bn_.max()) + 1
        g = rng_inputs(15)
        tn = torch.from_numpy(g.integers(0, 8, N))
        pos = torch.from_numpy(g.standard_normal((N, 3), dtype=np.float32) * 1.5)
        xn = F.one_hot(tn, 8).float()
        t = torch.full((B,), 321, dtype=torch.long)
        with torch.enable_grad():
            p = pos.clone().requires_grad_(True)
            logits = m_bond(xn, p, bn_, ei_, be_, t)
            u = torch.sigmoid(-torch.logsumexp(logits, -1)).log().sum()
            delta = -torch.autograd.grad(u, p)[0] * 1e-4
        od, ol = O.guidance_delta(P_bond, CFGB, xn, pos, bn_, ei_, be_, t, 1e-4)
        pin('guidance_delta', od, delta)
        pin('guidance_logits', ol, logits.detach())
        gd[f'{tag}_sizes'] = np.bincount(bn_.numpy())
        gd[f'{tag}_node_type'], gd[f'{tag}_pos'] = tn.numpy(), pos.numpy()
        gd[f'{tag}_logits'], gd[f'{tag}_delta'] = logits.detach().numpy(), delta.numpy()
    gd['t'] = np.array(321)
    np.savez_compressed(os.path.join(OUT, 'guidance.npz'), **gd)

    # ---- teacher-forced step replay (B=4: N=101, Eh=1260), simple + full(guided) -------------------
    # noise is captured by wrapping the torch RNG entry points the reference calls.
    ph, bn, hei, bh, ei, be = graph(4)
    N, Eh = len(bn), len(bh)
    rep = {'sizes': ph['n_nodes_list']}
    for tag, model, P, bond, Pb, guid in (('simple', m_simple, P_simple, None, None, None),
                                          ('guided', m_full, P_full, m_bond, P_bond, ['uncertainty', 1e-4])):
        tabs = {'pos': {k: P['pos_transition.' + k] for k in ('coef_x0', 'coef_xt', 'std')},
                'node': {k: P['node_transition.' + k] for k in ('q_mats', 'transpopse_q_onestep_mats')},
                'edge': {k: P['edge_transition.' + k] for k in ('q_mats', 'transpopse_q_onestep_mats')}}
        for window, steps in (('hi', [999, 998, 997]), ('lo', [2, 1, 0])):
            g = rng_inputs(16 + (window == 'lo'))
            st = {'h_node': F.one_hot(torch.from_numpy(g.integers(0, 8, N)), 8).float(),
                  'pos': torch.from_numpy(g.standard_normal((N, 3), dtype=np.float32) * (1.0 if window == 'hi' else 2.5)),
                  'h_halfedge': F.one_hot(torch.from_numpy(g.integers(0, 6, Eh)), 6).float()}
            st['log_node'] = F.log_softmax(torch.from_numpy(g.standard_normal((N, 8), dtype=np.float32)), -1)
            st['log_halfedge'] = F.log_softmax(torch.from_numpy(g.standard_normal((Eh, 6), dtype=np.float32)), -1)
            pre = f'{tag}_{window}'
            rep[pre + '_steps'] = np.array(steps)
            for k in ('pos', 'log_node', 'log_halfedge'):
                rep[f'{pre}_init_{k}'] = st[k].numpy()
            rep[f'{pre}_init_node_type'] = st['h_node'].argmax(-1).numpy()
            rep[f'{pre}_init_halfedge_type'] = st['h_halfedge'].argmax(-1).numpy()
            for j, s in enumerate(steps):
                noise = {'eps_pos': torch.from_numpy(g.standard_normal((N, 3), dtype=np.float32)),
                         'u_node': torch.from_numpy(g.random((N, 8), dtype=np.float32)),
                         'u_halfedge': torch.from_numpy(g.random((Eh, 6), dtype=np.float32))}
                # --- run one iteration of the REFERENCE loop body (model.py:272-372) with injected noise
                ref_state, ref_preds = reference_step(model, DF, st, bn, hei, bh, s, noise, bond, guid)
                o_state, o_preds = O.sample_step(P, CFG, tabs, st, {'batch_node': bn, 'halfedge_index': hei,
                                                                     'batch_halfedge': bh, 'n_graphs': 4}, s, noise,
                                                 Pb=Pb, cfgb=CFGB, guidance=guid)
                for k in ('pos', 'log_node', 'log_halfedge'):
                    pin(f'sample_step_{k}', o_state[k], ref_state[k])
                assert torch.equal(o_state['node_type'], ref_state['node_type'])
                assert torch.equal(o_state['halfedge_type'], ref_state['halfedge_type'])
                for k in ref_preds:
                    pin('sample_step_preds', o_preds[k], ref_preds[k])
                for k in noise:
                    rep[f'{pre}_{j}_{k}'] = noise[k].numpy()
                rep[f'{pre}_{j}_pos'] = ref_state['pos'].numpy()
                rep[f'{pre}_{j}_log_node'] = ref_state['log_node'].numpy()
                rep[f'{pre}_{j}_log_halfedge'] = ref_state['log_halfedge'].numpy()
                rep[f'{pre}_{j}_node_type'] = ref_state['node_type'].numpy()
                rep[f'{pre}_{j}_halfedge_type'] = ref_state['halfedge_type'].numpy()
                rep[f'{pre}_{j}_pred_pos'] = ref_preds['pred_pos'].numpy()
                rep[f'{pre}_{j}_pred_node'] = ref_preds['pred_node'].numpy()
                # top-2 margins of the Gumbel-perturbed logits (for margin-aware class-id comparison)
                for nm, lg, u in (('node', ref_state['log_node'], noise['u_node']),
                                  ('halfedge', ref_state['log_halfedge'], noise['u_halfedge'])):
                    z = lg - torch.log(-torch.log(u + 1e-30) + 1e-30)
                    top = z.topk(2, -1).values
                    rep[f'{pre}_{j}_{nm}_margin_min'] = np.array(float((top[:, 0] - top[:, 1]).min()))
                st = {k: ref_state[k] for k in ('h_node', 'pos', 'h_halfedge', 'log_node', 'log_halfedge')}
    np.savez_compressed(os.path.join(OUT, 'step_replay.npz'), **rep)

    # ---- init (sample_init) with explicit float64 uniforms ------------------------------------------
    g = rng_inputs(18)
    ini = {}
    for part, K, tr in (('node', 8, m_full.node_transition), ('edge', 6, m_full.edge_transition)):
        u = torch.from_numpy(g.random((64, K)))
        orig = torch.rand_like
        torch.rand_like = lambda x, *a, **k: u.to(x.dtype)
        try:
            c, oh, lv = tr.sample_init(64)
        finally:
            torch.rand_like = orig
        tab = {'q_mats': tr.q_mats, 'init_prob': tr.init_prob}
        oc, ooh, olv = O.cat_init(tab, 64, u)
        assert torch.equal(c, oc)
        pin('cat_init', olv, lv)
        ini[f'{part}_u'], ini[f'{part}_class'], ini[f'{part}_log'] = u.numpy(), c.numpy(), lv.numpy()
    np.savez_compressed(os.path.join(OUT, 'init.npz'), **ini)

    # ---- placeholder -----------------------------------------------------------------------------------
    plc = {}
    for B in (8, 256, 2048):
        np.random.seed(2920)
        ph = O.placeholder(B)
        plc[f'B{B}_sizes'] = ph['n_nodes_list']
        plc[f'B{B}_N'] = np.array(len(ph['batch_node']))
        plc[f'B{B}_Eh'] = np.array(len(ph['batch_halfedge']))
        plc[f'B{B}_he_first16'] = ph['halfedge_index'][:, :16].numpy()
        plc[f'B{B}_he_last16'] = ph['halfedge_index'][:, -16:].numpy()
    np.savez_compressed(os.path.join(OUT, 'placeholder.npz'), **plc)

    with open(os.path.join(OUT, 'PINNING.json'), 'w') as f:
        json.dump({'threads': torch.get_num_threads(), 'torch': torch.__version__,
                   'max_abs_diff_oracle_vs_reference': PIN}, f, indent=1, sort_keys=True)
    print(json.dumps(PIN, indent=1))
    for fn in sorted(os.listdir(OUT)):
        print(fn, os.path.getsize(os.path.join(OUT, fn)))


def reference_step(model, DF, st, bn, hei, bh, step, noise, bond, guid, B=4):
    """Drive the reference's own modules through one iteration of its sampling loop body
    (models/model.py:272-372) with the random draws replaced by `noise` (draw order pos, node, halfedge)."""
    draws = [noise['eps_pos'], noise['u_node'], noise['u_halfedge']]
    it = iter(draws)
    o_randn, o_rand = torch.randn_like, torch.rand_like
    torch.randn_like = lambda x, *a, **k: next(it)
    torch.rand_like = lambda x, *a, **k: next(it)
    try:
        with torch.no_grad():
            ei = torch.cat([hei, hei.flip(0)], 1)
            be = torch.cat([bh, bh])
            t = torch.full((B,), step, dtype=torch.long)
            preds = model(st['h_node'], st['pos'], bn, torch.cat([st['h_halfedge']] * 2), ei, be, t)
            pos_prev = model.pos_transition.get_prev_from_recon(x_t=st['pos'], x_recon=preds['pred_pos'], t=t, batch=bn)
            ln = model.node_transition.q_v_posterior(F.log_softmax(preds['pred_node'], -1), st['log_node'], t, bn, v0_prob=True)
            cn = DF.log_sample_categorical(ln)
            lh = model.edge_transition.q_v_posterior(F.log_softmax(preds['pred_halfedge'], -1), st['log_halfedge'], t, bh, v0_prob=True)
            ch = DF.log_sample_categorical(lh)
            if guid is not None:
                with torch.enable_grad():
                    p = st['pos'].detach().requires_grad_(True)
                    lg = bond(st['h_node'].detach(), p, bn, ei, be, t)
                    u = torch.sigmoid(-torch.logsumexp(lg, -1)).log().sum()
                    delta = -torch.autograd.grad(u, p)[0] * guid[1]
                pos_prev = pos_prev + delta
    finally:
        torch.randn_like, torch.rand_like = o_randn, o_rand
    return ({'h_node': model.node_transition.onehot_encode(cn), 'pos': pos_prev,
             'h_halfedge': model.edge_transition.onehot_encode(ch), 'log_node': ln, 'log_halfedge': lh,
             'node_type': cn, 'halfedge_type': ch}, preds)


if __name__ == '__main__':
    main()
