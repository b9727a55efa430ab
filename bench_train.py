#!/usr/bin/env python
"""bench_train.py -- molecules/sec of one optimisation step of MolDiff training on MI355X (BASELINE.json configs[4]:
train_MolDiff.yml, batch_size 256 per GPU, data-parallel).  NOT the headline metric (that is bench.py); same launch
contract:

    python bench_train.py [--gpus N] [--steps K] [--warmup W] [--batch 256] [--model MolDiff|bondpred] [--no-cpu-baseline]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench_train.py --gpus N ...

One step = scripts/train_drug3d.py:88-109: zero_grad, get_loss on a 256-molecule batch (random step per molecule,
forward through the 6-block denoiser), backward (every parameter gradient), gradient all-reduce over the ranks (RCCL,
one flat buffer), clip_grad_norm_(50), AdamW.  fp32 throughout (the reference autocasts to fp16).  Synthetic clean
molecules: sizes by the reference recipe (seed 2920), uniform atom types, 25 % bonded half-edges, N(0, 2^2) coordinates.
cpu_baseline = the torch-CPU oracle (forward + autograd backward + torch AdamW) on a 32-molecule slice of the batch,
scaled by the edge count.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def clean_batch(sizes, seed, device):
    from moldiff_amd.harness import placeholder_from_sizes
    ph = placeholder_from_sizes(sizes, device)
    g = np.random.Generator(np.random.PCG64(seed))
    N, Eh = int(ph['batch_node'].numel()), int(ph['batch_halfedge'].numel())
    node_type = torch.from_numpy(g.integers(0, 7, N)).to(device)
    pos = torch.from_numpy((g.standard_normal((N, 3)) * 2.0).astype(np.float32)).to(device)
    half = torch.from_numpy((g.random(Eh) < 0.25) * g.integers(1, 5, Eh)).to(device)
    return (node_type, pos, ph['batch_node'], half, ph['halfedge_index'], ph['batch_halfedge'], len(sizes))


def cpu_baseline(kind, model, sizes, budget_s):
    from oracle import moldiff_oracle as O
    sub = [int(s) for s in sizes[:32]]
    batch = clean_batch(sub, 5, 'cpu')
    P = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    names = [k for k, p in model.named_parameters() if p.requires_grad]
    for k in names:
        P[k].requires_grad_(True)
    opt = torch.optim.AdamW([P[k] for k in names], lr=1e-4, betas=(0.99, 0.999), weight_decay=1e-8)
    cores = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    threads = min(16, cores)
    torch.set_num_threads(threads)
    g = torch.Generator().manual_seed(1)
    N, Eh, B = batch[1].shape[0], batch[3].shape[0], batch[6]

    def step():
        t = torch.randint(0, 1000, (B,), generator=g)
        noise = dict(eps_pos=torch.randn(N, 3, generator=g), u_node=torch.rand(N, 8, generator=g), u_halfedge=torch.rand(Eh, 6, generator=g))
        opt.zero_grad(set_to_none=True)
        if kind == 'bondpred':
            tabs = {'pos': {'alphas_bar': P['pos_transition.alphas_bar']}, 'node': {'q_mats': P['node_transition.q_mats']}}
            loss = O.bondpred_loss(P, dict(num_timesteps=1000, num_blocks=8, cutoff=20), tabs, *batch, t, noise)['loss']
        else:
            tabs = {'pos': {k: P['pos_transition.' + k] for k in ('coef_x0', 'coef_xt', 'std', 'alphas_bar')},
                    'node': {k: P['node_transition.' + k] for k in ('q_mats', 'transpopse_q_onestep_mats')},
                    'edge': {k: P['edge_transition.' + k] for k in ('q_mats', 'transpopse_q_onestep_mats')}}
            loss = O.moldiff_loss(P, dict(num_timesteps=1000, num_blocks=6, cutoff=15), tabs, *batch, t, noise)['loss']
        loss.backward()
        torch.nn.utils.clip_grad_norm_([P[k] for k in names], 50.0)
        opt.step()

    t0 = time.perf_counter(); step(); est = time.perf_counter() - t0
    n = int(max(1, min(10, budget_s / max(est, 1e-3))))
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    per = (time.perf_counter() - t0) / n
    e_sub = sum(s * (s - 1) for s in sub)
    e_all = sum(int(s) * (int(s) - 1) for s in sizes)
    per_full = per * e_all / e_sub
    return {'value': len(sizes) / per_full, 'unit': 'molecules/sec', 'cores': threads, 'kind': 'port',
            'sample': f'{n} optimisation steps (after 1 warm-up) of the torch-CPU oracle (forward, autograd backward, clip, torch '
                      f'AdamW; fp32, {threads} threads) on the first 32 molecules of the batch ({e_sub} of {e_all} directed edges), '
                      f'scaled by the edge count', 'ms_per_step_scaled': per_full * 1e3}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--batch', type=int, default=256)
    ap.add_argument('--model', default='MolDiff', choices=['MolDiff', 'MolDiff_simple', 'bondpred'])
    ap.add_argument('--precision', default='f32', choices=['f32', 'bf16'], help="GEMM operand precision ('bf16' = mixed precision)")
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-budget', type=float, default=20.0)
    args = ap.parse_args()
    world, rank, local_rank = (int(os.environ.get(k, d)) for k, d in (('WORLD_SIZE', '1'), ('RANK', '0'), ('LOCAL_RANK', '0')))
    if args.gpus != world:
        raise SystemExit(f'--gpus {args.gpus} != WORLD_SIZE {world} (launch with python -m torch.distributed.run --nproc-per-node N)')
    if not torch.cuda.is_available():
        raise SystemExit('bench_train.py needs a ROCm GPU (no CPU fallback).')
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)

    import moldiff_amd as M
    from moldiff_amd.harness import default_config, GEOM_DRUGS_MEAN_ATOMS, GEOM_DRUGS_STD_ATOMS
    from moldiff_amd.trainer import Trainer
    np.random.seed(2920)
    sizes = np.random.normal(GEOM_DRUGS_MEAN_ATOMS, GEOM_DRUGS_STD_ATOMS, size=args.batch * (rank + 1)).astype('int64')
    sizes = np.maximum(sizes[args.batch * rank:], 2)
    if args.model == 'bondpred':
        model = M.BondPredictor(default_config('bondpred'), 8, 5)
        model.load_state_dict(M.recipe_state_dict(model, 20230808), strict=True)
    else:
        model = M.MolDiff(default_config(args.model), 8, 6)
        model.load_state_dict(M.recipe_state_dict(model, 20230807), strict=True)
    model = model.to(dev).train()
    tr = Trainer(model, lr=1e-4, betas=(0.99, 0.999), weight_decay=1e-8, max_grad_norm=50.0, precision=args.precision)
    batch = clean_batch([int(s) for s in sizes], 100 + rank, dev)
    torch.manual_seed(2023 + rank)
    losses = []
    for _ in range(args.warmup):
        tr.step(*batch)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    barrier()
    torch.cuda.reset_peak_memory_stats()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        losses.append(tr.step(*batch)['loss'])
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    ms = elapsed / args.steps * 1e3
    if rank == 0:
        N, E = int(batch[1].shape[0]), 2 * int(batch[3].shape[0])
        out = {'metric': 'molecules/sec (training step: forward + backward + all-reduce + clip + AdamW)', 'value': args.batch * world / (ms / 1e3),
               'unit': 'molecules/sec', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': ms,
               'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
               'dtype': 'f32' if args.precision == 'f32' else 'bf16 GEMM operands, f32 accumulate / elsewhere', 'data': 'synthetic',
               'config': {'workload': f'train_{args.model}.yml: batch_size={args.batch} molecules/GPU (rank 0: N={N} atoms, E={E} directed '
                                      f'edges), AdamW lr 1e-4 betas (0.99,0.999) wd 1e-8, max_grad_norm 50; recipe weights',
                          'parallelism': f'data-parallel x{world}, one flat-gradient all-reduce per step',
                          'parameters': tr.flat.numel},
               'peak_hbm_gb': torch.cuda.max_memory_allocated() / 2 ** 30,
               'loss_first_last': [float(losses[0]), float(losses[-1])]}
        if not args.no_cpu_baseline and world == 1:
            out['cpu_baseline'] = cpu_baseline(args.model, model, sizes, args.cpu_budget)
            out['speedup_vs_cpu_baseline'] = out['value'] / out['cpu_baseline']['value']
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
