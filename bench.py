#!/usr/bin/env python
"""bench.py -- molecules/sec of MolDiff's 1000-step denoising loop on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch 256] [--no-cpu-baseline]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One "step" = one iteration of the reverse chain (models/model.py:272-372: noise draw, denoiser forward,
3 posteriors, 2 Gumbel-max draws) over one packed batch of `--batch` molecules per GPU
(BASELINE.json configs[1]: sample_MolDiff_simple, batch_size=256, T=1000, no bond guidance; sizes drawn
by the reference's recipe with numpy seed 2920 -> N=6,279 atoms, E=154,666 directed edges on rank 0).
The per-step cost does not depend on the step index, so K timed steps give
    molecules/sec = (batch * n_gpus) / (ms_per_step * T / 1000),   T = 1000.
With the default K = 1000 the timed region IS one complete sampling run.
Each rank samples its own independent batch (weak scaling, no data-path collective; the only collective
is the barrier + the max-reduce of the elapsed time).  Inputs (index tensors, weights) are resident in
HBM before the timed region; weights are synthetic "recipe" weights (no checkpoint offline).

`python bench.py --train [--model MolDiff|bondpred] [--precision f32|fp16|bf16] ...` runs the training-step benchmark instead
(BASELINE.json configs[4]; same launch contract, its own JSON line; see main_train).

Extra objects on the JSON line:
  roofline     -- dominant kernel = fused edge kernel A (fp32 MFMA): algorithmic FLOPs (617,088 per directed
                  edge) / its average launch duration measured with hipEvents on its launch stream
                  during the timed region, against the 157.3 TFLOP/s fp32-matrix peak.
  aggregation  -- the reduction pass after edge kernel A (round 3: combines the partial rows the kernel's in-kernel segment sums emit
                  + the by-right BondFFN sum); segment_sum -- the scatter/gather primitive (E,256)->(N,256) by itself vs 8 TB/s.
  cpu_baseline -- the CPU oracle (oracle/moldiff_oracle.py, a torch-CPU restatement pinned bit-exact to the
                  reference here) timed on this box's host cores on a bounded sample of the same workload.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

T_STEPS = 1000
_ORIG_ARGV = list(sys.argv[1:])   # main_train() strips --train before parsing; a self-launch must pass it on
FLOP_EDGE_A = 617088        # per directed edge per launch (hoisted count, DESIGN.md / SURVEY Appendix D)
FLOP_EDGE_B = 221184        # edge kernel B (EdgeBlock tail + PosUpdate): 2 * (2*64*64 + 2*64*256 + 2*64*32 + 256*256)
# per directed edge per launch of the guidance backward's edge kernel: EXECUTED flops.  Round 3 reads the
# BondFFN intermediates from the tape instead of recomputing W_bl (64->128), W_1 (128->128) and W_2 (128->64) on both sides:
# 829,440 - 2 * 2 * (64*128 + 128*128 + 128*64) = 698,368.  Round 5: the launch of block i > 0 also runs block i-1's EdgeBlock-tail backward
# (3 GEMMs 64x64 = 24,576) and block 0's launch no longer forms dL/dHe_0 (8,192): averaged over a predictor's 8 launches
# 698,368 + 7/8 * 24,576 - 1/8 * 8,192 = 718,848
FLOP_EDGE_BWD = 718848
EDGE_A_NAME = ('edge_a2_kernel<15> (row-owner fused per-edge MLP chain + in-kernel segment sums of its messages, 16 rows x 2 waves per '
               'SIMD, v_mfma_f32_16x16x4_f32)')
EDGE_B_NAME = 'edge_b2_kernel (row-owner EdgeBlock tail + PosUpdate, v_mfma_f32_16x16x4_f32)'
REFERENCE_CPU_MOL_S = 0.097  # the REAL reference on 8 CPU cores, config #2 (2.64 s/step at 256 molecules)
PEAK_FP32_MFMA = 157.3      # TFLOP/s (MI355X_MICROARCH.md)
PEAK_HBM = 8000.0           # GB/s


def self_launch(n_gpus):
    """`python bench.py --gpus N` without a launcher: start the N ranks ourselves (one process per GPU through
    torch.distributed.run on 127.0.0.1 and a free port), stream their output through and exit with their status.  Under an
    external launcher (WORLD_SIZE set) this is never reached."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    env.setdefault('OMP_NUM_THREADS', '8')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n_gpus), '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + _ORIG_ARGV
    raise SystemExit(subprocess.run(cmd, env=env).returncode)


_REAL_STDOUT = None


def own_stdout():
    """The contract is ONE JSON line on stdout.  Native libraries write there too (RCCL prints "Librccl path : ..." into C stdio's
    buffer at communicator setup, and the buffer is flushed at process exit, i.e. AFTER the line): from here on file descriptor 1
    is stderr for everybody, and `emit` writes the line to the real stdout."""
    global _REAL_STDOUT
    if _REAL_STDOUT is None:
        sys.stdout.flush()
        _REAL_STDOUT = os.dup(1)
        os.dup2(2, 1)


def emit(obj):
    line = (json.dumps(obj) + '\n').encode()
    sys.stdout.flush()
    if _REAL_STDOUT is None:
        os.write(1, line)
    else:
        os.write(_REAL_STDOUT, line)


LINE_LIMIT = 6144     # bytes: the stdout line the driver parses stays below this; the full tree goes to bench_full.json + stderr


def _num(x, sig=6):
    if isinstance(x, float):
        return float('%.*g' % (sig, x))
    if isinstance(x, (list, tuple)):
        return [_num(v, sig) for v in x]
    if isinstance(x, dict):
        return {k: _num(v, sig) for k, v in x.items()}
    return x


def _pick(d, keys, sig=6):
    return {k: _num(d[k], sig) for k in keys if isinstance(d, dict) and k in d and d[k] is not None}


def _short(txt, n):
    txt = str(txt)
    return txt if len(txt) <= n else txt[:n - 3] + '...'


def compact_line(full, full_path='bench_full.json'):
    """The ONE stdout line (<= LINE_LIMIT bytes) from the full result tree: the contract's headline fields, `config`, `roofline`
    (with `traffic`), `cpu_baseline`, and one-line summaries {ms_per_step, value, frac} of the other configurations.  Everything
    else (kernel tables, notes, the per-configuration rooflines and CPU legs) lives in `full_path`.  Optional groups are dropped,
    last first, should a line ever exceed the limit -- the contract fields never are."""
    line = _pick(full, ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling'), 8)
    line['vs_baseline'] = full.get('vs_baseline')
    line.update(_pick(full, ('dtype', 'data', 'matrix_path')))
    line['dtype'] = _short(line.get('dtype', ''), 160)
    cfg = dict(full.get('config', {}))
    if 'workload' in cfg:
        cfg['workload'] = _short(cfg['workload'], 420)
    line['config'] = {k: _num(v) for k, v in cfg.items()}
    rf = full.get('roofline') or {}
    line['roofline'] = dict(_pick(rf, ('bound', 'achieved', 'peak', 'unit', 'frac', 'launches', 'avg_ms', 'flops_per_launch', 'flops_per_edge',
                                       'frac_of_fp32_mfma_peak')), traffic=_num(rf.get('traffic')))
    line['roofline']['kernel'] = _short(rf.get('kernel_symbol') or rf.get('kernel', ''), 120)
    if rf.get('traffic_source'):
        line['roofline']['traffic_source'] = _short(rf['traffic_source'], 120)
    cb = full.get('cpu_baseline')
    if cb:
        line['cpu_baseline'] = dict(_pick(cb, ('value', 'unit', 'cores', 'kind', 'ms_per_step', 'ms_per_step_scaled', 'sample_molecules',
                                               'pinned_physical_cores', 'host_logical_cpus', 'reference_cpu_mol_s')),
                                    sample=_short(cb.get('sample', ''), 360))
    optional = []      # (key, value) in the order they are kept; dropped from the end if the line is too long
    if 'sample_wall_s' in full:
        optional.append(('sample_wall_s', _pick(full['sample_wall_s'], ('value', 'molecules_per_sec'))))
    if 'configs' in full:
        summ = {}
        for name, c in full['configs'].items():
            if 'error' in c:
                summ[name] = {'error': _short(c['error'], 80)}
                continue
            e = _pick(c, ('ms_per_step', 'value', 'host_issue_ms_per_step', 'host_only_ms_per_step', 'launches_per_step'))
            r = c.get('roofline') or {}
            if r.get('frac') is not None:
                e['frac'] = _num(r['frac'])
                e['kernel'] = _short(r.get('kernel', ''), 24).split(' ')[0]
            ro = c.get('roofline_other') or {}
            if ro.get('frac') is not None:
                e['frac_' + _short(ro.get('kernel', 'other'), 24).split(' ')[0]] = _num(ro['frac'])
            if 'cpu_baseline' in c:
                e['cpu_value'] = _num(c['cpu_baseline'].get('value'))
            if isinstance(c.get('f32'), dict):
                e['f32_ms_per_step'] = _num(c['f32'].get('ms_per_step'))
            if isinstance(c.get('loss_fixed_probe_before_after'), list):
                e['loss_fixed_probe'] = [_num(x) if isinstance(x, float) else _short(str(x), 40) for x in c['loss_fixed_probe_before_after']]
            summ[name] = e
        optional.append(('configs', summ))
    for k in ('roofline_edge_b', 'segment_sum', 'aggregation_large'):
        if isinstance(full.get(k), dict) and full[k].get('frac') is not None:
            e = _pick(full[k], ('bound', 'frac', 'achieved', 'avg_ms', 'unit', 'molecules', 'ms_per_step', 'molecules_per_sec'))
            if 'unit' in e:
                e['unit'] = _short(e['unit'], 8)
            optional.append((k, e))
    if 'kernel_ms_per_step' in full:
        optional.append(('kernel_ms_per_step', _num(full['kernel_ms_per_step'])))
    # the process-group fields stay top-level keys (the multi-rank tests and the SCALE run read them there)
    line.update(_pick(full, ('ranks_seen', 'backend', 'per_rank_ms_per_step', 'gather_ms', 'gather_first_ms', 'gather_rows', 'gather_bytes',
                             'value_incl_gather'), 8))     # (same digits as ms_per_step: it is the maximum of per_rank_ms_per_step)
    line['full'] = full_path
    for k, v in optional:
        line[k] = v
    while len(json.dumps(line)) >= LINE_LIMIT and optional:
        k, _ = optional.pop()
        line.pop(k, None)
        line['dropped'] = line.get('dropped', []) + [k]
    if len(json.dumps(line)) >= LINE_LIMIT:   # contract fields alone can only get here through an absurd workload string
        line['config']['workload'] = _short(line['config'].get('workload', ''), 120)
        line.get('cpu_baseline', {}).pop('sample', None)
    return line


def emit_result(full, name='bench_full.json'):
    """Full tree -> <repo>/bench_full.json (best effort) and stderr; the compact line -> stdout."""
    path = os.path.join(ROOT, name)
    try:
        with open(path, 'w') as f:
            json.dump(full, f, indent=1)
            f.write('\n')
    except OSError:
        path = None
    sys.stderr.write('BENCH_FULL ' + json.dumps(full) + '\n')
    sys.stderr.flush()
    emit(compact_line(full, name if path else 'stderr (BENCH_FULL line)'))


def launch_env(args_gpus):
    """(world, rank, local_rank); starts the ranks when --gpus N > 1 was given to a plain `python bench.py`."""
    if 'WORLD_SIZE' not in os.environ and args_gpus > 1:
        self_launch(args_gpus)
    world, rank, local_rank = (int(os.environ.get(k, d)) for k, d in (('WORLD_SIZE', '1'), ('RANK', '0'), ('LOCAL_RANK', '0')))
    if args_gpus != world:
        raise SystemExit(f'--gpus {args_gpus} != WORLD_SIZE {world}')
    return world, rank, local_rank


def build_bond_predictor():
    import moldiff_amd as M
    from moldiff_amd.harness import default_config
    bp = M.BondPredictor(default_config('bondpred'), 8, 5).eval()
    bp.load_state_dict(M.recipe_state_dict(bp, 20230808), strict=True)
    return bp


def rank_molecules(batch, rank, world=1):
    """-> (sizes, global molecule ids) of the `batch` molecules rank `rank` samples.  The job is batch x world molecules whose sizes are
    the first batch x world draws of the reference's size recipe (utils/transforms.py:128-131) with seed 2920 (= 2023 +
    sum(ord('./outputs'))).  One rank: the draws in order.  Several ranks: the product's sharding (moldiff_amd/sample_drug3d.py:189) --
    a contiguous slice of the serpentine-by-size order (distributed.balanced_order), so every GPU gets the same number of molecules
    and, to 0.1 %, the same number of directed edges (consecutive blocks of draws differ by up to 4.8 % at 8 ranks, and the slowest
    rank is the job's time).  world=None: the r-th consecutive block of draws (development tools, per-rank fixtures)."""
    from moldiff_amd.harness import GEOM_DRUGS_MEAN_ATOMS, GEOM_DRUGS_STD_ATOMS
    from moldiff_amd.distributed import balanced_order, shard_bounds
    np.random.seed(2920)
    if world is None or world == 1:
        sizes = np.random.normal(GEOM_DRUGS_MEAN_ATOMS, GEOM_DRUGS_STD_ATOMS, size=batch * (rank + 1)).astype('int64')
        return sizes[batch * rank: batch * (rank + 1)], np.arange(batch * rank, batch * (rank + 1), dtype=np.int64)
    sizes = np.random.normal(GEOM_DRUGS_MEAN_ATOMS, GEOM_DRUGS_STD_ATOMS, size=batch * world).astype('int64')
    order = balanced_order(sizes, world)
    lo, hi = shard_bounds(len(sizes), world, rank)
    return sizes[order[lo:hi]], order[lo:hi].astype(np.int64)


def build_workload(batch, rank, device, kind='MolDiff_simple', world=1):
    import moldiff_amd as M
    from moldiff_amd.harness import default_config, placeholder_from_sizes
    sizes, _ = rank_molecules(batch, rank, None if world == 1 else world)
    ph = placeholder_from_sizes(sizes, device)
    model = M.MolDiff(default_config(kind), 8, 6).eval()
    model.load_state_dict(M.recipe_state_dict(model, 20230807), strict=True)
    return model, ph, sizes


def numa_physical_cores():
    """(node id, [one logical CPU per physical core of that NUMA node]) restricted to this process's affinity mask: the node with
    the most usable cores.  Read from sysfs; falls back to the affinity mask when the topology files are absent."""
    def parse(txt):
        out = []
        for part in txt.strip().split(','):
            if part:
                lo, _, hi = part.partition('-')
                out.extend(range(int(lo), int(hi or lo) + 1))
        return out
    try:
        allowed = set(os.sched_getaffinity(0))
    except Exception:
        allowed = set(range(os.cpu_count() or 1))
    best = (-1, sorted(allowed))
    try:
        import glob
        found = {}
        for nd in glob.glob('/sys/devices/system/node/node[0-9]*'):
            cpus = [c for c in parse(open(os.path.join(nd, 'cpulist')).read()) if c in allowed]
            phys = []
            for c in cpus:
                try:
                    sib = parse(open(f'/sys/devices/system/cpu/cpu{c}/topology/thread_siblings_list').read())
                except Exception:
                    sib = [c]
                if c == min(x for x in sib if x in allowed):
                    phys.append(c)
            found[int(os.path.basename(nd)[4:])] = phys
        if found:
            node = max(sorted(found), key=lambda k: len(found[k]))
            if found[node]:
                best = (node, found[node])
    except Exception:
        pass
    return best


def cpu_worker(spec):
    """Child process of cpu_baseline (already pinned by its parent): time the CPU oracle and print one JSON line."""
    import torch.nn.functional as F
    from oracle import moldiff_oracle as O
    from moldiff_amd.harness import placeholder_from_sizes
    kind, batch, budget_s, ncores = spec['kind'], spec['batch'], spec['budget_s'], spec['ncores']
    model, ph_cpu, sizes = build_workload(batch, 0, None, kind)
    gkw = {}
    if kind == 'MolDiff':
        gkw = dict(Pb={k: v.detach().cpu() for k, v in build_bond_predictor().state_dict().items()},
                   cfgb=dict(num_timesteps=1000, num_blocks=8, cutoff=20), guidance=['uncertainty', 1e-4])
    P = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    tabs = {'pos': {k: P['pos_transition.' + k] for k in ('coef_x0', 'coef_xt', 'std')},
            'node': {k: P['node_transition.' + k] for k in ('q_mats', 'transpopse_q_onestep_mats')},
            'edge': {k: P['edge_transition.' + k] for k in ('q_mats', 'transpopse_q_onestep_mats')}}
    cfg = dict(num_timesteps=1000, num_blocks=6, cutoff=15)
    sizes = np.asarray(sizes, dtype=np.int64)
    e_all = int((sizes * (sizes - 1)).sum())
    g = torch.Generator().manual_seed(0)

    def make(nmol):
        ph = placeholder_from_sizes(sizes[:nmol])
        N, Eh = len(ph['batch_node']), len(ph['batch_halfedge'])
        st = {'h_node': F.one_hot(torch.randint(0, 8, (N,), generator=g), 8).float(), 'pos': torch.randn(N, 3, generator=g),
              'h_halfedge': F.one_hot(torch.randint(0, 6, (Eh,), generator=g), 6).float()}
        st['log_node'] = torch.log(st['h_node'].clamp(min=1e-30))
        st['log_halfedge'] = torch.log(st['h_halfedge'].clamp(min=1e-30))
        return dict(ph, n_graphs=nmol), st, N, Eh

    def one(graph, st, N, Eh, step):
        noise = {'eps_pos': torch.randn(N, 3, generator=g), 'u_node': torch.rand(N, 8, generator=g),
                 'u_halfedge': torch.rand(Eh, 6, generator=g)}
        t0 = time.perf_counter()
        with torch.no_grad():
            new, _ = O.sample_step(P, cfg, tabs, st, graph, step, noise, **gkw)
        return {k: new[k] for k in ('h_node', 'pos', 'h_halfedge', 'log_node', 'log_halfedge')}, time.perf_counter() - t0

    # 1. thread ladder on the batch itself (round 5; it used to run on a 32-molecule probe, which picked 16 threads where the
    #    256-molecule batch wants more: per-op parallel efficiency depends on the row count).  One step per rung, rungs inside the
    #    parent's pinned set (physical cores of one NUMA node), smallest first; the ladder stops early when it has used 3x the budget.
    #    Config #3 (twice the cost per step) ladders on the first 96 molecules.
    nprobe = batch if kind != 'MolDiff' else min(96, batch)
    graph, st, N, Eh = make(nprobe)
    e_probe = int((sizes[:nprobe] * (sizes[:nprobe] - 1)).sum())
    torch.set_num_threads(min(16, ncores))
    one(graph, st, N, Eh, 999)          # page in / MKL init
    ladder, best, spent = [], (float('inf'), 1), 0.0
    for th in sorted({c for c in (8, 16, 32, 64, 128) if c <= ncores} | {ncores}):
        torch.set_num_threads(th)
        _, dt = one(graph, st, N, Eh, 997)
        ladder.append((th, round(dt, 3)))
        spent += dt
        if dt < best[0]:
            best = (dt, th)
        if spent > 3.0 * budget_s:
            break
    threads = best[1]
    torch.set_num_threads(threads)
    # 2. the sample: the FULL batch when warm-up + 3 steps fit ~2x the budget (estimated from the ladder), else the largest prefix
    est_full = best[0] * e_all / e_probe
    nmol = batch if 4 * est_full <= 2.0 * budget_s else int(max(min(32, batch), min(batch, batch * (2.0 * budget_s / 4) / est_full)))
    graph, st, N, Eh = make(nmol)
    e_sub = int((sizes[:nmol] * (sizes[:nmol] - 1)).sum())
    st, _ = one(graph, st, N, Eh, 996)  # warm-up at this size
    times = []
    for j in range(3):
        st, dt = one(graph, st, N, Eh, 995 - j)
        times.append(dt)
    w_probe = nprobe
    print('CPU_WORKER ' + json.dumps({'threads': threads, 'ladder': ladder, 'ladder_molecules': w_probe, 'nmol': nmol, 'e_sub': e_sub, 'e_all': e_all, 'times': times}))


def cpu_baseline(kind, batch, budget_s=20.0):
    """Time the CPU oracle (oracle/moldiff_oracle.py, pinned to the reference) on the host: a child process pinned (taskset) to one
    logical CPU per PHYSICAL core of ONE NUMA node, thread count chosen by a ladder inside that set, then 1 warm-up + 3 timed
    consecutive denoising steps of the full batch (config #2) or of the largest prefix of it that fits the budget (config #3; cost is
    proportional to the directed edges and scaled by that ratio)."""
    import shutil
    import subprocess
    node, cpus = numa_physical_cores()
    spec = {'kind': kind, 'batch': batch, 'budget_s': budget_s, 'ncores': len(cpus)}
    cmd = [sys.executable, os.path.abspath(__file__), '--cpu-worker', json.dumps(spec)]
    if shutil.which('taskset'):
        cmd = ['taskset', '-c', ','.join(map(str, cpus))] + cmd
    env = dict(os.environ, OMP_NUM_THREADS=str(len(cpus)), MKL_NUM_THREADS=str(len(cpus)), HIP_VISIBLE_DEVICES='')
    r = subprocess.run(cmd, env=env, capture_output=True, text=True)
    line = next((ln for ln in r.stdout.splitlines() if ln.startswith('CPU_WORKER ')), None)
    if line is None:
        raise RuntimeError('cpu baseline worker failed: ' + r.stderr[-2000:])
    w = json.loads(line[len('CPU_WORKER '):])
    per_step = float(np.mean(w['times'])) * w['e_all'] / w['e_sub']
    value = batch / (per_step * T_STEPS)
    guided = kind == 'MolDiff'
    return {'value': value, 'unit': 'molecules/sec', 'cores': w['threads'], 'kind': 'port',
            'sample': f"3 consecutive denoising steps (after 1 warm-up) of {'the full batch' if w['nmol'] == batch else 'the first %d' % w['nmol']} "
                      f"of the {batch} molecules ({w['e_sub']} of {w['e_all']} directed edges; cost scaled by that ratio) with the torch-CPU "
                      f"oracle, fp32, process pinned to the {len(cpus)} physical cores of NUMA node {node} (one logical CPU per core), "
                      f"{w['threads']} threads = best of the ladder {w['ladder']} (threads, s/step on {w.get('ladder_molecules', 32)} molecules of the batch); scaled to "
                      f"T=1000 (per-step cost is step-independent)",
            'ms_per_step': per_step * 1e3, 'step_seconds': w['times'], 'pinned_physical_cores': len(cpus), 'numa_node': node,
            'host_logical_cpus': os.cpu_count(), 'timed_steps': 3, 'sample_molecules': w['nmol'],
            'reference_cpu_mol_s': None if guided else REFERENCE_CPU_MOL_S,
            'reference_cpu_note': ('BASELINE.md has no measured CPU number for config #3' if guided else
                                   'BASELINE.md: the real reference (PyTorch CPU), config #2, 8 cores of the survey container: 2.64 s/step at '
                                   '256 molecules.  Port vs reference on the SAME host at the SAME time (build container, 8 cores, round 3): '
                                   'reference 12.0 s/step, port 10.5 s/step -- the port is not slower than the reference; hosts differ')}


# --------------------------------------------------------------------------------------------------
# Training-step benchmark (BASELINE.json configs[4]); `python bench.py --train ...`.
# --------------------------------------------------------------------------------------------------
def clean_batch(sizes, seed, device):
    from moldiff_amd.harness import placeholder_from_sizes
    ph = placeholder_from_sizes(sizes, device)
    g = np.random.Generator(np.random.PCG64(seed))
    N, Eh = int(ph['batch_node'].numel()), int(ph['batch_halfedge'].numel())
    node_type = torch.from_numpy(g.integers(0, 7, N)).to(device)
    pos = torch.from_numpy((g.standard_normal((N, 3)) * 2.0).astype(np.float32)).to(device)
    half = torch.from_numpy((g.random(Eh) < 0.25) * g.integers(1, 5, Eh)).to(device)
    return (node_type, pos, ph['batch_node'], half, ph['halfedge_index'], ph['batch_halfedge'], len(sizes))



def train_cpu_baseline(kind, model, sizes, budget_s):
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



TRAIN_DTYPE = {'f32': 'f32', 'bf16': 'bf16 GEMM operands, f32 accumulate / elsewhere',
               'fp16': 'f16 Linear operands and results (f32 accumulate), f32 LayerNorm / loss / sums, dynamic loss scale '
                       '(the reference\'s use_amp: True)',
               'fp16_f32store': 'as fp16, but the float16 values kept in fp32 containers',
               'bf16_autocast': 'bf16 Linear operands and results, f32 elsewhere'}
# executed matrix FLOPs of one training step = forward + data gradient + weight gradient of every Linear, each the hoisted forward
# count of SURVEY 8(d) (5,033,472 E + 5,935,104 N + 8,960 Eh for the 6-block denoiser; the O(rows x 8) loss tail is not counted)
PEAK_F16_MFMA = 2500.0   # TFLOP/s dense (MI355X_MICROARCH.md)


def train_measure(model_kind, precision, batch_size, steps, warmup, dev, rank=0, world=1, dist=None):
    """`warmup` + `steps` optimisation steps of BASELINE config #5's step (scripts/train_drug3d.py:88-109: get_loss, backward,
    gradient all-reduce, clip, AdamW) on one synthetic batch of the reference's size recipe.  Returns the measurement dict."""
    import moldiff_amd as M
    from moldiff_amd.harness import default_config, GEOM_DRUGS_MEAN_ATOMS, GEOM_DRUGS_STD_ATOMS
    from moldiff_amd.trainer import Trainer
    np.random.seed(2920)
    sizes = np.random.normal(GEOM_DRUGS_MEAN_ATOMS, GEOM_DRUGS_STD_ATOMS, size=batch_size * (rank + 1)).astype('int64')
    sizes = np.maximum(sizes[batch_size * rank:], 2)
    if model_kind == 'bondpred':
        model = M.BondPredictor(default_config('bondpred'), 8, 5)
        model.load_state_dict(M.recipe_state_dict(model, 20230808), strict=True)
    else:
        model = M.MolDiff(default_config(model_kind), 8, 6)
        model.load_state_dict(M.recipe_state_dict(model, 20230807), strict=True)
    model = model.to(dev).train()
    tr = Trainer(model, lr=1e-4, betas=(0.99, 0.999), weight_decay=1e-8, max_grad_norm=50.0, precision=precision)
    batch = clean_batch([int(s) for s in sizes], 100 + rank, dev)
    torch.manual_seed(2023 + rank)
    losses = []

    # fixed probe (outside the timed region): the loss at FIXED time steps and FIXED noise, evaluated under no_grad by the fp32 sampling
    # kernels before the first and after the last optimisation step.  `loss_first_last` compares two steps with different random t and
    # noise (the per-step loss of a diffusion model swings by 2x with t alone); this pair shows whether the weights descended.
    def fixed_probe_loss():
        try:
            g = torch.Generator(device='cpu').manual_seed(77)
            nn_, ne_, nm_ = int(batch[1].shape[0]), int(batch[3].shape[0]), int(batch[6])
            tfix = ((torch.arange(nm_, dtype=torch.int64) * 997) % 1000).to(dev)
            noise = dict(eps_pos=torch.randn(nn_, 3, generator=g).to(dev), u_node=torch.rand(nn_, 8, generator=g).to(dev))
            if model_kind != 'bondpred':
                noise['u_halfedge'] = torch.rand(ne_, 6, generator=g).to(dev)
            with torch.no_grad():
                return float(model.get_loss(*batch, time_step=tfix, noise=noise)['loss'])
        except Exception as e:   # never let the probe break the measurement
            return repr(e)

    probe_before = fixed_probe_loss()
    for _ in range(warmup):
        tr.step(*batch)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    barrier()
    torch.cuda.reset_peak_memory_stats()
    t0 = time.perf_counter()
    for _ in range(steps):
        losses.append(tr.step(*batch)['loss'])
    issued = time.perf_counter() - t0      # the host has ISSUED every step (nothing in a step waits for the device)
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    ms = elapsed / steps * 1e3
    # host-only time of a step: three more steps, each issued with the device idle (a synchronize before it).  `issued` above cannot be
    # shorter than the GPU's time by more than one step: Trainer.step waits for the PREVIOUS step's class-range check at its top
    host_only = 0.0
    for _ in range(3):
        torch.cuda.synchronize()
        h0 = time.perf_counter()
        tr.step(*batch)
        host_only += time.perf_counter() - h0
    torch.cuda.synchronize()
    launches = None
    try:
        from moldiff_amd import train_ops as _T
        F = _T._fast()
        if F is not None:
            n0 = F.launches()
            tr.step(*batch)
            torch.cuda.synchronize()
            launches = int(F.launches() - n0)
    except Exception:
        launches = None
    probe_after = fixed_probe_loss()
    N, Eh = int(batch[1].shape[0]), int(batch[3].shape[0])
    E = 2 * Eh
    nb = 8 if model_kind == 'bondpred' else 6
    fwd = (5033472.0 / 6 * nb) * E + (5935104.0 / 6 * nb) * N + 8960.0 * Eh if model_kind != 'bondpred' else \
        ((10240 + 590336 + 393344 - 442368.0) * nb) * E + (966656.0 * nb) * N
    flop = 3.0 * fwd
    half = precision in ('fp16', 'fp16_f32store', 'bf16', 'bf16_autocast')
    peak = PEAK_F16_MFMA if half else PEAK_FP32_MFMA
    ach = flop / (ms * 1e-3) / 1e12
    out = {'metric': 'molecules/sec (training step: forward + backward + all-reduce + clip + AdamW)', 'value': batch_size * world / (ms / 1e3),
           'unit': 'molecules/sec', 'n_gpus': world, 'steps': steps, 'warmup': warmup, 'ms_per_step': ms,
           'host_issue_ms_per_step': issued / steps * 1e3,
           'host_only_ms_per_step': host_only / 3 * 1e3,
           'host_path': ('C++ operator bodies (moldiff_amd/_mdx_fast.so)' if launches is not None else 'Python operator bodies (MDX_TRAIN_FAST=0)'),
           'library_launches_from_cpp_bodies_per_step': launches,
           'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
           'dtype': TRAIN_DTYPE[precision], 'data': 'synthetic',
           'config': {'workload': f'train_{model_kind}.yml: batch_size={batch_size} molecules/GPU (rank 0: N={N} atoms, E={E} directed '
                                  f'edges), AdamW lr 1e-4 betas (0.99,0.999) wd 1e-8, max_grad_norm 50; recipe weights',
                      'parallelism': f'data-parallel x{world}, one flat-gradient all-reduce per step',
                      'parameters': tr.flat.numel},
           'roofline': {'bound': 'mfma', 'kernel': 'whole step (layer-operator GEMMs: hgemm_nt_rows / hgemm_tn_tr / sgemm kernels of mdx_train.hip)',
                        'achieved': ach, 'peak': peak, 'unit': 'TFLOP/s', 'frac': ach / peak,
                        'frac_of_fp32_mfma_peak': ach / PEAK_FP32_MFMA, 'flops_per_step': flop,
                        'flops_what': '3 x the hoisted forward count (forward, data gradient, weight gradient of every Linear)',
                        'traffic': None,
                        'note': 'whole-step fraction: the step is a chain of ~690 launches (rocprofv3: profiles/r6_train_fp16_kernel_stats.csv), '
                                'most of its time in latency- / store-bound fused row-owner kernels (DESIGN.md section 3.3)'},
           'peak_hbm_gb': torch.cuda.max_memory_allocated() / 2 ** 30,
           'loss_first_last': [float(losses[0]), float(losses[-1])],
           'loss_fixed_probe_before_after': [probe_before, probe_after],
           'loss_fixed_probe_what': f'get_loss at fixed time steps and fixed noise (fp32 kernels, no_grad) before the first and after the last of '
                                    f'the {warmup + steps + 4} optimisation steps of this run; loss_first_last are two steps with different random t'}
    return out, model, sizes


def main_train():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--batch', type=int, default=256)
    ap.add_argument('--model', default='MolDiff', choices=['MolDiff', 'MolDiff_simple', 'bondpred'])
    ap.add_argument('--precision', default='f32', choices=['f32', 'bf16', 'fp16', 'fp16_f32store', 'bf16_autocast'],
                    help="'fp16' = the reference's use_amp arithmetic (autocast float16 + dynamic loss scale); 'bf16' = GEMM operands only")
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-budget', type=float, default=20.0)
    args = ap.parse_args()
    world, rank, local_rank = launch_env(args.gpus)
    own_stdout()
    if not torch.cuda.is_available():
        raise SystemExit('bench.py --train needs a ROCm GPU (no CPU fallback).')
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)

    out, model, sizes = train_measure(args.model, args.precision, args.batch, args.steps, args.warmup, dev, rank, world, dist)
    if rank == 0:
        if not args.no_cpu_baseline and world == 1:
            out['cpu_baseline'] = train_cpu_baseline(args.model, model, sizes, args.cpu_budget)
            out['speedup_vs_cpu_baseline'] = out['value'] / out['cpu_baseline']['value']
        emit(out)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()



def _profile(L, k):
    c, ms = ctypes.c_int64(), ctypes.c_double()
    from moldiff_amd import _lib
    _lib.check(L.mdx_profile_read(k, ctypes.byref(c), ctypes.byref(ms)))
    return c.value, ms.value


def run_chain(sm, steps, warmup, barrier, start=0, prof=1, whole_run=False):
    """`warmup` untimed + `steps` timed iterations of the reverse chain; hipEvent kernel timing on during the timed region
    (`prof`: 1 = every kernel, 2 << k = kernel k only -- an event pair costs ~3.4 us of stream time).
    whole_run (steps == T): the timed region is what MolDiff.sample runs after its one-off set-up -- the prior draw, then
    iterations 0..T-1 writing trajectory frames 1..T -- so `value` IS n_graphs / wall of a sampling run.
    Returns (elapsed seconds, {kernel: (launches, total ms)})."""
    from moldiff_amd import _lib
    L = _lib.lib()
    i = start
    for _ in range(warmup):
        sm.step(i % T_STEPS)
        i += 1
    barrier()
    L.mdx_profile_enable(prof)
    t0 = time.perf_counter()
    if whole_run:
        assert steps == T_STEPS
        sm.init()
        i = 0
    for _ in range(steps):
        sm.step(i % T_STEPS)
        i += 1
    barrier()
    elapsed = time.perf_counter() - t0
    L.mdx_profile_enable(0)
    names = (('edge_a', 0), ('edge_b', 1), ('node', 2), ('aggregate', 3), ('edge_bwd', 4))
    if not prof:
        return elapsed, {n: (0, 0.0) for n, _ in names}
    return elapsed, {n: _profile(L, k) for n, k in names}


def roofline_mfma(name, kernel, flop_per_edge, E, prof):
    c, ms = prof[name]
    avg = ms / max(c, 1)
    ach = flop_per_edge * E / (avg * 1e-3) / 1e12 if c else None
    return {'bound': 'mfma', 'kernel': kernel, 'achieved': ach, 'peak': PEAK_FP32_MFMA, 'unit': 'TFLOP/s',
            'frac': (ach / PEAK_FP32_MFMA) if ach else None, 'traffic': None, 'launches': c, 'avg_ms': avg,
            'flops_per_launch': flop_per_edge * E, 'flops_per_edge': flop_per_edge}


SPLIT_DTYPE = ('f32 operands split into float16 hi + lo halves (22 significand bits; lo scaled by 2^11), three '
               'v_mfma_f32_16x16x32_f16 per 32-wide k-group (Whi Xhi + Whi Xlo + Wlo Xhi), f32 accumulation; per-node layers, LayerNorm, '
               'gates, segment sums, transitions: f32 as in the exact path')


def roofline_split(name, kernel, flop_per_edge, E, prof):
    """The split float16 path's kernel against BOTH peaks: `achieved` counts the ALGORITHMIC fp32 FLOPs (the same count as the exact
    path's line, so `frac_of_fp32_mfma_peak` can exceed 1); the matrix pipe actually executes three float16 products per one,
    `frac` = 3 x achieved / the dense float16 MFMA peak."""
    r = roofline_mfma(name, kernel, flop_per_edge, E, prof)
    if r['achieved']:
        r['frac_of_fp32_mfma_peak'] = r['achieved'] / PEAK_FP32_MFMA
        r['executed_f16_tflops'] = 3.0 * r['achieved']
        r['peak'], r['frac'] = PEAK_F16_MFMA, 3.0 * r['achieved'] / PEAK_F16_MFMA
        r['bound_note'] = ('no longer bound by the matrix pipe: the per-wave weight stream (L2 -> VGPR, the same bytes as fp32) and the '
                           'operand conversions are what is left (DESIGN section 3.4)')
    return r


def partial_rows(sizes):
    """Partial rows edge kernel A's in-kernel aggregation emits for fully connected molecules of these sizes: node j of an n-atom
    molecule owns edges j(n-1) .. (j+1)(n-1) of the molecule, cut wherever one of the molecule's 16-edge units ends."""
    tot = 0
    for n in sizes:
        n = int(n)
        if n < 2:
            continue
        j = np.arange(n, dtype=np.int64)
        tot += int((((j + 1) * (n - 1) - 1) // 16 - (j * (n - 1)) // 16 + 1).sum())
    return tot


def aggregation_line(N, E, prof, sizes=None, agg=True):
    """The reduction pass after edge kernel A.  Round 3: the 256-wide message sums and the BondFFN-right sums are formed INSIDE
    edge kernel A (one partial row per node and 16-edge unit), so this pass only combines ~2.5 partial rows per node and still
    sums the BondFFN-left rows through the by-right index list."""
    c, ms = prof['aggregate']
    avg = ms / max(c, 1)
    if agg and c == 0:
        return {'bound': 'hbm', 'kernel': 'none: the reduction left after edge kernel A\'s in-kernel segment sums (combine ~2.5 partial rows per '
                                          'node, by-right BondFFN sum) runs inside node_kernel (its MID stage and a second set of workgroups); '
                                          '(as a launch of its own, seg_reduce_block2_kernel, it took 16 us). The '
                                          'scatter/gather primitive by itself: see segment_sum', 'achieved': None, 'peak': PEAK_HBM,
                'unit': 'GB/s', 'frac': None, 'launches': 0}
    if agg and sizes is not None:
        P = partial_rows(sizes)
        nbytes = P * (1024.0 + 256.0) + E * (256.0 + 4.0) + N * 1536.0
        what = ('seg_reduce_block2_kernel: %d partial rows (256- and 64-wide, written by edge kernel A) -> aggr (N,256), SR (N,64); '
                'FL (E,64) -> SL (N,64) through the by-right index list' % P)
    else:
        nbytes = 1536.0 * (E + N)   # reads M (E,256) + FL, FR (E,64 each), writes aggr (N,256) + SL, SR (N,64 each)
        what = 'seg_reduce_block_kernel: (E,256)->(N,256) message aggregation + the two (E,64)->(N,64) BondFFN sums'
    agg_bw = nbytes / (avg * 1e-3) / 1e9 if c else None
    return {'bound': 'hbm', 'kernel': what, 'achieved': agg_bw, 'peak': PEAK_HBM,
            'unit': 'GB/s (HBM or Infinity Cache: the operands were written by the preceding kernel and fit the 256 MiB MALL)',
            'frac': (agg_bw / PEAK_HBM) if agg_bw else None, 'bytes_per_launch': nbytes, 'launches': c, 'avg_ms': avg}


def segment_sum_line(sm, dev):
    """The scatter/gather primitive by itself (mdx_segment_sum = the deterministic replacement of torch_scatter.scatter_sum,
    models/graph.py:50): (E,256) -> (N,256) by left node, hipEvent-timed stand-alone on this workload's graph.  Inside the sampling
    step the same sum is now fused into edge kernel A; the primitive still serves NodeBlock.forward and the training operators."""
    from moldiff_amd import _lib
    L = _lib.lib()
    N, E = sm.N, 2 * sm.Eh
    src = torch.randn(E, 256, device=dev)
    out = torch.empty(N, 256, device=dev)
    ws, nb = sm.g.workspace(dev)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    for _ in range(3):
        _lib.check(L.mdx_segment_sum(sm.g.h, _lib.ptr(src), 256, 2, _lib.ptr(out), ws, nb, _lib.stream()))
    torch.cuda.synchronize()
    reps = 20
    ev[0].record()
    for _ in range(reps):
        _lib.check(L.mdx_segment_sum(sm.g.h, _lib.ptr(src), 256, 2, _lib.ptr(out), ws, nb, _lib.stream()))
    ev[1].record()
    torch.cuda.synchronize()
    avg = ev[0].elapsed_time(ev[1]) / reps
    nbytes = 1024.0 * (E + N)
    bw = nbytes / (avg * 1e-3) / 1e9
    return {'bound': 'hbm', 'kernel': 'seg_reduce_kernel<256> via mdx_segment_sum (rows already in plan order): (E,256)->(N,256) by left node, stand-alone, %d launches back to '
                                      'back on torch\'s current stream (torch events)' % reps,
            'achieved': bw, 'peak': PEAK_HBM, 'unit': 'GB/s', 'frac': bw / PEAK_HBM, 'bytes_per_launch': nbytes, 'avg_ms': avg,
            'operand_mib': E * 1024.0 / 2 ** 20}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=1000)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--batch', type=int, default=256, help='molecules per GPU')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-budget', type=float, default=20.0)
    ap.add_argument('--guided', action='store_true',
                    help="make BASELINE config #3 (full model + bond-predictor 'uncertainty' guidance) the headline instead of "
                         "config #2; by default config #3 is reported under configs.guided of the same line")
    ap.add_argument('--headline-only', action='store_true', help='skip the extra single-GPU measurements (configs, sample(), B=2048)')
    args = ap.parse_args()

    world, rank, local_rank = launch_env(args.gpus)
    own_stdout()
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a ROCm GPU (no CPU fallback).')
    # one process per GPU over RCCL.  MDX_BENCH_BACKEND=gloo (+ ranks sharing a GPU when there are fewer devices than
    # ranks) exists only so the multi-rank control flow can be exercised on a 1-GPU box; it is never the default.
    backend = os.environ.get('MDX_BENCH_BACKEND', 'nccl')
    dev_index = local_rank if backend == 'nccl' else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    dev = torch.device('cuda', dev_index)
    # The group is created for ONE rank too (round 4): the N = 1 line then runs the same RCCL code as N = 8 -- communicator
    # setup, the max-over-ranks all-reduce, the end-of-run gather on device tensors -- instead of skipping it.  A single rank
    # whose group cannot be created (no RCCL on the box) still reports its line, with `backend` saying why.
    dist, backend_note = None, None
    if world > 1 or backend != 'none':
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        if 'MASTER_PORT' not in os.environ:
            import socket
            with socket.socket() as sk:
                sk.bind(('127.0.0.1', 0))
                os.environ['MASTER_PORT'] = str(sk.getsockname()[1])
        try:
            if backend == 'nccl':
                dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
            else:
                dist.init_process_group(backend, rank=rank, world_size=world)
        except Exception as e:
            if world > 1:
                raise
            dist, backend_note = None, 'none (single rank; process group not created: %r)' % (e,)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    def sampler_for(kind, batch, rk, nranks=1, bp_path=None, traj=False, **kw):
        model, ph_cpu, sizes = build_workload(batch, rk, None, kind, nranks)
        model = model.to(dev)
        gkw = {}
        if kind == 'MolDiff':
            gkw = dict(bond_predictor=build_bond_predictor().to(dev), guidance=['uncertainty', 1e-4])
            gkw['bond_predictor'].matrix_path = bp_path     # None: the process default, like the denoiser
        ph = {k: v.to(dev) for k, v in ph_cpu.items()}
        mol_ids = rank_molecules(batch, rk, None if nranks == 1 else nranks)[1]     # noise is keyed by the global molecule id
        sm = model.sampler(batch, ph['batch_node'], ph['halfedge_index'], ph['batch_halfedge'], seed=2023, mol_ids=mol_ids,
                           return_traj=traj, **gkw, **kw)
        sm.init()
        return sm, model, ph, ph_cpu, gkw

    def config_line(kind, sm, steps, warmup, elapsed, prof, nranks):
        N, E = sm.N, 2 * sm.Eh
        ms = elapsed / steps * 1e3
        line = {'ms_per_step': ms, 'value': args.batch * nranks / (ms * T_STEPS / 1e3), 'unit': 'molecules/sec', 'steps': steps,
                'warmup': warmup,
                'workload': ('sample_MolDiff.yml: full model + bond_predictor guidance [uncertainty, 1e-4], ' if kind == 'MolDiff'
                             else 'sample_MolDiff_simple.yml: no bond guidance, ') +
                            'batch_size=%d molecules/GPU, T=1000 steps; sizes ~ reference recipe seed 2920%s (rank 0: N=%d atoms, '
                            'E=%d directed edges); recipe weights' % (args.batch, '' if nranks == 1 else
                                ', the %d draws dealt to the ranks in serpentine order by size' % (args.batch * nranks), N, E),
                'kernel_ms_per_step': {k: v[1] / steps for k, v in prof.items() if v[0]}}
        return line

    head_kind = 'MolDiff' if args.guided else 'MolDiff_simple'
    hkw = {'overlap_guidance': True} if (args.guided and os.environ.get('MDX_BENCH_OVERLAP')) else {}  # A/B: guidance on a side stream
    # the headline chain records the trajectory like a default model.sample() call (compact frames, moldiff_amd/traj.py): every
    # step writes its frame, and with --steps 1000 the timed region is the prior draw + the complete run (run_chain whole_run)
    whole = args.steps == T_STEPS
    sm, model, ph, ph_cpu, gkw = sampler_for(head_kind, args.batch, rank, world, traj=True, **hkw)
    sizes_head = torch.bincount(ph_cpu['batch_node'], minlength=args.batch).numpy()
    N, E = sm.N, 2 * sm.Eh
    # Timed region: only the roofline kernel (edge kernel A) carries hipEvent brackets -- 25 event pairs per step on every kernel
    # cost 0.17 ms of a 7.3 ms step (tools/profile_overhead.py).  The other kernels' durations come from a short second pass
    # OUTSIDE the timed region (same chain, continued).
    elapsed, prof = run_chain(sm, args.steps, args.warmup, barrier, prof=2 << 0, whole_run=whole)
    _, prof_all = run_chain(sm, min(args.steps, 40), 0, barrier, start=args.steps + args.warmup)
    steps_all = min(args.steps, 40)
    multi = None
    if dist is not None:
        comm_dev = dev if backend == 'nccl' else torch.device('cpu')
        mine = torch.tensor([elapsed / args.steps * 1e3], dtype=torch.float64, device=comm_dev)
        per_rank = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(per_rank, mine)
        tt = torch.tensor([elapsed], dtype=torch.float64, device=comm_dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        # the path's ONLY data collective: the end-of-run gather of every rank's last-step predictions to rank 0
        # (moldiff_amd/distributed.gather_pred = what sample_drug3d does once per batch).  Timed twice, outside the step
        # loop's timed region: the first call pays the communicator's lazy setup, the second is the steady cost.
        from moldiff_amd.distributed import gather_pred
        pred = sm.result()['pred']
        gt = []
        for _ in range(2):
            barrier()
            t0 = time.perf_counter()
            got = gather_pred([p.to(comm_dev) for p in pred], dst=0)
            torch.cuda.synchronize()
            g = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=comm_dev)
            dist.all_reduce(g, op=dist.ReduceOp.MAX)
            gt.append(float(g.item()) * 1e3)
        nrows = torch.tensor([sm.N, sm.Eh], dtype=torch.int64, device=comm_dev)
        dist.all_reduce(nrows)
        if rank == 0:
            assert got[0].shape[0] == int(nrows[0]) and got[2].shape[0] == int(nrows[1]), 'gather_pred lost rows'
        multi = {'ranks_seen': dist.get_world_size(), 'backend': 'rccl (torch "nccl")' if backend == 'nccl' else backend,
                 'per_rank_ms_per_step': [float(x.item()) for x in per_rank], 'gather_first_ms': gt[0], 'gather_ms': gt[1],
                 'gather_rows': [int(nrows[0]), int(nrows[1])],
                 'gather_bytes': int(nrows[0]) * (8 + 3) * 4 + int(nrows[1]) * 6 * 4,
                 'gather_what': 'one distributed.gather_pred of all ranks\' last-step predictions to rank 0 (1 count all_gather + ONE padded '
                                'gather-to-rank-0 of a flat buffer holding all three tensors), max over ranks; happens once per 1000-step run, outside ms_per_step'}
    ms_per_step = elapsed / args.steps * 1e3
    value = args.batch * world / (ms_per_step * T_STEPS / 1e3)
    # a timing of garbage is worth nothing: after the timed region, every rank checks what its chain left behind
    st, pr = sm.state(), sm.result()['pred']
    ok = all(bool(torch.isfinite(p).all()) for p in pr) and bool(torch.isfinite(st['pos']).all())
    ok = ok and bool((st['h_node'].sum(-1) == 1).all()) and bool((st['h_halfedge'].sum(-1) == 1).all())
    ok = ok and float(st['pos'].abs().max()) < 1e3
    if not ok:
        raise SystemExit('bench.py: rank %d: the chain left non-finite predictions / positions or a state that is not one-hot after '
                         '%d steps -- no line is reported for that' % (rank, args.steps + args.warmup))

    out = None
    if rank == 0:
        head = config_line(head_kind, sm, args.steps, args.warmup, elapsed, prof, world)
        head['kernel_ms_per_step'] = dict({k: v[1] / steps_all for k, v in prof_all.items() if v[0]}, edge_a=prof['edge_a'][1] / args.steps)
        head['kernel_ms_note'] = ('edge_a: hipEvents inside the timed region; the others: a %d-step pass right after it (event pairs on '
                                  'every kernel would add 0.17 ms to each timed step)' % steps_all)
        out = {
            'metric': 'molecules/sec (1000-step GEOM-Drugs sampling)', 'value': value, 'unit': 'molecules/sec',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': ms_per_step,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': head['workload'], 'molecules_per_gpu': args.batch, 'num_timesteps': T_STEPS,
                       'parallelism': f'independent streams x{world}', 'value_formula': 'batch*n_gpus / (ms_per_step*T/1000)',
                       'timed_region': ('prior draw + all 1000 iterations with the trajectory recorded = one MolDiff.sample run after its one-off '
                                        'set-up (graph plan, buffers)' if whole else
                                        '%d iterations of the reverse chain, trajectory frames recorded as in MolDiff.sample' % args.steps)},
            'roofline': roofline_mfma('edge_a', EDGE_A_NAME, FLOP_EDGE_A, E, prof),
            'roofline_edge_b': roofline_mfma('edge_b', EDGE_B_NAME, FLOP_EDGE_B, E, prof_all),
            'aggregation': aggregation_line(N, E, prof_all, sizes_head, True),
            'kernel_ms_per_step': head['kernel_ms_per_step'], 'kernel_ms_note': head['kernel_ms_note'],
            'ranks_seen': 1, 'backend': backend_note or 'none',
            'outputs_checked': 'after the timed region, on every rank: predictions and positions finite, |pos| < 1e3, atom / bond states one-hot',
        }
        from moldiff_amd import _lib as _lp
        if _lp.resolve_matrix_path(None) != 'exact_f32':
            # MOLDIFF_MATRIX_PATH selected the opt-in split path for the WHOLE run (profiling / A-B use; never the driver's command):
            # say so on the line instead of reporting split numbers under the exact path's labels
            out['dtype'], out['matrix_path'] = SPLIT_DTYPE, _lp.resolve_matrix_path(None)
            out['roofline'] = roofline_split('edge_a', 'edge_a2s_kernel (split float16 build of edge kernel A)', FLOP_EDGE_A, E, prof)
            out['roofline_edge_b'] = roofline_split('edge_b', 'edge_b2s_kernel (split float16 build of edge kernel B)', FLOP_EDGE_B, E, prof_all)
        if multi is not None:
            out.update(multi)
            # a complete 1000-step run of every rank plus the gather, i.e. what the entry point's batch loop costs
            out['value_incl_gather'] = args.batch * world / (ms_per_step * T_STEPS / 1e3 + multi['gather_ms'] / 1e3)
        # HBM traffic per launch comes from the PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in their own runs,
        # gfx950 correction applied) committed under profiles/; a Python process cannot collect PMCs on itself.  The figure is
        # quoted only when the summary was taken with THIS build's kernels (the library names the kernel each timing slot
        # brackets; a summary that does not know that exact name -- template arguments included -- is stale and refused).
        try:
            import glob
            from moldiff_amd import _lib
            L = _lib.lib()
            name_a, name_g = (L.mdx_profile_kernel_name(k).decode() for k in (0, 3))
            out['roofline']['kernel_symbol'], out['aggregation']['kernel_symbol'] = name_a, name_g
            # newest summary of THIS configuration first: the round-4 directory also holds summaries of the guided step and of the
            # split-float16 build, which know other kernels
            pm = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_pmc_summary.json')))
            pm = [f for f in pm if 'guided' not in os.path.basename(f)]
            pm.sort(key=lambda f: ('split' in os.path.basename(f)) == ('matrix_path' in out))   # stable: the matching build's newest round last
            if pm and args.batch == 256 and not args.guided:
                ks = json.load(open(pm[-1]))['kernels']
                src = os.path.relpath(pm[-1], ROOT)
                if name_a in ks:
                    out['roofline']['traffic'] = ks[name_a]['hbm_bytes_per_launch']
                    out['roofline']['traffic_source'] = src + ' (PMC passes of the same command; not measured in this run)'
                else:
                    out['roofline']['traffic_source'] = f'refused: {src} has no kernel {name_a} (taken with another kernel set: {sorted(ks)})'
                if name_g in ks:
                    out['aggregation']['traffic'] = ks[name_g]['hbm_bytes_per_launch']
        except Exception:
            pass

    # ---- single-GPU extras, all OUTSIDE the headline's timed region --------------------------------------------------
    if world == 1 and dist is not None:
        # the single-rank group has done its work (barriers, the max-reduce, the gather); release it before the rank-0-only extras so
        # that no communicator / watchdog is alive while the training measurement runs in processes of its own
        dist.barrier()
        dist.destroy_process_group()
        dist = None
    if world == 1 and not args.headline_only:
        configs = {'guided' if args.guided else 'simple': dict(head, roofline=out['roofline'])}
        out['segment_sum'] = segment_sum_line(sm, dev)
        other_kind = 'MolDiff_simple' if args.guided else 'MolDiff'
        del sm
        torch.cuda.empty_cache()
        # the other configuration: a run without kernel events for the step time, then a short run with events on every kernel
        # for the kernel durations (both with the guidance chain in line on one stream, the default)
        osteps, owarm = max(10, min(args.steps, 200)), min(args.warmup, 10)
        sm2, model2, ph2, ph_cpu2, gkw2 = sampler_for(other_kind, args.batch, 0)
        el2, prof2 = run_chain(sm2, osteps, owarm, barrier, prof=(0 if other_kind == 'MolDiff' else 2 << 0))  # step time without event overhead
        line2 = config_line(other_kind, sm2, osteps, owarm, el2, prof2, 1)
        guided_sm_kind = 'MolDiff'
        if other_kind == 'MolDiff':
            del sm2
            torch.cuda.empty_cache()
            sm3, *_ = sampler_for('MolDiff', args.batch, 0, overlap_guidance=False)
            el3, prof3 = run_chain(sm3, 20, 3, barrier)
            ra = roofline_mfma('edge_a', EDGE_A_NAME + ' (14 launches per guided step: 6 denoiser + 8 predictor blocks)', FLOP_EDGE_A,
                               2 * sm3.Eh, prof3)
            rb = roofline_mfma('edge_bwd', 'edge_bwd2_kernel (row-owner guidance backward in by-right edge order: dgrad chain over the forward tape, first-layer recompute '
                               'only, in-kernel sums of the by-right payloads, previous block\'s EdgeBlock-tail backward fused; v_mfma_f32_16x16x4_f32)', FLOP_EDGE_BWD, 2 * sm3.Eh, prof3)
            tot_a, tot_b = prof3['edge_a'][1], prof3['edge_bwd'][1]
            line2['roofline'] = dict(ra if tot_a >= tot_b else rb,
                                     note='kernel durations from a 20-step run with hipEvent brackets on every block kernel '
                                          '(%.2f ms/step with that overhead); ms_per_step is the run without them' % (el3 / 20 * 1e3))
            line2['roofline_other'] = rb if tot_a >= tot_b else ra
            line2['kernel_ms_per_step_inline'] = {k: v[1] / 20 for k, v in prof3.items() if v[0]}
            try:   # HBM traffic per launch from the guided step's own committed PMC passes (same rule as the headline's): the edge_a
                # slot is 6 launches of kernel A and 8 of its tape variant per step -> their launch-weighted mean
                import glob
                pmG = sorted(f for f in glob.glob(os.path.join(ROOT, 'profiles', 'r*_guided_pmc_summary.json')) if 'split' not in os.path.basename(f))
                ksG = json.load(open(pmG[-1]))['kernels'] if pmG and args.batch == 256 else {}
                if all(k in ksG for k in ('edge_a2_kernel<15>', 'edge_a2_kernel<63>', 'edge_bwd2_kernel<true>')):
                    srcG = os.path.relpath(pmG[-1], ROOT) + ' (PMC passes of the guided run; not measured in this run)'
                    ta = (6 * ksG['edge_a2_kernel<15>']['hbm_bytes_per_launch'] + 8 * ksG['edge_a2_kernel<63>']['hbm_bytes_per_launch']) / 14
                    tb = ksG['edge_bwd2_kernel<true>']['hbm_bytes_per_launch']   # <true>: with the fused tail of block i-1 (7 of 8 launches)
                    if 'edge_bwd2_kernel<false>' in ksG:
                        tb = (7 * tb + ksG['edge_bwd2_kernel<false>']['hbm_bytes_per_launch']) / 8
                    for r_ in (line2['roofline'], line2['roofline_other']):
                        r_['traffic'] = tb if 'edge_bwd2' in r_['kernel'] else ta
                        r_['traffic_source'] = srcG
            except Exception:
                pass
            del sm3
        else:
            line2['roofline'] = roofline_mfma('edge_a', EDGE_A_NAME, FLOP_EDGE_A, 2 * sm2.Eh, prof2)
            del sm2
        torch.cuda.empty_cache()
        configs['simple' if args.guided else 'guided'] = line2
        # the stated metric: wall time of one real model.sample() call with default arguments (prior draw + 1000 steps +
        # trajectory, synchronised at the end), scripts/sample_drug3d.py:117-125
        m_s, ph_s, _ = build_workload(args.batch, 0, None, 'MolDiff_simple')
        m_s = m_s.to(dev)
        ph_s = {k: v.to(dev) for k, v in ph_s.items()}
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res = m_s.sample(n_graphs=args.batch, batch_node=ph_s['batch_node'], halfedge_index=ph_s['halfedge_index'],
                         batch_halfedge=ph_s['batch_halfedge'])
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        traj_bytes = sum(int(t.ids.numel()) if hasattr(t, 'ids') else int(t.numel()) * 4 for t in res['traj'])
        out['sample_wall_s'] = {'value': wall, 'molecules_per_sec': args.batch / wall,
                                'what': 'one model.sample(n_graphs=%d, ...) call with default arguments on config #2: graph plan, '
                                        'prior draw, 1000 steps, compact trajectory (%.0f MB on the device; the reference layout '
                                        'is expanded lazily), synchronised at the end' % (args.batch, traj_bytes / 1e6)}
        del res, m_s
        torch.cuda.empty_cache()
        # the aggregation pass with an operand that cannot sit in the Infinity Cache: 2048 molecules, M = 1.2 GiB
        try:
            big = 2048
            mb, phb, _ = build_workload(big, 0, None, 'MolDiff_simple')
            mb = mb.to(dev)
            phb = {k: v.to(dev) for k, v in phb.items()}
            smb = mb.sampler(big, phb['batch_node'], phb['halfedge_index'], phb['batch_halfedge'], seed=5, return_traj=False)
            smb.init()
            elb, profb = run_chain(smb, 5, 2, barrier)
            szb = torch.bincount(phb['batch_node'], minlength=big).cpu().numpy()
            out['aggregation_large'] = dict(aggregation_line(smb.N, 2 * smb.Eh, profb, szb, True), molecules=big,
                                            ms_per_step=elb / 5 * 1e3, molecules_per_sec=big / (elb / 5 * T_STEPS))
            out['aggregation_large']['segment_sum'] = segment_sum_line(smb, dev)
            out['aggregation_large']['roofline_edge_a'] = roofline_mfma('edge_a', EDGE_A_NAME, FLOP_EDGE_A, 2 * smb.Eh, profb)['frac']
            del smb, mb
            torch.cuda.empty_cache()
        except Exception as e:  # measurement extra: never fail the headline line
            out['aggregation_large'] = {'error': repr(e)}
        # ---- the opt-in split float16 matrix path (csrc/mdx_split.h): the same two configurations, reported beside the exact ones;
        # the headline `value` above is and stays the exact fp32 path
        try:
            from moldiff_amd import _lib as _L
            for cname, kind in (('simple_split', 'MolDiff_simple'), ('guided_split', 'MolDiff')):
                with _L.default_matrix_path('split_f16'):
                    ssteps, swarm = max(10, min(args.steps, 200)), min(args.warmup, 10)
                    smS, *_ = sampler_for(kind, args.batch, 0)
                    elS, _ = run_chain(smS, ssteps, swarm, barrier, prof=0)            # step time without event overhead
                    lineS = config_line(kind, smS, ssteps, swarm, elS, {}, 1)
                    elP, profS = run_chain(smS, 20, 0, barrier, start=ssteps + swarm)  # kernel durations, events on every kernel
                    ES = 2 * smS.Eh
                    del smS
                lineS['dtype'] = SPLIT_DTYPE
                lineS['matrix_path'] = 'split_f16 (opt-in: module.matrix_path / MOLDIFF_MATRIX_PATH / mdx_model_set_matrix_path)'
                lineS['kernel_ms_per_step'] = {k: v[1] / 20 for k, v in profS.items() if v[0]}
                lineS['roofline'] = roofline_split('edge_a', 'edge_a2s_kernel (split float16 build of edge kernel A, mdx_edge2s.hip)', FLOP_EDGE_A, ES, profS)
                lineS['roofline_edge_b'] = roofline_split('edge_b', 'edge_b2s_kernel (split float16 build of edge kernel B)', FLOP_EDGE_B, ES, profS)
                try:   # HBM traffic per launch of the split build's kernel A from ITS committed PMC passes (same rule as the headline's)
                    import glob
                    pat = 'r*_split_guided_pmc_summary.json' if cname == 'guided_split' else 'r*_split_pmc_summary.json'
                    pmS = sorted(glob.glob(os.path.join(ROOT, 'profiles', pat)))
                    ksS = json.load(open(pmS[-1]))['kernels'] if pmS else {}
                    hit = [k for k in ksS if k.startswith('edge_a2s_kernel<15>')]
                    if hit and args.batch == 256:
                        lineS['roofline']['traffic'] = ksS[hit[0]]['hbm_bytes_per_launch']
                        lineS['roofline']['traffic_source'] = os.path.relpath(pmS[-1], ROOT) + ' (PMC passes of the split run; not measured in this run)'
                except Exception:
                    pass
                exact = configs['simple' if cname == 'simple_split' else 'guided']
                lineS['speedup_vs_exact_path'] = exact['ms_per_step'] / lineS['ms_per_step']
                lineS['parity'] = ('tests/test_gpu_round4.py: the exact path\'s golden / fp64-arbitrated parity tests re-run on this path '
                                   '(class ids bit-exact, positions and logits inside the same contract)')
                configs[cname] = lineS
                torch.cuda.empty_cache()
        except Exception as e:  # measurement extra: never fail the headline line
            configs['simple_split'] = {'error': repr(e)}
        # ---- config #3 with ONLY the guidance predictor on the split path: the denoiser (what decides atom / bond classes and the
        # posterior mean) stays on the exact fp32 MFMA; the predictor's forward + backward -- 61 % of the exact guided step -- produce an
        # increment scaled by 1e-4 (bond_predictor.matrix_path = 'split_f16' / MOLDIFF_GUIDANCE_MATRIX_PATH)
        try:
            msteps, mwarm = max(10, min(args.steps, 200)), min(args.warmup, 10)
            smM, *_ = sampler_for('MolDiff', args.batch, 0, bp_path='split_f16')
            elM, _ = run_chain(smM, msteps, mwarm, barrier, prof=0)
            lineM = config_line('MolDiff', smM, msteps, mwarm, elM, {}, 1)
            del smM
            torch.cuda.empty_cache()
            lineM['dtype'] = 'denoiser: f32 (fp32 MFMA); guidance predictor: ' + SPLIT_DTYPE
            lineM['matrix_path'] = "denoiser exact_f32, bond_predictor.matrix_path = 'split_f16' (opt-in)"
            lineM['speedup_vs_exact_path'] = configs['guided']['ms_per_step'] / lineM['ms_per_step']
            lineM['parity'] = 'tests/test_gpu_fullsize.py::test_one_full_size_guided_step_mixed_paths_matches_oracle, tests/test_gpu_round5.py (mixed)'
            lineM.pop('kernel_ms_per_step', None)
            configs['guided_mixed'] = lineM
        except Exception as e:
            configs['guided_mixed'] = {'error': repr(e)}
        # ---- BASELINE config #5's step (one optimisation step of train_MolDiff.yml at its own batch size) on the same line.  Measured
        # by `python bench.py --train` in a FRESH process each: the step is issued launch by launch from Python (host time ~ GPU time), and
        # at the end of this process -- process group, a dozen samplers and their streams behind it -- the same step takes the host 2-5 ms
        # longer than in a process of its own (same box: 33.5 / 36.5 ms here against 31.1 / 32.2 there, profiles/HISTORY.md round 5).
        try:
            import subprocess
            torch.cuda.empty_cache()
            tr_lines = {}
            for prec, st_, wu_ in (('fp16', 20, 6), ('f32', 8, 3)):
                cmd = [sys.executable, os.path.abspath(__file__), '--train', '--precision', prec, '--steps', str(st_), '--warmup', str(wu_),
                       '--batch', str(args.batch), '--cpu-budget', str(min(args.cpu_budget, 15.0))]
                if prec != 'fp16' or args.no_cpu_baseline:
                    cmd.append('--no-cpu-baseline')
                env = dict(os.environ)
                for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_PORT'):
                    env.pop(k, None)
                r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
                ln = [x for x in r.stdout.splitlines() if x.startswith('{')]
                if r.returncode != 0 or not ln:
                    raise RuntimeError('bench.py --train failed: ' + r.stderr[-500:])
                tl = json.loads(ln[-1])
                tl.pop('metric', None)
                tr_lines[prec] = tl
            configs['train'] = dict(tr_lines['fp16'], what="config #5: train_MolDiff.yml's step (use_amp: True = precision 'fp16'), one GPU, measured by "
                                                           "`python bench.py --train --precision fp16` in a process of its own; "
                                                           "the 8-GPU data-parallel half adds one 22 MB all-reduce per step",
                                    f32=tr_lines['f32'])
        except Exception as e:
            configs['train'] = {'error': repr(e)}
        out['configs'] = configs
        if not args.no_cpu_baseline:
            for name, kind in (('simple', 'MolDiff_simple'), ('guided', 'MolDiff')):
                cb = cpu_baseline(kind, args.batch, args.cpu_budget)
                configs[name]['cpu_baseline'] = cb
                configs[name]['speedup_vs_cpu_baseline'] = configs[name]['value'] / cb['value']
            hk = 'guided' if args.guided else 'simple'
            out['cpu_baseline'] = configs[hk]['cpu_baseline']
            out['speedup_vs_cpu_baseline'] = value / out['cpu_baseline']['value']
            if not args.guided:
                out['speedup_vs_reference_cpu'] = value / REFERENCE_CPU_MOL_S
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        emit_result(out)


if __name__ == '__main__':
    if '--cpu-worker' in sys.argv:
        cpu_worker(json.loads(sys.argv[sys.argv.index('--cpu-worker') + 1]))
    elif '--train' in sys.argv:
        sys.argv.remove('--train')
        main_train()
    else:
        main()
