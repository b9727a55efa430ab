"""Phase-level timing of the row-owner edge kernels on the bench workload (development tool, not product).

    tools/build_variant.sh trace2 -DMDX_TRACE2
    python tools/trace_edge2.py a          # on the GPU box

Lane 0 of every wave stamps clock64() at the phase boundaries (STAMP in csrc/mdx_edge2.hip) and the 100 MHz wall clock at
entry and exit.  Printed per phase: mean duration over all 32-edge units next to the time the phase's MFMAs alone need on
the wave's SIMD (64 flop/clk at the measured shader clock).
"""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import moldiff_amd._lib as _lib  # noqa: E402

_lib.LIB_PATH = os.path.join(ROOT, 'moldiff_amd', 'libmoldiff_hip_trace2.so')
import bench  # noqa: E402

ROWS = 16
PH_A = [('tile load + He + smear', 0, 1, 0), ('emb GEMM 80->64', 1, 2, 2 * 80 * 64), ('He store + gate init gathers', 2, 3, 0),
        ('gate GEMM1 64->256', 3, 4, 2 * 64 * 256), ('LN + bias', 4, 5, 0), ('gate GEMM2 256->256', 5, 6, 2 * 256 * 256),
        ('sigmoid + park + bias', 6, 7, 0), ('en GEMM1 64->256', 7, 8, 2 * 64 * 256), ('LN + bias', 8, 9, 0),
        ('en GEMM2 256->256', 9, 10, 2 * 256 * 256), ('H[r] gather * + bias', 10, 11, 0),
        ('msg GEMM 256->256', 11, 12, 2 * 256 * 256), ('unpark * + M store', 12, 13, 0)]
PH_B = [('gathers SL,SR,nfl,nfr,Lf,Rf + sum', 0, 1, 0), ('self_ffn GEMM 64->64', 1, 2, 2 * 64 * 64), ('LN64 + bias', 2, 3, 0),
        ('out GEMM 64->64', 3, 4, 2 * 64 * 64), ('He store + next tile + a=Lf*Rf', 4, 5, 0), ('Wbl GEMM 64->256', 5, 6, 2 * 64 * 256),
        ('Wnl GEMM 64->256 pairwise *', 6, 7, 2 * 64 * 256), ('gate GEMMs 2x(64->32)', 7, 8, 2 * 2 * 64 * 32),
        ('gate LN32 + dot + bias', 8, 9, 0), ('inter GEMM 256->256', 9, 10, 2 * 256 * 256), ('LN256 + dot + force + store', 10, 40, 0)]
for s in (0, 1):
    o = 10 * s
    PH_A += [(f'ffn{s}: gathers', 13 + 9 * s if s == 0 else 22, 14 + o, 0), (f'ffn{s}: bl GEMM 64->128', 14 + o, 15 + o, 2 * 64 * 128),
             (f'ffn{s}: *nl + gate1 GEMM 64->32', 15 + o, 16 + o, 2 * 64 * 32), (f'ffn{s}: LN32 + bias', 16 + o, 17 + o, 0),
             (f'ffn{s}: W1 GEMM 128->128', 17 + o, 18 + o, 2 * 128 * 128), (f'ffn{s}: LN128 + bias', 18 + o, 19 + o, 0),
             (f'ffn{s}: W2 GEMM 128->64', 19 + o, 20 + o, 2 * 128 * 64), (f'ffn{s}: gate2 GEMM 32->64', 20 + o, 21 + o, 2 * 32 * 64),
             (f'ffn{s}: sigmoid + store', 21 + o, 22 + o, 0)]


PH_W = [('tile + SG, M, dL/daggr[l] loads', 0, 1, 0), ('gate W2^T GEMM 256->256', 1, 2, 2 * 256 * 256),
        ('gx[r], He gathers + bias', 2, 3, 0), ('gate W1 recompute 64->256', 3, 4, 2 * 64 * 256),
        ('LN backward 256 + GGX segment sums', 4, 5, 0), ('gate W1^T GEMM 256->64', 5, 6, 2 * 256 * 64),
        ('SG, dL/daggr[l] loads', 6, 7, 0), ('msg^T GEMM 256->256', 7, 8, 2 * 256 * 256),
        ('HE load, H[r] gather, GH to LDS', 8, 9, 0), ('en W2^T GEMM 256->256', 9, 10, 2 * 256 * 256),
        ('GH segment sums + en W1 recompute 64->256', 10, 11, 2 * 64 * 256), ('LN backward 256', 11, 12, 0),
        ('en W1^T GEMM 256->64', 12, 13, 2 * 256 * 64),
        ('ffn0 tape loads + gate recompute', 13, 14, 2 * (64 * 32 + 32 * 64)),
        ('ffn0 backward inter', 14, 15, 2 * (64 * 128 + 128 * 128 + 128 * 64)), ('ffn0 backward gate', 15, 21, 2 * (64 * 32 + 32 * 64)),
        ('ffn1 tape loads + gate recompute', 21, 17, 2 * (64 * 32 + 32 * 64)),
        ('ffn1 backward inter + GNL1 segment sums', 17, 18, 2 * (64 * 128 + 128 * 128 + 128 * 64)),
        ('ffn1 backward gate', 18, 20, 2 * (64 * 32 + 32 * 64)), ('emb backward + gdist', 20, 40, 2 * (64 * 64 + 64 * 32))]


def main_bwd():
    dev = torch.device('cuda:0')
    model, ph, sizes = bench.build_workload(256, 0, dev, 'MolDiff')
    model = model.to(dev)
    L = _lib.lib()
    sm = model.sampler(256, ph['batch_node'], ph['halfedge_index'], ph['batch_halfedge'], seed=2023, return_traj=False,
                       bond_predictor=bench.build_bond_predictor().to(dev), guidance=['uncertainty', 1e-4], overlap_guidance=False)
    sm.init()
    for i in range(3):
        sm.step(i)
    torch.cuda.synchronize()
    E = 2 * ph['halfedge_index'].shape[1]
    rows = int(os.environ.get('MDX_BWD_ROWS', 16))
    nunits = (E + rows - 1) // rows + 256      # graph-aligned units of the by-right order: at most one more per molecule
    buf = torch.zeros(nunits * 48, dtype=torch.int64, device=dev)
    assert L.mdx_debug_set_trace_bwd(ctypes.c_void_p(buf.data_ptr())) == 0
    sm.step(3)
    torch.cuda.synchronize()
    L.mdx_debug_set_trace_bwd(ctypes.c_void_p(0))
    report('bwd', buf, nunits, PH_W, 40, rows)


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else 'a'
    if which == 'w':
        return main_bwd()
    phases, last = (PH_A, 40) if which == 'a' else (PH_B, 40)
    dev = torch.device('cuda:0')
    model, ph, sizes = bench.build_workload(256, 0, dev)
    model = model.to(dev)
    L = _lib.lib()
    sm = model.sampler(256, ph['batch_node'], ph['halfedge_index'], ph['batch_halfedge'], seed=2023, return_traj=False)
    sm.init()
    for i in range(3):
        sm.step(i)
    torch.cuda.synchronize()
    E = 2 * ph['halfedge_index'].shape[1]
    rows = int(os.environ.get('MDX_ROWS', 16))   # rows per wave the kernels are built with (MDX_RR = 1)
    nunits = (E + rows - 1) // rows
    if which == 'a' and os.environ.get('MDX_NO_AGG') != '1':   # EA_AGG: units are aligned to each molecule's first edge
        nunits = int(sum((int(n) * (int(n) - 1) + 15) // 16 for n in sizes))
    buf = torch.zeros(nunits * 48, dtype=torch.int64, device=dev)
    setter = (lambda p: L.mdx_debug_set_trace2(p, 0)) if which == 'a' else L.mdx_debug_set_trace2b
    assert setter(ctypes.c_void_p(buf.data_ptr())) == 0
    sm.step(3)
    torch.cuda.synchronize()
    setter(ctypes.c_void_p(0))
    report(which, buf, nunits, phases, last, rows)


def report(which, buf, nunits, phases, last, ROWS=ROWS):
    tr = buf.cpu().numpy().reshape(nunits, 48).astype(np.int64)
    os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
    np.save(os.path.join(ROOT, 'gpurun_out', f'trace_edge2_{which}.npy'), tr)
    # units cut by section (kernel A's tail) are stamped by two waves on CUs whose shader clocks are not synchronised: keep the
    # units whose stamps are monotonic, i.e. come from one wave
    span = tr[:, last] - tr[:, 0]
    ok = (span > 0) & (span < 10_000_000) & (tr[:, 47] > tr[:, 46])
    for _, a, b, _ in phases:
        ok &= tr[:, b] >= tr[:, a]
    print(f'{int(ok.sum())} of {nunits} units traced by a single wave (the others are cut by section between two waves)')
    tr = tr[ok]
    clk, wall0, wall1 = tr[:, :46], tr[:, 46], tr[:, 47]
    scale = (clk[:, last] - clk[:, 0]).sum() / ((wall1 - wall0).sum() / 100.0)   # shader clocks per us
    dur = (clk[:, last] - clk[:, 0]) / scale
    start = (wall0 - wall0.min()) / 100.0
    end = start + dur
    print(f'edge_{which}2: units {nunits}; shader clock {scale:.0f} MHz; kernel span {end.max():.1f} us (last of 6 launches); '
          f'unit duration mean {dur.mean():.1f} median {np.median(dur):.1f} min {dur.min():.1f} max {dur.max():.1f} us; '
          f'mean resident waves {dur.sum() / end.max():.0f} of 1024')
    tg = tn = ti = 0.0
    print(f'{"phase":40s} {"mean us":>8s} {"median":>8s} {"MFMA-only":>10s} {"ratio":>6s}')
    for name, a, b, fl in phases:
        d = (clk[:, b] - clk[:, a]) / scale
        ideal = fl * ROWS / 64.0 / scale
        ti += ideal
        tg, tn = (tg + d.mean(), tn) if fl else (tg, tn + d.mean())
        print(f'{name:40s} {d.mean():8.2f} {np.median(d):8.2f} {ideal:10.2f} {(ideal / d.mean() if fl else 0):6.2f}')
    print(f'GEMM phases {tg:.1f} us + other phases {tn:.1f} us = {tg + tn:.1f} us per unit; MFMA-only {ti:.1f} us => pipe busy {ti / (tg + tn):.3f}')


if __name__ == '__main__':
    main()
