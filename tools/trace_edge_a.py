"""Phase-level timing of edge_a_kernel on the bench workload (development tool, not part of the product).

    make -C moldiff_amd/csrc clean && make -C moldiff_amd/csrc -j8 EXTRA=-DMDX_TRACE OUT=../libmoldiff_hip_trace.so
    make -C moldiff_amd/csrc clean && make -C moldiff_amd/csrc -j8
    python tools/trace_edge_a.py            # on the GPU box

Thread 0 of every workgroup stamps clock64() at 22+8 phase boundaries (see MDX_STAMP in csrc/mdx_edge.hip).  The
script prints, per phase, the mean/median duration over all tiles next to the time the phase's MFMAs need on a
quarter of a CU's matrix pipes (what one workgroup gets when it shares the CU with a second one).
"""
import ctypes
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import moldiff_amd._lib as _lib  # noqa: E402

_lib.LIB_PATH = os.path.join(ROOT, 'moldiff_amd', 'libmoldiff_hip_trace.so')
import bench  # noqa: E402

PHASES = [  # (name, stamp_from, stamp_to, MFMA flops per edge in the phase)
    ('load He + smear', 0, 1, 0), ('emb GEMM 80->64 + store', 1, 2, 2 * 80 * 64),
    ('gate: gathers/bias', 2, 3, 0), ('gate GEMM1 64->256', 3, 4, 2 * 64 * 256), ('gate LN+lds+bar', 4, 5, 0),
    ('gate GEMM2 256->256', 5, 6, 2 * 256 * 256), ('gate sigmoid+bar', 6, 7, 0),
    ('en GEMM1 64->256', 7, 8, 2 * 64 * 256), ('en LN+lds+bar', 8, 9, 0), ('en GEMM2 256->256', 9, 10, 2 * 256 * 256),
    ('*h[r] + lds + 2 bar', 10, 11, 0), ('msg GEMM 256->256', 11, 12, 2 * 256 * 256), ('M store + bar', 12, 13, 0),
]
PHASES += [('ffn A: gathers + GEMM 64->320', 13, 14, 2 * 64 * 320), ('ffn gate LN + *nl + lds + bar', 14, 15, 0),
           ('ffn B: W1 GEMM 2x(128->128)', 15, 16, 2 * 2 * 128 * 128), ('ffn LN + lds + bar', 16, 17, 0),
           ('ffn C: W2 GEMM 2x(128->64)', 17, 18, 2 * 2 * 128 * 64), ('ffn D: gate GEMM 2x(32->64)', 18, 19, 2 * 2 * 32 * 64),
           ('ffn sigmoid + F store', 19, 20, 0)]
LAST = 20


def main():
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
    ntiles = (E + 47) // 48
    buf = torch.zeros(ntiles * 32, dtype=torch.int64, device=dev)
    assert L.mdx_debug_set_trace(ctypes.c_void_p(buf.data_ptr())) == 0
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record(); sm.step(3); t1.record()
    torch.cuda.synchronize()
    L.mdx_debug_set_trace(ctypes.c_void_p(0))
    tr = buf.cpu().numpy().reshape(ntiles, 32).astype(np.int64)
    np.save(os.path.join(ROOT, 'gpurun_out', 'trace_edge_a.npy'), tr)
    clk, wall, hw = tr[:, :30], tr[:, 30], tr[:, 31]
    xcc = (hw >> 32) & 0xf
    # clock64 counters are not synchronised across the chip: only differences within one workgroup are used, scaled by
    # the 100 MHz wall clock stamped at the first and last boundary of the same workgroup
    wall_end = tr[:, 29]
    scale_all = (clk[:, LAST] - clk[:, 0]).sum() / ((wall_end - wall).sum() / 100.0)
    scale = np.full(16, scale_all)
    t = clk / scale[xcc][:, None]
    start = (wall - wall.min()) / 100.0
    dur = t[:, LAST] - t[:, 0]
    end = start + dur
    print(f'step {t0.elapsed_time(t1):.3f} ms; tiles {ntiles}; clock64 per us by XCD {np.round(scale[:8])}')
    print(f'kernel span {end.max():.1f} us; tile duration mean {dur.mean():.1f} median {np.median(dur):.1f} min {dur.min():.1f} '
          f'max {dur.max():.1f} us; mean resident workgroups {dur.sum() / end.max():.0f} of 512')
    pipe = 4 * 64 * 2.4e3   # MFMA flop per us of one CU (4 SIMDs x 64 flop/clk at 2.4 GHz)
    rows, tg, tn = [], 0.0, 0.0
    print(f'{"phase":40s} {"mean us":>8s} {"median":>8s} {"MFMA-only, CU alone":>20s} {"CU shared by 2":>15s}')
    for name, a, b, fl in PHASES:
        d = t[:, b] - t[:, a]
        ideal = fl * 48 / pipe
        rows.append((name, float(d.mean()), float(np.median(d)), ideal))
        tg, tn = (tg + d.mean(), tn) if fl else (tg, tn + d.mean())
        print(f'{name:40s} {d.mean():8.2f} {np.median(d):8.2f} {ideal:20.2f} {2 * ideal:15.2f}')
    tot_ideal = sum(r[3] for r in rows)
    print(f'GEMM phases {tg:.1f} us + other phases {tn:.1f} us = {tg + tn:.1f} us per tile; MFMA-only {tot_ideal:.1f} us '
          f'=> matrix pipe busy {2 * tot_ideal / (tg + tn):.2f} while two workgroups share a CU')
    mhz = 0
    json.dump({'rows': rows, 'mhz': mhz}, open(os.path.join(ROOT, 'gpurun_out', 'trace_edge_a.json'), 'w'))


if __name__ == '__main__':
    os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
    main()
