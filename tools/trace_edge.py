"""Phase-level timing of edge_a_kernel / edge_b_kernel on the bench workload (development tool, not product).

    make -C moldiff_amd/csrc clean && make -C moldiff_amd/csrc -j8 EXTRA=-DMDX_TRACE OUT=../libmoldiff_hip_trace.so
    make -C moldiff_amd/csrc clean && make -C moldiff_amd/csrc -j8
    python tools/trace_edge.py a|b          # on the GPU box

Thread 0 of every workgroup stamps clock64() at the phase boundaries (MDX_STAMP / MDX_STAMPB in csrc/mdx_edge.hip) and
the 100 MHz wall clock at its first and last boundary.  Printed per phase: mean / median duration over all tiles next
to the time the phase's MFMAs alone need on one CU (4 SIMDs x 64 flop/clk x 2.4 GHz) and on half of one.
"""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import moldiff_amd._lib as _lib  # noqa: E402

_lib.LIB_PATH = os.path.join(ROOT, 'moldiff_amd', 'libmoldiff_hip_trace.so')
import bench  # noqa: E402

PHASES_A = [  # (name, stamp_from, stamp_to, MFMA flops per edge in the phase)
    ('load He + smear', 0, 1, 0), ('emb GEMM 80->64 + store', 1, 2, 2 * 80 * 64),
    ('gate: gathers/bias', 2, 3, 0), ('gate GEMM1 64->256', 3, 4, 2 * 64 * 256), ('gate LN+lds+bar', 4, 5, 0),
    ('gate GEMM2 256->256', 5, 6, 2 * 256 * 256), ('gate sigmoid', 6, 7, 0),
    ('en GEMM1 64->256', 7, 8, 2 * 64 * 256), ('en LN+lds+bar', 8, 9, 0), ('en GEMM2 256->256', 9, 10, 2 * 256 * 256),
    ('*h[r] + lds + 2 bar', 10, 11, 0), ('msg GEMM 256->256', 11, 12, 2 * 256 * 256), ('M store', 12, 13, 0),
    ('ffn A: gathers + GEMM 64->320', 13, 14, 2 * 64 * 320), ('ffn gate LN + *nl + lds + bar', 14, 15, 0),
    ('ffn B: W1 GEMM 2x(128->128)', 15, 16, 2 * 2 * 128 * 128), ('ffn LN + lds + bar', 16, 17, 0),
    ('ffn C: W2 GEMM 2x(128->64)', 17, 18, 2 * 2 * 128 * 64), ('ffn D: gate GEMM 2x(32->64)', 18, 19, 2 * 2 * 32 * 64),
    ('ffn sigmoid + F store', 19, 20, 0)]
PHASES_B = [
    ('load He\'', 0, 1, 0), ('gathers SL,SR,nfl,nfr', 1, 2, 0), ('self_ffn GEMM 64->64', 2, 3, 2 * 64 * 64),
    ('LN + lds + bar', 3, 4, 0), ('out GEMM 64->64', 4, 5, 2 * 64 * 64), ('He\'\' store + 2 bar', 5, 6, 0),
    ('a = Lf*Rf gather + bar', 6, 7, 0), ('gate GEMMs 2x(64->32), 2 waves', 7, 8, 2 * 2 * 64 * 32), ('gate LN + dot', 8, 9, 0),
    ('Wbl, Wnl GEMMs 2x(64->256) + product', 9, 10, 2 * 2 * 64 * 256), ('2 bar + lds', 10, 11, 0),
    ('inter GEMM 256->256', 11, 12, 2 * 256 * 256), ('LN + dot', 12, 13, 0), ('force + store', 13, 14, 0)]


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else 'a'
    phases, last = (PHASES_A, 20) if which == 'a' else (PHASES_B, 14)
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
    assert L.mdx_debug_set_trace(ctypes.c_void_p(buf.data_ptr()), 0 if which == 'a' else 1) == 0
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record(); sm.step(3); t1.record()
    torch.cuda.synchronize()
    L.mdx_debug_set_trace(ctypes.c_void_p(0), 0)
    tr = buf.cpu().numpy().reshape(ntiles, 32).astype(np.int64)
    os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
    np.save(os.path.join(ROOT, 'gpurun_out', f'trace_edge_{which}.npy'), tr)
    clk, wall0, wall1 = tr[:, :29], tr[:, 30], tr[:, 29]
    # clock64 counters are not synchronised across the chip: only differences within one workgroup are used, scaled by
    # the wall clock stamped at the first and last boundary of the same workgroups
    scale = (clk[:, last] - clk[:, 0]).sum() / ((wall1 - wall0).sum() / 100.0)
    t = clk / scale
    start = (wall0 - wall0.min()) / 100.0
    dur = t[:, last] - t[:, 0]
    end = start + dur
    print(f'edge_{which}: step {t0.elapsed_time(t1):.3f} ms; tiles {ntiles}; clock64 = {scale:.0f} per us')
    print(f'kernel span {end.max():.1f} us (last of 6 launches); tile duration mean {dur.mean():.1f} median {np.median(dur):.1f} '
          f'min {dur.min():.1f} max {dur.max():.1f} us; mean resident workgroups {dur.sum() / end.max():.0f} of 512')
    pipe = 4 * 64 * 2.4e3
    tg = tn = tot_ideal = 0.0
    print(f'{"phase":40s} {"mean us":>8s} {"median":>8s} {"MFMA-only, CU alone":>20s} {"CU shared by 2":>15s}')
    for name, a, b, fl in phases:
        d = t[:, b] - t[:, a]
        ideal = fl * 48 / pipe
        tot_ideal += ideal
        tg, tn = (tg + d.mean(), tn) if fl else (tg, tn + d.mean())
        print(f'{name:40s} {d.mean():8.2f} {np.median(d):8.2f} {ideal:20.2f} {2 * ideal:15.2f}')
    print(f'GEMM phases {tg:.1f} us + other phases {tn:.1f} us = {tg + tn:.1f} us per tile; MFMA-only {tot_ideal:.1f} us '
          f'=> matrix pipe busy {2 * tot_ideal / (tg + tn):.2f} while two workgroups share a CU')


if __name__ == '__main__':
    main()
