"""Phase stamps of the split-precision edge kernel A on the bench workload (development tool, not product).

    tools/build_variant.sh trace2s -DMDX_TRACE2S mdx_edge2s.hip
    MOLDIFF_MATRIX_PATH=split_f16 python tools/trace_edge2s.py          # on the GPU box

Lane 0 of every wave stamps clock64() around the GEMMs of the NodeBlock message path (STAMPS in csrc/mdx_edge2s.hip).  Printed per
phase: mean cycles per work item (32 rows), and for GEMM phases the cycles per MFMA next to the ~17 the instruction needs.
"""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import moldiff_amd._lib as _lib  # noqa: E402

_lib.LIB_PATH = os.path.join(ROOT, 'moldiff_amd', 'libmoldiff_hip_trace2s.so')
import bench  # noqa: E402

RR = 2
# (name, first stamp, last stamp, MFMAs per wave: feature tiles x row tiles x k-groups x 3)
PH = [('emb GEMM 80->64 (+ smear, split)', 0, 1, 4 * RR * 3 * 3), ('gate init: gx[r] gather, bias', 1, 2, 0),
      ('gate GEMM1 64->256', 2, 3, 16 * RR * 2 * 3), ('LN256 + split + bias', 3, 4, 0), ('gate GEMM2 256->256', 4, 5, 16 * RR * 8 * 3),
      ('sigmoid + park + bias', 5, 6, 0), ('en GEMM1 64->256', 6, 7, 16 * RR * 2 * 3), ('LN256 + split + bias', 7, 8, 0),
      ('en GEMM2 256->256', 8, 9, 16 * RR * 8 * 3), ('H[r] gather, *, split, bias', 9, 10, 0), ('msg GEMM 256->256', 10, 11, 16 * RR * 8 * 3)]


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
    nunits = int(sum((int(n) * (int(n) - 1) + 15) // 16 for n in sizes))
    nitems = (nunits + RR - 1) // RR + 8
    buf = torch.zeros(nitems * 32, dtype=torch.int64, device=dev)
    assert L.mdx_debug_set_trace2s(ctypes.c_void_p(buf.data_ptr())) == 0
    sm.step(3)
    torch.cuda.synchronize()
    L.mdx_debug_set_trace2s(ctypes.c_void_p(0))
    tr = buf.cpu().numpy().reshape(nitems, 32).astype(np.int64)
    ok = (tr[:, 11] > tr[:, 0]) & (tr[:, 11] - tr[:, 0] < 10_000_000)
    for _, a, b, _ in PH:
        ok &= tr[:, b] >= tr[:, a]
    tr = tr[ok]
    print(f'{len(tr)} of {nitems} work items traced whole by one wave (last of 6 launches)')
    print(f'{"phase":40s} {"cycles":>9s} {"median":>9s} {"per MFMA":>9s}')
    tot = 0.0
    for name, a, b, nm in PH:
        d = tr[:, b] - tr[:, a]
        tot += d.mean()
        print(f'{name:40s} {d.mean():9.0f} {np.median(d):9.0f} {(d.mean() / nm if nm else 0):9.1f}')
    print(f'stamped part of the message path: {tot:.0f} cycles per item')


if __name__ == '__main__':
    main()
