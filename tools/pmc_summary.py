"""Collect the PMC evidence behind bench.py's `roofline.traffic` (run on the GPU box; development/measurement tool).

    python tools/pmc_summary.py profiles/r1_c_pmc_summary.json

Three separate rocprofv3 --pmc passes over `bench.py --steps 6 --warmup 2 --headline-only [extra args, e.g. --guided]` (never combined with a trace
domain), exactly as MI355X_MICROARCH.md prescribes: (1) SQ / GRBM counters, (2) FETCH_SIZE, (3) WRITE_SIZE.  Per-kernel,
per-launch averages are written as JSON.  FETCH_SIZE / WRITE_SIZE are in KB; on gfx950 FETCH_SIZE under-reports wide
(16 B/lane) coalesced reads by 2x, so hbm_bytes = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024 (checked on the pure streaming
kernel seg_reduce<256>: 165.0 MB measured vs 164.8 MB algorithmic).
"""
import csv
import glob
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PASSES = [['SQ_WAVE_CYCLES', 'SQ_VALU_MFMA_BUSY_CYCLES', 'SQ_ACTIVE_INST_ANY', 'SQ_WAIT_INST_ANY', 'GRBM_GUI_ACTIVE'], ['FETCH_SIZE'], ['WRITE_SIZE']]
import re


def kernel_key(name):
    """'void (anonymous namespace)::edge_a2_kernel<15>(EdgeAArgs, ...)' -> 'edge_a2_kernel<15>' for the library's own block kernels
    (the keys bench.py looks up through mdx_profile_kernel_name); None for everything else."""
    m = re.search(r'((?:edge_a2s|edge_b2s|edge_bwd2s|edge_tail_bwd2s|node_bwd_s|node_s|edge_a2|edge_b2|edge_a|edge_b|node|node_bwd|seg_reduce_block2|seg_reduce_block|edge_bwd2|edge_bwd|edge_tail_bwd2|'
                  r'seg_reduce_bwd_block|seg_reduce)_kernel(?:<[0-9a-z, ]+>)?)\(', name)
    return m.group(1) if m else None

def main():
    out_path = sys.argv[1]
    work = tempfile.mkdtemp(dir='/tmp')
    env = dict(os.environ, TMPDIR='/tmp')
    acc = {}
    for i, ctrs in enumerate(PASSES):
        d = os.path.join(work, f'pass{i}')
        cmd = ['rocprofv3', '--pmc'] + ctrs + ['--output-format', 'csv', '-d', d, '--', sys.executable, os.path.join(ROOT, 'bench.py'),
                                               '--steps', '6', '--warmup', '2', '--headline-only'] + sys.argv[2:]
        subprocess.run(cmd, cwd='/tmp', env=env, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
            for row in csv.DictReader(open(f)):
                name = row['Kernel_Name']
                key = kernel_key(name)
                if key is None:
                    continue
                a = acc.setdefault(key, {}).setdefault(row['Counter_Name'], [0.0, set()])
                a[0] += float(row['Counter_Value'])
                a[1].add(row['Dispatch_Id'])
    res = {}
    for k, ctr in acc.items():
        r = {c: v[0] / max(len(v[1]), 1) for c, v in ctr.items()}
        r['launches_sampled'] = max(len(v[1]) for v in ctr.values())
        if 'FETCH_SIZE' in r and 'WRITE_SIZE' in r:
            r['hbm_bytes_per_launch'] = 2 * r['FETCH_SIZE'] * 1024 + r['WRITE_SIZE'] * 1024
            r['hbm_bytes_per_launch_uncorrected'] = (r['FETCH_SIZE'] + r['WRITE_SIZE']) * 1024
        if 'SQ_VALU_MFMA_BUSY_CYCLES' in r and r.get('GRBM_GUI_ACTIVE'):
            r['mfma_util'] = r['SQ_VALU_MFMA_BUSY_CYCLES'] / (r['GRBM_GUI_ACTIVE'] / 8 * 1024)
        res[k] = r
    # bench.py quotes a kernel's traffic only when ITS name (mdx_profile_kernel_name) is a key here: a summary taken with an older
    # kernel set is refused instead of silently reused
    json.dump({'note': __doc__.strip().split('\n\n', 1)[1] if '\n\n' in __doc__ else '', 'kernel_names': sorted(res), 'kernels': res},
              open(out_path, 'w'), indent=1)
    print(json.dumps({k: {c: v for c, v in r.items() if c in ('hbm_bytes_per_launch', 'mfma_util', 'launches_sampled')} for k, r in res.items()}))


if __name__ == '__main__':
    main()
