"""Where do the waves of the fused kernels spend their cycles?  One rocprofv3 --pmc pass (8 SQ counters) over a short bench run.
    python tools/pmc_stalls.py out.json [bench args]
SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles per wave; SQ_VALU_MFMA_* count cycles (MI355X_MICROARCH.md)."""
import csv, glob, json, os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBV = os.environ.get('MDX_LIB_VARIANT')   # e.g. moldiff_amd/libmoldiff_hip_ring4.so -> run through tools/bench_with_lib.py
CTRS = ['SQ_WAVE_CYCLES', 'SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_VALU', 'SQ_ACTIVE_INST_VMEM', 'SQ_ACTIVE_INST_LDS',
        'SQ_VALU_MFMA_COEXEC_CYCLES', 'SQ_VALU_MFMA_BUSY_CYCLES']
KERNELS = ['edge_a2_kernel', 'edge_b2_kernel', 'edge_bwd2_kernel', 'edge_bwd_kernel', 'node_kernel(']
work = tempfile.mkdtemp(dir='/tmp')
cmd = ['rocprofv3', '--pmc'] + CTRS + ['--output-format', 'csv', '-d', work, '--', sys.executable] + ([os.path.join(ROOT, 'tools', 'bench_with_lib.py'), os.path.join(ROOT, LIBV)] if LIBV else [os.path.join(ROOT, 'bench.py')]) + [
                                       '--steps', '6', '--warmup', '2', '--headline-only'] + sys.argv[2:]
subprocess.run(cmd, cwd='/tmp', env=dict(os.environ, TMPDIR='/tmp'), check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
acc = {}
for f in glob.glob(os.path.join(work, '**', '*counter_collection.csv'), recursive=True):
    for row in csv.DictReader(open(f)):
        key = next((k for k in KERNELS if k in row['Kernel_Name']), None)
        if key:
            a = acc.setdefault(key.rstrip('('), {}).setdefault(row['Counter_Name'], [0.0, set()])
            a[0] += float(row['Counter_Value']); a[1].add(row['Dispatch_Id'])
res = {}
for k, c in acc.items():
    r = {n: v[0] / len(v[1]) for n, v in c.items()}
    wc = r['SQ_WAVE_CYCLES']
    res[k] = dict(r, frac_of_wave_cycles={n: round(v / wc, 4) for n, v in r.items() if n.startswith('SQ_W') or n.startswith('SQ_ACTIVE')},
                  mfma_busy_per_wave_cycle=round(r['SQ_VALU_MFMA_BUSY_CYCLES'] / (4 * wc), 4),
                  coexec_over_mfma_busy=round(r['SQ_VALU_MFMA_COEXEC_CYCLES'] / max(r['SQ_VALU_MFMA_BUSY_CYCLES'], 1), 4))
json.dump(res, open(sys.argv[1], 'w'), indent=1)
print(json.dumps({k: {kk: v[kk] for kk in ('frac_of_wave_cycles', 'mfma_busy_per_wave_cycle', 'coexec_over_mfma_busy')} for k, v in res.items()}, indent=1))
