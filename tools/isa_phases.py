"""Per-phase instruction mix of a row-owner kernel: compile mdx_edge2.hip with -DMDX_TRACE2 -S and split at s_memtime.
   python tools/isa_phases.py [a|b]"""
import re, subprocess, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
which = sys.argv[1] if len(sys.argv) > 1 else 'a'
extra = sys.argv[2:]
src = os.path.join(ROOT, 'moldiff_amd', 'csrc', 'mdx_edge2.hip')
subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-DMDX_TRACE2', '-S',
                '--cuda-device-only', src, '-o', '/tmp/edge2t.s'] + extra, check=True, stderr=subprocess.DEVNULL)
txt = open('/tmp/edge2t.s').read().split('\n')
starts = [i for i, l in enumerate(txt) if re.match(r'^_ZN.*edge_[ab]2_kernel.*:', l)]
lo, hi = (starts[0], starts[1]) if which == 'a' else (starts[1], len(txt))
seg, cur = [], []
for l in txt[lo:hi]:
    t = l.strip()
    if not t or t.startswith(';') or t.startswith('.'):
        continue
    if t.startswith('s_memtime') or t.startswith('s_memrealtime'):
        seg.append(cur); cur = []; continue
    cur.append(t.split()[0])
seg.append(cur)
tot = dict(mfma=0, valu=0, acc=0, vmem=0, scr=0)
for i, s in enumerate(seg):
    mf = sum(x.startswith('v_mfma') for x in s)
    va = sum(x.startswith('v_') and not x.startswith('v_mfma') for x in s)
    acc = sum('accvgpr' in x for x in s)
    vm = sum(x.startswith('global_') for x in s)
    sc = sum(x.startswith('scratch_') for x in s)
    ds = sum(x.startswith('ds_') for x in s)
    wt = sum(x.startswith('s_waitcnt') for x in s)
    nop = sum(x.startswith('s_nop') for x in s)
    tot['mfma'] += mf; tot['valu'] += va; tot['acc'] += acc; tot['vmem'] += vm; tot['scr'] += sc
    if len(s) > 30:
        print(f'{i:3d} n={len(s):5d} mfma={mf:5d} valu={va:5d} (acc {acc:4d}) vmem={vm:4d} scratch={sc:4d} ds={ds:4d} wait={wt:4d} nop={nop:4d}')
print(tot)
