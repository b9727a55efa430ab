import os, sys, collections
os.environ['MDX_TRAIN_FAST'] = '0'
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch
from moldiff_amd import train_ops as T
orig = T._flush_wgrads
seen = []
def wrap():
    sk = T._SINK
    if sk is not None and sk['wq'] and not seen:
        c = collections.Counter()
        for g, x, plan, dims, dst_w, ldw, dst_b, rk in sk['wq']:
            c[(plan[0], dims[0], dims[1], dims[2], dims[3], dst_b is not None)] += 1
        for k in sorted(c):
            print('JOB kind %d M %d N %d K %d dt %d bias %s x%d' % (*k, c[k]))
        seen.append(1)
    return orig()
T._flush_wgrads = wrap
sys.argv = ['bench.py', '--train', '--precision', 'fp16', '--no-cpu-baseline', '--steps', '1', '--warmup', '1']
import runpy
runpy.run_path(os.path.join(os.environ.get('GRAFT_REPO_ROOT', '/root/repo'), 'bench.py'), run_name='__main__')
