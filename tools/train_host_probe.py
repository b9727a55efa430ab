"""Why is the fp16 training step slower inside the default `python bench.py` run than in `python bench.py --train`?  (development tool)
Measures train_measure (20 steps, 6 warm-up) fresh, after a sampling run in the same process, and after gc.freeze()."""
import gc
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def tm(tag):
    tl, m, _ = bench.train_measure('MolDiff', 'fp16', 256, 20, 6, torch.device('cuda:0'))
    print(tag, 'ms/step %.2f host %.2f' % (tl['ms_per_step'], tl['host_issue_ms_per_step']), 'gc counts', gc.get_count(),
          'tracked objects', len(gc.get_objects()), flush=True)
    del m
    torch.cuda.empty_cache()


def main():
    dev = torch.device('cuda:0')
    tm('fresh          ')
    tm('again          ')
    model, ph, sizes = bench.build_workload(256, 0, dev)
    sm = model.to(dev).sampler(256, ph['batch_node'], ph['halfedge_index'], ph['batch_halfedge'], seed=1, return_traj=False)
    sm.init()
    bench.run_chain(sm, 200, 5, torch.cuda.synchronize, prof=0)
    tm('after sampling ')
    del sm, model
    torch.cuda.empty_cache()
    tm('sampler freed  ')
    gc.collect()
    gc.freeze()
    tm('gc.freeze      ')
    gc.disable()
    tm('gc.disable     ')
    gc.enable()
    torch.set_num_threads(32)
    x = torch.randn(2048, 2048)
    for _ in range(5):
        x @ x                                   # spin up the intra-op pool (what the in-process CPU baselines leave behind)
    tm('after cpu GEMMs')
    torch.set_num_threads(1)
    tm('threads = 1    ')


if __name__ == '__main__':
    main()
