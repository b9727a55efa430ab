import sys, time, torch
sys.path.insert(0, '/root/repo')
import bench
from moldiff_amd import _lib
dev = torch.device('cuda:0')
model, ph, sizes = bench.build_workload(256, 0, dev)
model = model.to(dev)
sm = model.sampler(256, ph['batch_node'], ph['halfedge_index'], ph['batch_halfedge'], seed=2023, return_traj=False)
sm.init()
L = _lib.lib()
for prof in (0, 1, 0, 1):
    for i in range(10): sm.step(i)
    torch.cuda.synchronize()
    L.mdx_profile_enable(prof)
    t0 = time.perf_counter()
    for i in range(200): sm.step(10 + i)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    L.mdx_profile_enable(0)
    print('profile', prof, 'ms/step', el / 200 * 1e3)
