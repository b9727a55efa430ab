# bit-compare two library builds on one forward + one guided step:  python cmp_libs.py libA libB
import os, sys, subprocess, torch
if len(sys.argv) == 4:
    sys.path.insert(0, '/root/repo' if os.path.exists('/root/repo/bench.py') else os.getcwd())
    import moldiff_amd._lib as _lib
    _lib.LIB_PATH = os.path.abspath(sys.argv[1])
    import bench
    dev = torch.device('cuda:0')
    model, ph, sizes = bench.build_workload(64, 0, dev, 'MolDiff')
    model = model.to(dev)
    sm = model.sampler(64, ph['batch_node'], ph['halfedge_index'], ph['batch_halfedge'], seed=3, return_traj=False,
                       bond_predictor=bench.build_bond_predictor().to(dev), guidance=['uncertainty', 1e-4])
    sm.init()
    for i in range(3):
        sm.step(i)
    st = sm.state()
    torch.save({k: v.cpu() for k, v in st.items()}, sys.argv[2])
else:
    a, b = sys.argv[1], sys.argv[2]
    for lib, out in ((a, '/tmp/cmp_a.pt'), (b, '/tmp/cmp_b.pt')):
        subprocess.run([sys.executable, __file__, lib, out, 'child'], check=True)
    A, B = torch.load('/tmp/cmp_a.pt'), torch.load('/tmp/cmp_b.pt')
    for k in A:
        print(k, 'equal' if torch.equal(A[k], B[k]) else 'DIFFERENT max %.3e' % float((A[k] - B[k]).abs().max()))
