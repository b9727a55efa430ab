import sys, numpy as np, torch
sys.path.insert(0,'.')
import bench, moldiff_amd as M
from moldiff_amd.harness import default_config, GEOM_DRUGS_MEAN_ATOMS, GEOM_DRUGS_STD_ATOMS
from moldiff_amd import train_ops
dev='cuda:0'
np.random.seed(2920)
sizes=np.maximum(np.random.normal(GEOM_DRUGS_MEAN_ATOMS, GEOM_DRUGS_STD_ATOMS, size=256).astype('int64'),2)
model=M.MolDiff(default_config('MolDiff'),8,6); model.load_state_dict(M.recipe_state_dict(model,20230807),strict=True); model=model.to(dev).train()
batch=bench.clean_batch([int(s) for s in sizes],100,dev)
for seed in range(4):
    out={}
    for mode in ('f32','fp16','bf16_autocast'):
        torch.manual_seed(seed)
        with torch.no_grad():
            pass
        with train_ops.precision(mode):
            l=model.get_loss(*batch)
        out[mode]={k:round(float(v),4) for k,v in l.items()}
    print(seed,out)
