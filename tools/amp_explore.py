import sys, numpy as np, torch
sys.path.insert(0,'.')
from tests import util as U
from tests.test_loss import _case
from moldiff_amd import train_ops
z=U.gold('loss_amp.npz'); z32=U.gold('loss_grads.npz')
S=float(z['scale'])
for nm,kind in (('full','MolDiff'),('simple','MolDiff_simple')):
    args,t,noise,want=_case(nm,'cuda')
    m=U.moldiff(kind,'cuda')
    for tag,mode in (('fp16','fp16'),('fp16','fp16_f32store'),('fp16','f32')):
        m.zero_grad(set_to_none=True)
        with train_ops.precision(mode):
            got=m.get_loss(*args,time_step=t,noise=noise)
            (got['loss']*S).backward()
        names=[k[len(nm)+len(tag)+7:] for k in z.files if k.startswith(f'{nm}/{tag}/norm/')]
        P=dict(m.named_parameters())
        gmax=max(float(z[f'{nm}/{tag}/norm/{k}']) for k in names)
        rel=[];cos=[]
        for k in names:
            g=P[k].grad.detach()/S
            w=float(z[f'{nm}/{tag}/norm/{k}'])
            if w>1e-2*gmax: rel.append(abs(float(g.double().norm())-w)/w)
            fk=f'{nm}/{tag}/full/{k}'
            if fk in z.files and w>1e-2*gmax:
                wv=torch.from_numpy(z[fk]).cuda().flatten().double(); gv=g.flatten().double()
                cos.append(float((wv*gv).sum()/wv.norm()/gv.norm()))
        rel=np.sort(rel); cos=np.sort(cos)
        print(nm, 'golden',tag,'mode',mode, 'loss', float(got['loss']), 'gold', float(z[f'{nm}/{tag}/loss']), 'fp32', want['loss'],
              'norm rel median %.4f p95 %.4f max %.4f'%(rel[len(rel)//2], rel[int(.95*len(rel))], rel[-1]), 'cos min %.4f p5 %.4f med %.4f n=%d'%(cos[0],cos[int(.05*len(cos))],cos[len(cos)//2],len(cos)),
              'finite', all(bool(torch.isfinite(p.grad).all()) for p in P.values() if p.grad is not None))
    # reference fp32 vs reference autocast
    for tag in ('fp16','bf16'):
        names=[k[len(nm)+len(tag)+7:] for k in z.files if k.startswith(f'{nm}/{tag}/norm/')]
        gmax=max(float(z[f'{nm}/{tag}/norm/{k}']) for k in names)
        rel=np.sort([abs(float(z32[f'{nm}/norm/{k}'])-float(z[f'{nm}/{tag}/norm/{k}']))/float(z[f'{nm}/{tag}/norm/{k}']) for k in names if float(z[f'{nm}/{tag}/norm/{k}'])>1e-2*gmax])
        print(nm,'REF fp32 vs REF',tag,'norm rel median %.4f p95 %.4f max %.4f'%(rel[len(rel)//2], rel[int(.95*len(rel))], rel[-1]))
