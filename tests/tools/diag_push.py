import sys; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/drl-on-robot-arm_amd')
import numpy as np, torch
from armenv import envs
from oracle import oracle as O
n=2048
kuka=O.make_chain('kuka'); cfg=O.default_config('push')
for form in (0,1):
    cfg.ik_form=form
    e=envs.BatchedPushEnv(n, device='cuda:0', seed=2, auto_reset=False)
    st=O.PushState(n); obs_r=O.push_reset(kuka,cfg,st,seed=2); e.reset()
    rng=np.random.default_rng(70)
    for t in range(6):
        a=rng.normal(0,0.39,(n,3)).astype(np.float32)
        e.set_state(q=st.q, aux=st.aux, step=st.step, ep_return=st.ep_return)
        obs,rew,done,succ=e.step(torch.from_numpy(a).cuda())
        obs=obs.cpu().numpy().copy()
        obs_r,rew_r,done_r,succ_r,it=O.push_step(kuka,cfg,st,a)
        dq=np.abs(e.get_state()['q'].cpu().numpy()-st.q).max(1)
        dp=np.abs(obs[:,:3]-obs_r[:,:3]).max(1)
        print('form',form,'t',t,'iters',np.bincount(it)[:21].tolist(),'dq q50 %.1e q99 %.1e max %.1e'%(np.median(dq),np.quantile(dq,0.99),dq.max()),'dp max %.1e'%dp.max())
    e.close()
