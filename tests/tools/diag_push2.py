import sys; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/drl-on-robot-arm_amd')
import numpy as np, torch
from armenv import envs
from oracle import oracle as O
n=1029
kuka=O.make_chain('kuka'); cfg=O.default_config('push')
e=envs.BatchedPushEnv(n, device='cuda:0', seed=2, auto_reset=False, precision=int(sys.argv[1]) if len(sys.argv)>1 else 64)
st=O.PushState(n); obs_r=O.push_reset(kuka,cfg,st,seed=2); e.reset()
rng=np.random.default_rng(70)
chase=np.arange(n)<n//2
for t in range(5):
    a=rng.normal(0,0.39,(n,3)).astype(np.float32)
    want=st.aux[:,0:3].copy(); want[:,2]=0.015
    c=np.clip((want-obs_r[:,:3].astype(np.float64))/0.08,-1,1).astype(np.float32); a[chase]=c[chase]
    e.set_state(q=st.q, aux=st.aux, step=st.step, ep_return=st.ep_return)
    pre=st.aux.copy()
    obs,rew,done,succ=e.step(torch.from_numpy(a).cuda()); obs=obs.cpu().numpy().copy()
    obs_r,rew_r,done_r,succ_r,it=O.push_step(kuka,cfg,st,a)
    s=e.get_state(); aux=s['aux'].cpu().numpy()
    d=np.abs(aux[:,:7]-st.aux[:,:7])
    bad=np.where(d.max(1)>(1e-9 if len(sys.argv)<2 else 2e-4))[0]
    print('t',t,'bad',len(bad), 'of chase', (bad<n//2).sum())
    for i in bad[:4]:
        print(' env',i,'it',it[i],'pre',pre[i,:7].round(6),'\n   gpu',aux[i,:7].round(6),'\n   ref',st.aux[i,:7].round(6),'\n   eef gpu',obs[i,:3],'ref',obs_r[i,:3], 'dq', np.abs(s['q'].cpu().numpy()[i]-st.q[i]).max())
