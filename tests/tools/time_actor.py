import sys, time; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/drl-on-robot-arm_amd')
import numpy as np, torch
from armenv import _lib as L
import os
if len(sys.argv)>2: L.LIB_PATH=sys.argv[2]
from armenv import envs
g=np.load('/root/repo/tests/golden/td3_actor_seed0.npz')
sd={k: torch.from_numpy(g[k.replace('.','_')]) for k in ("fc1.weight","fc1.bias","fc2.weight","fc2.bias","fc3.weight","fc3.bias")}
for prec in (64,32):
    e=envs.BatchedReachEnv(65536, device='cuda:0', precision=prec)
    e.set_policy(sys.argv[1] if len(sys.argv)>1 else 'actor', actor_state_dict=sd)
    for n in (65536, 131072, 262144):
        s=torch.rand(n,6,device='cuda:0')
        for _ in range(3): e.actor_forward(s)
        torch.cuda.synchronize(); ev0=torch.cuda.Event(enable_timing=True); ev1=torch.cuda.Event(enable_timing=True)
        ev0.record()
        for _ in range(20): e.actor_forward(s)
        ev1.record(); torch.cuda.synchronize()
        us=ev0.elapsed_time(ev1)*1e3/20
        print('prec',prec,'n',n,'actor_forward us',round(us,1),'TF', round(2*(6*256+256*256+256*3)*n/us/1e6,1))
    e.close()
