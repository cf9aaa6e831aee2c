import sys, os, ctypes as C
import numpy as np, torch
ROOT=os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT,'..','..','..','drl-on-robot-arm_amd'))
from armenv import _lib as L
L.LIB_PATH=os.path.join(ROOT,'libarmenv_tl.so')
from armenv import envs
prec=int(sys.argv[1]) if len(sys.argv)>1 else 64
n=int(sys.argv[2]) if len(sys.argv)>2 else 65536
e=envs.BatchedReachEnv(n, device='cuda:0', precision=prec)
lib=L.load()
tl=torch.zeros((16,n//64,8),dtype=torch.int64,device='cuda:0')
lib.armenv_dbg_set_timeline.argtypes=[C.c_void_p]
assert lib.armenv_dbg_set_timeline(C.c_void_p(tl.data_ptr()))==0
g=torch.Generator(device='cuda:0'); g.manual_seed(0)
NA=512 if n<=65536 else 64   # i.i.d. actions per step (a short cycle would walk every env into a corner of the box)
acts=(torch.randn((NA,n,3),device='cuda:0',generator=g)*0.686).clamp_(-0.7,0.7)
e.reset()
_c=[0]
def nxt():
    _c[0]+=1; return (_c[0]-1)%NA
for k in range(30): e.step(acts[nxt()])
torch.cuda.synchronize()
for k in range(3):
    e.step(acts[nxt()]); torch.cuda.synchronize()
    t=tl.cpu().numpy().astype(np.int64)[(30+k)%16]
    t0=t[:,0].min()
    st=(t[:,0]-t0)/100.0; ld=(t[:,1]-t[:,0])/100.0; lp=(t[:,2]-t[:,1])/100.0; fin=(t[:,3]-t[:,2])/100.0; end=(t[:,3]-t0)/100.0
    print(f"step {k}: kernel span {end.max():.2f} us | start: mean {st.mean():.2f} max {st.max():.2f} | load: mean {ld.mean():.2f} max {ld.max():.2f} | loop: mean {lp.mean():.2f} min {lp.min():.2f} max {lp.max():.2f} | epilogue mean {fin.mean():.2f} max {fin.max():.2f}")
    for tr in np.unique(t[:,4]):
        m=t[:,4]==tr; print(f"   trips={tr}: waves {m.sum()} loop mean {lp[m].mean():.2f} us; end mean {end[m].mean():.2f}")
    m=t[:,5]>0; print(f"   waves with done lanes: {m.sum()}, epilogue mean {fin[m].mean() if m.any() else 0:.2f} vs {fin[~m].mean():.2f}")
    hw=t[:,7]; xcc=t[:,6]; cu=(hw>>8)&0xf; sh=(hw>>12)&1; se=(hw>>13)&0x7; simd=(hw>>4)&3
    key=xcc*100000+se*1000+sh*100+cu
    u,c=np.unique(key,return_counts=True); print("   distinct CUs used:",len(u)," waves/CU hist:",np.bincount(c))
    key2=key*10+simd; u2,c2=np.unique(key2,return_counts=True); print("   distinct SIMDs:",len(u2)," waves/SIMD hist:",np.bincount(c2))

# back-to-back launches (no host sync): kernel k's last wave exit -> kernel k+1's first wave entry, on the shared 100 MHz clock
torch.cuda.synchronize()
base=30+3
for k in range(12): e.step(acts[nxt()])
torch.cuda.synchronize()
t=tl.cpu().numpy().astype(np.int64)
rows=[t[(base+k)%16] for k in range(12)]
print("back-to-back launches: [first entry -> last exit] span, gap to the next kernel's first entry, entry-to-entry period (us)")
for k in range(11):
    a,b=rows[k],rows[k+1]
    print(f"   launch {k}: span {(a[:,3].max()-a[:,0].min())/100:.2f}  gap {(b[:,0].min()-a[:,3].max())/100:.2f}  period {(b[:,0].min()-a[:,0].min())/100:.2f}  start skew max {(a[:,0].max()-a[:,0].min())/100:.2f}  slowest wave {((a[:,3]-a[:,0]).max())/100:.2f} mean {((a[:,3]-a[:,0]).mean())/100:.2f}")

# persistence of slow waves: are the waves that need extra IK trips the same ones launch after launch?
warm=int(sys.argv[3]) if len(sys.argv)>3 else 0
for k in range(warm): e.step(acts[nxt()])
torch.cuda.synchronize()
tot=np.zeros(n//64); trips=[]
for rnd in range(4):
    for k in range(12): e.step(acts[nxt()])
    torch.cuda.synchronize()
    t=tl.cpu().numpy().astype(np.int64)
    # launch counter position is unknown after warm-up: use every ring slot once (12 of the 16 slots are fresh)
    fresh=np.argsort(-t[:,:,0].min(axis=1))[:12]
    for s_ in fresh:
        tot+= (t[s_][:,3]-t[s_][:,0])/100.0; trips.append(t[s_][:,4].copy())
trips=np.array(trips)   # [48 launches][waves]
# where does the spread between waves come from: their own trip counts, or where they run?
tsum=trips.sum(axis=0).astype(float)
A=np.vstack([np.ones_like(tsum),tsum]).T
coef,_,_,_=np.linalg.lstsq(A,tot,rcond=None)
resid=tot-A@coef
hw=t[0][:,7]; xcc=t[0][:,6]; cu=(hw>>8)&0xf; sh=(hw>>12)&1; se=(hw>>13)&0x7; simd=(hw>>4)&3
print(f"   time = {coef[0]:.1f} + {coef[1]:.2f} us x trips; residual std {resid.std():.2f} us, max {resid.max():.1f}")
print("   residual mean by XCC:", " ".join(f"{resid[xcc==x].mean():+.1f}" for x in np.unique(xcc)))
print("   residual mean by SIMD:", " ".join(f"{resid[simd==x].mean():+.1f}" for x in np.unique(simd)))
print("   residual mean by SE:", " ".join(f"{resid[se==x].mean():+.1f}" for x in np.unique(se)))
slow=np.argsort(-tot)[:8]
print("   slowest waves: " + "; ".join(f"w{w} {tot[w]:.0f}us trips {int(tsum[w])} xcc{xcc[w]} se{se[w]} cu{cu[w]} simd{simd[w]}" for w in slow))

print(f"after {warm} more steps, 48 launches: in-kernel time per wave summed: mean {tot.mean():.1f} us, std {tot.std():.1f}, max {tot.max():.1f} (+{100*(tot.max()/tot.mean()-1):.1f} %), sum of per-launch maxima n/a")
print("   wave-trips histogram over all wave-launches:", np.bincount(trips.ravel()))
extra=(trips>3).sum(axis=0)
print("   launches (of 48) in which a wave needed >3 trips: histogram over waves", np.bincount(extra))
print("   mean trips per wave-launch", trips.mean(), " mean of per-launch max", trips.max(axis=1).mean())
