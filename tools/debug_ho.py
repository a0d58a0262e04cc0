import sys, os
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
import gym_xarm_amd
from oracle import oracle as O
E=64
env=gym_xarm_amd.make("XarmPDHandover-v0", num_envs=E, seed=2, auto_reset=False)
ora=O.OracleHandover(E,seed=2)
print('init diff', np.abs(env.get_state().cpu().numpy()-ora.state).max())
obs=env.reset(); o=ora.reset()
st=env.get_state().cpu().numpy().astype(np.float64)
print('reset diff state %.2e obs %.2e'%(np.abs(st-ora.state).max(), np.abs(obs["observation"].cpu().numpy()-o[0]).max()))
def ezpolicy(o):
    obj=o[0:3]; g1=o[13:16]; q1=o[19]; g2=o[21:24]; q2=o[27]
    ig1 = q1<0.25 and np.linalg.norm(obj-g1)<0.05; ig2 = q2<0.25 and np.linalg.norm(obj-g2)<0.05
    d1=obj-g1+[-0.07,0,0]; n1=np.linalg.norm(d1); d2=obj-g2+[0.07,0,0]; n2=np.linalg.norm(d2)
    a=[0.0]*8
    a[3] = -0.5 if np.linalg.norm(obj-g1)<0.1 else 0.5
    a[7] = -0.5 if np.linalg.norm(obj-g2)<0.1 else 0.5
    if not ig1: a[0:3]=list(d1/n1)
    else:
        if not ig2: a[0:3]=[0.5,0,0.5]; a[4:7]=list(d2/n2)
        else: a[4]=-0.5
    return a
ob=o[0]
for k in range(24):
    a=np.array([ezpolicy(ob[e]) for e in range(E)])
    s0=ora.get_state()
    env.set_state(s0)
    obs,rew,done,info=env.step(torch.tensor(a,dtype=torch.float32))
    r=ora.step(a); ob=r[0]
    dev=env.get_state().cpu().numpy().astype(np.float64)
    d=np.abs(dev-ora.state).max(1)
    if k%3==0: print(k,'dev-oracle max %.2e median %.2e'%(d.max(),np.median(d)),'touch',ora.state[:,70:72].sum(0),'rew eq',np.array_equal(rew.cpu().numpy(),r[3].astype(np.float32)),'obs %.1e'%np.abs(obs["observation"].cpu().numpy()-r[0]).max())
print('handed over (x>0.05 & z>0.08):', ((ora.state[:,38]>0.05)&(ora.state[:,40]>0.08)).sum(), 'of', E)
