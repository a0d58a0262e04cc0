import sys, os, ctypes as C
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
import gym_xarm_amd
from oracle import oracle as O
H = C.CDLL(os.path.join(ROOT,'tests/hostbuild/libxarm_host.so'))
dp=C.POINTER(C.c_double)
def P(a): return a.ctypes.data_as(dp)
names=['q']*9+['qd']*9+['bp']*3+['bq']*4+['bv']*3+['bw']*3+['goal']*3+['lt']*8+['lp']*8+['touch','mug','steps','ep']
E=64
env=gym_xarm_amd.make("XarmPDPickAndPlace-v0", num_envs=E, seed=3, auto_reset=False)
st0=env.get_state().cpu().numpy().astype(np.float64)
print('init state row0', st0[0,:34])
qt=np.zeros((E,9)); qt[:,3]=0.5; qt[:,7:]=0.02
for n in (1,2,15):
    env.set_state(st0)
    env.debug_substeps(qt, n)
    dev=env.get_state().cpu().numpy().astype(np.float64)
    hs=st0.copy(); H.xh_substep(1, C.c_int64(E), P(hs), P(qt), n)
    d=np.abs(dev-hs).max(0)
    print('n=%d max diff dev vs host-f32: %.3e at field %d (%s)'%(n,d.max(),d.argmax(),names[d.argmax()]))
    if d.max()>1e-3:
        for f in range(54):
            if d[f]>1e-4: print('  field',f,names[f],'diff',d[f],'dev',dev[0,f],'host',hs[0,f])
# ---- reset and step vs host-f32
u8=C.POINTER(C.c_uint8)
def U(a): return a.ctypes.data_as(u8)
env.set_state(st0)
env.reset()
dev=env.get_state().cpu().numpy().astype(np.float64)
hs=st0.copy(); obs=np.zeros((E,24)); ag=np.zeros((E,3)); dg=np.zeros((E,3))
H.xh_reset(1, C.c_uint64(3), C.c_int64(0), C.c_double(0), C.c_double(0), 0, 0, C.c_int64(E), P(hs), None, P(obs), P(ag), P(dg))
d=np.abs(dev-hs).max(0)
print('reset: max diff dev vs host-f32: %.3e at field %d (%s)'%(d.max(),d.argmax(),names[d.argmax()]))
for f in range(54):
    if d[f]>1e-3: print('  field',f,names[f],'diff',d[f],'dev',dev[0,f],'host',hs[0,f])
print('dev row0 q', dev[0,:9]); print('host row0 q', hs[0,:9])
# step from host reset state
env.set_state(hs)
a=(np.random.default_rng(0).random((E,4))*2-1)
o,r,dn,info=env.step(torch.tensor(a,dtype=torch.float32).cuda())
dev=env.get_state().cpu().numpy().astype(np.float64)
hs2=hs.copy(); rew=np.zeros(E); dd=np.zeros(E,np.uint8); ss=np.zeros(E,np.uint8)
H.xh_step(1, C.c_uint64(3), C.c_int64(0), C.c_double(0), C.c_double(0), 0, 0, C.c_int64(E), P(hs2), P(a), P(obs), P(ag), P(dg), P(rew), U(dd), U(ss))
d=np.abs(dev-hs2).max(0)
print('step: max diff dev vs host-f32: %.3e at field %d (%s)'%(d.max(),d.argmax(),names[d.argmax()]))
for f in range(54):
    if d[f]>1e-3: print('  field',f,names[f],'diff',d[f],'dev',dev[0,f],'host',hs2[0,f])
# ---- substeps from the post-reset states (some envs have pad contact / deep penetration)
qt2=hs[:, :9].copy()
for n in (1,2,5,15):
    env.set_state(hs)
    env.debug_substeps(qt2, n)
    dev=env.get_state().cpu().numpy().astype(np.float64)
    h2=hs.copy(); H.xh_substep(1, C.c_int64(E), P(h2), P(qt2), n)
    h3=hs.copy(); H.xh_substep(0, C.c_int64(E), P(h3), P(qt2), n)
    d=np.abs(dev-h2).max(1); d64=np.abs(h3-h2).max(1)
    worst=np.argsort(-d)[:4]
    print('post-reset n=%d: dev-vs-hostf32 max %.3e ; hostf64-vs-hostf32 max %.3e'%(n,d.max(),d64.max()))
    for w in worst: print('   env',w,'dev diff %.3e f64 diff %.3e'%(d[w],d64[w]),'lam_p',np.round(h2[w,42:50],4),'box',np.round(hs[w,18:21],3))
