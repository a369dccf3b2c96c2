import sys, os, ctypes, numpy as np
sys.path.insert(0,'/root/repo')
from oracle import oracle_py as O, urdf_model, crosscheck_np as X
O._LIB = '/root/repo/oracle/_build/libwbc_oracle_asan.so'
O._lib = None
O.build = lambda force=False: O._LIB
import wbc_quadruped_dob_amd as W
from wbc_quadruped_dob_amd import synth
flat = urdf_model.load_urdf('/root/repo/wbc_quadruped_dob_amd/assets/synthetic_quadruped.urdf')
orc = O.Oracle(flat)
for cfg, obs in ((2,0),(3,1),(4,2)):
    for dt in (np.float64, np.float32):
        n=257
        B = synth.make_batch(cfg, n, float(flat['mass'].sum()))
        P = synth.default_params(observer_order=obs, dtype='f64' if dt==np.float64 else 'f32')
        c=lambda a:a.astype(dt)
        integ = orc.dynamics(c(B['q']),c(B['v']))['p'].copy(); r=np.zeros((n,18),dt)
        B['mask'][:16]=np.arange(16)
        o=orc.step(P,c(B['q']),c(B['v']),c(B['w_des']),c(B['vdot_des']),c(B['normals']),c(B['mu']),B['mask'],c(B['tau_prev']),c(B['f_prev']),integ,r,nthreads=4)
        q,v=c(B['q']).copy(),c(B['v']).copy()
        orc.rollout(P,5,q,v,c(B['w_des']),c(B['vdot_des']),c(B['normals']),c(B['mu']),B['mask'],integ=integ,r=r,want_traj=True,nthreads=4)
        print(cfg,obs,dt.__name__,'status',np.bincount(o['status']))
# degenerate QPs
H=np.eye(2); g=np.zeros(2); C=np.array([[1.0,0],[-1.0,0]]); d=np.array([1.0,0.0])
print(O.qp_solve(H,g,C,d)[2])
print('asan run complete')
# reference generator, tracking rollout, op count, per-QP timing (added after the first ASan pass)
G = synth.default_ref_params()
for dt in (np.float64, np.float32):
    n = 65
    B = synth.make_batch(3, n, float(flat['mass'].sum()))
    plan = synth.make_plan(B).astype(dt)
    c = lambda a: a.astype(dt)
    orc.reference(G, c(B['q']), c(B['v']), plan, 0.01)
    P = synth.default_params(observer_order=1, dtype='f64' if dt == np.float64 else 'f32')
    q, v = c(B['q']).copy(), c(B['v']).copy()
    integ = orc.dynamics(q, v)['p'].copy(); r = np.zeros((n, 18), dt)
    orc.rollout_tracking(P, G, 4, q, v, plan, c(B['normals']), c(B['mu']), B['mask'], integ=integ, r=r, want_traj=True, want_com=True, nthreads=3)
B = synth.make_batch(3, 16, float(flat['mass'].sum()))
P = synth.default_params(observer_order=2)
for s in range(16):
    orc.op_count(P, B['q'][s], B['v'][s], B['w_des'][s], B['vdot_des'][s], B['normals'][s], B['mu'][s], int(B['mask'][s]),
                 B['tau_prev'][s], B['f_prev'][s], np.zeros(18), np.zeros(18))
orc.qp_time(P, B['q'], B['v'], B['w_des'], B['normals'], B['mu'], B['mask'])
print('asan run 2 complete')
