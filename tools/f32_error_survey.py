import sys, numpy as np, torch
sys.path.insert(0,'.')
import wbc_quadruped_dob_amd as W
from wbc_quadruped_dob_amd import synth
from oracle import oracle_py, urdf_model
from tests.util import relerr, to_dev, to_host
m=W.Model.from_urdf(W.SYNTHETIC_URDF); orc=oracle_py.Oracle(urdf_model.load_urdf(W.SYNTHETIC_URDF))
n=32768
for cfg,obs in ((2,0),(4,1)):
    P32=synth.default_params(observer_order=obs,dtype='f32'); P64=synth.default_params(observer_order=obs)
    B=synth.make_batch(cfg,n,m.total_mass)
    f32=lambda a:a.astype(np.float32)
    integ=orc.dynamics(B['q'],B['v'],nthreads=8)['p']; r=np.zeros((n,18))
    i32,r32=f32(integ),f32(r); i64,r64=integ.copy(),r.copy()
    ref32=orc.step(P32,f32(B['q']),f32(B['v']),f32(B['w_des']),f32(B['vdot_des']),f32(B['normals']),f32(B['mu']),B['mask'],f32(B['tau_prev']),f32(B['f_prev']),i32,r32,nthreads=8)
    ref64=orc.step(P64,B['q'],B['v'],B['w_des'],B['vdot_des'],B['normals'],B['mu'],B['mask'],B['tau_prev'],B['f_prev'],i64,r64,nthreads=8)
    s=W.Solver(m,W.Params.from_dict(P32,'f32'),dtype='f32',max_batch=n)
    td=torch.float32; dv=lambda k: to_dev(B[k],torch,td)
    ig=to_dev(integ,torch,td); rr=to_dev(r,torch,td)
    out=s.step(dv('q'),dv('v'),dv('w_des'),dv('vdot_des'),dv('normals'),dv('mu'),torch.from_numpy(B['mask']).cuda(),dv('tau_prev'),dv('f_prev'),ig,rr)
    torch.cuda.synchronize()
    tau=to_host(out['tau']); f=to_host(out['f']); st=out['status'].cpu().numpy()
    ok=(st==0)&(ref32['status']==0)&(ref64['status']==0)
    sc=np.abs(ref64['tau']).max()
    e_g32=np.abs(tau-ref32['tau'])[ok].max(1)/sc; e_g64=np.abs(tau-ref64['tau'])[ok].max(1)/sc; e_o=np.abs(ref32['tau']-ref64['tau'])[ok].max(1)/sc
    pr=lambda e: 'p50 %.1e p99 %.1e max %.1e'%(np.median(e),np.percentile(e,99),e.max())
    print('cfg',cfg,'ok frac',ok.mean(),'status mism', (st!=ref32['status']).mean())
    print('  gpu32 vs oracle32:',pr(e_g32)); print('  gpu32 vs oracle64:',pr(e_g64)); print('  oracle32 vs oracle64:',pr(e_o))
