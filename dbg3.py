import sys, os; sys.path.insert(0,'.')
os.environ['TCSFM_DEBUG_STAMPS']='1'
import numpy as np, torch, ctypes as C
from tightly_coupled_sfm_amd import synth
from tightly_coupled_sfm_amd.engine import Engine, default_opts
H,W,N=192,640,2
b=synth.make_batch(N,H,W,seed0=0,both_directions=True)
t=lambda a: torch.as_tensor(np.ascontiguousarray(a,dtype=np.float32)).cuda()
e=Engine(H,W,N)
d=(t(b['tgt']),t(b['src']),t(b['depth_t']),t(b['depth_s']),t(b['K']))
for it in range(3):
    pose,_,st=e.refine(*d,t(b['pose_init']),default_opts(n_iters=4))
    torch.cuda.synchronize()
    out=(C.c_longlong*8)(); e.lib.tcsfm_debug_stamps(e._h,out)
    v=np.array(list(out)); print('stamps (us from start):', ((v[1:7]-v[0])/100.0).round(2))
