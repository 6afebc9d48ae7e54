import sys; sys.path.insert(0,'/root/repo')
import numpy as np, torch, time, ctypes as C
from ur_gym_amd import make_vec, _abi, _native
from scipy.spatial.transform import Rotation as Rot
env=make_vec("UR5DynReach-v1",num_envs=64,seed=1)
rng=np.random.default_rng(0)
n=327680
dev=env.device
ta=torch.zeros(n,dtype=torch.int32,device=dev); tb=torch.ones(n,dtype=torch.int32,device=dev)
link=np.repeat(np.arange(2,7),n//5)          # wave-uniform links like the step kernel (64-lane groups share a hull)
pa=np.zeros((n,3)); pa[:,0]=link
pb=np.tile([0.05,0.4,0],(n,1))
xa=np.c_[rng.uniform(-0.6,0.6,(n,3))+[0.5,0,0.4], Rot.random(n,random_state=1).as_quat()]
xb=np.c_[rng.uniform(-0.3,0.3,(n,3))+[0.7,0,0.4], Rot.random(n,random_state=2).as_quat()]
T=lambda a: torch.as_tensor(a,device=dev).contiguous()
pa,pb,xa,xb=T(pa),T(pb),T(xa),T(xb)
out=torch.zeros(n,dtype=torch.float64,device=dev); info=torch.zeros(n,dtype=torch.int32,device=dev)
p=lambda t: C.c_void_p(t.data_ptr())
def run():
    _native.check(env.lib.urgym_probe_closest(env._h,n,p(ta),p(pa),p(xa),p(tb),p(pb),p(xb),5.0,p(out),p(info),env._stream()),env._h)
for _ in range(3): run()
torch.cuda.synchronize(); t0=time.perf_counter()
for _ in range(10): run()
torch.cuda.synchronize(); dt=(time.perf_counter()-t0)/10
print("probe: %d hull-cyl queries in %.1f us  (%.1f M queries/s)"%(n,dt*1e6,n/dt/1e6), "penetrating",int((info&1).sum()))
