import sys, os, math, time, torch
ROOT='/root/repo'
sys.path[:0]=[ROOT,os.path.join(ROOT,'vtgaussian-slam_amd'),os.path.join(ROOT,'tests')]
from oracle import gs_oracle as go
from parity_util import to_settings
import diff_gaussian_rasterization as dgr
dev=torch.device('cuda:0')
N,W,H=1000000,1200,680
scene,cam=go.view_tied_scene(N,W,H,seed=0)
st=to_settings(cam,dev)
g=torch.rand(3,H,W,device=dev)
def pose(t):
    a=math.radians(0.15*t)/2; ax=torch.tensor([0.3,1.0,0.1]); ax=ax/ax.norm()
    q=torch.cat([torch.tensor([math.cos(a)]), math.sin(a)*ax]); tr=torch.tensor([0.004*t,-0.002*t,0.003*t])
    R=go.quat_to_rotmat(q[None])[0]
    return R,tr
for t in (0,10,30,60):
    R,tr=pose(t)
    m=(scene['means3D']@R.T+tr)
    leaves={k:v.to(dev).requires_grad_(True) for k,v in dict(scene,means3D=m).items()}
    rast=dgr.GaussianRasterizer(raster_settings=st)
    def step():
        c,r,d=rast(**leaves); c.backward(g)
    for _ in range(5): step()
    torch.cuda.synchronize(); dgr.profile_enable(True)
    for _ in range(10): step()
    p=dgr.profile_collect(); dgr.profile_enable(False)
    info=dgr.last_forward_info()
    print(t, {k:round(v[0]/v[1]*1e3,1) for k,v in p.items()}, {k:info[k] for k in ('instances','visible','max_tile_list','capacity')}, flush=True)
