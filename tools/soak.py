"""Memory-stability soak: 3,000 forward+backward steps through the drop-in operator, reserved / allocated bytes every 1,000."""
import sys, os, torch, time
ROOT='/root/repo'
sys.path[:0]=[ROOT,os.path.join(ROOT,'vtgaussian-slam_amd'),os.path.join(ROOT,'tests')]
from oracle import gs_oracle as go
from parity_util import to_settings
import diff_gaussian_rasterization as dgr
dev=torch.device('cuda:0')
scene,cam=go.view_tied_scene(300000,640,480,seed=0)
leaves={k:v.to(dev).requires_grad_(True) for k,v in scene.items()}
st=to_settings(cam,dev); g=torch.rand(3,480,640,device=dev)
t0=time.time()
for it in range(3000):
    for t in leaves.values(): t.grad=None
    c,r,d=dgr.GaussianRasterizer(raster_settings=st)(**leaves); c.backward(g)
    if it%1000==0:
        torch.cuda.synchronize(); print(it, round(torch.cuda.memory_reserved()/1e6), 'MB reserved', round(torch.cuda.memory_allocated()/1e6), 'MB allocated', flush=True)
torch.cuda.synchronize(); print('done', round((time.time()-t0)/3000*1e3,3),'ms/step', round(torch.cuda.memory_reserved()/1e6),'MB reserved', torch.cuda.memory_stats().get('num_alloc_retries',0),'retries')
