import sys, os, time, torch
sys.path[:0]=['/root/repo','/root/repo/vtgaussian-slam_amd','/root/repo/tests']
from oracle import gs_oracle as go
from parity_util import to_settings
import diff_gaussian_rasterization as dgr
dev=torch.device('cuda:0')
scene,cam=go.view_tied_scene(1_000_000,1200,680,seed=0)
leaves={k:v.to(dev) for k,v in scene.items()}
rast=dgr.GaussianRasterizer(raster_settings=to_settings(cam,dev))
dgr.profile_enable(True)
for it in range(6):
    try: rast(**leaves)
    except Exception as ex: print('err',ex); break
p=dgr.profile_collect()
print(os.environ.get('VTGS_LIBRARY','base').split('/')[-1], {k:round(v[0]/v[1]*1e3,1) for k,v in p.items()}, flush=True)
