"""Per-kernel timing of the fwd+bwd step at the headline shape (HIP events inside the library).

    python tools/kernel_timing.py            # env: ABL_N (Gaussians), ABL_BWD=0 forward only, ABL_TAG label,
                                             #      VTGS_FWD_IMPL / VTGS_BWD_IMPL = 0 scalar kernels, 1 matrix-core kernels; ABL_W / ABL_H image size; ABL_OPACITY constant opacity; ABL_BAND=r/w one band of a w-way partition
"""
import sys, os, time, torch
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0]=[ROOT,os.path.join(ROOT,'vtgaussian-slam_amd'),os.path.join(ROOT,'tests')]
from oracle import gs_oracle as go
from parity_util import to_settings
import diff_gaussian_rasterization as dgr
dev=torch.device('cuda:0')
if os.environ.get('ABL_POISON','0')=='1': dgr.poison_workspaces(True)   # every workspace / scratch starts as 0xFF bytes: a kernel that reads a slot nobody wrote shows up
N=int(os.environ.get('ABL_N','1000000'))
W=int(os.environ.get('ABL_W','1200')); H=int(os.environ.get('ABL_H','680'))
scene,cam=go.view_tied_scene(N,W,H,seed=0)
if os.environ.get('ABL_TAIL','0')!='0':
    # a map AFTER some frames of mapping (round 6): the optimiser has grown a heavy tail of splat sizes -- at frame 21 of the synthetic
    # Replica sequence 6 % of the Gaussians have 5-6 instances, 8 % have 7..64, 0.1 % hundreds (gpurun_out/r6/slam_c.log).  Here:
    # 14 % of the scales x exp(U(0.3, 1.6)), 1 % x exp(U(1.6, 3)), 0.03 % x exp(U(3, 4.3))
    gt=torch.Generator().manual_seed(5); u=torch.rand(N,generator=gt); f=torch.ones(N)
    r=torch.rand(N,generator=gt)
    f=torch.where(u<0.14, torch.exp(0.3+1.3*r), f); f=torch.where(u<0.01, torch.exp(1.6+1.4*r), f); f=torch.where(u<0.0003, torch.exp(3.0+1.3*r), f)
    scene['scales']=scene['scales']*f[:,None]
if 'ABL_OPACITY' in os.environ: scene['opacities']=torch.full_like(scene['opacities'], float(os.environ['ABL_OPACITY']))   # saturating scenes
leaves={k:v.to(dev).requires_grad_(True) for k,v in scene.items()}
band=None
if 'ABL_BAND' in os.environ:                                   # "r/w": time rank r's band of a w-way tile-row partition
    from diff_gaussian_rasterization.partition import band_for_rank
    r_,w_=os.environ['ABL_BAND'].split('/'); band=band_for_rank(H,int(w_),int(r_))
rast=dgr.GaussianRasterizer(raster_settings=to_settings(cam,dev),tile_rows=band)
g=torch.rand(3,H,W,device=dev)
bwd=os.environ.get('ABL_BWD','1')=='1'
def step():
    c,r,d=rast(**leaves)
    if bwd: c.backward(g)
if os.environ.get('ABL_SINGLE_THREAD_AUTOGRAD')=='1': torch.autograd.set_multithreading_enabled(False)   # backward on the calling thread: no hand-off to the engine's device thread
for it in range(5): step()
torch.cuda.synchronize()
dgr.profile_enable(True)
for it in range(20): step()
p=dgr.profile_collect()
torch.cuda.synchronize()
dgr.profile_enable(False)
for it in range(10): step()
torch.cuda.synchronize(); t=time.time()
e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True); e0.record()
for it in range(20): step()
e1.record(); torch.cuda.synchronize(); dt=(time.time()-t)/20*1e3
print('policy: run-ahead', [bool(v) for v in dgr._async_ok.values()], 'capacities', list(dgr._caps_in_use.values()), 'info', dgr.last_forward_info(),
      'events %.3f ms per step'%(e0.elapsed_time(e1)/20), flush=True)
if os.environ.get('ABL_TAIL','0')!='0':
    offs,gid,_=dgr.debug_tile_lists(rast); per=torch.bincount(gid,minlength=N)
    edges=[0,1,2,3,5,7,9,13,17,33,65,257,1<<30]
    print('instances',int(per.sum()),{f"{a}..{b-1}":int(((per>=a)&(per<b)).sum()) for a,b in zip(edges[:-1],edges[1:])}, flush=True)
print(os.environ.get('ABL_TAG',''), 'step %.3f ms'%dt, {k:round(v[0]/v[1]*1e3,1) for k,v in p.items()}, flush=True)
