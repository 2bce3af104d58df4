import sys, os
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, ROOT+'/sw-nerf_amd', ROOT+'/tests/golden'): sys.path.insert(0,p)
import numpy as np, torch, cases
from oracle import nerf_oracle as O
import swnerf.render as R, swnerf.model as M, swnerf.embedder as E
torch.set_grad_enabled(False)
dev=torch.device('cuda:0'); T=lambda a: torch.from_numpy(np.ascontiguousarray(a))
e10,_=E.get_embedder(10,3,0)
dn=M.NeRF.get_by_name("direct_temporal", D=8, W=256, input_ch=63, output_ch=5, skips=[4], input_ch_views=27, input_ch_time=21, use_viewdirs=True, embed_fn=e10, zero_canonical=True)
sdn=cases.weights_dnerf(); dn.load_state_dict({k:T(v) for k,v in sdn.items()}); dn=dn.to(dev).eval()
sd=O.to_torch_sd(sdn)
g=cases.g8_inputs(n=256)
for tv in (0.0, 0.5):
    rb=O.make_ray_batch(T(g['rays_o']),T(g['rays_d']),2.,6.,frame_time=tv)
    ref=O.render_rays_dnerf(rb, sd, 64, 0, white_bkgd=True, retraw=True)
    p=R.render_pass(rb.to(dev), dn, 64, white_bkgd=True, want=['rgb_map','acc_map','raw','dx','z_out'], run_deform=(tv!=0))
    f=lambda a,b: float((a.cpu()-b).abs().max())
    mse=float(((p['rgb_map'].cpu()-ref['rgb_map'])**2).mean())
    print('t=%.1f no-resample: dx max diff %.3e (|dx|max %.3f)  raw %.3e  rgb %.3e  acc %.3e  psnr %.1f'%(tv, f(p['dx'],ref['position_delta']), float(ref['position_delta'].abs().max()), f(p['raw'],ref['raw']), f(p['rgb_map'],ref['rgb_map']), f(p['acc_map'],ref['acc_map']), -10*np.log10(max(mse,1e-20))))
    # feed the ORACLE's dx-shifted points through canonical only: isolates canonical parity
# sensitivity of the ORACLE itself: perturb dx by 1 ulp-ish (1e-7 relative) and see output change
rb=O.make_ray_batch(T(g['rays_o']),T(g['rays_d']),2.,6.,frame_time=0.5)
z=O.coarse_z(rb[:,6:7], rb[:,7:8], 64); pts=rb[:,None,0:3]+rb[:,None,3:6]*z[...,None]
raw,dx=O.run_network_dnerf(sd, pts, rb[:,-3:], rb[:,8:9])
pts2=(pts+dx)
e=O.embed(pts2.reshape(-1,3),10); e2=O.embed((pts2+2e-7*torch.sign(torch.randn_like(pts2))).reshape(-1,3),10)
dirs=O.embed(rb[:,None,-3:].expand(pts.shape).reshape(-1,3),4)
ca={k[5:]:v for k,v in sd.items() if k.startswith('_occ.')}
r1=O.nerf_mlp(ca, torch.cat([e,dirs],-1)); r2=O.nerf_mlp(ca, torch.cat([e2,dirs],-1))
print('oracle self-sensitivity: +-2e-7 shift of (x+dx) -> max |d raw| %.3e, mean %.3e'%(float((r1-r2).abs().max()), float((r1-r2).abs().mean())))
