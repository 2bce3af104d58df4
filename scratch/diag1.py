import sys, os
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, ROOT+'/sw-nerf_amd', ROOT+'/tests/golden'): sys.path.insert(0,p)
import numpy as np, torch, cases
from oracle import nerf_oracle as O
import swnerf.render as R, swnerf.model as M, swnerf.embedder as E, swnerf.ray as RAY
torch.set_grad_enabled(False)
dev=torch.device('cuda:0'); T=lambda a: torch.from_numpy(np.ascontiguousarray(a))
sd_c, sd_f = cases.weights_static()
def load(sd):
    m=M.vallina_NeRF(D=8,W=256,input_ch=63,input_ch_views=27,output_ch=5,skips=[4],use_viewdirs=True); m.load_state_dict({k:T(v) for k,v in sd.items()}); return m.to(dev).eval()
nc, nf = load(sd_c), load(sd_f)
oc, of = O.to_torch_sd(sd_c), O.to_torch_sd(sd_f)
g=cases.g7_inputs()
rb=O.make_ray_batch(T(g['rays_o']),T(g['rays_d']),2.,6.)
ref=O.render_rays(rb, oc, of, 64, 128, white_bkgd=True, retraw=True)
p0=R.render_pass(rb.to(dev), nc, 64, white_bkgd=True, want=['rgb_map','acc_map','weights','z_out','raw'], n_importance=128)
# coarse weights
z0=O.coarse_z(rb[:,6:7], rb[:,7:8], 64)
pts=rb[:,None,0:3]+rb[:,None,3:6]*z0[...,None]
raw0=O.run_network(oc, pts, rb[:,-3:])
_,_,_,w0,_=O.raw2outputs(raw0,z0,rb[:,3:6],0.,True)
print('coarse raw max diff', float((p0['raw'].cpu()-raw0).abs().max()), ' weights max diff', float((p0['weights'].cpu()-w0).abs().max()), 'z_out exact', bool(torch.equal(p0['z_out'].cpu(), z0)))
dz=(p0['z_fine'].cpu()-ref['z_vals']).abs()
print('z_fine diff: max %.3e  p99.9 %.3e  p99 %.3e  frac>1e-5 %.4f'%(dz.max(), dz.flatten().kthvalue(int(dz.numel()*0.999))[0], dz.flatten().kthvalue(int(dz.numel()*0.99))[0], (dz>1e-5).float().mean()))
# fine pass with the oracle's exact z
p1=R.render_pass(rb.to(dev), nf, 192, z_vals=ref['z_vals'].to(dev), white_bkgd=True, want=['rgb_map','acc_map','raw','weights'])
print('fine pass @oracle z: rgb max diff %.3e  acc %.3e raw %.3e'%(float((p1['rgb_map'].cpu()-ref['rgb_map']).abs().max()), float((p1['acc_map'].cpu()-ref['acc_map']).abs().max()), float((p1['raw'].cpu()-ref['raw']).abs().max())))
# end-to-end
p1b=R.render_pass(rb.to(dev), nf, 192, z_vals=p0['z_fine'], white_bkgd=True, want=['rgb_map'])
d=(p1b['rgb_map'].cpu()-ref['rgb_map']).abs()
print('end-to-end rgb: max %.3e frac>2e-4 %.4f  psnr %.1f'%(d.max(), (d>2e-4).float().mean(), -10*np.log10(float((d**2).mean()))))
# sample_pdf given identical weights (oracle's) -> isolates the kernel
zs=RAY.sample_pdf((.5*(z0[:,1:]+z0[:,:-1])).to(dev), w0[:,1:-1].contiguous().to(dev), 128, det=True).cpu()
zr=O.sample_pdf(.5*(z0[:,1:]+z0[:,:-1]), w0[:,1:-1], 128, det=True)
d=(zs-zr).abs(); print('sample_pdf @identical weights: max %.3e frac>1e-5 %.5f'%(d.max(), (d>1e-5).float().mean()))
# raw2outputs S=1
rng=np.random.default_rng(7); N,S=5,1
raw=T(rng.standard_normal((N,S,4)).astype(np.float32)); z=T(np.sort(rng.uniform(2,6,(N,S)).astype(np.float32),-1)); dd=T(rng.standard_normal((N,3)).astype(np.float32))
r=O.raw2outputs(raw,z,dd,0.,True); gq=RAY.raw2outputs(raw.to(dev),z.to(dev),dd.to(dev),0,True)
print('S=1 raw', raw[:,0].tolist()); print('oracle rgb', r[0].tolist(), 'acc', r[2].tolist()); print('gpu rgb', gq[0].cpu().tolist(), 'acc', gq[2].cpu().tolist())
