import sys, json, numpy as np, torch
sys.path.insert(0,'.')
from raygun_amd import synth, ops
from raygun_amd.torch import networks as N
from oracle import torch_ref as O
dev=torch.device('cuda')
def T(a): return torch.from_numpy(np.ascontiguousarray(a))
def rel(a,b):
    a=a.detach().float().cpu().numpy().astype(np.float64); b=b.detach().float().numpy().astype(np.float64)
    return float(np.sqrt(((a-b)**2).sum())/(np.sqrt((b**2).sum())+1e-30)), float(np.abs(a-b).max())
kw=dict(input_nc=1, ngf=4, fmap_inc_factor=2, downsample_factors=[(2,2,2)]*3, constant_upsample=True)
m=N.UNet(**kw); o=O.UNet(**kw)
sd=synth.fill_state_dict(m.state_dict(),41); m.load_state_dict(sd); o.load_state_dict(sd); m=m.to(dev); O.emulate_bf16(o)
caps={}
def hook(store,name):
    def f(mod,inp,out): store[name]=out.detach()
    return f
hm,ho={}, {}
for name,mod in m.named_modules():
    if name and name.count('.')<=2 and not name.endswith('conv_pass'): mod.register_forward_hook(hook(hm,name))
for name,mod in o.named_modules():
    if name and name.count('.')<=2 and not name.endswith('conv_pass'): mod.register_forward_hook(hook(ho,name))
x=T(synth.uniform((1,1,100,100,100),42))
ym=m(x.to(dev)); yo=o(x.to(torch.bfloat16).float())
for k in hm:
    if k in ho and torch.is_tensor(hm[k]) and torch.is_tensor(ho[k]) and hm[k].shape==ho[k].shape:
        print(k, tuple(hm[k].shape), rel(hm[k],ho[k]))
