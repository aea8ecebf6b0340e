import sys, json, functools, numpy as np, torch, time
sys.path.insert(0, '.')
from raygun_amd import synth
from raygun_amd.torch import networks as N
from raygun_amd.torch.networks.UNet import ConvPass
from raygun_amd.torch.networks.ResNet import ResnetBlock3D
from raygun_amd.torch.models import MTLSDModel
from oracle import torch_ref as O
import copy
G='tests/golden/'
dev=torch.device('cuda')
def T(a): return torch.from_numpy(np.ascontiguousarray(a))
def rel(a,b):
    a=a.detach().float().cpu().numpy().astype(np.float64); b=np.asarray(b,dtype=np.float64)
    return float(np.sqrt(((a-b)**2).sum())/(np.sqrt((b**2).sum())+1e-30))
def run(name, mod, fx, omod=None):
    meta=json.loads(str(fx['meta'])); seed=meta['seed']; shp=tuple(meta['in_shape'])
    mod.load_state_dict(synth.fill_state_dict(mod.state_dict(), seed)); mod=mod.to(dev)
    x=T(synth.uniform(shp, seed+1)).to(dev).requires_grad_(True)
    y=mod(x); ys=y if isinstance(y,(list,tuple)) else [y]
    gs=[T(synth.uniform(tuple(t.shape), seed+2+i)).to(dev) for i,t in enumerate(ys)]
    torch.autograd.backward(ys, gs)
    out={f'y{i}': rel(t, fx[f'y{i}']) for i,t in enumerate(ys)}
    out['gx']=rel(x.grad, fx['gx'])
    for k in fx.files:
        if k.startswith('gw::'):
            out[k]=rel(dict(mod.named_parameters())[k[4:]].grad, fx[k])
    print(name, 'vs fp32 golden', {k: round(v,4) for k,v in out.items() if 'bias' not in k})
    if omod is not None:
        omod.load_state_dict(synth.fill_state_dict(omod.state_dict(), seed)); O.emulate_bf16(omod)
        xo=T(synth.uniform(shp, seed+1)).to(torch.bfloat16).float().requires_grad_(True)
        yo=omod(xo); yos=yo if isinstance(yo,(list,tuple)) else [yo]
        torch.autograd.backward(yos, [g.cpu().to(torch.bfloat16).float() for g in gs])
        o2={f'y{i}': rel(t, yos[i].detach().numpy()) for i,t in enumerate(ys)}
        o2['gx']=rel(x.grad, xo.grad.numpy())
        po=dict(omod.named_parameters())
        for k,p in mod.named_parameters():
            if 'bias' not in k: o2[k]=rel(p.grad, po[k].grad.numpy())
        print(name, 'vs bf16-emulating oracle', {k: round(v,4) for k,v in o2.items()})
IN3=functools.partial(torch.nn.InstanceNorm3d)
run('convpass_valid', ConvPass(3,5,[(3,3,3),(3,3,3)],'ReLU'), np.load(G+'convpass_valid.npz'), O.ConvPass(3,5,[(3,3,3),(3,3,3)],'ReLU'))
run('convpass_same', ConvPass(3,5,[(3,3,3),(3,3,3)],'ReLU',padding='same'), np.load(G+'convpass_same.npz'), O.ConvPass(3,5,[(3,3,3),(3,3,3)],'ReLU',padding='same'))
run('convpass_residual', ConvPass(2,4,[(3,3,3),(3,3,3)],'ReLU',residual=True), np.load(G+'convpass_residual.npz'))
run('convpass_sigmoid', ConvPass(12,3,[[1,1,1]],'Sigmoid'), np.load(G+'convpass_sigmoid_1x1.npz'))
fx=np.load(G+'unet_small.npz'); kw=json.loads(str(fx['meta']))['unet_kwargs']; kw['downsample_factors']=[tuple(f) for f in kw['downsample_factors']]
run('unet_small', N.UNet(**kw), fx, O.UNet(**kw))
fx=np.load(G+'unet_same_small.npz'); kw=json.loads(str(fx['meta']))['unet_kwargs']; kw['downsample_factors']=[tuple(f) for f in kw['downsample_factors']]
run('unet_same_small', N.UNet(**kw), fx, O.UNet(**kw))
for pt in ('reflect','valid','zeros'):
    run('resblock_'+pt, ResnetBlock3D(6,pt,IN3,False,True), np.load(G+f'resblock3d_{pt}.npz'), O.ResnetBlock(3,6,pt,IN3,False,True))
run('resnet3d_small', N.ResNet(3,input_nc=1,ngf=8,n_blocks=2,n_downsampling=2,norm_layer=IN3,padding_type='same'), np.load(G+'resnet3d_small.npz'), O.ResNet(3,input_nc=1,ngf=8,n_blocks=2,n_downsampling=2,norm_layer=IN3,padding_type='same'))
run('resnet3d_valid', N.ResNet(3,input_nc=1,ngf=4,n_blocks=1,n_downsampling=1,norm_layer=IN3,padding_type='valid'), np.load(G+'resnet3d_valid_small.npz'))
dn=functools.partial(torch.nn.InstanceNorm3d,affine=False,track_running_stats=False)
run('patchD', N.NLayerDiscriminator(3,input_nc=1,ngf=8,n_layers=3,norm_layer=dn), np.load(G+'patchD3d_small.npz'), O.NLayerDiscriminator(3,input_nc=1,ngf=8,n_layers=3,norm_layer=dn))
