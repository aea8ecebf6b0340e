import sys, numpy as np, torch, torch.nn.functional as F
sys.path.insert(0,'.')
from raygun_amd import synth, ops
dev=torch.device('cuda')
def T(a): return torch.from_numpy(np.ascontiguousarray(a))
def bf(t): return t.to(torch.bfloat16).float()
def rel(a,b):
    a=a.detach().float().cpu(); b=b.detach().float().cpu(); return float((a-b).abs().max()/(b.abs().max()+1e-30))
x=bf(T(synth.uniform((1,3,10,10,10),1))); w1=bf(T(synth.normal((5,3,3,3,3),2,0.15))); b1=bf(T(synth.uniform((5,),3,-.05,.05)))
w2=bf(T(synth.normal((5,5,3,3,3),4,0.12))); b2=bf(T(synth.uniform((5,),5,-.05,.05))); g=bf(T(synth.uniform((1,5,10,10,10),6)))
# reference
xr=x.clone().requires_grad_(True)
p1=F.pad(xr,(1,)*6,mode='reflect'); p1.retain_grad(); y1=F.relu(F.conv3d(p1,w1,b1)); y1.retain_grad()
p2=F.pad(y1,(1,)*6,mode='reflect'); p2.retain_grad(); y2=F.relu(F.conv3d(p2,w2,b2)); y2.backward(g)
# hip
xh=x.clone().to(dev).requires_grad_(True)
q1=ops.pad3d(xh,1,'reflect'); q1.retain_grad(); z1=ops.conv3d(q1,w1.to(dev),b1.to(dev),1,0,'relu'); z1.retain_grad()
q2=ops.pad3d(z1,1,'reflect'); q2.retain_grad(); z2=ops.conv3d(q2,w2.to(dev),b2.to(dev),1,0,'relu'); z2.backward(g.to(dev).to(z2.dtype))
for n,a,b in [('y2',z2,y2),('y1',z1,y1),('g_p2',q2.grad,p2.grad),('g_y1',z1.grad,y1.grad),('g_p1',q1.grad,p1.grad),('g_x',xh.grad,xr.grad)]:
    print(n, rel(a,b))
