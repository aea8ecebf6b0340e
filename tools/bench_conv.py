import sys, time, torch, ctypes as C
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raygun_amd import ops, _lib
sfx=os.environ.get('RG_LIB_SUFFIX')
if sfx: _lib.LIB_PATHS['bf16']=_lib.LIB_PATH.replace('.so', sfx+'.so')
dev=torch.device('cuda')
shapes=[ # cin,cout,in_spatial,k
 (3024,432,34,3),(2592,2592,19,3),(432,432,44,3),(432,2592,21,3),(504,72,60,3),(72,72,94,3),(432,432,32,3),(72,432,46,3),
 (12,12,194,3),(1,12,196,3),(84,12,108,3),(12,72,96,3)]
sel=[int(v) for v in sys.argv[1].split(',')] if len(sys.argv)>1 else None
if sel is not None: shapes=[shapes[i] for i in sel]
for cin,cout,s,k in shapes:
    zero=bool(os.environ.get('RG_ZERO_DATA'))
    x=torch.randn(1,cin,s,s,s,device=dev).to(torch.bfloat16); x=ops.to_cl(x)
    if zero: x.zero_()
    w=(torch.randn(cout,cin,k,k,k,device=dev)*0.05).requires_grad_(True); b=torch.zeros(cout,device=dev)
    xg=x.detach().requires_grad_(cin>1)
    if zero:
        with torch.no_grad(): w.zero_()
    y=ops.conv3d(xg,w,b,1,0,'relu'); g=torch.randn_like(y)
    if zero: g.zero_()
    for rep in range(2):
        ops.profile_start()
        y=ops.conv3d(xg,w,b,1,0,'relu')
        y.backward(g)
        pr=ops.profile_stop()
    out=[]
    for name,d in pr.items():
        out.append(f"{name}: {d['ms']:.3f} ms {d['flops']/d['ms']/1e9:.0f} TF")
    print(f"{cin}->{cout} @{s}: "+' | '.join(out), flush=True)
