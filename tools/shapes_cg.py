import sys, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from raygun_amd import ops
wl = sys.argv[1] if len(sys.argv) > 1 else "cyclegan"
dev = torch.device("cuda")
step, info = (bench.build_unet if wl == "unet" else bench.build_cyclegan)(dev, 196 if wl == "unet" else 128, seed=0)
from raygun_amd import parallel
sync = parallel.GradSync()
for _ in range(2): step(sync)
torch.cuda.synchronize()
ops.profile_shapes = True
ops.profile_start(); step(sync); pr = ops.profile_stop()
rows = sorted(pr.items(), key=lambda kv: -kv[1]["ms"])
tot = sum(v["ms"] for _, v in rows)
print("total conv ms", tot)
for k, v in rows[:60]:
    print(f"{v['ms']:8.3f} ms x{v['launches']:3d} {v['flops']/v['ms']/1e9:6.0f} TF  {k}")
