"""Per-shape conv timing of the UNet 196^3 train step (tools)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from raygun_amd import ops, parallel
dev = torch.device("cuda", 0)
step, info = bench.build_unet(dev, 196, 0)
sync = parallel.GradSync()
for _ in range(3): step(sync)
torch.cuda.synchronize()
ops.profile_shapes = True
ops.profile_start()
for _ in range(5): step(sync)
prof = ops.profile_stop()
tot = 0
for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"])[:80]:
    ms = v["ms"] / 5; tot += ms
    print(f"{ms:7.3f} ms {v['flops']/5/1e12:6.3f} TF {v['flops']/(v['ms']*1e-3)/1e12:7.1f} TF/s  x{v['launches']//5}  {k}")
print("sum", tot)
