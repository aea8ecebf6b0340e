import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from raygun_amd import parallel
from torch.profiler import profile, ProfilerActivity
dev = torch.device("cuda", 0)
step, info = bench.build_unet(dev, 196, 0)
sync = parallel.GradSync()
for _ in range(3): step(sync)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    step(sync); torch.cuda.synchronize()
rows = [e for e in prof.key_averages(group_by_input_shape=True) if e.key.startswith("aten::")]
rows.sort(key=lambda e: -e.device_time_total)
for e in rows[:14]:
    print(f"{e.device_time_total/1e3:8.3f} ms x{e.count:3d} {e.key:28s} {str(e.input_shapes)[:110]}")
