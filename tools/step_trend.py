"""Per-step wall time of the U-Net train step from a cold start (GPU box): how many steps the device / allocator need to settle.
usage: python tools/step_trend.py [steps]   -> one line of per-step ms (HIP events around each step, no host synchronisation inside)"""
import sys, os, contextlib, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from raygun_amd import parallel
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
dev = torch.device("cuda")
with contextlib.redirect_stdout(sys.stderr):
    step, info = bench.build_unet(dev, 196, seed=0)
sync = parallel.GradSync()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
ev[0].record()
for i in range(n):
    step(sync); ev[i + 1].record()
torch.cuda.synchronize()
ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(n)]
print("per-step ms:", " ".join(f"{m:.1f}" for m in ms))
print("mean of steps 4-13: %.2f   14-33: %.2f   last 20: %.2f" % (sum(ms[3:13]) / 10, sum(ms[13:33]) / 20, sum(ms[-20:]) / 20))
