"""Per-step time of the U-Net train step from a cold start (GPU box): GPU time between per-step events, and the HOST's own time per
step (enqueue only) beside it — a slow GPU step with a normal host step is a device-side stall, one with a slow host step a host pause.
usage: python tools/step_trend.py [steps] [gc: default | freeze | off]"""
import sys, os, contextlib, gc, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from raygun_amd import parallel
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
mode = sys.argv[2] if len(sys.argv) > 2 else "default"
dev = torch.device("cuda")
with contextlib.redirect_stdout(sys.stderr):
    step, info = bench.build_unet(dev, 196, seed=0)
if mode == "freeze":
    gc.collect(); gc.freeze()
elif mode == "off":
    gc.disable()
sync = parallel.GradSync()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
host = [time.perf_counter()]
gcs = []
gc.callbacks.append(lambda phase, info: gcs.append((phase, info.get("generation"), time.perf_counter())))
ev[0].record()
for i in range(n):
    step(sync); ev[i + 1].record(); host.append(time.perf_counter())
torch.cuda.synchronize()
ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(n)]
print(f"gc {mode}: GPU per-step ms:", " ".join(f"{m:.1f}" for m in ms))
print("host per-step ms:", " ".join(f"{(host[i + 1] - host[i]) * 1e3:.1f}" for i in range(n)))
starts = {}
for phase, gen, t in gcs:
    if phase == "start":
        starts[gen] = t
    elif gen in starts and (t - starts[gen]) > 2e-3:
        print(f"  gc generation {gen}: {(t - starts[gen]) * 1e3:.1f} ms at host t={(t - host[0]) * 1e3:.0f} ms")
print("mean of steps 4-13: %.2f   last 10: %.2f" % (sum(ms[3:13]) / 10, sum(ms[-10:]) / 10))
