"""Which torch (non-library) kernels run inside one train step: python3 tools/prof_aten.py [unet|cyclegan]  (GPU box); index / scatter ops with their Python stack"""
import sys, os, torch, contextlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from raygun_amd import parallel
dev = torch.device("cuda")
with contextlib.redirect_stdout(sys.stderr):
    wl = sys.argv[1] if len(sys.argv) > 1 else "unet"
    step, info = bench.build_unet(dev, 196, seed=0) if wl == "unet" else bench.build_cyclegan(dev, 128, seed=0)
sync = parallel.GradSync()
for _ in range(2): step(sync)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    step(sync); torch.cuda.synchronize()
rows = [e for e in prof.key_averages(group_by_input_shape=True) if e.device_time_total > 15 and (e.key.startswith('aten::') or 'at::native' in e.key)]
rows.sort(key=lambda e: -e.device_time_total)
for e in rows[:25]:
    print(f"{e.key[:40]:40s} n={e.count:4d} dev={e.device_time_total/1e3:8.3f} ms  shapes={str(e.input_shapes)[:150]}")

for e in prof.events():
    if ("index" in e.name or "scatter" in e.name) and e.name.startswith("aten::") and e.stack:
        print(e.name, e.input_shapes, "|", " <- ".join(f.split("/")[-1] for f in e.stack[:6]))
