"""Host-side (Python / ctypes / autograd) cost of one train step: python3 tools/prof_host.py [unet|cyclegan]  (GPU box)"""
import sys, os, time, torch, cProfile, pstats, contextlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from raygun_amd import parallel
dev = torch.device("cuda")
with contextlib.redirect_stdout(sys.stderr):
    wl = sys.argv[1] if len(sys.argv) > 1 else "unet"
    step, info = (bench.build_unet(dev, 196, seed=0) if wl == "unet" else bench.build_cyclegan(dev, 128, seed=0))
sync = parallel.GradSync()
for _ in range(3): step(sync)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5): step(sync)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"enqueue {1e3*(t1-t0)/5:.2f} ms/step, wall {1e3*(t2-t0)/5:.2f} ms/step")
pr = cProfile.Profile(); pr.enable()
for _ in range(5): step(sync)
pr.disable(); torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(45)
