"""CycleGAN 128^3 step: how long the HOST needs to enqueue one step (up to the first host<-device read of a loss value) against the
step's wall time (GPU box).  usage: python tools/cg_host_probe.py [steps]"""
import builtins, contextlib, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import raygun_amd.torch.losses   # noqa: F401
M = sys.modules["raygun_amd.torch.losses.LinkCycleLoss"]   # (the package attribute of that name is the class)
from raygun_amd import parallel
marks = []


def _float(x):   # the module's float(): the four loss values read at the end of LinkCycleLoss.forward
    if torch.is_tensor(x) and x.is_cuda:
        marks.append(time.perf_counter())
    return builtins.float(x)


M.float = _float
n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
dev = torch.device("cuda")
with contextlib.redirect_stdout(sys.stderr):
    step, info = bench.build_cyclegan(dev, 128, seed=0)
sync = parallel.GradSync()
for _ in range(4): step(sync)
torch.cuda.synchronize()
enq, wall = [], []
for _ in range(n):
    del marks[:]
    t0 = time.perf_counter(); step(sync); torch.cuda.synchronize(); t1 = time.perf_counter()
    enq.append((marks[0] - t0) * 1e3); wall.append((t1 - t0) * 1e3)
print("host enqueue ms/step (to the first loss read):", " ".join(f"{e:.1f}" for e in enq))
print("wall ms/step:", " ".join(f"{w:.1f}" for w in wall))
