"""Long-run check of the train steps (GPU box): memory must stay flat and the step time stable over hundreds of steps."""
import sys, os, time, contextlib, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from raygun_amd import parallel
dev = torch.device("cuda")
for wl, size, n in (("unet", 196, 200), ("cyclegan", 128, 40)):
    with contextlib.redirect_stdout(sys.stderr):
        step, info = (bench.build_unet if wl == "unet" else bench.build_cyclegan)(dev, size, seed=0)
    sync = parallel.GradSync()
    for _ in range(3): step(sync)
    torch.cuda.synchronize(); torch.cuda.reset_peak_memory_stats()
    m0 = torch.cuda.memory_allocated(); t0 = time.perf_counter(); marks = []
    for i in range(n):
        loss = step(sync)
        if (i + 1) % (n // 4) == 0:
            torch.cuda.synchronize(); marks.append((i + 1, round((time.perf_counter() - t0) / (i + 1) * 1e3, 2), torch.cuda.memory_allocated() - m0))
    print(wl, "steps, ms/step so far, bytes allocated above the start:", marks, "peak GB", round(torch.cuda.max_memory_allocated() / 2**30, 2), "reserved GB", round(torch.cuda.memory_reserved() / 2**30, 2), flush=True)
    del step, sync
    import gc; gc.collect(); torch.cuda.empty_cache()
