"""File-system write rate of the bench's scratch directory (GPU box): 128 files of 1.9 MB, as the zarr predict flow writes per volume.  python tools/fs_probe.py"""
import os, tempfile, time, threading
d = tempfile.mkdtemp(prefix="raygun_amd_fsprobe_")
buf = os.urandom(1900000)
for threads in (1, 4, 16):
    t0 = time.perf_counter()
    def work(k):
        for i in range(k, 128, threads):
            with open(os.path.join(d, f"c{threads}_{i}"), "wb") as f:
                f.write(buf)
    ts = [threading.Thread(target=work, args=(k,)) for k in range(threads)]
    [t.start() for t in ts]; [t.join() for t in ts]
    dt = time.perf_counter() - t0
    print(f"{threads:2d} writer threads: 128 x 1.9 MB in {dt * 1e3:.0f} ms = {128 * 1.9 / dt:.0f} MB/s")
for f in os.listdir(d): os.remove(os.path.join(d, f))
os.rmdir(d)
