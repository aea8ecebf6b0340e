"""Microbenchmark + check of rg_gemm_batched (GPU box): python3 tools/bench_gemm.py [check]
Every shape is run with the LDS-DMA GEMM (gemm5.hip) forced on and off, interleaved in ONE process (A/B rule), on random data."""
import sys, os, torch, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raygun_amd import _lib, ops
L = _lib.lib(); dev = torch.device("cuda")
def run(B, m, k, n, reps=5, check=False):
    kp = (k + 63) // 64 * 64; nr = (n + 31) // 32 * 32; ms_ = (n + 63) // 64 * 64
    A = torch.zeros(B, m, kp, device=dev, dtype=torch.bfloat16); A[:, :, :k] = (torch.randn(B, m, k, device=dev) * 0.5).to(torch.bfloat16)
    W = torch.zeros(B, nr, kp, device=dev, dtype=torch.bfloat16); W[:, :n, :k] = (torch.randn(B, n, k, device=dev) * 0.05).to(torch.bfloat16)
    Cm = torch.empty(B, m, ms_, dtype=torch.bfloat16, device=dev)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    def go():
        _lib.check(L.rg_gemm_batched(C.c_void_p(A.data_ptr()), C.c_void_p(W.data_ptr()), C.c_void_p(Cm.data_ptr()), B, m, k, n, kp, m * kp, nr * kp, ms_, m * ms_, st), "gemm")
    res = {}
    for rnd in range(3):
        for opt in (1, -1):
            with ops.option("RG_GEMM5", opt):
                go(); torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(reps): go()
                e1.record(); torch.cuda.synchronize()
                res.setdefault(opt, []).append((e0.elapsed_time(e1) / reps, L.rg_last_kernel_name().decode()))
                if check and rnd == 0:
                    ref = torch.matmul(A[:, :, :k].float(), W[:, :n, :k].float().transpose(1, 2))
                    got = Cm[:, :, :n].float()
                    err = float((got - ref).abs().max() / ref.abs().max())
                    print(f"   check opt={opt:2d} {res[opt][-1][1]}: max rel err {err:.2e}", flush=True)
    for opt in (1, -1):
        t = min(r[0] for r in res[opt]); tm = sorted(r[0] for r in res[opt])[1]
        print(f"B={B:3d} m={m:5d} k={k:6d} n={n:5d} opt={opt:2d}: min {t:.3f} ms median {tm:.3f} ms  {2.0*B*m*k*n/t/1e9:7.0f} TF/s  {res[opt][0][1]}", flush=True)
if len(sys.argv) > 2 and sys.argv[1] == "shape":      # python3 tools/bench_gemm.py shape B,m,k,n [B,m,k,n ...]
    for a in sys.argv[2:]:
        run(*[int(v) for v in a.split(",")])
    sys.exit(0)
chk = len(sys.argv) > 1 and sys.argv[1] == "check"
if chk:
    for cfg in [(2, 300, 200, 288), (3, 700, 2592, 576), (1, 257, 64, 144), (2, 1000, 448, 432)]:
        run(*cfg, reps=1, check=True)
for cfg in [(64, 729, 2592, 2592), (64, 1000, 2592, 2592), (1, 32768, 2592, 2592), (1, 27000, 432 * 9, 432), (1, 4096, 2592 * 8, 3456)]:
    run(*cfg)
