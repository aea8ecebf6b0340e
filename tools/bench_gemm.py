"""Microbenchmark of rg_gemm_batched (GPU box): python3 tools/bench_gemm.py"""
import sys, os, torch, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raygun_amd import _lib
L = _lib.lib(); dev = torch.device("cuda")
def run(B, m, k, n, reps=5):
    kp = (k + 63) // 64 * 64; nr = (n + 31) // 32 * 32; ms_ = (n + 63) // 64 * 64
    A = (torch.randn(B * m * kp, device=dev) * 0.5).to(torch.bfloat16)
    W = (torch.randn(B * nr * kp, device=dev) * 0.05).to(torch.bfloat16)
    Cm = torch.empty(B * m * ms_, dtype=torch.bfloat16, device=dev)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    def go():
        _lib.check(L.rg_gemm_batched(C.c_void_p(A.data_ptr()), C.c_void_p(W.data_ptr()), C.c_void_p(Cm.data_ptr()), B, m, k, n, kp, m * kp, nr * kp, ms_, m * ms_, st), "gemm")
    go(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): go()
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / reps
    print(f"B={B:3d} m={m:5d} k={k:6d} n={n:5d}: {t:.3f} ms  {2.0*B*m*k*n/t/1e9:7.0f} TF/s  {L.rg_last_kernel_name().decode()}", flush=True)
for cfg in [(64, 729, 2592, 2592), (64, 1000, 2592, 2592), (64, 2592, 768, 2592), (64, 2592, 729, 2592), (64, 2592, 1024, 2592)]:
    run(*cfg)
