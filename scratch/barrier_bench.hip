// grid-barrier latency microbenchmark (scratch): hipcc --offload-arch=gfx950 -O3 scratch/barrier_bench.hip -o scratch/barrier_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
template <int MODE>
__device__ __forceinline__ void bar(unsigned* b, unsigned& target, unsigned nwg) {
  if (MODE == 0) __threadfence();
  __syncthreads();
  target += nwg;
  if (threadIdx.x == 0) {
    if (MODE != 0) __threadfence();
    atomicAdd(b, 1u);
    while (__hip_atomic_load(b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
    if (MODE != 0) __threadfence();
  }
  __syncthreads();
  if (MODE == 0) __threadfence();
}
// MODE 2: two-level: 8 groups
__device__ __forceinline__ void bar2(unsigned* b, unsigned& gen, unsigned nwg) {
  __syncthreads();
  gen += 1;
  if (threadIdx.x == 0) {
    __threadfence();
    const unsigned g = blockIdx.x & 7, gsize = (nwg - g + 7) / 8;
    if (atomicAdd(&b[16 * (1 + g)], 1u) == gen * gsize - 1) {
      if (atomicAdd(&b[16 * 9], 1u) == gen * 8 - 1) __hip_atomic_store(&b[0], gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    while (__hip_atomic_load(&b[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gen) __builtin_amdgcn_s_sleep(1);
    __threadfence();
  }
  __syncthreads();
}
template <int MODE>
__global__ __launch_bounds__(1024) void k(unsigned* b, int iters, unsigned* data) {
  unsigned t = 0;
  for (int i = 0; i < iters; ++i) {
    data[(blockIdx.x * blockDim.x + threadIdx.x)] = i;  // a store per thread so the fences have something to do
    if (MODE == 2) bar2(b, t, gridDim.x); else bar<MODE>(b, t, gridDim.x);
  }
}
int main() {
  unsigned* b; unsigned* data;
  hipMalloc(&b, 4096); hipMalloc(&data, 4 << 20);
  const int iters = 2000;
  for (int mode = 0; mode < 3; ++mode)
    for (int threads : {256, 1024})
      for (int nwg : {16, 32, 64, 128, 256}) {
        hipMemset(b, 0, 4096);
        void* args[] = {&b, (void*)&iters, &data};
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipError_t err = hipLaunchCooperativeKernel(mode == 0 ? (const void*)k<0> : mode == 1 ? (const void*)k<1> : (const void*)k<2>, dim3(nwg), dim3(threads), args, 0, 0);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("mode %d threads %4d wgs %3d: %.2f us/barrier (%s)\n", mode, threads, nwg, ms * 1e3 / iters, hipGetErrorString(err));
        fflush(stdout);
      }
  return 0;
}
