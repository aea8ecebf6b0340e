// Mutex watershed on gfx950 — replaces affogato.segmentation.compute_mws_segmentation as called by
// raygun.segment.mutex_segment (reference: src/raygun/segment.py:197-199).
//
// Semantics = the build-defined restatement in oracle/mws.c: Kruskal over all grid-graph edges in
// DESCENDING weight, ties by ASCENDING edge id (k*N + u); attractive edges merge two clusters unless a mutex
// separates them, repulsive edges (kept only on the stride grid) add a mutex.  The result is the unique partition
// that order defines, so labels are bit-exact against the oracle.
//
// Pipeline (all on the caller's stream, caller-owned workspace):
//   1. mws_edges_kernel     — HBM-bound pass over the (n_offsets, Z, Y, X) weights: every valid edge becomes one
//                             64-bit key  [~orderable(weight) : 32][edge id : 32]  appended with a wave-aggregated atomic.
//   2. rocprim::radix_sort_keys — one device-wide ascending sort (= descending weight, ascending id).
//   3. mws_rounds_kernel    — persistent grid (cooperative launch, hand-made grid barrier) that walks a window over the
//                             sorted edges in rounds and executes, per round, every edge whose outcome the sequential
//                             order already fixes (see the kernel).  Mutex sets are per-root singly linked lists of
//                             (possibly stale) partner voxels, resolved lazily with find(); a union splices two lists in O(1).
//   4. mws_label kernels    — first voxel of every cluster in raster order (atomicMin), exclusive scan of the "is
//                             first" flags, labels = 1 + rank; masked-out voxels 0.
#include <cstring>
#include <string.h>
#include "common.h"
#include <rocprim/rocprim.hpp>

typedef unsigned long long u64;

__device__ __forceinline__ unsigned order_desc(float w) {
  unsigned b = __float_as_uint(w);
  b = (b & 0x80000000u) ? ~b : (b | 0x80000000u);  // ascending-orderable
  return ~b;                                        // descending weight first
}

__global__ void mws_edges_kernel(const float* affs, const int* offsets, const unsigned char* mask, u64* keys, unsigned* count,
                                 int Z, int Y, int X, int n_off, int n_attr, int sz, int sy, int sx) {
  const long N = (long)Z * Y * X, total = N * n_off;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int k = (int)(i / N); const long u = i % N;
    const int x = (int)(u % X), y = (int)((u / X) % Y), z = (int)(u / ((long)X * Y));
    const int z2 = z + offsets[3 * k], y2 = y + offsets[3 * k + 1], x2 = x + offsets[3 * k + 2];
    bool ok = (unsigned)z2 < (unsigned)Z && (unsigned)y2 < (unsigned)Y && (unsigned)x2 < (unsigned)X;
    if (k >= n_attr) ok = ok && (z % sz == 0) && (y % sy == 0) && (x % sx == 0);
    if (ok && mask) ok = mask[u] && mask[((long)z2 * Y + y2) * X + x2];
    if (ok) {
      const unsigned pos = atomicAdd(count, 1u);
      keys[pos] = ((u64)order_desc(affs[i]) << 32) | (u64)(unsigned)i;
    }
  }
}

struct MwsState {
  int* parent; int* root; int* head; int* tail; int* cnt;  // per voxel (roots carry the mutex list; root[] is scratch for labelling)
  int2* pool;                                              // list nodes: (partner voxel, next node)
};

__device__ __forceinline__ int mws_find(int* parent, int x) {
  int p = parent[x];
  while (p != x) { const int g = parent[p]; parent[x] = g; x = g; p = parent[x]; }  // path halving (races are benign)
  return x;
}
__device__ bool mws_has_mutex(const MwsState& s, int ra, int rb) {
  if (s.cnt[ra] > s.cnt[rb]) { const int t = ra; ra = rb; rb = t; }
  for (int n = s.head[ra]; n >= 0; n = s.pool[n].y) if (mws_find(s.parent, s.pool[n].x) == rb) return true;
  return false;
}

// ---- round-based Kruskal: the whole chip walks a WINDOW over the sorted edge list -----------------------------------------
// A window entry is a pending edge: its slot in the sorted list (= its rank in the order), the last known roots of its two
// ends, and the mutex-list lengths seen at its last mutex check.  Every round (two grid barriers):
//   A  find both roots.  Same root -> retired (clusters only grow).  Attractive edge between two clusters that already hold a
//      mutex -> retired (mutexes are never removed).  Otherwise claim both roots: claim[r] = earliest pending edge at r.
//   B  an edge that is the earliest pending edge at BOTH its roots executes exactly as in the sequential order (nothing earlier
//      can touch either cluster any more).  An attractive edge that is the earliest at a root whose cluster holds NO mutex
//      executes too, whatever waits at the other end: until its turn that cluster stays untouched and mutex-free, so the
//      sequential algorithm would merge it then, and merging it early changes nothing an earlier edge can observe (it
//      contributes no mutex and no earlier edge ends in it) — the Boruvka step.  Everything else stays pending.
// The earliest pending edge always executes, so every round makes progress; the partition is the sequential one.
struct MwsCtl { unsigned bar, surv[2], pool_used, stalled, rounds, pad[2]; };
#define MWS_RETIRED 0xffffffffu
#define MWS_ATTR 0x80000000u
#define MWS_THREADS 1024

// Grid barrier of the persistent kernel.  One thread per workgroup does the device-scope release / acquire (the workgroup
// barriers around it order the other waves' accesses with it); measured on MI355X: 2.2 us at 32 workgroups, 3.7 us at 64,
// 8.5 us at 128 — and 62 us at 128 when every wave fences, so the grid stays small: rounds are latency-bound.
__device__ int g_mws_fence_all;
__device__ __forceinline__ void mws_grid_barrier(unsigned* bar, unsigned& target, unsigned nwg) {
  if (g_mws_fence_all) __threadfence();
  __syncthreads();
  target += nwg;
  if (threadIdx.x == 0) {
    __threadfence();
    atomicAdd(bar, 1u);
    while (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
    __threadfence();
  }
  __syncthreads();
  if (g_mws_fence_all) __threadfence();
}

__global__ __launch_bounds__(MWS_THREADS) void mws_rounds_kernel(const u64* keys, const unsigned* count, const int* offsets, MwsState s,
                                                                u64* claim, uint4* win0, uint4* win1, MwsCtl* ctl, int Z, int Y, int X,
                                                                int n_off, int n_attr, unsigned cap, unsigned pool_cap) {
  __shared__ int lin[64];
  if (threadIdx.x < n_off && threadIdx.x < 64)
    lin[threadIdx.x] = (offsets[3 * threadIdx.x] * Y + offsets[3 * threadIdx.x + 1]) * X + offsets[3 * threadIdx.x + 2];
  __syncthreads();
  const unsigned T = gridDim.x * blockDim.x, tid = blockIdx.x * blockDim.x + threadIdx.x, lane = threadIdx.x & 63;
  const unsigned N = (unsigned)((long)Z * Y * X);
  const unsigned ne = *count;
  unsigned bar_target = 0, next = 0, nsurv = 0;
  uint4* win = win0; uint4* wout = win1;
  for (unsigned round = 1;; ++round) {
    // fresh edges this round: top the window up to ~2 entries per thread (the survivors already hold the earliest pending
    // edge, so progress never depends on new ones; a window full of blocked edges is re-examined every round, so keep it small)
    unsigned want = nsurv < 2 * T ? 2 * T - nsurv : 0;
    if (want > cap - nsurv) want = cap - nsurv;
    if (want > ne - next) want = ne - next;
    const unsigned n = nsurv + want;
    if (n == 0) break;
    const u64 tag = (u64)round << 32;
    // ---- phase A
    for (unsigned i = tid; i < n; i += T) {
      uint4 e;
      if (i < nsurv) e = win[i];
      else {
        const unsigned slot = next + (i - nsurv);
        const unsigned id = (unsigned)keys[slot];
        const unsigned k = id / N, u = id - k * N;
        e = make_uint4(slot, u, (unsigned)((int)u + lin[k]), (k < (unsigned)n_attr ? MWS_ATTR : 0u) | 0x7fffffffu);
      }
      const int a = mws_find(s.parent, (int)e.y), b = mws_find(s.parent, (int)e.z);
      bool retired = a == b;
      if (!retired && (e.w & MWS_ATTR)) {
        const int ca = s.cnt[a], cb = s.cnt[b];
        if (ca && cb) {
          const unsigned sig = MWS_ATTR | (unsigned)(ca + cb);
          if ((int)e.y != a || (int)e.z != b || e.w != sig) retired = mws_has_mutex(s, a, b);
          e.w = sig;
        }
      }
      if (!retired) {
        const u64 mine = tag | (u64)(0xffffffffu - e.x);
        atomicMax(&claim[a], mine); atomicMax(&claim[b], mine);
      }
      e.y = (unsigned)a; e.z = (unsigned)b;
      if (retired) e.x = MWS_RETIRED;
      win[i] = e;
    }
    mws_grid_barrier(&ctl->bar, bar_target, gridDim.x);
    // ---- phase B
    unsigned* surv = &ctl->surv[round & 1];
    for (unsigned i0 = tid - lane; i0 < n; i0 += T) {  // whole waves iterate together (ballots below)
      const unsigned i = i0 + lane;
      uint4 e = make_uint4(MWS_RETIRED, 0, 0, 0);
      if (i < n) e = win[i];
      bool keep = false;
      if (e.x != MWS_RETIRED) {
        const int a = (int)e.y, b = (int)e.z;
        const u64 mine = tag | (u64)(0xffffffffu - e.x);
        const bool fa = claim[a] == mine, fb = claim[b] == mine;
        if (e.w & MWS_ATTR) {
          if (fa && fb) {
            s.parent[b] = a;
            if (s.head[b] >= 0) {  // splice b's mutex list onto a's
              if (s.head[a] >= 0) s.pool[s.tail[a]].y = s.head[b]; else s.head[a] = s.head[b];
              s.tail[a] = s.tail[b]; s.cnt[a] += s.cnt[b];
            }
          } else if (fb && s.cnt[b] == 0) s.parent[b] = a;
          else if (fa && s.cnt[a] == 0) s.parent[a] = b;
          else keep = true;
        } else if (fa && fb) {
          const unsigned na = atomicAdd(&ctl->pool_used, 2u), nb = na + 1;
          if (nb < pool_cap) {
            s.pool[na] = make_int2(b, s.head[a]); if (s.head[a] < 0) s.tail[a] = (int)na; s.head[a] = (int)na; s.cnt[a] += 1;
            s.pool[nb] = make_int2(a, s.head[b]); if (s.head[b] < 0) s.tail[b] = (int)nb; s.head[b] = (int)nb; s.cnt[b] += 1;
          }
        } else keep = true;
      }
      const u64 km = __ballot(keep);
      if (km) {
        unsigned base = 0;
        if (lane == 0) base = atomicAdd(surv, (unsigned)__popcll(km));
        base = __shfl(base, 0);
        if (keep) wout[base + __popcll(km & ((1ull << lane) - 1ull))] = e;
      }
    }
    mws_grid_barrier(&ctl->bar, bar_target, gridDim.x);
    const unsigned ns2 = __hip_atomic_load(surv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    next += want;
    if (tid == 0) { ctl->surv[(round + 1) & 1] = 0; ctl->rounds = round; }  // the other counter: everyone read it before the last barrier
    if (ns2 == n || round > (1u << 23)) {  // no edge retired although the earliest one always can: never expected; leave instead of spinning
      if (tid == 0) ctl->stalled = 1;
      break;
    }
    nsurv = ns2;
    uint4* t = win; win = wout; wout = t;
  }
}

__global__ void mws_init_kernel(MwsState s, int* first, unsigned* flags, long N) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (long)gridDim.x * blockDim.x) {
    s.parent[i] = (int)i; s.root[i] = (int)i; s.head[i] = -1; s.tail[i] = -1; s.cnt[i] = 0; first[i] = 0x7fffffff; flags[i] = 0;
  }
}
__global__ void mws_first_kernel(MwsState s, const unsigned char* mask, int* first, long N) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (long)gridDim.x * blockDim.x) {
    if (mask && !mask[i]) continue;
    const int r = mws_find(s.parent, (int)i);
    s.root[i] = r;
    atomicMin(&first[r], (int)i);
  }
}
__global__ void mws_flag_kernel(MwsState s, const unsigned char* mask, const int* first, unsigned* flags, long N) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (long)gridDim.x * blockDim.x)
    flags[i] = (!(mask && !mask[i]) && first[s.root[i]] == (int)i) ? 1u : 0u;
}
__global__ void mws_label_kernel(MwsState s, const unsigned char* mask, const int* first, const unsigned* rank, u64* labels, long N,
                                 const MwsCtl* ctl) {
  const bool bad = ctl->stalled != 0;  // never expected; all-ones labels make the caller fail loudly
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (long)gridDim.x * blockDim.x)
    labels[i] = bad ? ~0ull : (mask && !mask[i]) ? 0ull : (u64)rank[first[s.root[i]]] + 1ull;
}

#define MWS_WIN_CAP (1u << 20)
#define MWS_MAX_WGS 32
struct MwsPlan { long N, ne_max, pool_nodes; size_t off_claim, off_win, off_keys2, off_cnt, off_state, off_pool, off_first, off_flags, off_rank, off_tmp, tmp_bytes, total; };
static inline size_t al(size_t v) { return (v + 255) / 256 * 256; }
static int mws_plan(const rg_mws_params* p, MwsPlan* q) {
  if (!p || p->shape[0] <= 0 || p->shape[1] <= 0 || p->shape[2] <= 0 || p->n_offsets <= 0 || p->n_attractive < 0 ||
      p->n_attractive > p->n_offsets || p->strides[0] <= 0 || p->strides[1] <= 0 || p->strides[2] <= 0 || p->randomize_strides)
    return -1;
  q->N = (long)p->shape[0] * p->shape[1] * p->shape[2];
  if ((double)q->N * p->n_offsets >= 4294967295.0 || q->N >= 0x7fffffff) return -2;  // 32-bit edge ids
  const long grid = (long)((p->shape[0] + p->strides[0] - 1) / p->strides[0]) * ((p->shape[1] + p->strides[1] - 1) / p->strides[1]) *
                    ((p->shape[2] + p->strides[2] - 1) / p->strides[2]);
  q->ne_max = q->N * p->n_attractive + grid * (p->n_offsets - p->n_attractive);
  if (q->ne_max < 1) q->ne_max = 1;
  size_t tb = 0, tb2 = 0;
  rocprim::radix_sort_keys(nullptr, tb, (u64*)nullptr, (u64*)nullptr, (size_t)q->ne_max, 0, 64, (hipStream_t)0);
  rocprim::exclusive_scan(nullptr, tb2, (unsigned*)nullptr, (unsigned*)nullptr, 0u, (size_t)q->N, rocprim::plus<unsigned>(), (hipStream_t)0);
  q->tmp_bytes = tb > tb2 ? tb : tb2;
  size_t o = 0;
  o += al((size_t)q->ne_max * 8); q->off_keys2 = o;
  o += al((size_t)q->ne_max * 8); q->off_cnt = o;
  o += 256; q->off_state = o;
  o += al((size_t)q->N * 4) * 5; q->off_pool = o;
  q->pool_nodes = 2 * grid * (p->n_offsets - p->n_attractive) + 2;
  o += al((size_t)q->pool_nodes * 8); q->off_claim = o;
  o += al((size_t)q->N * 8); q->off_win = o;
  o += 2 * al((size_t)MWS_WIN_CAP * 16); q->off_first = o;
  o += al((size_t)q->N * 4); q->off_flags = o;
  o += al((size_t)q->N * 4); q->off_rank = o;
  o += al((size_t)q->N * 4); q->off_tmp = o;
  o += al(q->tmp_bytes); q->total = o;
  return 0;
}

extern "C" int64_t rg_mws_workspace_bytes(const rg_mws_params* p) {
  MwsPlan q;
  if (mws_plan(p, &q)) return -1;
  return (int64_t)q.total;
}

extern "C" int rg_mws_segment(const float* affs, const int32_t* offsets, const uint8_t* mask, uint64_t* labels,
                              const rg_mws_params* p, void* workspace, size_t ws_bytes, rg_stream s) {
  MwsPlan q;
  const int rc = mws_plan(p, &q);
  if (rc == -2) RG_FAIL(RG_ERR_UNSUPPORTED, "mws: volume too large for one segment (n_offsets*N must be < 2^32): split into blocks");
  if (rc) RG_FAIL(RG_ERR_BAD_SHAPE, "mws: bad parameters (randomize_strides is not supported)");
  if (!affs || !offsets || !labels) RG_FAIL(RG_ERR_BAD_SHAPE, "mws: null pointer");
  if (!workspace || ws_bytes < q.total) RG_FAIL(RG_ERR_WORKSPACE, "mws: workspace %zu < %zu bytes", ws_bytes, q.total);
  hipStream_t st = (hipStream_t)s;
  char* w = (char*)workspace;
  u64* keys = (u64*)w; u64* keys2 = (u64*)(w + q.off_keys2);
  unsigned* count = (unsigned*)(w + q.off_cnt);
  MwsState S;
  const size_t nb = al((size_t)q.N * 4);
  S.parent = (int*)(w + q.off_state); S.root = (int*)(w + q.off_state + nb); S.head = (int*)(w + q.off_state + 2 * nb);
  S.tail = (int*)(w + q.off_state + 3 * nb); S.cnt = (int*)(w + q.off_state + 4 * nb);
  S.pool = (int2*)(w + q.off_pool);
  int* first = (int*)(w + q.off_first); unsigned* flags = (unsigned*)(w + q.off_flags); unsigned* rank = (unsigned*)(w + q.off_rank);
  void* tmp = w + q.off_tmp;
  const int Z = p->shape[0], Y = p->shape[1], X = p->shape[2];
  int blocks = (int)((q.N + 255) / 256); if (blocks > 8192) blocks = 8192;
  if (hipMemsetAsync(keys, 0xff, (size_t)q.ne_max * 8, st) != hipSuccess || hipMemsetAsync(count, 0, 256, st) != hipSuccess)
    RG_FAIL(RG_ERR_HIP, "mws: memset failed");
  hipLaunchKernelGGL(mws_init_kernel, dim3(blocks), dim3(256), 0, st, S, first, flags, q.N);
  RG_CHECK_LAUNCH();
  int eblocks = (int)((q.N * p->n_offsets + 255) / 256); if (eblocks > 16384) eblocks = 16384;
  hipLaunchKernelGGL(mws_edges_kernel, dim3(eblocks), dim3(256), 0, st, affs, offsets, mask, keys, count, Z, Y, X, p->n_offsets,
                     p->n_attractive, p->strides[0], p->strides[1], p->strides[2]);
  RG_CHECK_LAUNCH();
  size_t tb = q.tmp_bytes;
  if (rocprim::radix_sort_keys(tmp, tb, keys, keys2, (size_t)q.ne_max, 0, 64, st) != hipSuccess) RG_FAIL(RG_ERR_HIP, "mws: radix sort failed");
  {
    // persistent grid with hand-made grid barriers: every workgroup must be resident, hence the cooperative launch
    static int wgs = 0;
    if (!wgs) {
      int dev = 0, cus = 0, per_cu = 0;
      if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
          hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, mws_rounds_kernel, MWS_THREADS, 0) != hipSuccess || cus * per_cu < 1)
        RG_FAIL(RG_ERR_HIP, "mws: cannot size the persistent grid");
      wgs = cus * per_cu < MWS_MAX_WGS ? cus * per_cu : MWS_MAX_WGS;
      if (const char* e = getenv("RG_MWS_WGS")) { const int v = atoi(e); if (v >= 1 && v <= cus * per_cu) wgs = v; }
    }
    { int fa = getenv("RG_MWS_FENCE_ALL") ? 1 : 0; hipMemcpyToSymbolAsync(HIP_SYMBOL(g_mws_fence_all), &fa, 4, 0, hipMemcpyHostToDevice, st); hipStreamSynchronize(st); }
    const u64* kp = keys2; const unsigned* cp = count; const int* op = offsets;
    u64* claim = (u64*)(w + q.off_claim);
    uint4* win0 = (uint4*)(w + q.off_win); uint4* win1 = (uint4*)(w + q.off_win + al((size_t)MWS_WIN_CAP * 16));
    MwsCtl* ctl = (MwsCtl*)(w + q.off_cnt + 64);
    int zz = Z, yy = Y, xx = X, no = p->n_offsets, na = p->n_attractive;
    unsigned cap = MWS_WIN_CAP, pool_cap = (unsigned)q.pool_nodes;
    if (p->n_offsets > 64) RG_FAIL(RG_ERR_UNSUPPORTED, "mws: at most 64 offsets");
    if (hipMemsetAsync(claim, 0, (size_t)q.N * 8, st) != hipSuccess) RG_FAIL(RG_ERR_HIP, "mws: memset failed");
    void* args[] = {&kp, &cp, &op, &S, &claim, &win0, &win1, &ctl, &zz, &yy, &xx, &no, &na, &cap, &pool_cap};
    if (hipLaunchCooperativeKernel((const void*)mws_rounds_kernel, dim3(wgs), dim3(MWS_THREADS), args, 0, st) != hipSuccess)
      RG_FAIL(RG_ERR_HIP, "mws: cooperative launch failed: %s", hipGetErrorString(hipGetLastError()));
    if (getenv("RG_MWS_STATS")) {
      MwsCtl h; unsigned ne = 0;
      hipStreamSynchronize(st);
      hipMemcpy(&h, ctl, sizeof(h), hipMemcpyDeviceToHost); hipMemcpy(&ne, count, 4, hipMemcpyDeviceToHost);
      fprintf(stderr, "[mws] N=%ld edges=%u rounds=%u pool=%u stalled=%u wgs=%d\n", q.N, ne, h.rounds, h.pool_used, h.stalled, wgs);
    }
  }
  hipLaunchKernelGGL(mws_first_kernel, dim3(blocks), dim3(256), 0, st, S, mask, first, q.N);
  RG_CHECK_LAUNCH();
  hipLaunchKernelGGL(mws_flag_kernel, dim3(blocks), dim3(256), 0, st, S, mask, (const int*)first, flags, q.N);
  RG_CHECK_LAUNCH();
  tb = q.tmp_bytes;
  if (rocprim::exclusive_scan(tmp, tb, flags, rank, 0u, (size_t)q.N, rocprim::plus<unsigned>(), st) != hipSuccess) RG_FAIL(RG_ERR_HIP, "mws: scan failed");
  hipLaunchKernelGGL(mws_label_kernel, dim3(blocks), dim3(256), 0, st, S, mask, (const int*)first, (const unsigned*)rank, (u64*)labels, q.N,
                     (const MwsCtl*)(w + q.off_cnt + 64));
  RG_CHECK_LAUNCH();
  return RG_OK;
}
