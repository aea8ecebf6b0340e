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
//   3. mws_kruskal_kernel   — ONE wavefront walks the sorted edges 64 at a time with sequential semantics:
//                             every lane finds the roots of its edge (path halving); an edge may execute as soon as no
//                             EARLIER unexecuted edge of the batch shares a root with it (64-step broadcast scan), all
//                             such edges then run in parallel (they touch disjoint clusters), the rest re-evaluate.
//                             Mutex sets are per-root singly linked lists of (possibly stale) partner voxels, resolved
//                             lazily with find(); a union splices the two lists in O(1).
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
  int* parent; int* size; int* head; int* tail; int* cnt;  // per voxel (roots carry size and the mutex list)
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

__global__ __launch_bounds__(64) void mws_kruskal_kernel(const u64* keys, const unsigned* count, const int* offsets, MwsState s,
                                                        int Z, int Y, int X, int n_attr) {
  const int lane = threadIdx.x;
  const long N = (long)Z * Y * X;
  const unsigned ne = *count;
  for (unsigned base = 0; base < ne; base += 64) {
    const unsigned e = base + lane;
    bool pending = e < ne;
    int u = 0, v = 0; bool attractive = false;
    if (pending) {
      const unsigned id = (unsigned)keys[e];
      const int k = (int)(id / N); u = (int)(id % N);
      const int x = u % X, y = (u / X) % Y, z = u / (X * Y);
      v = ((z + offsets[3 * k]) * Y + (y + offsets[3 * k + 1])) * X + (x + offsets[3 * k + 2]);
      attractive = k < n_attr;
    }
    while (__any(pending)) {
      int ru = -1, rv = -2;
      if (pending) { ru = mws_find(s.parent, u); rv = mws_find(s.parent, v); if (ru == rv) pending = false; }
      // blocked = an earlier still-pending edge of this batch shares one of my roots
      bool blocked = false;
      const u64 pmask = __ballot(pending);
      for (int j = 0; j < 63; ++j) {
        if (!((pmask >> j) & 1)) continue;  // wave-uniform
        const int ju = __shfl(ru, j), jv = __shfl(rv, j);
        if (lane > j && pending && (ru == ju || ru == jv || rv == ju || rv == jv)) blocked = true;
      }
      if (pending && !blocked) {
        if (attractive) {
          if (!mws_has_mutex(s, ru, rv)) {
            int a = ru, b = rv;
            if (s.size[a] < s.size[b]) { a = rv; b = ru; }
            s.parent[b] = a; s.size[a] += s.size[b];
            if (s.head[b] >= 0) {  // splice b's mutex list onto a's
              if (s.head[a] >= 0) s.pool[s.tail[a]].y = s.head[b]; else s.head[a] = s.head[b];
              s.tail[a] = s.tail[b]; s.cnt[a] += s.cnt[b];
            }
          }
        } else {
          const int na = 2 * (int)e, nb = na + 1;  // two nodes per edge slot: no allocator needed
          s.pool[na] = make_int2(rv, s.head[ru]); if (s.head[ru] < 0) s.tail[ru] = na; s.head[ru] = na; s.cnt[ru] += 1;
          s.pool[nb] = make_int2(ru, s.head[rv]); if (s.head[rv] < 0) s.tail[rv] = nb; s.head[rv] = nb; s.cnt[rv] += 1;
        }
        pending = false;
      }
      __threadfence_block();  // this wave's stores are visible to its own later loads
    }
  }
}

__global__ void mws_init_kernel(MwsState s, int* first, unsigned* flags, long N) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (long)gridDim.x * blockDim.x) {
    s.parent[i] = (int)i; s.size[i] = 1; s.head[i] = -1; s.tail[i] = -1; s.cnt[i] = 0; first[i] = 0x7fffffff; flags[i] = 0;
  }
}
__global__ void mws_first_kernel(MwsState s, const unsigned char* mask, int* first, long N) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (long)gridDim.x * blockDim.x) {
    if (mask && !mask[i]) continue;
    int r = (int)i; while (s.parent[r] != r) r = s.parent[r];
    s.size[i] = r;  // size[] is dead after the Kruskal pass: reuse it as the root of every voxel
    atomicMin(&first[r], (int)i);
  }
}
__global__ void mws_flag_kernel(MwsState s, const unsigned char* mask, const int* first, unsigned* flags, long N) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (long)gridDim.x * blockDim.x)
    flags[i] = (!(mask && !mask[i]) && first[s.size[i]] == (int)i) ? 1u : 0u;
}
__global__ void mws_label_kernel(MwsState s, const unsigned char* mask, const int* first, const unsigned* rank, u64* labels, long N) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (long)gridDim.x * blockDim.x)
    labels[i] = (mask && !mask[i]) ? 0ull : (u64)rank[first[s.size[i]]] + 1ull;
}

struct MwsPlan { long N, ne_max; size_t off_keys2, off_cnt, off_state, off_pool, off_first, off_flags, off_rank, off_tmp, tmp_bytes, total; };
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
  o += al((size_t)q->ne_max * 2 * 8); q->off_first = o;
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
  S.parent = (int*)(w + q.off_state); S.size = (int*)(w + q.off_state + nb); S.head = (int*)(w + q.off_state + 2 * nb);
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
  hipLaunchKernelGGL(mws_kruskal_kernel, dim3(1), dim3(64), 0, st, (const u64*)keys2, (const unsigned*)count, offsets, S, Z, Y, X, p->n_attractive);
  RG_CHECK_LAUNCH();
  hipLaunchKernelGGL(mws_first_kernel, dim3(blocks), dim3(256), 0, st, S, mask, first, q.N);
  RG_CHECK_LAUNCH();
  hipLaunchKernelGGL(mws_flag_kernel, dim3(blocks), dim3(256), 0, st, S, mask, (const int*)first, flags, q.N);
  RG_CHECK_LAUNCH();
  tb = q.tmp_bytes;
  if (rocprim::exclusive_scan(tmp, tb, flags, rank, 0u, (size_t)q.N, rocprim::plus<unsigned>(), st) != hipSuccess) RG_FAIL(RG_ERR_HIP, "mws: scan failed");
  hipLaunchKernelGGL(mws_label_kernel, dim3(blocks), dim3(256), 0, st, S, mask, (const int*)first, (const unsigned*)rank, (u64*)labels, q.N);
  RG_CHECK_LAUNCH();
  return RG_OK;
}
