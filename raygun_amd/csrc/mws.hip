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

// One thread per voxel: the (z, y, x) decode and the stride-grid test happen once, only the channels that can hold an edge
// are read (3 attractive ones for 999 voxels in 1000 at stride 10), and a 1024-voxel workgroup reserves its keys with ONE
// atomic (one per wavefront was the bottleneck: 262 144 serialised atomics on one counter at 256^3).
// Where an edge weight comes from: the float32 weight array of the reference pipeline (scale == 0: segment.py:188-194 ran before), or the block
// runner's uint8 affinities with that preprocessing applied on the fly, in the same float32 roundings — a = u8 * scale, w = 1 - a, and the first
// `prep_sep` channels flipped back as w * -1 + 1 — so that the 4-byte-per-value weight array (87 GB for config 5's 2048^3 / 8 part) never exists.
// `cstride`: elements between channels (a Z-slab of a larger volume is a view, not a copy).
struct MwsSrc { const void* p; long cstride; float scale; int prep_sep; };
__device__ __forceinline__ float mws_src(const MwsSrc& a, int k, long u) {
  if (a.scale == 0.f) return ((const float*)a.p)[(long)k * a.cstride + u];
  const float v = __fmul_rn((float)((const unsigned char*)a.p)[(long)k * a.cstride + u], a.scale);
  float w = __fsub_rn(1.f, v);
  if (k < a.prep_sep) w = __fadd_rn(__fmul_rn(w, -1.f), 1.f);
  return w;
}
#define MWS_EDGE_THREADS 1024
__global__ __launch_bounds__(MWS_EDGE_THREADS) void mws_edges_kernel(MwsSrc affs, const int* offsets, const unsigned char* mask, u64* keys,
                                                                    unsigned* count, int Z, int Y, int X, int n_off, int n_attr, int sz, int sy, int sx,
                                                                    int oz, int oy, int ox, int invert, float bias) {
  __shared__ unsigned wave_base[MWS_EDGE_THREADS / 64];
  const unsigned N = (unsigned)((long)Z * Y * X);
  const unsigned u = blockIdx.x * blockDim.x + threadIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  u64 valid = 0;
  if (u < N && (!mask || mask[u])) {
    const int x = (int)(u % (unsigned)X), y = (int)((u / (unsigned)X) % (unsigned)Y), z = (int)(u / ((unsigned)X * (unsigned)Y));
    const int kmax = ((z + oz) % sz == 0 && (y + oy) % sy == 0 && (x + ox) % sx == 0) ? n_off : n_attr;
    for (int k = 0; k < kmax; ++k) {
      const int z2 = z + offsets[3 * k], y2 = y + offsets[3 * k + 1], x2 = x + offsets[3 * k + 2];
      if ((unsigned)z2 < (unsigned)Z && (unsigned)y2 < (unsigned)Y && (unsigned)x2 < (unsigned)X &&
          (!mask || mask[((long)z2 * Y + y2) * X + x2]))
        valid |= 1ull << k;
    }
  }
  const int nv = __popcll(valid);
  int incl = nv;
  for (int d = 1; d < 64; d <<= 1) { const int t = __shfl_up(incl, d); if ((int)lane >= d) incl += t; }
  if (lane == 63) wave_base[wave] = (unsigned)incl;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned total = 0;
    for (int w = 0; w < MWS_EDGE_THREADS / 64; ++w) { const unsigned t = wave_base[w]; wave_base[w] = total; total += t; }
    const unsigned base = total ? atomicAdd(count, total) : 0u;
    for (int w = 0; w < MWS_EDGE_THREADS / 64; ++w) wave_base[w] += base;
  }
  __syncthreads();
  unsigned pos = wave_base[wave] + (unsigned)(incl - nv);
  while (valid) {
    const int k = __builtin_ctzll(valid); valid &= valid - 1;
    const unsigned id = (unsigned)k * N + u;
    float w = mws_src(affs, k, (long)u);  // affogato's wrapper, in float32: w[n_attr:] = w[n_attr:] * -1 + 1 (invert_repulsive_weights), w[:n_attr] += bias_cut
    if (k >= n_attr) { if (invert) w = __fadd_rn(__fmul_rn(w, -1.f), 1.f); }
    else w = __fadd_rn(w, bias);
    keys[pos++] = ((u64)order_desc(w) << 32) | (u64)id;
  }
}

// segment.py:188-194 on the device, bit for bit as numpy computes it in float32: mask = mean over the channels (sequential
// sum, then one division) > thresh ; weights = 1 - affs, the attractive channels flipped back as (w * -1) + 1.
__global__ __launch_bounds__(256) void mws_prepare_kernel(const float* affs, float* weights, unsigned char* mask, long N, int n_off,
                                                         int n_attr, float thresh) {
  const long u = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (u >= N) return;
  float sum = 0.f;
  for (int k = 0; k < n_off; ++k) {
    const float a = affs[(long)k * N + u];
    sum = k ? __fadd_rn(sum, a) : a;
    float w = __fsub_rn(1.f, a);
    if (k < n_attr) w = __fadd_rn(__fmul_rn(w, -1.f), 1.f);
    weights[(long)k * N + u] = w;
  }
  mask[u] = __fdiv_rn(sum, (float)n_off) > thresh ? 1 : 0;
}
// The same from the uint8 affinities the block runner stores for integer destinations (worker.py:133-144: value * 255, clamped, cast):
// a = (float)u8 * scale in float32 — what `affs8.astype(float32) * scale` gives — without the 4-byte-per-value intermediate (config 5's
// 2048^3 / 8 part: 86 GB of float affinities that exist only to be read once).
__global__ __launch_bounds__(256) void mws_prepare_u8_kernel(const unsigned char* affs, float scale, float* weights, unsigned char* mask, long N,
                                                            int n_off, int n_attr, float thresh) {
  const long u = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (u >= N) return;
  float sum = 0.f;
  for (int k = 0; k < n_off; ++k) {
    const float a = __fmul_rn((float)affs[(long)k * N + u], scale);
    sum = k ? __fadd_rn(sum, a) : a;
    if (weights) {
      float w = __fsub_rn(1.f, a);
      if (k < n_attr) w = __fadd_rn(__fmul_rn(w, -1.f), 1.f);
      weights[(long)k * N + u] = w;
    }
  }
  mask[u] = __fdiv_rn(sum, (float)n_off) > thresh ? 1 : 0;
}
extern "C" int rg_mws_prepare_u8(const uint8_t* affs, float scale, float* weights, uint8_t* mask, int64_t n_voxels, int32_t n_offsets,
                                 int32_t n_attractive, float thresh, rg_stream s) {
  if (!affs || !mask || n_voxels <= 0 || n_offsets <= 0 || n_attractive < 0 || n_attractive > n_offsets)   // weights == NULL: the mask only
    RG_FAIL(RG_ERR_BAD_SHAPE, "mws_prepare_u8: bad arguments");
  if ((n_voxels + 255) / 256 > 0x7fffffffL) RG_FAIL(RG_ERR_UNSUPPORTED, "mws_prepare_u8: volume too large");
  hipLaunchKernelGGL(mws_prepare_u8_kernel, dim3((unsigned)((n_voxels + 255) / 256)), dim3(256), 0, (hipStream_t)s, affs, scale, weights, mask,
                     (long)n_voxels, n_offsets, n_attractive, thresh);
  RG_CHECK_LAUNCH();
  return RG_OK;
}
extern "C" int rg_mws_prepare(const float* affs, float* weights, uint8_t* mask, int64_t n_voxels, int32_t n_offsets, int32_t n_attractive,
                              float thresh, rg_stream s) {
  if (!affs || !weights || !mask || n_voxels <= 0 || n_offsets <= 0 || n_attractive < 0 || n_attractive > n_offsets)
    RG_FAIL(RG_ERR_BAD_SHAPE, "mws_prepare: bad arguments");
  if ((n_voxels + 255) / 256 > 0x7fffffffL) RG_FAIL(RG_ERR_UNSUPPORTED, "mws_prepare: volume too large");
  hipLaunchKernelGGL(mws_prepare_kernel, dim3((unsigned)((n_voxels + 255) / 256)), dim3(256), 0, (hipStream_t)s, affs, weights, mask,
                     (long)n_voxels, n_offsets, n_attractive, thresh);
  RG_CHECK_LAUNCH();
  return RG_OK;
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
  for (int n = s.head[ra]; n >= 0;) {
    const int2 nd = s.pool[n];
    const int rp = mws_find(s.parent, nd.x);
    if (rp == rb) return true;
    if (rp != nd.x) s.pool[n].x = rp;  // any voxel of the partner cluster stands for it: keep the walk one hop long next time
    n = nd.y;
  }
  return false;
}

// ---- round-based Kruskal: the whole chip walks a WINDOW over the sorted edge list -----------------------------------------
// A window entry is a pending edge: its slot in the sorted list (= its rank in the order), the last known roots of its two
// ends, and the mutex-list lengths seen at its last mutex check.  Every round:
//   A  find both roots.  Same root -> retired (clusters only grow).  Attractive edge between two clusters that already hold a
//      mutex -> retired (mutexes are never removed).  Otherwise claim both roots: claim[r] = earliest pending edge at r.
//   B  decide, with parents and lists as phase A saw them (lists are only changed in C):
//      * an edge that is the earliest pending edge at BOTH roots executes as in the sequential order: nothing earlier can
//        touch either cluster any more;
//      * a repulsive edge that is the earliest at ONE of its roots, b, executes: b cannot merge with anything before its turn,
//        so the mutex it adds separates two clusters that are distinct at its turn, and the only edges that could notice it
//        early (attractive edges between the two clusters) end in b, hence come later in the order;
//      * an attractive edge (a, b) that is the earliest at b executes (b joins a) when every cluster p that holds a mutex
//        with b is QUIET — no pending edge at p is earlier than this edge — or already holds a mutex with a.  Then p cannot
//        become part of a's cluster before the edge's turn, so the sequential algorithm finds no mutex and merges; and no
//        earlier edge can observe b's mutexes inside a's cluster, because it would have to end in such a p.  A cluster
//        without mutexes is the trivial case (the Boruvka step).
//      Everything else stays pending.  The earliest pending edge always executes, so every round makes progress.
//   C  (only in rounds that need it) mutex-list surgery, by atomic appends to the CURRENT roots: a merged cluster's list is
//      appended to its root's list, a repulsive edge appends one node to either side.
// The partition is the sequential one; root ids and list order differ, labels are canonical.
struct MwsCtl { unsigned bar, surv[2], nops[2], pool_used, stalled, rounds; unsigned long long t[4], visits; };  // t: 100 MHz ticks in A, B, C (tid 0)
#define MWS_RETIRED 0xffffffffu
#define MWS_ATTR 0x80000000u
#define MWS_THREADS 1024
#ifndef MWS_NESTED_MAX
#define MWS_NESTED_MAX 16  // longest list walked to show that a busy partner already holds a mutex with the other side
#endif
#ifndef MWS_QUIET_MAX
#define MWS_QUIET_MAX 192  // 48 -> 192: pure-noise volume 102 -> 93 ms, structured volume unchanged (sweep 4 / 16 / 48 / 200 x nested 4 / 16 / 64)
#endif
//         // longest mutex list examined for the quiet-partner rule
#define MWS_PAIR_BITS 20          // cache of root pairs known to hold a mutex (sound: roots never come back, mutexes stay)

__device__ __forceinline__ u64 mws_pair_key(int x, int y) { return (x < y ? ((u64)(unsigned)x << 32) | (unsigned)y : ((u64)(unsigned)y << 32) | (unsigned)x) + 1ull; }
__device__ __forceinline__ unsigned mws_pair_hash(u64 k) { k *= 0x9E3779B97F4A7C15ull; return (unsigned)(k >> (64 - MWS_PAIR_BITS)); }
__device__ bool mws_pair_known(const u64* pairs, int x, int y) {
  const u64 k = mws_pair_key(x, y);
  unsigned h = mws_pair_hash(k);
  for (int i = 0; i < 4; ++i, h = (h + 1) & ((1u << MWS_PAIR_BITS) - 1)) {
    const u64 v = pairs[h];
    if (v == k) return true;
    if (v == 0) return false;
  }
  return false;
}
__device__ void mws_pair_insert(u64* pairs, int x, int y) {
  const u64 k = mws_pair_key(x, y);
  unsigned h = mws_pair_hash(k);
  for (int i = 0; i < 4; ++i, h = (h + 1) & ((1u << MWS_PAIR_BITS) - 1)) {
    const u64 v = atomicCAS(&pairs[h], 0ull, k);
    if (v == 0 || v == k) return;
  }
}
// exact when parents and lists are stable (phase A); in phase B a concurrent merge can only turn a "yes" into a "no"
__device__ bool mws_mutex_between(const MwsState& s, u64* pairs, int ra, int rb) {
  if (mws_pair_known(pairs, ra, rb)) return true;
  if (mws_has_mutex(s, ra, rb)) { mws_pair_insert(pairs, ra, rb); return true; }
  return false;
}
// may the attractive edge `slot` merge cluster c into `other` ahead of its turn?  (c's claim is this edge.)
// 1 = yes, 0 = not now, -1 = never: the two clusters hold a mutex (c's whole list is walked, so this is the exact test)
__device__ int mws_quiet(const MwsState& s, u64* pairs, const u64* claim, int c, int other, unsigned slot, unsigned round) {
  const int n = s.cnt[c];
  if (n == 0) return 1;
  if (n > MWS_QUIET_MAX) return 0;
  int ok = 1;
  for (int node = s.head[c]; node >= 0;) {
    const int2 nd = s.pool[node];
    const int rp = mws_find(s.parent, nd.x);
    if (rp != nd.x) s.pool[node].x = rp;
    node = nd.y;
    if (rp == other) { mws_pair_insert(pairs, c, other); return -1; }
    const u64 cl = claim[rp];
    if (ok && (unsigned)(cl >> 32) == round && 0xffffffffu - (unsigned)cl < slot)  // an earlier edge is pending at this partner
      if (s.cnt[other] == 0 || !(mws_pair_known(pairs, other, rp) ||
                                 (min(s.cnt[other], s.cnt[rp]) <= MWS_NESTED_MAX && mws_mutex_between(s, pairs, other, rp)))) ok = 0;
  }
  return ok;
}
__device__ __forceinline__ void mws_list_append(const MwsState& s, int r, int h, int t, int c) {
  const int old = atomicExch(&s.tail[r], t);
  if (old >= 0) s.pool[old].y = h; else s.head[r] = h;
  atomicAdd(&s.cnt[r], c);
}

// Grid barrier of the persistent kernel.  One thread per workgroup does the device-scope release / acquire (the workgroup
// barriers around it order the other waves' accesses with it).  The arrivals are counted in a two-level tree — groups of 16
// workgroups on counters of their own (64 bytes apart), the last arriver of a group bumps the top counter everybody polls —
// because 128 atomic increments of ONE word serialise: measured on MI355X with a flat counter 2.2 us at 32 workgroups, 3.7 us
// at 64, 8.5 us at 128 (and 62 us at 128 when every wave fences, so the grid stays small: rounds are latency-bound).
// Counters only ever grow (epoch e is complete when the top counter reaches e * groups): no reset, no ABA.
#define MWS_BAR_GROUP 16
#define MWS_BAR_STRIDE 16   // unsigned words between counters
__device__ __forceinline__ void mws_grid_barrier(unsigned* bar, unsigned& epoch, unsigned nwg) {
  __syncthreads();
  ++epoch;
  if (threadIdx.x == 0) {
    __threadfence();
    const unsigned g = blockIdx.x / MWS_BAR_GROUP, ng = (nwg + MWS_BAR_GROUP - 1) / MWS_BAR_GROUP;
    const unsigned gsize = (g + 1) * MWS_BAR_GROUP <= nwg ? MWS_BAR_GROUP : nwg - g * MWS_BAR_GROUP;
    if (atomicAdd(&bar[(1 + g) * MWS_BAR_STRIDE], 1u) + 1 == epoch * gsize) atomicAdd(&bar[0], 1u);   // last of my group this epoch
    while (__hip_atomic_load(&bar[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < epoch * ng) __builtin_amdgcn_s_sleep(1);
    __threadfence();
  }
  __syncthreads();
}

__global__ __launch_bounds__(MWS_THREADS) void mws_rounds_kernel(const u64* keys, const unsigned* count, const int* offsets, MwsState s,
                                                                u64* claim, u64* pairs, uint4* win0, uint4* win1, int2* ops, MwsCtl* ctl,
                                                                int Z, int Y, int X, int n_off, int n_attr, unsigned cap, unsigned pool_cap, unsigned wmul,
                                                                const unsigned* perm, const unsigned* eu, const unsigned* ev, const unsigned char* eattr, unsigned* bar) {
  // grid mode (perm == nullptr): edge id = k*N + u on the (Z,Y,X) grid, v = u + offset k.  graph mode: slot -> edge perm[slot] of an
  // explicit edge list (eu, ev, eattr) over N = Z*Y*X abstract nodes (the second level of the slab decomposition).
  __shared__ int lin[64];
  if (!perm && threadIdx.x < n_off && threadIdx.x < 64)
    lin[threadIdx.x] = (offsets[3 * threadIdx.x] * Y + offsets[3 * threadIdx.x + 1]) * X + offsets[3 * threadIdx.x + 2];
  __syncthreads();
  const unsigned T = gridDim.x * blockDim.x, tid = blockIdx.x * blockDim.x + threadIdx.x, lane = threadIdx.x & 63;
  const unsigned N = (unsigned)((long)Z * Y * X);
  const unsigned ne = *count;
  unsigned bar_target = 0, next = 0, nsurv = 0, pend = 0;   // pend: list operations of the previous round, applied at the head of this one
  uint4* win = win0; uint4* wout = win1;
  for (unsigned round = 1;; ++round) {
    // fresh edges this round: top the window up to ~2 entries per thread (the survivors already hold the earliest pending
    // edge, so progress never depends on new ones; a window full of blocked edges is re-examined every round, so keep it small)
    unsigned want = nsurv < wmul * T ? wmul * T - nsurv : 0;
    if (want > cap - nsurv) want = cap - nsurv;
    if (want > ne - next) want = ne - next;
    const unsigned n = nsurv + want;
    if (n == 0) break;
    const u64 tag = (u64)round << 32;
    const u64 c0 = wall_clock64();
    // ---- phase C of the PREVIOUS round, in the same barrier interval as phase A: mutex-list surgery by atomic appends to the CURRENT
    // roots.  Phase A does not walk lists (it reads cnt only to decide whether the pair cache is worth a look, and a cache hit is a
    // fact whatever cnt says), parents are not changed by either part (path halving aside), and phase B — the list walker — starts
    // behind the barrier: the third grid barrier of a round and its own chain of dependent loads are gone.
    if (pend) {
      const int2* po = ops + (size_t)((round - 1) & 1) * cap;
      for (unsigned i = tid; i < pend; i += T) {
        const int2 op = po[i];
        if (op.y < 0) mws_list_append(s, mws_find(s.parent, op.x), s.head[op.x], s.tail[op.x], s.cnt[op.x]);
        else {
          const unsigned na = atomicAdd(&ctl->pool_used, 2u), nb = na + 1;
          if (nb < pool_cap) {
            // only .x is written: .y is -1 from the initial fill, and once the node is published by the exchange below another
            // thread may already be linking its own chain behind it
            s.pool[na].x = op.y; mws_list_append(s, mws_find(s.parent, op.x), (int)na, (int)na, 1);
            s.pool[nb].x = op.x; mws_list_append(s, mws_find(s.parent, op.y), (int)nb, (int)nb, 1);
          }
        }
      }
    }
    // ---- phase A
    for (unsigned i = tid; i < n; i += T) {
      uint4 e;
      if (i < nsurv) e = win[i];
      else {
        const unsigned slot = next + (i - nsurv);
        if (perm) {
          const unsigned id = perm[slot];
          e = make_uint4(slot, eu[id], ev[id], (eattr[id] ? MWS_ATTR : 0u) | 0x7fffffffu);
        } else {
          const unsigned id = (unsigned)keys[slot];
          const unsigned k = id / N, u = id - k * N;
          e = make_uint4(slot, u, (unsigned)((int)u + lin[k]), (k < (unsigned)n_attr ? MWS_ATTR : 0u) | 0x7fffffffu);
        }
      }
      const int a = mws_find(s.parent, (int)e.y), b = mws_find(s.parent, (int)e.z);
      bool retired = a == b;
      // known mutex between the two clusters: the edge can never merge them (the exact test runs in phase B, once, when the
      // edge is about to execute; the pair cache lets every other edge between the same two clusters leave right here)
      if (!retired && (e.w & MWS_ATTR) && s.cnt[a] && s.cnt[b]) retired = mws_pair_known(pairs, a, b);
      if (!retired) {
        const u64 mine = tag | (u64)(0xffffffffu - e.x);
        // a claim only grows within a round: a root that already holds an earlier edge needs no read-modify-write from this one.  With one giant
        // cluster (a random-initialised network's output: half a million window entries end in the same root) every entry used to queue an
        // atomicMax on ONE word — 7.1 of the 7.5 s of that case; the entries arrive roughly earliest first, so almost all of them now stop at the load
        if (__hip_atomic_load(&claim[a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < mine) atomicMax(&claim[a], mine);
        if (__hip_atomic_load(&claim[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < mine) atomicMax(&claim[b], mine);
      }
      e.y = (unsigned)a; e.z = (unsigned)b;
      if (retired) e.x = MWS_RETIRED;
      win[i] = e;
    }
    mws_grid_barrier(bar, bar_target, gridDim.x);
    const u64 c1 = wall_clock64();
    // ---- phase B
    unsigned* surv = &ctl->surv[round & 1];
    unsigned* nops = &ctl->nops[round & 1];
    int2* const ops_w = ops + (size_t)(round & 1) * cap;
    for (unsigned i0 = tid - lane; i0 < n; i0 += T) {  // whole waves iterate together (ballots below)
      const unsigned i = i0 + lane;
      uint4 e = make_uint4(MWS_RETIRED, 0, 0, 0);
      if (i < n) e = win[i];
      bool keep = false;
      int2 op = make_int2(-1, -1);  // (child, -1): append child's list to its root ; (a, b) with b >= 0: new mutex a | b
      if (e.x != MWS_RETIRED) {
        const int a = (int)e.y, b = (int)e.z;
        const u64 mine = tag | (u64)(0xffffffffu - e.x);
        const bool fa = claim[a] == mine, fb = claim[b] == mine;
        if (e.w & MWS_ATTR) {
          int child = -1, q = 0;
          if (fa && fb) {  // nothing else can touch either cluster this round: the list walk is exact
            if (s.cnt[a] && s.cnt[b] && mws_mutex_between(s, pairs, a, b)) q = -1;
            else child = s.cnt[b] <= s.cnt[a] ? b : a;
          } else {
            if (fb) { q = mws_quiet(s, pairs, claim, b, a, e.x, round); if (q > 0) child = b; }
            if (fa && q == 0) { q = mws_quiet(s, pairs, claim, a, b, e.x, round); if (q > 0) child = a; }
          }
          if (child >= 0) {
            s.parent[child] = child == b ? a : b;
            if (s.cnt[child] > 0) op = make_int2(child, -1);
          } else keep = q == 0;
        } else if (fa || fb) op = make_int2(a, b);
        else keep = true;
      }
      const u64 km = __ballot(keep);
      if (km) {
        unsigned base = 0;
        if (lane == 0) base = atomicAdd(surv, (unsigned)__popcll(km));
        base = __shfl(base, 0);
        if (keep) wout[base + __popcll(km & ((1ull << lane) - 1ull))] = e;
      }
      const u64 om = __ballot(op.x >= 0);
      if (om) {
        unsigned base = 0;
        if (lane == 0) base = atomicAdd(nops, (unsigned)__popcll(om));
        base = __shfl(base, 0);
        if (op.x >= 0) ops_w[base + __popcll(om & ((1ull << lane) - 1ull))] = op;
      }
    }
    mws_grid_barrier(bar, bar_target, gridDim.x);
    const u64 c2 = wall_clock64();
    const unsigned ns2 = __hip_atomic_load(surv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned no = __hip_atomic_load(nops, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    next += want;
    if (tid == 0) { ctl->surv[(round + 1) & 1] = 0; ctl->nops[(round + 1) & 1] = 0; ctl->rounds = round; }  // everyone read those before the last barrier
    pend = no;   // applied at the head of the next round (the last round's operations are never needed: labels come from the parents)
    if (tid == 0) { ctl->t[0] += c1 - c0; ctl->t[1] += c2 - c1; ctl->visits += n; }
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
    if (first[r] > (int)i) atomicMin(&first[r], (int)i);  // a stale read only costs a redundant atomic
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
#define MWS_MAX_WGS 256
struct MwsPlan { long N, ne_max, pool_nodes; size_t off_claim, off_pairs, off_ops, off_win, off_keys2, off_cnt, off_state, off_pool, off_first, off_flags, off_rank, off_tmp, tmp_bytes, total; };
static inline size_t al(size_t v) { return (v + 255) / 256 * 256; }
static int mws_plan(const rg_mws_params* p, MwsPlan* q) {
  if (!p || p->shape[0] <= 0 || p->shape[1] <= 0 || p->shape[2] <= 0 || p->n_offsets <= 0 || p->n_attractive < 0 ||
      p->n_attractive > p->n_offsets || p->strides[0] <= 0 || p->strides[1] <= 0 || p->strides[2] <= 0 || p->randomize_strides || p->origin[0] < 0 || p->origin[1] < 0 ||
      p->origin[2] < 0 || !(p->bias_cut == p->bias_cut))
    return -1;
  q->N = (long)p->shape[0] * p->shape[1] * p->shape[2];
  if ((double)q->N * p->n_offsets >= 4294967295.0 || q->N >= 0x7fffffff) return -2;  // 32-bit edge ids
  // voxels on the stride grid: an upper bound for any origin
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
  o += 2048; q->off_state = o;   // edge count (+0), MwsCtl (+64), barrier counters (+512: 1 + MWS_MAX_WGS / 16 words, 64 B apart)
  o += al((size_t)q->N * 4) * 5; q->off_pool = o;
  q->pool_nodes = 2 * grid * (p->n_offsets - p->n_attractive) + 2;
  o += al((size_t)q->pool_nodes * 8); q->off_claim = o;
  o += al((size_t)q->N * 8); q->off_pairs = o;
  o += al((size_t)8 << MWS_PAIR_BITS); q->off_ops = o;
  o += 2 * al((size_t)MWS_WIN_CAP * 8); q->off_win = o;   // two operation buffers (round parity)
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

static int mws_segment_src(MwsSrc affs, const int32_t* offsets, const uint8_t* mask, uint64_t* labels, const rg_mws_params* p, void* workspace,
                           size_t ws_bytes, rg_stream s);
extern "C" int rg_mws_segment(const float* affs, const int32_t* offsets, const uint8_t* mask, uint64_t* labels,
                              const rg_mws_params* p, void* workspace, size_t ws_bytes, rg_stream s) {
  if (!p) RG_FAIL(RG_ERR_BAD_SHAPE, "mws: null parameters");
  const MwsSrc src = {affs, (long)p->shape[0] * p->shape[1] * p->shape[2], 0.f, 0};
  return mws_segment_src(src, offsets, mask, labels, p, workspace, ws_bytes, s);
}
extern "C" int rg_mws_segment_u8(const uint8_t* affs8, float scale, int32_t prepare_sep, int64_t channel_stride, const int32_t* offsets,
                                 const uint8_t* mask, uint64_t* labels, const rg_mws_params* p, void* workspace, size_t ws_bytes, rg_stream s) {
  if (!p || scale == 0.f || prepare_sep < 0 || channel_stride < (int64_t)p->shape[0] * p->shape[1] * p->shape[2])
    RG_FAIL(RG_ERR_BAD_SHAPE, "mws_segment_u8: bad scale / prepare_sep / channel stride");
  const MwsSrc src = {affs8, (long)channel_stride, scale, prepare_sep};
  return mws_segment_src(src, offsets, mask, labels, p, workspace, ws_bytes, s);
}
static int mws_segment_src(MwsSrc affs, const int32_t* offsets, const uint8_t* mask, uint64_t* labels, const rg_mws_params* p, void* workspace,
                           size_t ws_bytes, rg_stream s) {
  MwsPlan q;
  const int rc = mws_plan(p, &q);
  if (rc == -2) RG_FAIL(RG_ERR_UNSUPPORTED, "mws: volume too large for one segment (n_offsets*N must be < 2^32): split into blocks");
  if (rc) RG_FAIL(RG_ERR_BAD_SHAPE, "mws: bad parameters (randomize_strides is not supported)");
  if (!affs.p || !offsets || !labels) RG_FAIL(RG_ERR_BAD_SHAPE, "mws: null pointer");
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
  if (hipMemsetAsync(count, 0, 2048, st) != hipSuccess) RG_FAIL(RG_ERR_HIP, "mws: memset failed");
  if (p->n_offsets > 64) RG_FAIL(RG_ERR_UNSUPPORTED, "mws: at most 64 offsets");
  hipLaunchKernelGGL(mws_edges_kernel, dim3((unsigned)((q.N + MWS_EDGE_THREADS - 1) / MWS_EDGE_THREADS)), dim3(MWS_EDGE_THREADS), 0, st, affs, offsets, mask, keys, count, Z, Y, X, p->n_offsets,
                     p->n_attractive, p->strides[0], p->strides[1], p->strides[2], p->origin[0], p->origin[1], p->origin[2],
                     p->invert_repulsive_weights, p->bias_cut);
  RG_CHECK_LAUNCH();
  // The sort is sized by the edges that EXIST (a masked volume holds well under half of ne_max = every in-range edge: 22.5 M of
  // 50.6 M in the 256^3 benchmark volume, and the radix sort was 2.1 ms of 12): the count comes back to the host while the GPU
  // runs the initialisation passes queued behind it — this call therefore waits for the stream once (like rg_mws_stats).
  unsigned n_edges = 0;
  if (hipMemcpyAsync(&n_edges, count, sizeof(n_edges), hipMemcpyDeviceToHost, st) != hipSuccess) RG_FAIL(RG_ERR_HIP, "mws: count read-back failed");
  hipLaunchKernelGGL(mws_init_kernel, dim3(blocks), dim3(256), 0, st, S, first, flags, q.N);
  RG_CHECK_LAUNCH();
  {
    u64* claim0 = (u64*)(w + q.off_claim); u64* pairs0 = (u64*)(w + q.off_pairs);
    if (hipMemsetAsync(S.pool, 0xff, (size_t)q.pool_nodes * 8, st) != hipSuccess || hipMemsetAsync(claim0, 0, (size_t)q.N * 8, st) != hipSuccess ||
        hipMemsetAsync(pairs0, 0, (size_t)8 << MWS_PAIR_BITS, st) != hipSuccess) RG_FAIL(RG_ERR_HIP, "mws: memset failed");
  }
  if (hipStreamSynchronize(st) != hipSuccess) RG_FAIL(RG_ERR_HIP, "mws: synchronisation failed");
  if (n_edges > (unsigned long)q.ne_max) RG_FAIL(RG_ERR_HIP, "mws: edge count %u beyond the bound %ld", n_edges, q.ne_max);
  size_t tb = q.tmp_bytes;
  if (n_edges && rocprim::radix_sort_keys(tmp, tb, keys, keys2, (size_t)n_edges, 0, 64, st) != hipSuccess) RG_FAIL(RG_ERR_HIP, "mws: radix sort failed");
  {
    // persistent grid with hand-made grid barriers: every workgroup must be resident, hence the cooperative launch
    // sized per call from the device the caller made current: no process-wide state (header contract)
    int wgs = 0;
    {
      int dev = 0, cus = 0, per_cu = 0;
      if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
          hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, mws_rounds_kernel, MWS_THREADS, 0) != hipSuccess || cus * per_cu < 1)
        RG_FAIL(RG_ERR_HIP, "mws: cannot size the persistent grid");
      const int mx = rg_opt(RG_OPT_MWS_WGS) > 0 && rg_opt(RG_OPT_MWS_WGS) <= MWS_MAX_WGS ? rg_opt(RG_OPT_MWS_WGS) : 128;
      wgs = cus * per_cu < mx ? cus * per_cu : mx;
    }
    const u64* kp = keys2; const unsigned* cp = count; const int* op = offsets;
    u64* claim = (u64*)(w + q.off_claim); u64* pairs = (u64*)(w + q.off_pairs); int2* ops = (int2*)(w + q.off_ops);
    uint4* win0 = (uint4*)(w + q.off_win); uint4* win1 = (uint4*)(w + q.off_win + al((size_t)MWS_WIN_CAP * 16));
    MwsCtl* ctl = (MwsCtl*)(w + q.off_cnt + 64);
    unsigned* bar = (unsigned*)(w + q.off_cnt + 512);
    int zz = Z, yy = Y, xx = X, no = p->n_offsets, na = p->n_attractive;
    unsigned cap = MWS_WIN_CAP, pool_cap = (unsigned)q.pool_nodes, wmul = rg_opt(RG_OPT_MWS_WMUL) > 0 ? (unsigned)rg_opt(RG_OPT_MWS_WMUL) : 4u;
    const unsigned* gperm = nullptr; const unsigned* geu = nullptr; const unsigned* gev = nullptr; const unsigned char* gattr = nullptr;
    void* args[] = {&kp, &cp, &op, &S, &claim, &pairs, &win0, &win1, &ops, &ctl, &zz, &yy, &xx, &no, &na, &cap, &pool_cap, &wmul,
                    &gperm, &geu, &gev, &gattr, &bar};
    rg_set_kernel_name("mws_rounds_kernel");
    rg_prof_mark(st);
    if (hipLaunchCooperativeKernel((const void*)mws_rounds_kernel, dim3(wgs), dim3(MWS_THREADS), args, 0, st) != hipSuccess)
      RG_FAIL(RG_ERR_HIP, "mws: cooperative launch failed: %s", hipGetErrorString(hipGetLastError()));
    rg_prof_mark(st);
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

// Diagnostics of the last rg_mws_segment run in `workspace`: out = {edges, rounds, window visits, 100-MHz ticks spent in phase A, B, C
// (thread 0's view, barriers included), mutex-list nodes used, 0}.  Synchronises the stream.
extern "C" int rg_mws_stats(const void* workspace, const rg_mws_params* p, uint64_t out[8], rg_stream s) {
  MwsPlan q;
  if (!workspace || !out || mws_plan(p, &q)) RG_FAIL(RG_ERR_BAD_SHAPE, "mws_stats: bad arguments");
  const char* w = (const char*)workspace;
  unsigned count = 0; MwsCtl ctl;
  if (hipMemcpyAsync(&count, w + q.off_cnt, sizeof(count), hipMemcpyDeviceToHost, (hipStream_t)s) != hipSuccess ||
      hipMemcpyAsync(&ctl, w + q.off_cnt + 64, sizeof(ctl), hipMemcpyDeviceToHost, (hipStream_t)s) != hipSuccess ||
      hipStreamSynchronize((hipStream_t)s) != hipSuccess) RG_FAIL(RG_ERR_HIP, "mws_stats: copy failed");
  out[0] = count; out[1] = ctl.rounds; out[2] = ctl.visits; out[3] = ctl.t[0]; out[4] = ctl.t[1]; out[5] = ctl.t[2]; out[6] = ctl.pool_used; out[7] = 0;
  return RG_OK;
}

// ==== second level of the slab decomposition: mutex watershed of an explicit edge list ============
// Nodes are first-level clusters, edges the grid edges that cross a slab boundary plus the mutexes the first level left between
// clusters of one slab (wkey 0: before everything else).  Order: ascending wkey (= descending transformed weight, the same 32-bit
// order_desc() value the grid pass ranks by), ties by ascending 64-bit global edge id `gid`.  Same Kruskal rules, same kernel;
// result: rep[i] = smallest node id of i's cluster.
__global__ void mws_iota_kernel(unsigned* p, unsigned n) {
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) p[i] = i;
}
__global__ void mws_gather_u32_kernel(const unsigned* src, const unsigned* idx, unsigned* dst, unsigned n) {
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) dst[i] = src[idx[i]];
}
__global__ void mws_rep_kernel(MwsState s, const int* first, int* rep, long N, const MwsCtl* ctl) {
  const bool bad = ctl->stalled != 0;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (long)gridDim.x * blockDim.x) rep[i] = bad ? -1 : first[s.root[i]];
}
struct MwsGraphPlan { long N, E; size_t off_gid2, off_perm, off_perm2, off_wk, off_wk2, off_state, off_pool, off_claim, off_pairs, off_ops, off_win, off_first, off_flags, off_ctl, off_tmp, tmp_bytes, total; long pool_nodes; };
static int mws_graph_plan(int64_t n_nodes, int64_t n_edges, MwsGraphPlan* q) {
  if (n_nodes <= 0 || n_nodes >= 0x7fffffffL || n_edges < 0 || n_edges >= 0xffffffffL) return -1;
  q->N = n_nodes; q->E = n_edges < 1 ? 1 : n_edges;
  size_t t1 = 0, t2 = 0;
  rocprim::radix_sort_pairs(nullptr, t1, (u64*)nullptr, (u64*)nullptr, (unsigned*)nullptr, (unsigned*)nullptr, (size_t)q->E, 0, 64, (hipStream_t)0);
  rocprim::radix_sort_pairs(nullptr, t2, (unsigned*)nullptr, (unsigned*)nullptr, (unsigned*)nullptr, (unsigned*)nullptr, (size_t)q->E, 0, 32, (hipStream_t)0);
  q->tmp_bytes = t1 > t2 ? t1 : t2;
  q->pool_nodes = 2 * q->E + 2;
  size_t o = 0;
  q->off_gid2 = o; o += al((size_t)q->E * 8);
  q->off_perm = o; o += al((size_t)q->E * 4);
  q->off_perm2 = o; o += al((size_t)q->E * 4);
  q->off_wk = o; o += al((size_t)q->E * 4);
  q->off_wk2 = o; o += al((size_t)q->E * 4);
  q->off_state = o; o += al((size_t)q->N * 4) * 5;
  q->off_pool = o; o += al((size_t)q->pool_nodes * 8);
  q->off_claim = o; o += al((size_t)q->N * 8);
  q->off_pairs = o; o += al((size_t)8 << MWS_PAIR_BITS);
  q->off_ops = o; o += 2 * al((size_t)MWS_WIN_CAP * 8);
  q->off_win = o; o += 2 * al((size_t)MWS_WIN_CAP * 16);
  q->off_first = o; o += al((size_t)q->N * 4);
  q->off_flags = o; o += al((size_t)q->N * 4);
  q->off_ctl = o; o += 2048;
  q->off_tmp = o; o += al(q->tmp_bytes);
  q->total = o;
  return 0;
}
extern "C" int64_t rg_mws_graph_workspace_bytes(int64_t n_nodes, int64_t n_edges) {
  MwsGraphPlan q;
  if (mws_graph_plan(n_nodes, n_edges, &q)) return -1;
  return (int64_t)q.total;
}
extern "C" int rg_mws_graph(const uint32_t* wkey, const uint64_t* gid, const uint32_t* eu, const uint32_t* ev, const uint8_t* attractive,
                            int64_t n_edges, int64_t n_nodes, int32_t* rep, void* workspace, size_t ws_bytes, rg_stream s) {
  MwsGraphPlan q;
  if (mws_graph_plan(n_nodes, n_edges, &q)) RG_FAIL(RG_ERR_BAD_SHAPE, "mws_graph: bad sizes (nodes %ld, edges %ld)", (long)n_nodes, (long)n_edges);
  if (!rep || (n_edges > 0 && (!wkey || !gid || !eu || !ev || !attractive))) RG_FAIL(RG_ERR_BAD_SHAPE, "mws_graph: null pointer");
  if (!workspace || ws_bytes < q.total) RG_FAIL(RG_ERR_WORKSPACE, "mws_graph: workspace %zu < %zu bytes", ws_bytes, q.total);
  hipStream_t st = (hipStream_t)s;
  char* w = (char*)workspace;
  u64* gid2 = (u64*)(w + q.off_gid2);
  unsigned* perm = (unsigned*)(w + q.off_perm); unsigned* perm2 = (unsigned*)(w + q.off_perm2);
  unsigned* wk = (unsigned*)(w + q.off_wk); unsigned* wk2 = (unsigned*)(w + q.off_wk2);
  MwsState S;
  const size_t nb = al((size_t)q.N * 4);
  S.parent = (int*)(w + q.off_state); S.root = (int*)(w + q.off_state + nb); S.head = (int*)(w + q.off_state + 2 * nb);
  S.tail = (int*)(w + q.off_state + 3 * nb); S.cnt = (int*)(w + q.off_state + 4 * nb);
  S.pool = (int2*)(w + q.off_pool);
  int* first = (int*)(w + q.off_first); unsigned* flags = (unsigned*)(w + q.off_flags);
  MwsCtl* ctl = (MwsCtl*)(w + q.off_ctl + 64);
  unsigned* count = (unsigned*)(w + q.off_ctl);
  void* tmp = w + q.off_tmp;
  int blocks = (int)((q.N + 255) / 256); if (blocks > 8192) blocks = 8192;
  int eblocks = (int)((q.E + 255) / 256); if (eblocks > 8192) eblocks = 8192;
  if (hipMemsetAsync(w + q.off_ctl, 0, 2048, st) != hipSuccess) RG_FAIL(RG_ERR_HIP, "mws_graph: memset failed");
  hipLaunchKernelGGL(mws_init_kernel, dim3(blocks), dim3(256), 0, st, S, first, flags, q.N);
  RG_CHECK_LAUNCH();
  if (n_edges > 0) {
    // order: stable LSD — by gid, then by wkey
    hipLaunchKernelGGL(mws_iota_kernel, dim3(eblocks), dim3(256), 0, st, perm, (unsigned)n_edges);
    RG_CHECK_LAUNCH();
    size_t tb = q.tmp_bytes;
    if (rocprim::radix_sort_pairs(tmp, tb, (const u64*)gid, gid2, perm, perm2, (size_t)n_edges, 0, 64, st) != hipSuccess) RG_FAIL(RG_ERR_HIP, "mws_graph: sort failed");
    hipLaunchKernelGGL(mws_gather_u32_kernel, dim3(eblocks), dim3(256), 0, st, wkey, (const unsigned*)perm2, wk, (unsigned)n_edges);
    RG_CHECK_LAUNCH();
    tb = q.tmp_bytes;
    if (rocprim::radix_sort_pairs(tmp, tb, wk, wk2, perm2, perm, (size_t)n_edges, 0, 32, st) != hipSuccess) RG_FAIL(RG_ERR_HIP, "mws_graph: sort failed");
    const unsigned ne = (unsigned)n_edges;
    if (hipMemcpyAsync(count, &ne, 4, hipMemcpyHostToDevice, st) != hipSuccess) RG_FAIL(RG_ERR_HIP, "mws_graph: memcpy failed");
    int wgs = 0;
    {
      int dev = 0, cus = 0, per_cu = 0;
      if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
          hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, mws_rounds_kernel, MWS_THREADS, 0) != hipSuccess || cus * per_cu < 1)
        RG_FAIL(RG_ERR_HIP, "mws_graph: cannot size the persistent grid");
      const int mx = rg_opt(RG_OPT_MWS_WGS) > 0 && rg_opt(RG_OPT_MWS_WGS) <= MWS_MAX_WGS ? rg_opt(RG_OPT_MWS_WGS) : 128;
      wgs = cus * per_cu < mx ? cus * per_cu : mx;
    }
    const u64* kp = nullptr; const unsigned* cp = count; const int* op = nullptr;
    u64* claim = (u64*)(w + q.off_claim); u64* pairs = (u64*)(w + q.off_pairs); int2* ops = (int2*)(w + q.off_ops);
    uint4* win0 = (uint4*)(w + q.off_win); uint4* win1 = (uint4*)(w + q.off_win + al((size_t)MWS_WIN_CAP * 16));
    unsigned* bar = (unsigned*)(w + q.off_ctl + 512);
    int zz = 1, yy = 1, xx = (int)q.N, no = 0, na = 0;
    unsigned cap = MWS_WIN_CAP, pool_cap = (unsigned)q.pool_nodes, wmul = rg_opt(RG_OPT_MWS_WMUL) > 0 ? (unsigned)rg_opt(RG_OPT_MWS_WMUL) : 4u;
    const unsigned* gperm = perm; const unsigned* geu = eu; const unsigned* gev = ev; const unsigned char* gattr = attractive;
    if (hipMemsetAsync(S.pool, 0xff, (size_t)q.pool_nodes * 8, st) != hipSuccess || hipMemsetAsync(claim, 0, (size_t)q.N * 8, st) != hipSuccess ||
        hipMemsetAsync(pairs, 0, (size_t)8 << MWS_PAIR_BITS, st) != hipSuccess) RG_FAIL(RG_ERR_HIP, "mws_graph: memset failed");
    void* args[] = {&kp, &cp, &op, &S, &claim, &pairs, &win0, &win1, &ops, &ctl, &zz, &yy, &xx, &no, &na, &cap, &pool_cap, &wmul,
                    &gperm, &geu, &gev, &gattr, &bar};
    rg_set_kernel_name("mws_rounds_kernel(graph)");
    if (hipLaunchCooperativeKernel((const void*)mws_rounds_kernel, dim3(wgs), dim3(MWS_THREADS), args, 0, st) != hipSuccess)
      RG_FAIL(RG_ERR_HIP, "mws_graph: cooperative launch failed: %s", hipGetErrorString(hipGetLastError()));
  }
  hipLaunchKernelGGL(mws_first_kernel, dim3(blocks), dim3(256), 0, st, S, (const unsigned char*)nullptr, first, q.N);
  RG_CHECK_LAUNCH();
  hipLaunchKernelGGL(mws_rep_kernel, dim3(blocks), dim3(256), 0, st, S, (const int*)first, rep, q.N, (const MwsCtl*)ctl);
  RG_CHECK_LAUNCH();
  return RG_OK;
}

// ---- building the second-level graph: per-edge passes over a slab's boundary zones (labels are int64 first-level labels) -------
// Edges (u, v = u + offset k) whose u lies in THIS slab and whose v lies in the neighbour above (dir = +1: offsets with dz > 0)
// or below (dir = -1: dz < 0).  `nzone` = the neighbour's H boundary planes nearest to this slab (H >= max |dz|), already as
// GLOBAL cluster ids (0 = masked out).  Validity = the grid pass's rules (in bounds, stride grid in global coordinates for
// repulsive edges, both ends unmasked).  Records are appended in no particular order (the second level sorts by wkey, gid).
__global__ void mws_cross_kernel(MwsSrc w, const long* lab, long base_own, const long* nzone, int H, int Zs, int Y, int X, int z0,
                                 long Ng, int dir, const int* offsets, int n_off, int n_attr, int sz, int sy, int sx, int invert, float bias,
                                 unsigned* wkey, u64* gid, long* gu, long* gv, unsigned char* attr, unsigned* count, unsigned cap) {
  const long plane = (long)Y * X, total = (long)n_off * H * plane;
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
    const int x = (int)(t % X), y = (int)((t / X) % Y), j = (int)((t / plane) % H), k = (int)(t / (plane * H));
    const int dz = offsets[3 * k], a = dz < 0 ? -dz : dz;
    if (dz == 0 || (dz > 0) != (dir > 0) || j >= a || a > Zs) continue;
    const int zl = dir > 0 ? Zs - a + j : j;                 // u's plane in this slab
    const int zn = dir > 0 ? j : H + j - a;                  // v's plane in the neighbour zone
    const int y2 = y + offsets[3 * k + 1], x2 = x + offsets[3 * k + 2];
    if ((unsigned)y2 >= (unsigned)Y || (unsigned)x2 >= (unsigned)X) continue;
    if (k >= n_attr && ((z0 + zl) % sz || y % sy || x % sx)) continue;
    const long ul = (long)zl * plane + (long)y * X + x;
    const long lu = lab[ul], lv = nzone[(long)zn * plane + (long)y2 * X + x2];
    if (lu <= 0 || lv <= 0) continue;
    float ww = mws_src(w, k, ul);
    if (k >= n_attr) { if (invert) ww = __fadd_rn(__fmul_rn(ww, -1.f), 1.f); }
    else ww = __fadd_rn(ww, bias);
    const unsigned pos = atomicAdd(count, 1u);
    if (pos >= cap) continue;
    wkey[pos] = order_desc(ww);
    gid[pos] = (u64)k * (u64)Ng + (u64)((long)(z0 + zl) * plane + (long)y * X + x);
    gu[pos] = base_own + lu; gv[pos] = lv; attr[pos] = k < n_attr ? 1 : 0;
  }
}
// mutexes the first level left between clusters of this slab that both belong to the second-level node set (is_node[label]):
// every valid repulsive edge inside the slab whose ends carry different labels (see DESIGN.md 6 for why that is the mutex set).
__global__ void mws_slab_mutex_kernel(const long* lab, const unsigned char* is_node, int Zs, int Y, int X, int z0, const int* offsets, int n_off,
                                      int n_attr, int sz, int sy, int sx, u64* pairs, unsigned* count, unsigned cap) {
  const long plane = (long)Y * X, Ns = (long)Zs * plane;
  for (long u = (long)blockIdx.x * blockDim.x + threadIdx.x; u < Ns; u += (long)gridDim.x * blockDim.x) {
    const int x = (int)(u % X), y = (int)((u / X) % Y), z = (int)(u / plane);
    if ((z0 + z) % sz || y % sy || x % sx) continue;
    const long lu = lab[u];
    if (lu <= 0 || !is_node[lu]) continue;
    for (int k = n_attr; k < n_off; ++k) {
      const int z2 = z + offsets[3 * k], y2 = y + offsets[3 * k + 1], x2 = x + offsets[3 * k + 2];
      if ((unsigned)z2 >= (unsigned)Zs || (unsigned)y2 >= (unsigned)Y || (unsigned)x2 >= (unsigned)X) continue;
      const long lv = lab[(long)z2 * plane + (long)y2 * X + x2];
      if (lv <= 0 || lv == lu || !is_node[lv]) continue;
      const unsigned pos = atomicAdd(count, 1u);
      if (pos < cap) pairs[pos] = lu < lv ? ((u64)lu << 32) | (u64)lv : ((u64)lv << 32) | (u64)lu;
    }
  }
}
__global__ void mws_relabel_kernel(long* labels, const long* lut, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) labels[i] = lut[labels[i]];
}
static int mws_slab_args_ok(const rg_mws_params* p) {
  return p && p->shape[0] > 0 && p->shape[1] > 0 && p->shape[2] > 0 && p->n_offsets > 0 && p->n_offsets <= 64 && p->n_attractive >= 0 &&
         p->n_attractive <= p->n_offsets && p->strides[0] > 0 && p->strides[1] > 0 && p->strides[2] > 0 && p->origin[0] >= 0;
}
static int mws_cross_src(MwsSrc weights, const int64_t* labels, int64_t base_own, const int64_t* nzone, int32_t H, int32_t dir,
                         int64_t n_global_voxels, const int32_t* offsets, const rg_mws_params* p, uint32_t* wkey, uint64_t* gid,
                         int64_t* gu, int64_t* gv, uint8_t* attractive, uint32_t* count, uint32_t cap, rg_stream s);
extern "C" int rg_mws_cross_edges(const float* weights, const int64_t* labels, int64_t base_own, const int64_t* nzone, int32_t H, int32_t dir,
                                  int64_t n_global_voxels, const int32_t* offsets, const rg_mws_params* p, uint32_t* wkey, uint64_t* gid,
                                  int64_t* gu, int64_t* gv, uint8_t* attractive, uint32_t* count, uint32_t cap, rg_stream s) {
  if (!p) RG_FAIL(RG_ERR_BAD_SHAPE, "mws_cross_edges: null parameters");
  const MwsSrc src = {weights, (long)p->shape[0] * p->shape[1] * p->shape[2], 0.f, 0};
  return mws_cross_src(src, labels, base_own, nzone, H, dir, n_global_voxels, offsets, p, wkey, gid, gu, gv, attractive, count, cap, s);
}
extern "C" int rg_mws_cross_edges_u8(const uint8_t* affs8, float scale, int32_t prepare_sep, int64_t channel_stride, const int64_t* labels,
                                     int64_t base_own, const int64_t* nzone, int32_t H, int32_t dir, int64_t n_global_voxels, const int32_t* offsets,
                                     const rg_mws_params* p, uint32_t* wkey, uint64_t* gid, int64_t* gu, int64_t* gv, uint8_t* attractive,
                                     uint32_t* count, uint32_t cap, rg_stream s) {
  if (!p || scale == 0.f || prepare_sep < 0 || channel_stride < (int64_t)p->shape[0] * p->shape[1] * p->shape[2])
    RG_FAIL(RG_ERR_BAD_SHAPE, "mws_cross_edges_u8: bad scale / prepare_sep / channel stride");
  const MwsSrc src = {affs8, (long)channel_stride, scale, prepare_sep};
  return mws_cross_src(src, labels, base_own, nzone, H, dir, n_global_voxels, offsets, p, wkey, gid, gu, gv, attractive, count, cap, s);
}
static int mws_cross_src(MwsSrc weights, const int64_t* labels, int64_t base_own, const int64_t* nzone, int32_t H, int32_t dir,
                         int64_t n_global_voxels, const int32_t* offsets, const rg_mws_params* p, uint32_t* wkey, uint64_t* gid,
                         int64_t* gu, int64_t* gv, uint8_t* attractive, uint32_t* count, uint32_t cap, rg_stream s) {
  if (!mws_slab_args_ok(p) || !weights.p || !labels || !nzone || !offsets || !wkey || !gid || !gu || !gv || !attractive || !count || H <= 0 ||
      (dir != 1 && dir != -1)) RG_FAIL(RG_ERR_BAD_SHAPE, "mws_cross_edges: bad arguments");
  const long total = (long)p->n_offsets * H * p->shape[1] * p->shape[2];
  int blocks = (int)((total + 255) / 256); if (blocks > 65535) blocks = 65535;
  hipLaunchKernelGGL(mws_cross_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)s, weights, (const long*)labels, (long)base_own, (const long*)nzone, H,
                     p->shape[0], p->shape[1], p->shape[2], p->origin[0], (long)n_global_voxels, dir, offsets, p->n_offsets, p->n_attractive,
                     p->strides[0], p->strides[1], p->strides[2], p->invert_repulsive_weights, p->bias_cut, wkey, (u64*)gid, (long*)gu, (long*)gv,
                     attractive, count, cap);
  RG_CHECK_LAUNCH();
  return RG_OK;
}
extern "C" int rg_mws_slab_mutexes(const int64_t* labels, const uint8_t* is_node, const int32_t* offsets, const rg_mws_params* p, uint64_t* pairs,
                                   uint32_t* count, uint32_t cap, rg_stream s) {
  if (!mws_slab_args_ok(p) || !labels || !is_node || !offsets || !pairs || !count) RG_FAIL(RG_ERR_BAD_SHAPE, "mws_slab_mutexes: bad arguments");
  const long Ns = (long)p->shape[0] * p->shape[1] * p->shape[2];
  int blocks = (int)((Ns + 255) / 256); if (blocks > 65535) blocks = 65535;
  hipLaunchKernelGGL(mws_slab_mutex_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)s, (const long*)labels, is_node, p->shape[0], p->shape[1], p->shape[2],
                     p->origin[0], offsets, p->n_offsets, p->n_attractive, p->strides[0], p->strides[1], p->strides[2], (u64*)pairs, count, cap);
  RG_CHECK_LAUNCH();
  return RG_OK;
}
extern "C" int rg_mws_relabel(int64_t* labels, const int64_t* lut, int64_t n, rg_stream s) {
  if (!labels || !lut || n < 0) RG_FAIL(RG_ERR_BAD_SHAPE, "mws_relabel: bad arguments");
  if (n == 0) return RG_OK;
  int blocks = (int)((n + 255) / 256); if (blocks > 65535) blocks = 65535;
  hipLaunchKernelGGL(mws_relabel_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)s, (long*)labels, (const long*)lut, (long)n);
  RG_CHECK_LAUNCH();
  return RG_OK;
}
