// HBM-bound kernels of the hot path: layout conversion, weight packing, activation backward +
// bias gradient, max-pool, nearest-upsample+crop+concat, residual add, losses, Adam.
// All of them move 16 bytes (8 bf16 channel lanes) per thread-access on channels-last data.
#include "common.h"
#include <stdarg.h>

// ---------------------------------------------------------------------------- error string / version
static thread_local char g_err[512] = "";
void rg_set_error(const char* fmt, ...) {
  va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof(g_err), fmt, ap); va_end(ap);
}
extern "C" const char* rg_last_error_string(void) { return g_err; }
static thread_local char g_kname[128] = "";
void rg_set_kernel_name(const char* fmt, ...) {
  va_list ap; va_start(ap, fmt); vsnprintf(g_kname, sizeof(g_kname), fmt, ap); va_end(ap);
}
extern "C" const char* rg_last_kernel_name(void) { return g_kname; }
#ifdef RG_STORE_F16
extern "C" int rg_version(void) { return 101 | (1 << 16); }  // bit 16: this build stores activations / packed weights as IEEE half
#else
extern "C" int rg_version(void) { return 101; }
#endif

// ---- diagnostic switches + per-device latches ------------------------------------------------------------------------------
#include <atomic>
#include <mutex>
#include <stdlib.h>
static const char* const g_opt_names[RG_OPT_COUNT] = {"RG_FM", "RG_NO_HALO", "RG_NO_THIN", "RG_NO_THIN3", "RG_NO_WLINES", "RG_NO_KWP", "RG_DEBUG_BOUNDS", "RG_GEMM5", "RG_NO_WGEMM", "RG_MWS_WGS", "RG_MWS_WMUL", "RG_SK"};
static std::atomic<int> g_opt[RG_OPT_COUNT];
static const bool g_opt_init = [] {   // environment -> switches, once, when the library is loaded
  for (int i = 0; i < RG_OPT_COUNT; ++i) { const char* e = getenv(g_opt_names[i]); g_opt[i].store(e ? ((i == RG_OPT_FM || i == RG_OPT_GEMM5 || i == RG_OPT_MWS_WGS || i == RG_OPT_MWS_WMUL || i == RG_OPT_SK) ? atoi(e) : (e[0] && e[0] != '0')) : 0); }
  return true;
}();
int rg_opt(int which) { return which >= 0 && which < RG_OPT_COUNT ? g_opt[which].load(std::memory_order_relaxed) : 0; }
extern "C" int rg_set_option(const char* name, int32_t value) {
  for (int i = 0; name && i < RG_OPT_COUNT; ++i)
    if (!strcmp(name, g_opt_names[i])) { g_opt[i].store(value); return RG_OK; }
  RG_FAIL(RG_ERR_BAD_SHAPE, "rg_set_option: unknown option %s", name ? name : "(null)");
}
extern "C" int32_t rg_get_option(const char* name) {
  for (int i = 0; name && i < RG_OPT_COUNT; ++i) if (!strcmp(name, g_opt_names[i])) return g_opt[i].load();
  return -1;
}
// Launch errors are otherwise silent (hipLaunchKernelGGL returns nothing): the binding peeks at the runtime's sticky error after
// every entry point and raises.  0 = no error pending; else the hipError_t, with rg_last_error_string() set.
extern "C" int rg_hip_status(void) {
  const hipError_t e = hipGetLastError();   // reading CLEARS it: one stale error (possibly the caller's own, e.g. torch's) must not fail every later call
  if (e == hipSuccess) return 0;
  rg_set_error("HIP runtime error pending after %s (it may predate this call): %s", g_kname[0] ? g_kname : "(no kernel)", hipGetErrorString(e));
  return (int)e;
}
static std::mutex g_latch_mu;
void rg_once_per_device(rg_device_latch* l, void (*fn)()) {
  int d = 0;
  if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= 256) d = 0;
  std::lock_guard<std::mutex> g(g_latch_mu);
  const unsigned long long bit = 1ull << (d & 63);
  if (l->done[d >> 6] & bit) return;
  fn();
  l->done[d >> 6] |= bit;
}

// ---- kernel timing aid (bench.py's roofline): HIP events around the DOMINANT kernel of a call, on the launch stream -------
// A weight-gradient call launches its main kernel plus small reduction passes; timing the whole call from outside would not
// agree with rocprofv3's per-kernel average.  Off by default (no events, no overhead); not thread-safe (one profiling host
// thread).
#include <vector>
static bool g_prof_on = false;
static std::vector<hipEvent_t> g_prof_ev;   // begin/end pairs in call order
static std::vector<hipEvent_t> g_prof_pool;
static hipEvent_t prof_event() {
  hipEvent_t e = nullptr;
  if (!g_prof_pool.empty()) { e = g_prof_pool.back(); g_prof_pool.pop_back(); }
  else if (hipEventCreate(&e) != hipSuccess) e = nullptr;
  return e;
}
void rg_prof_mark(hipStream_t st) {  // called by a launcher right before and right after its dominant kernel
  if (!g_prof_on) return;
  hipEvent_t e = prof_event();
  if (e && hipEventRecord(e, st) == hipSuccess) g_prof_ev.push_back(e);
}
extern "C" void rg_profile_enable(int on) { g_prof_on = on != 0; }
extern "C" int rg_profile_drain(float* ms, int cap) {  // elapsed ms of every marked kernel since the last drain, in call order
  int n = 0;
  for (size_t i = 0; i + 1 < g_prof_ev.size(); i += 2) {
    float t = 0.f;
    if (hipEventSynchronize(g_prof_ev[i + 1]) == hipSuccess && hipEventElapsedTime(&t, g_prof_ev[i], g_prof_ev[i + 1]) == hipSuccess && n < cap) ms[n++] = t;
  }
  for (hipEvent_t e : g_prof_ev) g_prof_pool.push_back(e);
  g_prof_ev.clear();
  return n;
}

static inline int grid_for(long n, int block = 256, int cap = 1 << 20) {
  long g = (n + block - 1) / block; if (g > cap) g = cap; if (g < 1) g = 1; return (int)g;
}

// ---------------------------------------------------------------------------- copy / convert
struct AnyT { void* p; int dtype; long s[5]; };  // strides for (n,c,d,h,w)

__device__ __forceinline__ float load_any(const AnyT& t, long off) {
  switch (t.dtype) {
    case RG_F32: return ((const float*)t.p)[off];
    case RG_U8: return (float)((const unsigned char*)t.p)[off];
    default: return bf2f(((const bf16_t*)t.p)[off]);
  }
}
// `sat` (uint8 destinations — the block runner's integer outputs, worker.py:133-144): the value is clamped to [lo, hi] and truncated like
// numpy's / torch's float -> uint8 cast.  `flag` (optional): set to 1 when a SOURCE value is inf / NaN (never cleared here).
__global__ void copy_convert_kernel(AnyT src, AnyT dst, int N, int C, int D, int H, int W, int lanes, float scale, float shift, float lo, float hi,
                                    float* flag) {
  const int G8 = (lanes + 7) / 8;
  const long nvox = (long)N * D * H * W, total = nvox * G8;
  // the channels-last side (bf16, channel stride 1, 16-byte voxel granules) moves as ONE 16-byte access per thread; the dense
  // side stays one element per lane and channel (consecutive threads = consecutive voxels: coalesced)
  const bool dst_cl = dst.dtype == RG_ACT_DTYPE && dst.s[1] == 1 && (lanes & 7) == 0;
  // (source: only WHOLE 8-channel groups are read as one vector — a view's voxel pitch need not cover zero-pad lanes, e.g. a
  // channel slice of a wider tensor or a caller's tightly packed channels-last tensor: reading lanes >= C there runs into the
  // next voxel and, at the last voxel, past the end of the allocation)
  const bool src_cl = src.dtype == RG_ACT_DTYPE && src.s[1] == 1;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long v = i % nvox; const int grp = (int)(i / nvox);
    long so, dof;
    if (nvox < (1L << 31)) {  // 32-bit index arithmetic (the common case)
      unsigned t = (unsigned)v;
      const unsigned w = t % (unsigned)W; t /= (unsigned)W; const unsigned h = t % (unsigned)H; t /= (unsigned)H;
      const unsigned d = t % (unsigned)D; const unsigned n = t / (unsigned)D;
      so = n * src.s[0] + d * src.s[2] + h * src.s[3] + w * src.s[4];
      dof = n * dst.s[0] + d * dst.s[2] + h * dst.s[3] + w * dst.s[4];
    } else {
      long t = v;
      const int w = (int)(t % W); t /= W; const int h = (int)(t % H); t /= H; const int d = (int)(t % D); const int n = (int)(t / D);
      so = n * src.s[0] + d * src.s[2] + h * src.s[3] + w * src.s[4];
      dof = n * dst.s[0] + d * dst.s[2] + h * dst.s[3] + w * dst.s[4];
    }
    float val[8];
    bool bad = false;
    if (src_cl && grp * 8 + 8 <= C && (((unsigned long)((const bf16_t*)src.p + so + grp * 8)) & 15) == 0) {
      const bf16x8 raw = *(const bf16x8*)((const bf16_t*)src.p + so + grp * 8);
#pragma unroll
      for (int k = 0; k < 8; ++k) { const float x = bf2f((bf16_t)raw[k]); bad |= (x - x) != 0.f; val[k] = x * scale + shift; }
    } else {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int c = grp * 8 + k;
        const float x = c < C ? load_any(src, so + c * src.s[1]) : 0.f;
        bad |= (x - x) != 0.f;
        val[k] = c < C ? x * scale + shift : 0.f;
      }
    }
    if (flag && bad) *flag = 1.f;
    if (dst_cl) {
      bf16x8 o;
#pragma unroll
      for (int k = 0; k < 8; ++k) o[k] = (short)f2bf(val[k]);
      *(bf16x8*)((bf16_t*)dst.p + dof + grp * 8) = o;
    } else {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int c = grp * 8 + k;
        if (c >= lanes) break;
        if (dst.dtype == RG_F32) ((float*)dst.p)[dof + c * dst.s[1]] = val[k];
        else if (dst.dtype == RG_U8) ((unsigned char*)dst.p)[dof + c * dst.s[1]] = (unsigned char)(int)fminf(fmaxf(val[k], lo), hi);
        else ((bf16_t*)dst.p)[dof + c * dst.s[1]] = f2bf(val[k]);
      }
    }
  }
}

extern "C" int rg_copy_convert(const rg_tensor* src, const rg_tensor* dst, int dst_lanes, float scale, float shift, rg_stream s) {
  return rg_copy_convert_store(src, dst, dst_lanes, scale, shift, 0.f, 255.f, nullptr, s);
}
extern "C" int rg_copy_convert_store(const rg_tensor* src, const rg_tensor* dst, int dst_lanes, float scale, float shift, float lo, float hi,
                                     float* nonfinite_flag, rg_stream s) {
  if (!src || !dst || !src->ptr || !dst->ptr) RG_FAIL(RG_ERR_BAD_SHAPE, "copy_convert: null tensor");
  if (src->rank != 5 || dst->rank != 5) RG_FAIL(RG_ERR_BAD_SHAPE, "copy_convert: rank-5 tensors expected");
  for (int i = 0; i < 5; ++i) if (src->sizes[i] != dst->sizes[i]) RG_FAIL(RG_ERR_BAD_SHAPE, "copy_convert: shape mismatch on dim %d", i);
  if (dst->dtype != RG_ACT_DTYPE && dst->dtype != RG_F32 && dst->dtype != RG_U8) RG_FAIL(RG_ERR_UNSUPPORTED, "copy_convert: dst must be this build's 16-bit type, f32 or u8");
  if (dst->dtype == RG_U8 && (lo < 0.f || hi > 255.f || !(lo <= hi))) RG_FAIL(RG_ERR_BAD_SHAPE, "copy_convert: u8 saturation bounds");
  if (src->dtype != RG_ACT_DTYPE && src->dtype != RG_F32 && src->dtype != RG_U8) RG_FAIL(RG_ERR_UNSUPPORTED, "copy_convert: src dtype");
  const int C = (int)src->sizes[1];
  if (dst_lanes < C) dst_lanes = C;
  if (dst_lanes > C && dst->strides[1] != 1) RG_FAIL(RG_ERR_BAD_SHAPE, "copy_convert: padded lanes need channel stride 1");
  AnyT a, b;
  a.p = src->ptr; a.dtype = src->dtype; b.p = dst->ptr; b.dtype = dst->dtype;
  for (int i = 0; i < 5; ++i) { a.s[i] = src->strides[i]; b.s[i] = dst->strides[i]; }
  const long total = (long)src->sizes[0] * src->sizes[2] * src->sizes[3] * src->sizes[4] * ((dst_lanes + 7) / 8);
  if (total == 0) return RG_OK;
  hipLaunchKernelGGL(copy_convert_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)s, a, b, (int)src->sizes[0], C,
                     (int)src->sizes[2], (int)src->sizes[3], (int)src->sizes[4], dst_lanes, scale, shift, lo, hi, nonfinite_flag);
  RG_CHECK_LAUNCH();
  return RG_OK;
}

// ---------------------------------------------------------------------------- weight packing
// fp32 master weights (dim0, dim1, KD, KH, KW) -> bf16 [rows16][tap][redK] (rows = dim0, reduced channel = dim1; swap: the
// other way round).  A workgroup takes PK_R packed rows x 64 reduced channels x a chunk of <= 32 SOURCE taps through LDS, so
// that both sides move long runs: the read is PK_R (swap: 64) contiguous runs of 64*kvol (PK_R*kvol) floats, the write
// one 16-byte store (8 reduced channels of one row and tap) per thread-iteration.  The tile grid covers the padded
// buffer (rows up to rows16, channels up to redK), pad entries are written as zeros: no memset pass.
#define PK_R 4
#define PK_MAXTAPS 32
__global__ __launch_bounds__(256) void pack_weights_kernel(const float* w, bf16_t* out, rg_pack_params p, int rows, int red, int rows16, int red32, long rowlen) {
  __shared__ float tile[PK_R * 64][PK_MAXTAPS + 1];  // [rr*64 + kk][source tap - t0]
  const int ntap = p.ntap[0] * p.ntap[1] * p.ntap[2];
  const int kvol = p.K[0] * p.K[1] * p.K[2];
  const int nkt = (red32 + 63) / 64;
  const int rb = (blockIdx.x / nkt) * PK_R, kb = (blockIdx.x % nkt) * 64;  // first packed row / reduced channel of the tile
  const int ngrp = (red32 - kb < 64 ? red32 - kb : 64) / 8;                  // 8-channel groups to write (pads included)
  if (kb == 0) {  // flat rows (red_align 1) are padded once, after the last tap: zero that tail
    const int tail = (int)(rowlen - (long)ntap * red32) / 8;
    for (int i = threadIdx.x; i < PK_R * tail; i += 256) {
      const int rr = i / tail, g = i - rr * tail;
      if (rb + rr < rows16) *(bf16x8*)(out + (long)(rb + rr) * rowlen + (long)ntap * red32 + g * 8) = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
    }
  }
  for (int t0 = 0; t0 < kvol; t0 += PK_MAXTAPS) {
    const int nt = kvol - t0 < PK_MAXTAPS ? kvol - t0 : PK_MAXTAPS;
    // fill: lane = source tap (contiguous in memory), 8 (row, channel) pairs per pass; no per-element index divisions
    {
      const int t = threadIdx.x & 31, e0 = threadIdx.x >> 5;
#pragma unroll 8
      for (int e = e0; e < PK_R * 64; e += 8) {
        const int rr = e >> 6, kk = e & 63;
        const int row = rb + rr, k = kb + kk;
        float v = 0.f;
        if (t < nt && row < rows && k < red) v = p.swap ? w[((long)k * p.dim1 + row) * kvol + t0 + t] : w[((long)row * p.dim1 + k) * kvol + t0 + t];
        tile[e][t] = v;
      }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < PK_R * ntap * ngrp; i += 256) {
      const int g = i % ngrp, tap = (i / ngrp) % ntap, rr = i / (ngrp * ntap);
      const int jw = tap % p.ntap[2], jh = (tap / p.ntap[2]) % p.ntap[1], jd = tap / (p.ntap[2] * p.ntap[1]);
      const int kd = p.k0[0] + p.kstep[0] * jd, kh = p.k0[1] + p.kstep[1] * jh, kw = p.k0[2] + p.kstep[2] * jw;
      const int kidx = (kd * p.K[1] + kh) * p.K[2] + kw;
      if (kidx < t0 || kidx >= t0 + nt || rb + rr >= rows16) continue;
      bf16x8 o;
#pragma unroll
      for (int q = 0; q < 8; ++q) o[q] = (short)f2bf(tile[rr * 64 + g * 8 + q][kidx - t0]);
      *(bf16x8*)(out + (long)(rb + rr) * rowlen + (long)tap * red32 + kb + g * 8) = o;
    }
    __syncthreads();
  }
}
// Source in channels-last memory [dim0][source tap][dim1] (torch.channels_last_3d master weights: the layout the
// weight-gradient kernels produce and Adam therefore updates in).
//  swap 0 (rows = dim0, reduced = dim1): every (row, tap) is one contiguous run of dim1 floats -> convert + pad, 8 per thread.
//  swap 1 (rows = dim1, reduced = dim0): a dim0 x dim1 transpose per tap, 64x64 tiles through LDS (256-B reads, 128-B writes).
__device__ __forceinline__ int pk_src_tap(const rg_pack_params& p, int tap) {
  const int jw = tap % p.ntap[2], jh = (tap / p.ntap[2]) % p.ntap[1], jd = tap / (p.ntap[2] * p.ntap[1]);
  return ((p.k0[0] + p.kstep[0] * jd) * p.K[1] + p.k0[1] + p.kstep[1] * jh) * p.K[2] + p.k0[2] + p.kstep[2] * jw;
}
__global__ __launch_bounds__(256) void pack_cl_rows_kernel(const float* w, bf16_t* out, rg_pack_params p, int rows, int red, int rows16, int red32, long rowlen) {
  const int ntap = p.ntap[0] * p.ntap[1] * p.ntap[2], kvol = p.K[0] * p.K[1] * p.K[2], ng = red32 / 8;
  const long total = (long)rows16 * ntap * ng;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int g = (int)(i % ng); long t = i / ng; const int tap = (int)(t % ntap); const int row = (int)(t / ntap);
    bf16x8 o = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
    if (row < rows) {
      const float* src = w + ((long)row * kvol + pk_src_tap(p, tap)) * p.dim1 + g * 8;
#pragma unroll
      for (int k = 0; k < 8; ++k) if (g * 8 + k < red) o[k] = (short)f2bf(src[k]);
    }
    *(bf16x8*)(out + (long)row * rowlen + (long)tap * red32 + g * 8) = o;
  }
}
__global__ __launch_bounds__(256) void pack_cl_transpose_kernel(const float* w, bf16_t* out, rg_pack_params p, int rows, int red, int rows16, int red32, long rowlen) {
  __shared__ __attribute__((aligned(16))) float tile[64][68];  // [k (dim0)][r (dim1)], rows 16-byte aligned for the float4 stores
  const int ntap = p.ntap[0] * p.ntap[1] * p.ntap[2], kvol = p.K[0] * p.K[1] * p.K[2];
  const int nkt = (red32 + 63) / 64, nrt = (rows16 + 63) / 64;
  (void)ntap;
  int b = blockIdx.x;
  const int kt = b % nkt; b /= nkt; const int rt = b % nrt; const int tap = b / nrt;
  const int k0 = kt * 64, r0 = rt * 64, st = pk_src_tap(p, tap);
  const bool v4 = (p.dim1 & 3) == 0 && ((uintptr_t)w & 15) == 0;  // source rows (dim1 floats) keep 16-byte alignment
  if (v4) {
#pragma unroll
    for (int i = threadIdx.x; i < 64 * 16; i += 256) {   // four 16-byte loads per thread in flight
      const int kk = i >> 4, rr = (i & 15) * 4;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (k0 + kk < red && r0 + rr < rows) {            // rows % 4 == 0: a piece is wholly inside or outside
        v = *(const float4*)(w + ((long)(k0 + kk) * kvol + st) * p.dim1 + r0 + rr);
      }
      *(float4*)&tile[kk][rr] = v;
    }
  } else {
    for (int i = threadIdx.x; i < 64 * 64; i += 256) {
      const int kk = i >> 6, rr = i & 63;
      float v = 0.f;
      if (k0 + kk < red && r0 + rr < rows) v = w[((long)(k0 + kk) * kvol + st) * p.dim1 + r0 + rr];
      tile[kk][rr] = v;
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 64 * 8; i += 256) {
    const int rr = i >> 3, g = i & 7;
    if (r0 + rr >= rows16 || k0 + g * 8 >= red32) continue;
    bf16x8 o;
#pragma unroll
    for (int q = 0; q < 8; ++q) o[q] = (short)f2bf(tile[g * 8 + q][rr]);
    *(bf16x8*)(out + (long)(r0 + rr) * rowlen + (long)tap * red32 + k0 + g * 8) = o;
  }
}
static int pack_align(const rg_pack_params* p) { return p->red_align == 64 ? 64 : ((p->red_align == 8 || p->red_align == 1) ? 8 : 32); }
static long pack_rowlen(const rg_pack_params* p, int red) {
  const long n = (long)p->ntap[0] * p->ntap[1] * p->ntap[2] * round_up(red, pack_align(p));
  return p->red_align == 1 ? (n + 63) / 64 * 64 : n;
}
static int pack_dims(const rg_pack_params* p, int* rows, int* red) {
  if (!p || p->dim0 <= 0 || p->dim1 <= 0 || (p->red_align != 0 && p->red_align != 1 && p->red_align != 8 && p->red_align != 32 && p->red_align != 64)) return -1;
  for (int d = 0; d < 3; ++d) {
    if (p->K[d] <= 0 || p->ntap[d] <= 0 || p->k0[d] < 0 || p->kstep[d] <= 0) return -1;
    if (p->k0[d] + p->kstep[d] * (p->ntap[d] - 1) >= p->K[d]) return -1;
  }
  *rows = p->swap ? p->dim1 : p->dim0; *red = p->swap ? p->dim0 : p->dim1;
  return 0;
}
extern "C" int64_t rg_pack_weights_bytes(const rg_pack_params* p) {
  int rows, red;
  if (pack_dims(p, &rows, &red)) return -1;
  return (int64_t)round_up(rows, 16) * pack_rowlen(p, red) * 2;
}
extern "C" int rg_pack_weights(const float* w, void* packed, const rg_pack_params* p, rg_stream s) {
  int rows, red;
  if (!w || !packed || pack_dims(p, &rows, &red)) RG_FAIL(RG_ERR_BAD_SHAPE, "pack_weights: bad params");
  const int rows16 = round_up(rows, 16), red32 = round_up(red, pack_align(p));
  const long total = (long)rows16 * pack_rowlen(p, red);
  (void)total;
  if (((uintptr_t)packed) & 15) RG_FAIL(RG_ERR_BAD_SHAPE, "pack_weights: packed buffer must be 16B aligned");
  if (p->src_channels_last) {
    const long rowlen = pack_rowlen(p, red);
    const int ntap = p->ntap[0] * p->ntap[1] * p->ntap[2];
    if (rowlen != (long)ntap * red32 && hipMemsetAsync(packed, 0, (size_t)total * 2, (hipStream_t)s) != hipSuccess)  // FLAT row tails
      RG_FAIL(RG_ERR_HIP, "pack_weights: memset failed");
    if (!p->swap) {
      const long n = (long)rows16 * ntap * (red32 / 8);
      hipLaunchKernelGGL(pack_cl_rows_kernel, dim3(grid_for(n, 256, 1 << 16)), dim3(256), 0, (hipStream_t)s, w, (bf16_t*)packed, *p, rows, red, rows16, red32, rowlen);
    } else {
      const long nblk = (long)((red32 + 63) / 64) * ((rows16 + 63) / 64) * ntap;
      if (nblk >= (1L << 31)) RG_FAIL(RG_ERR_UNSUPPORTED, "pack_weights: grid too large");
      hipLaunchKernelGGL(pack_cl_transpose_kernel, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)s, w, (bf16_t*)packed, *p, rows, red, rows16, red32, rowlen);
    }
    RG_CHECK_LAUNCH();
    return RG_OK;
  }
  const int nblk = ((rows16 + PK_R - 1) / PK_R) * ((red32 + 63) / 64);
  hipLaunchKernelGGL(pack_weights_kernel, dim3(nblk), dim3(256), 0, (hipStream_t)s, w, (bf16_t*)packed, *p, rows, red, rows16, red32, pack_rowlen(p, red));
  RG_CHECK_LAUNCH();
  return RG_OK;
}

// ---------------------------------------------------------------------------- activation backward + bias grad
// block (GX, GY): x = 8-lane channel group, y = voxel sub-index.  partial[blockIdx.x][lanes] fp32.
__global__ void act_bwd_bias_kernel(CL5 gy, CL5 y, CL5 g, float* partial, int act, float slope, int vox_per_block, int has_y) {
  extern __shared__ float red[];  // [GY][GX*8]
  const int GX = blockDim.x, GY = blockDim.y;
  const int grp = blockIdx.y * GX + threadIdx.x;
  const long nvox = (long)gy.N * gy.D * gy.H * gy.W;
  const long v0 = (long)blockIdx.x * vox_per_block;
  long v1 = v0 + vox_per_block; if (v1 > nvox) v1 = nvox;
  float sum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (grp * 8 < gy.lanes) {
    for (long v = v0 + threadIdx.y; v < v1; v += GY) {
      long t = v;
      const int w = (int)(t % gy.W); t /= gy.W; const int h = (int)(t % gy.H); t /= gy.H; const int d = (int)(t % gy.D); const int n = (int)(t / gy.D);
      const bf16x8 a = *(const bf16x8*)(gy.p + n * gy.sn + d * gy.sd + h * gy.sh + w * gy.sw + grp * 8);
      bf16x8 o;
      if (has_y) {
        const bf16x8 yy = *(const bf16x8*)(y.p + n * y.sn + d * y.sd + h * y.sh + w * y.sw + grp * 8);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const float gv = bf2f((bf16_t)a[k]) * act_bwd_from_y(bf2f((bf16_t)yy[k]), act, slope);
          const bf16_t r = f2bf(gv); o[k] = (short)r; sum[k] += bf2f(r);
        }
      } else {
        o = a;
#pragma unroll
        for (int k = 0; k < 8; ++k) sum[k] += bf2f((bf16_t)a[k]);
      }
      if (g.p) *(bf16x8*)(g.p + n * g.sn + d * g.sd + h * g.sh + w * g.sw + grp * 8) = o;
    }
  }
  if (!partial) return;
#pragma unroll
  for (int k = 0; k < 8; ++k) red[(threadIdx.y * GX + threadIdx.x) * 8 + k] = sum[k];
  __syncthreads();
  if (threadIdx.y == 0 && grp * 8 < gy.lanes) {
    for (int yy = 1; yy < GY; ++yy)
#pragma unroll
      for (int k = 0; k < 8; ++k) sum[k] += red[(yy * GX + threadIdx.x) * 8 + k];
#pragma unroll
    for (int k = 0; k < 8; ++k) partial[(long)blockIdx.x * gy.lanes + grp * 8 + k] = sum[k];
  }
}
// Dense fast path: gy, y and g are whole channels-last buffers with one voxel stride (no crops), so the tensor is a flat array of
// 16-byte pieces, piece i = (voxel i / S8, channel group i % S8), S8 = voxel stride / 8.  A thread's stride through the array is a
// multiple of S8: its channel group never changes (no per-piece division), pieces of pad groups (>= G8) are skipped by the
// threads that own them, and four pieces are in flight per thread.  partial[blockIdx.x][lanes] as in the general kernel.
// ACT: RG_ACT_RELU compiled in (the U-Net's layers: the run-time switch of act_bwd_from_y costs a scalar branch per ELEMENT), -1 = the run-time switch
template <int ACT>
__global__ __launch_bounds__(256) void act_bwd_bias_dense_kernel(const bf16_t* gy, const bf16_t* y, bf16_t* g, float* partial, long npieces, int S8,
                                                                int G8, int lanes, int act, float slope, int has_y, long step) {
  __shared__ float red[256 * 8];
  const long t0 = (long)blockIdx.x * 256 + threadIdx.x;   // (gridDim.x * 256) % S8 == 0 and 256 % S8 == 0 or handled by `step`
  const int grp = (int)(t0 % S8);
  float sum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (grp < G8) {
    for (long i = t0; i < npieces; i += 4 * step) {
      bf16x8 a[4], yy[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const long j = i + u * step;
        if (j < npieces) { a[u] = *(const bf16x8*)(gy + j * 8); if (has_y) yy[u] = *(const bf16x8*)(y + j * 8); }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const long j = i + u * step;
        if (j >= npieces) break;
        bf16x8 o = a[u];
        if (has_y) {
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            const float yv = bf2f((bf16_t)yy[u][k]);
            const float gv = ACT == RG_ACT_RELU ? (yv > 0.f ? bf2f((bf16_t)a[u][k]) : 0.f) : bf2f((bf16_t)a[u][k]) * act_bwd_from_y(yv, act, slope);
            const bf16_t r = f2bf(gv); o[k] = (short)r; sum[k] += bf2f(r);
          }
        } else {
#pragma unroll
          for (int k = 0; k < 8; ++k) sum[k] += bf2f((bf16_t)a[u][k]);
        }
        if (g) *(bf16x8*)(g + j * 8) = o;
      }
    }
  }
  if (!partial) return;
#pragma unroll
  for (int k = 0; k < 8; ++k) red[threadIdx.x * 8 + k] = sum[k];
  __syncthreads();
  // threads of one workgroup with equal (threadIdx.x % S8) share a channel group: thread grp < S8 sums them in a fixed order
  if ((int)threadIdx.x < S8 && (int)threadIdx.x < G8) {
    const int first = (int)(((long)threadIdx.x - (long)blockIdx.x * 256 % S8 + S8) % S8);  // smallest thread index of this block whose group == threadIdx.x
    float tot[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int t = first; t < 256; t += S8)
#pragma unroll
      for (int k = 0; k < 8; ++k) tot[k] += red[t * 8 + k];
#pragma unroll
    for (int k = 0; k < 8; ++k) partial[(long)blockIdx.x * lanes + threadIdx.x * 8 + k] = tot[k];
  }
}
static bool abb_dense(const CL5& t) {  // one voxel stride for the whole buffer, no gaps between lines / planes / samples
  return t.sh == (long)t.W * t.sw && t.sd == (long)t.H * t.sh && (t.N == 1 || t.sn == (long)t.D * t.sd) && t.sw % 8 == 0 && t.sw / 8 <= 256;
}
// out[c] = sum_b partial[b][c]: one workgroup per channel, fixed-order tree (deterministic)
__global__ void reduce_partials_kernel(const float* partial, float* out, int nblk, int lanes, int C) {
  __shared__ float red[256];
  const int c = blockIdx.x;
  float s = 0.f;
  for (int b = threadIdx.x; b < nblk; b += 256) s += partial[(long)b * lanes + c];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int k = 128; k > 0; k >>= 1) {
    if ((int)threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[c] = red[0];
}
static void abb_plan(const CL5& t, int* gx, int* gy, int* vpb, int* nblk) {
  const int G8 = t.lanes / 8;
  int x = 1; while (x < G8 && x < 32) x <<= 1;
  *gx = x; *gy = 256 / x;
  const long nvox = (long)t.N * t.D * t.H * t.W;
  long want = 1024; long v = (nvox + want - 1) / want; if (v < *gy * 4) v = *gy * 4;
  *vpb = (int)v; *nblk = (int)((nvox + v - 1) / v);
}
extern "C" int64_t rg_act_bwd_bias_workspace_bytes(const rg_tensor* gy) {
  CL5 t; if (rg_as_cl5(gy, &t, "act_bwd.gy")) return -1;
  int gx, gyy, vpb, nblk; abb_plan(t, &gx, &gyy, &vpb, &nblk);
  return (int64_t)nblk * t.lanes * 4;
}
extern "C" int rg_act_bwd_bias(const rg_tensor* gy, const rg_tensor* y, const rg_tensor* g, float* db, int act, float slope,
                               void* workspace, size_t ws_bytes, rg_stream s) {
  CL5 GY, Y, G; int rc;
  if ((rc = rg_as_cl5(gy, &GY, "act_bwd.gy"))) return rc;
  const int has_y = (act != RG_ACT_NONE);
  memset(&Y, 0, sizeof(Y)); memset(&G, 0, sizeof(G));
  if (has_y) { if ((rc = rg_as_cl5(y, &Y, "act_bwd.y"))) return rc;
    if (Y.N != GY.N || Y.C != GY.C || Y.D != GY.D || Y.H != GY.H || Y.W != GY.W) RG_FAIL(RG_ERR_BAD_SHAPE, "act_bwd: y/gy shape mismatch"); }
  if (g && g->ptr) { if ((rc = rg_as_cl5(g, &G, "act_bwd.g"))) return rc;
    if (G.N != GY.N || G.C != GY.C || G.D != GY.D || G.H != GY.H || G.W != GY.W) RG_FAIL(RG_ERR_BAD_SHAPE, "act_bwd: g/gy shape mismatch"); }
  int gx, gyy, vpb, nblk; abb_plan(GY, &gx, &gyy, &vpb, &nblk);
  float* partial = nullptr;
  if (db) {
    const size_t need = (size_t)nblk * GY.lanes * 4;
    if (!workspace || ws_bytes < need) RG_FAIL(RG_ERR_WORKSPACE, "act_bwd: workspace %zu < %zu", ws_bytes, need);
    partial = (float*)workspace;
  }
  const bool has_g = g && g->ptr;
  if (abb_dense(GY) && (!has_y || (abb_dense(Y) && Y.sw == GY.sw)) && (!has_g || (abb_dense(G) && G.sw == GY.sw))) {
    const int S8 = (int)(GY.sw / 8), G8 = GY.lanes / 8;
    const long npieces = (long)GY.N * GY.D * GY.H * GY.W * S8;
    // grid: at most nblk workgroups (the workspace holds nblk partial rows); total threads a multiple of S8
    long blocks = (npieces + 256 * 4 - 1) / (256 * 4); if (blocks > nblk) blocks = nblk; if (blocks < 1) blocks = 1;
    long step = blocks * 256; step = (step + S8 - 1) / S8 * S8;    // >= the thread count: some pieces are then visited by nobody? no: see below
    // step must EQUAL the thread count for full coverage; make the thread count a multiple of S8 instead: 256 * blocks is one iff
    // blocks % (S8 / gcd(256, S8)) == 0 — round blocks down to such a count (S8 <= 256)
    int gcd = 256, b2 = S8; while (b2) { const int t2 = gcd % b2; gcd = b2; b2 = t2; }
    const long q = S8 / gcd;
    if (blocks >= q) blocks = blocks / q * q; else blocks = q <= nblk ? q : 0;
    if (blocks > 0) {
      step = blocks * 256;
      if (partial && blocks < nblk && hipMemsetAsync(partial + blocks * GY.lanes, 0, (size_t)(nblk - blocks) * GY.lanes * 4, (hipStream_t)s) != hipSuccess)
        RG_FAIL(RG_ERR_HIP, "act_bwd: memset failed");
      if (has_y && act == RG_ACT_RELU)
        hipLaunchKernelGGL(act_bwd_bias_dense_kernel<RG_ACT_RELU>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)s, (const bf16_t*)GY.p, (const bf16_t*)Y.p,
                           has_g ? G.p : nullptr, partial, npieces, S8, G8, GY.lanes, act, slope, has_y, step);
      else
        hipLaunchKernelGGL(act_bwd_bias_dense_kernel<-1>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)s, (const bf16_t*)GY.p, has_y ? (const bf16_t*)Y.p : nullptr,
                           has_g ? G.p : nullptr, partial, npieces, S8, G8, GY.lanes, act, slope, has_y, step);
      RG_CHECK_LAUNCH();
      if (db) {
        hipLaunchKernelGGL(reduce_partials_kernel, dim3(GY.C), dim3(256), 0, (hipStream_t)s, partial, db, nblk, GY.lanes, GY.C);
        RG_CHECK_LAUNCH();
      }
      return RG_OK;
    }
  }
  dim3 grid(nblk, (GY.lanes / 8 + gx - 1) / gx), block(gx, gyy);
  hipLaunchKernelGGL(act_bwd_bias_kernel, grid, block, 256 * 8 * sizeof(float), (hipStream_t)s, GY, Y, G, partial, act, slope, vpb, has_y);
  RG_CHECK_LAUNCH();
  if (db) {
    hipLaunchKernelGGL(reduce_partials_kernel, dim3(GY.C), dim3(256), 0, (hipStream_t)s, partial, db, nblk, GY.lanes, GY.C);
    RG_CHECK_LAUNCH();
  }
  return RG_OK;
}

// ---------------------------------------------------------------------------- max pool
__global__ void maxpool_fwd_kernel(CL5 x, CL5 y, int fd, int fh, int fw) {
  const int G8 = y.lanes / 8;
  const long total = (long)y.N * y.D * y.H * y.W * G8;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int grp = (int)(i % G8); long t = i / G8;
    const int w = (int)(t % y.W); t /= y.W; const int h = (int)(t % y.H); t /= y.H; const int d = (int)(t % y.D); const int n = (int)(t / y.D);
    float m[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) m[k] = -INFINITY;
    for (int a = 0; a < fd; ++a) for (int b = 0; b < fh; ++b) for (int c = 0; c < fw; ++c) {
      const bf16x8 v = *(const bf16x8*)(x.p + n * x.sn + (long)(d * fd + a) * x.sd + (long)(h * fh + b) * x.sh + (long)(w * fw + c) * x.sw + grp * 8);
#pragma unroll
      for (int k = 0; k < 8; ++k) { const float f = bf2f((bf16_t)v[k]); if (f > m[k] || f != f) m[k] = f; }
    }
    bf16x8 o;
#pragma unroll
    for (int k = 0; k < 8; ++k) o[k] = (short)f2bf(m[k]);
    *(bf16x8*)(y.p + n * y.sn + d * y.sd + h * y.sh + w * y.sw + grp * 8) = o;
  }
}
__global__ void maxpool_bwd_kernel(CL5 x, CL5 gy, CL5 gx, int fd, int fh, int fw) {
  const int G8 = gy.lanes / 8;
  const long total = (long)gy.N * gy.D * gy.H * gy.W * G8;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int grp = (int)(i % G8); long t = i / G8;
    const int w = (int)(t % gy.W); t /= gy.W; const int h = (int)(t % gy.H); t /= gy.H; const int d = (int)(t % gy.D); const int n = (int)(t / gy.D);
    float m[8]; int arg[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { m[k] = -INFINITY; arg[k] = 0; }
    int idx = 0;
    for (int a = 0; a < fd; ++a) for (int b = 0; b < fh; ++b) for (int c = 0; c < fw; ++c, ++idx) {
      const bf16x8 v = *(const bf16x8*)(x.p + n * x.sn + (long)(d * fd + a) * x.sd + (long)(h * fh + b) * x.sh + (long)(w * fw + c) * x.sw + grp * 8);
#pragma unroll
      for (int k = 0; k < 8; ++k) { const float f = bf2f((bf16_t)v[k]); if (f > m[k] || f != f) { m[k] = f; arg[k] = idx; } }
    }
    const bf16x8 gv = *(const bf16x8*)(gy.p + n * gy.sn + d * gy.sd + h * gy.sh + w * gy.sw + grp * 8);
    idx = 0;
    for (int a = 0; a < fd; ++a) for (int b = 0; b < fh; ++b) for (int c = 0; c < fw; ++c, ++idx) {
      bf16x8 o;
#pragma unroll
      for (int k = 0; k < 8; ++k) o[k] = (arg[k] == idx) ? gv[k] : (short)0;
      *(bf16x8*)(gx.p + n * gx.sn + (long)(d * fd + a) * gx.sd + (long)(h * fh + b) * gx.sh + (long)(w * fw + c) * gx.sw + grp * 8) = o;
    }
  }
}
// zero every voxel of t outside the box [0,bd)x[0,bh)x[0,bw) (used for floor-cropped borders)
__global__ void zero_outside_kernel(CL5 t, int bd, int bh, int bw) {
  const int G8 = t.lanes / 8;
  const long total = (long)t.N * t.D * t.H * t.W * G8;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int grp = (int)(i % G8); long q = i / G8;
    const int w = (int)(q % t.W); q /= t.W; const int h = (int)(q % t.H); q /= t.H; const int d = (int)(q % t.D); const int n = (int)(q / t.D);
    if (d < bd && h < bh && w < bw) continue;
    *(bf16x8*)(t.p + n * t.sn + d * t.sd + h * t.sh + w * t.sw + grp * 8) = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
  }
}
static int pool_check(const CL5& X, const CL5& Y, const int32_t f[3], const char* what) {
  if (f[0] <= 0 || f[1] <= 0 || f[2] <= 0) RG_FAIL(RG_ERR_BAD_SHAPE, "%s: bad factor", what);
  if (Y.N != X.N || Y.C != X.C || Y.D != X.D / f[0] || Y.H != X.H / f[1] || Y.W != X.W / f[2])
    RG_FAIL(RG_ERR_BAD_SHAPE, "%s: output must be floor(input/factor): in (%d,%d,%d) out (%d,%d,%d)", what, X.D, X.H, X.W, Y.D, Y.H, Y.W);
  return RG_OK;
}
extern "C" int rg_maxpool3d_fwd(const rg_tensor* x, const rg_tensor* y, const int32_t factor[3], rg_stream s) {
  CL5 X, Y; int rc;
  if ((rc = rg_as_cl5(x, &X, "maxpool.x")) || (rc = rg_as_cl5(y, &Y, "maxpool.y")) || (rc = pool_check(X, Y, factor, "maxpool_fwd"))) return rc;
  const long total = (long)Y.N * Y.D * Y.H * Y.W * (Y.lanes / 8);
  hipLaunchKernelGGL(maxpool_fwd_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)s, X, Y, factor[0], factor[1], factor[2]);
  RG_CHECK_LAUNCH();
  return RG_OK;
}
extern "C" int rg_maxpool3d_bwd(const rg_tensor* x, const rg_tensor* gy, const rg_tensor* gx, const int32_t factor[3], rg_stream s) {
  CL5 X, GY, GX; int rc;
  if ((rc = rg_as_cl5(x, &X, "maxpool.x")) || (rc = rg_as_cl5(gy, &GY, "maxpool.gy")) || (rc = rg_as_cl5(gx, &GX, "maxpool.gx")) ||
      (rc = pool_check(X, GY, factor, "maxpool_bwd"))) return rc;
  if (GX.N != X.N || GX.C != X.C || GX.D != X.D || GX.H != X.H || GX.W != X.W) RG_FAIL(RG_ERR_BAD_SHAPE, "maxpool_bwd: gx shape != x shape");
  if (X.D % factor[0] || X.H % factor[1] || X.W % factor[2]) {
    const long tot = (long)GX.N * GX.D * GX.H * GX.W * (GX.lanes / 8);
    hipLaunchKernelGGL(zero_outside_kernel, dim3(grid_for(tot)), dim3(256), 0, (hipStream_t)s, GX, GY.D * factor[0], GY.H * factor[1], GY.W * factor[2]);
    RG_CHECK_LAUNCH();
  }
  const long total = (long)GY.N * GY.D * GY.H * GY.W * (GY.lanes / 8);
  hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)s, X, GY, GX, factor[0], factor[1], factor[2]);
  RG_CHECK_LAUNCH();
  return RG_OK;
}

// ---------------------------------------------------------------------------- upsample + crop + concat
struct UpcatGeom { int fd, fh, fw, ofd, ofh, ofw, ogd, ogh, ogw; };
__global__ void upcat_fwd_kernel(CL5 f, CL5 g, CL5 o, UpcatGeom q) {
  const int G8 = o.lanes / 8;
  const long total = (long)o.N * o.D * o.H * o.W * G8;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int grp = (int)(i % G8); long t = i / G8;
    const int w = (int)(t % o.W); t /= o.W; const int h = (int)(t % o.H); t /= o.H; const int d = (int)(t % o.D); const int n = (int)(t / o.D);
    const bf16_t* fp = f.p + n * f.sn + (long)(d + q.ofd) * f.sd + (long)(h + q.ofh) * f.sh + (long)(w + q.ofw) * f.sw;
    const bf16_t* gp = g.p + n * g.sn + (long)((d + q.ogd) / q.fd) * g.sd + (long)((h + q.ogh) / q.fh) * g.sh + (long)((w + q.ogw) / q.fw) * g.sw;
    bf16x8 v;
    const int c0 = grp * 8;
    if (c0 + 8 <= f.C) v = *(const bf16x8*)(fp + c0);
    else if (c0 >= f.C && ((c0 - f.C) & 7) == 0 && c0 - f.C + 8 <= g.lanes) v = *(const bf16x8*)(gp + (c0 - f.C));
    else {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int c = c0 + k;
        v[k] = c < f.C ? (short)fp[c] : (c < o.C ? (short)gp[c - f.C] : (short)0);
      }
    }
    *(bf16x8*)(o.p + n * o.sn + d * o.sd + h * o.sh + w * o.sw + c0) = v;
  }
}
// gf: full skip-sized gradient (zero outside the crop window)
__global__ void upcat_bwd_f_kernel(CL5 go, CL5 gf, UpcatGeom q) {
  const int G8 = gf.lanes / 8;
  const long total = (long)gf.N * gf.D * gf.H * gf.W * G8;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int grp = (int)(i % G8); long t = i / G8;
    const int w = (int)(t % gf.W); t /= gf.W; const int h = (int)(t % gf.H); t /= gf.H; const int d = (int)(t % gf.D); const int n = (int)(t / gf.D);
    const int od = d - q.ofd, oh = h - q.ofh, ow = w - q.ofw;
    bf16x8 v = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
    if ((unsigned)od < (unsigned)go.D && (unsigned)oh < (unsigned)go.H && (unsigned)ow < (unsigned)go.W) {
      const bf16_t* p = go.p + n * go.sn + (long)od * go.sd + (long)oh * go.sh + (long)ow * go.sw + grp * 8;
      if (grp * 8 + 8 <= gf.C) v = *(const bf16x8*)p;
      else {
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = (grp * 8 + k < gf.C) ? (short)p[k] : (short)0;
      }
    }
    *(bf16x8*)(gf.p + n * gf.sn + d * gf.sd + h * gf.sh + w * gf.sw + grp * 8) = v;
  }
}
// gg: sum over the factor^3 upsampled positions that survive the crop
__global__ void upcat_bwd_g_kernel(CL5 go, CL5 gg, int Cf, UpcatGeom q) {
  const int G8 = gg.lanes / 8;
  const long total = (long)gg.N * gg.D * gg.H * gg.W * G8;
  const bool vec = (Cf & 7) == 0;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int grp = (int)(i % G8); long t = i / G8;
    const int w = (int)(t % gg.W); t /= gg.W; const int h = (int)(t % gg.H); t /= gg.H; const int d = (int)(t % gg.D); const int n = (int)(t / gg.D);
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int a = 0; a < q.fd; ++a) { const int od = d * q.fd + a - q.ogd; if ((unsigned)od >= (unsigned)go.D) continue;
      for (int b = 0; b < q.fh; ++b) { const int oh = h * q.fh + b - q.ogh; if ((unsigned)oh >= (unsigned)go.H) continue;
        for (int c = 0; c < q.fw; ++c) { const int ow = w * q.fw + c - q.ogw; if ((unsigned)ow >= (unsigned)go.W) continue;
          const bf16_t* p = go.p + n * go.sn + (long)od * go.sd + (long)oh * go.sh + (long)ow * go.sw + Cf + grp * 8;
          if (vec) { const bf16x8 v = *(const bf16x8*)p;
#pragma unroll
            for (int k = 0; k < 8; ++k) acc[k] += bf2f((bf16_t)v[k]);
          } else {
#pragma unroll
            for (int k = 0; k < 8; ++k) if (grp * 8 + k < gg.C) acc[k] += bf2f(p[k]);
          }
        } } }
    bf16x8 o;
#pragma unroll
    for (int k = 0; k < 8; ++k) o[k] = (grp * 8 + k < gg.C) ? (short)f2bf(acc[k]) : (short)0;
    *(bf16x8*)(gg.p + n * gg.sn + d * gg.sd + h * gg.sh + w * gg.sw + grp * 8) = o;
  }
}
static int upcat_check(const CL5& F, const CL5& G, const CL5& O, const int32_t* fac, const int32_t* of, const int32_t* og, UpcatGeom* q) {
  if (O.C != F.C + G.C || O.N != F.N || O.N != G.N) RG_FAIL(RG_ERR_BAD_SHAPE, "upcat: channels %d != %d + %d", O.C, F.C, G.C);
  const int Od[3] = {O.D, O.H, O.W}, Fd[3] = {F.D, F.H, F.W}, Gd[3] = {G.D, G.H, G.W};
  for (int d = 0; d < 3; ++d) {
    if (fac[d] <= 0 || of[d] < 0 || og[d] < 0) RG_FAIL(RG_ERR_BAD_SHAPE, "upcat: bad factor/offset");
    if (of[d] + Od[d] > Fd[d]) RG_FAIL(RG_ERR_BAD_SHAPE, "upcat: skip crop outside f along dim %d", d);
    if (og[d] + Od[d] > Gd[d] * fac[d]) RG_FAIL(RG_ERR_BAD_SHAPE, "upcat: upsample crop outside g along dim %d", d);
  }
  q->fd = fac[0]; q->fh = fac[1]; q->fw = fac[2]; q->ofd = of[0]; q->ofh = of[1]; q->ofw = of[2]; q->ogd = og[0]; q->ogh = og[1]; q->ogw = og[2];
  return RG_OK;
}
extern "C" int rg_upcat_fwd(const rg_tensor* f, const rg_tensor* g, const rg_tensor* out, const int32_t factor[3],
                            const int32_t off_f[3], const int32_t off_g[3], rg_stream s) {
  CL5 F, G, O; int rc; UpcatGeom q;
  if ((rc = rg_as_cl5(f, &F, "upcat.f")) || (rc = rg_as_cl5(g, &G, "upcat.g")) || (rc = rg_as_cl5(out, &O, "upcat.out")) ||
      (rc = upcat_check(F, G, O, factor, off_f, off_g, &q))) return rc;
  const long total = (long)O.N * O.D * O.H * O.W * (O.lanes / 8);
  hipLaunchKernelGGL(upcat_fwd_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)s, F, G, O, q);
  RG_CHECK_LAUNCH();
  return RG_OK;
}
extern "C" int rg_upcat_bwd(const rg_tensor* gout, const rg_tensor* gf, const rg_tensor* gg, const int32_t factor[3],
                            const int32_t off_f[3], const int32_t off_g[3], rg_stream s) {
  CL5 F, G, O; int rc; UpcatGeom q;
  if ((rc = rg_as_cl5(gf, &F, "upcat.gf")) || (rc = rg_as_cl5(gg, &G, "upcat.gg")) || (rc = rg_as_cl5(gout, &O, "upcat.gout")) ||
      (rc = upcat_check(F, G, O, factor, off_f, off_g, &q))) return rc;
  long total = (long)F.N * F.D * F.H * F.W * (F.lanes / 8);
  hipLaunchKernelGGL(upcat_bwd_f_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)s, O, F, q);
  RG_CHECK_LAUNCH();
  total = (long)G.N * G.D * G.H * G.W * (G.lanes / 8);
  hipLaunchKernelGGL(upcat_bwd_g_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)s, O, G, F.C, q);
  RG_CHECK_LAUNCH();
  return RG_OK;
}

// ---------------------------------------------------------------------------- residual add
__global__ void add_kernel(CL5 a, CL5 b, CL5 y) {
  const int G8 = y.lanes / 8;
  const long total = (long)y.N * y.D * y.H * y.W * G8;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int grp = (int)(i % G8); long t = i / G8;
    const int w = (int)(t % y.W); t /= y.W; const int h = (int)(t % y.H); t /= y.H; const int d = (int)(t % y.D); const int n = (int)(t / y.D);
    const bf16x8 u = *(const bf16x8*)(a.p + n * a.sn + d * a.sd + h * a.sh + w * a.sw + grp * 8);
    const bf16x8 v = *(const bf16x8*)(b.p + n * b.sn + d * b.sd + h * b.sh + w * b.sw + grp * 8);
    bf16x8 o;
#pragma unroll
    for (int k = 0; k < 8; ++k) o[k] = (short)f2bf(bf2f((bf16_t)u[k]) + bf2f((bf16_t)v[k]));
    *(bf16x8*)(y.p + n * y.sn + d * y.sd + h * y.sh + w * y.sw + grp * 8) = o;
  }
}
extern "C" int rg_add(const rg_tensor* a, const rg_tensor* b, const rg_tensor* y, rg_stream s) {
  CL5 A, B, Y; int rc;
  if ((rc = rg_as_cl5(a, &A, "add.a")) || (rc = rg_as_cl5(b, &B, "add.b")) || (rc = rg_as_cl5(y, &Y, "add.y"))) return rc;
  if (A.N != Y.N || A.C != Y.C || A.D != Y.D || A.H != Y.H || A.W != Y.W || B.N != Y.N || B.C != Y.C || B.D != Y.D || B.H != Y.H || B.W != Y.W)
    RG_FAIL(RG_ERR_BAD_SHAPE, "add: shape mismatch");
  const long total = (long)Y.N * Y.D * Y.H * Y.W * (Y.lanes / 8);
  hipLaunchKernelGGL(add_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)s, A, B, Y);
  RG_CHECK_LAUNCH();
  return RG_OK;
}

// y = a + pad(b): b is a crop of a's extent starting at (od, oh, ow); outside it y = a.  The gradient of a tensor that is used
// whole by one consumer and centre-cropped by another (the U-Net's skip connections) in ONE pass instead of zero-fill + copy + a
// strided torch add.  a may be NULL (y = pad(b)).
__global__ void add_pad_kernel(CL5 a, CL5 b, CL5 y, int od, int oh, int ow, int has_a) {
  const int G8 = y.lanes / 8;
  const long total = (long)y.N * y.D * y.H * y.W * G8;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int grp = (int)(i % G8); long t = i / G8;
    const int w = (int)(t % y.W); t /= y.W; const int h = (int)(t % y.H); t /= y.H; const int d = (int)(t % y.D); const int n = (int)(t / y.D);
    bf16x8 u = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0}, v = u;
    if (has_a) u = *(const bf16x8*)(a.p + n * a.sn + d * a.sd + h * a.sh + w * a.sw + grp * 8);
    const int bd = d - od, bh = h - oh, bw = w - ow;
    const bool in = (unsigned)bd < (unsigned)b.D && (unsigned)bh < (unsigned)b.H && (unsigned)bw < (unsigned)b.W;
    if (in) v = *(const bf16x8*)(b.p + n * b.sn + bd * b.sd + bh * b.sh + bw * b.sw + grp * 8);
    bf16x8 o;
#pragma unroll
    for (int k = 0; k < 8; ++k) o[k] = has_a ? (in ? (short)f2bf(bf2f((bf16_t)u[k]) + bf2f((bf16_t)v[k])) : u[k]) : v[k];
    *(bf16x8*)(y.p + n * y.sn + d * y.sd + h * y.sh + w * y.sw + grp * 8) = o;
  }
}
extern "C" int rg_add_pad(const rg_tensor* a, const rg_tensor* b, const rg_tensor* y, const int32_t offset[3], rg_stream s) {
  CL5 A, B, Y; int rc;
  if ((rc = rg_as_cl5(b, &B, "add_pad.b")) || (rc = rg_as_cl5(y, &Y, "add_pad.y"))) return rc;
  const int has_a = a && a->ptr;
  memset(&A, 0, sizeof(A));
  if (has_a) {
    if ((rc = rg_as_cl5(a, &A, "add_pad.a"))) return rc;
    if (A.N != Y.N || A.C != Y.C || A.D != Y.D || A.H != Y.H || A.W != Y.W) RG_FAIL(RG_ERR_BAD_SHAPE, "add_pad: a / y shape mismatch");
  }
  if (B.N != Y.N || B.C != Y.C || offset[0] < 0 || offset[1] < 0 || offset[2] < 0 || offset[0] + B.D > Y.D || offset[1] + B.H > Y.H ||
      offset[2] + B.W > Y.W) RG_FAIL(RG_ERR_BAD_SHAPE, "add_pad: b does not fit inside y at the given offset");
  const long total = (long)Y.N * Y.D * Y.H * Y.W * (Y.lanes / 8);
  hipLaunchKernelGGL(add_pad_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)s, A, B, Y, offset[0], offset[1], offset[2], has_a);
  RG_CHECK_LAUNCH();
  return RG_OK;
}

// ---------------------------------------------------------------------------- (kh,kw) taps <-> channels of a 1-channel volume
// A C-channel conv whose other side has ONE channel (the generator's 7^3 stem and its composed 64->1 tail) re-reads the
// C-channel tensor once per tap.  Folding the KH*KW in-plane taps of the one-channel side into channels leaves a KD-tap
// convolution between C and KH*KW channels (49x fewer passes over the wide tensor for 7^3) plus these two cheap maps:
//   spread: y[n,d,h,w][kh*KW+kw] = x[n,d,h + s*kh, w + s*kw][0]   (0 outside x)
//   sum   : y[n,d,h,w][0] = bias + sum_{kh,kw} x[n,d,h + s*kh, w + s*kw][kh*KW+kw]   (terms outside x dropped)
// s = +1: im2col over (h,w) / the gather that finishes a conv;  s = -1: their adjoints.
// spread: one workgroup per (n, d, SP_TH x SP_TW output tile).  The (SP_TH+KH-1) x (SP_TW+KW-1) window of the one-channel side is staged
// in LDS as plain 2-byte scalars (one strided global load each, instead of 8 per 16-byte output group: the thread-per-group gather ran
// at 1.5 TB/s of output), then consecutive threads write consecutive 16-byte channel groups (voxel-major): fully coalesced stores.
#define SP_TH 4
#define SP_TW 64
__global__ __launch_bounds__(256) void diag_spread_kernel(CL5 x, CL5 y, int KH, int KW, int sgn, int tiles_w, int tiles_h) {
  extern __shared__ __attribute__((aligned(16))) bf16_t stile[];   // [RH*RW] window, then [lanes] per-channel window offsets (int)
  const int RW = SP_TW + KW - 1, RH = SP_TH + KH - 1, G8 = y.lanes / 8;
  int* coff = (int*)(stile + ((RH * RW + 7) & ~7));
  int b = blockIdx.x;
  const int tw = b % tiles_w; b /= tiles_w; const int th = b % tiles_h; b /= tiles_h; const int d = b % y.D; const int n = b / y.D;
  const int h0 = th * SP_TH, w0 = tw * SP_TW;
  const int ih0 = sgn > 0 ? h0 : h0 - (KH - 1), iw0 = sgn > 0 ? w0 : w0 - (KW - 1);
  for (int i = threadIdx.x; i < RH * RW; i += 256) {
    const int r = i / RW, c = i - r * RW;
    const int hh = ih0 + r, ww = iw0 + c;
    bf16_t v = 0;
    if ((unsigned)hh < (unsigned)x.H && (unsigned)ww < (unsigned)x.W) v = x.p[n * x.sn + d * x.sd + hh * x.sh + ww * x.sw];
    stile[i] = v;
  }
  if (threadIdx.x < y.lanes) {
    const int c = threadIdx.x, kh = c / KW, kw = c - kh * KW;
    coff[c] = c < KH * KW ? sgn * (kh * RW + kw) : (1 << 30);
  }
  __syncthreads();
  const int nitem = SP_TH * SP_TW * G8;
  for (int it = threadIdx.x; it < nitem; it += 256) {
    const int v = it / G8, g = it - v * G8;
    const int lh = v / SP_TW, lw = v - lh * SP_TW;
    const int h = h0 + lh, w = w0 + lw;
    if (h >= y.H || w >= y.W) continue;
    const int base = (sgn > 0 ? lh : lh + KH - 1) * RW + (sgn > 0 ? lw : lw + KW - 1);
    bf16x8 o;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int off = coff[g * 8 + k];
      o[k] = off == (1 << 30) ? (short)0 : (short)stile[base + off];
    }
    *(bf16x8*)(y.p + n * y.sn + d * y.sd + h * y.sh + w * y.sw + g * 8) = o;
  }
}
// sum: one workgroup per (n, d, TH x DS_TW output tile): the (TH+KH-1) x (DS_TW+KW-1) input voxel records are staged in LDS with
// coalesced 16-byte loads — ALL of a thread's loads of a batch are issued before the first LDS store (one load per iteration, waited
// for at once, was 21 serialised HBM latencies per workgroup: 0.6 TB/s) — then every thread sums its KH*KW diagonal entries.
// LDS voxel stride lanes + 4 (120 B for 49 channels): the 32 voxels of an output row start in 32 different banks.
#define DS_TW 32
template <int TH>
__global__ __launch_bounds__(TH * DS_TW) void diag_sum_kernel(CL5 x, CL5 y, int KH, int KW, int sgn, const float* bias, int tiles_w, int tiles_h) {
  extern __shared__ __attribute__((aligned(16))) bf16_t dtile[];
  constexpr int NT = TH * DS_TW, U = 8;
  const int RW = DS_TW + KW - 1, RH = TH + KH - 1, G8 = x.lanes / 8, LS = x.lanes + 4;
  int b = blockIdx.x;
  const int tw = b % tiles_w; b /= tiles_w; const int th = b % tiles_h; b /= tiles_h; const int d = b % y.D; const int n = b / y.D;
  const int h0 = th * TH, w0 = tw * DS_TW;
  const int ih0 = sgn > 0 ? h0 : h0 - (KH - 1), iw0 = sgn > 0 ? w0 : w0 - (KW - 1);
  const int total = RH * RW * G8;
  const bf16_t* xb = x.p + n * x.sn + d * x.sd;
  for (int base = 0; base < total; base += NT * U) {
    bf16x8 val[U]; int dst[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = base + u * NT + (int)threadIdx.x;
      val[u] = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0}; dst[u] = -1;
      if (i < total) {
        const int v = i / G8, g = i - v * G8, r = v / RW, c = v - r * RW;
        const int hh = ih0 + r, ww = iw0 + c;
        dst[u] = v * LS + g * 8;
        if ((unsigned)hh < (unsigned)x.H && (unsigned)ww < (unsigned)x.W) val[u] = *(const bf16x8*)(xb + hh * x.sh + ww * x.sw + g * 8);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (dst[u] >= 0) {   // (8-byte aligned: two 64-bit stores)
        const uint2 lo = {((const unsigned*)&val[u])[0], ((const unsigned*)&val[u])[1]}, hi = {((const unsigned*)&val[u])[2], ((const unsigned*)&val[u])[3]};
        *(uint2*)(dtile + dst[u]) = lo; *(uint2*)(dtile + dst[u] + 4) = hi;
      }
  }
  __syncthreads();
  const int lh = threadIdx.x / DS_TW, lw = threadIdx.x % DS_TW;
  const int h = h0 + lh, w = w0 + lw;
  if (h >= y.H || w >= y.W) return;
  float acc = bias ? bias[0] : 0.f;
  const int rb = sgn > 0 ? lh : lh + KH - 1, cb = sgn > 0 ? lw : lw + KW - 1;
  for (int kh = 0; kh < KH; ++kh) {
    const bf16_t* row = dtile + ((rb + sgn * kh) * RW + cb) * LS + kh * KW;
    for (int kw = 0; kw < KW; ++kw) acc += bf2f(row[sgn * kw * LS + kw]);
  }
  bf16x8 o = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
  o[0] = (short)f2bf(acc);
  *(bf16x8*)(y.p + n * y.sn + d * y.sd + h * y.sh + w * y.sw) = o;
}
static int diag_check(const rg_tensor* one, const rg_tensor* many, CL5* O, CL5* M, int KH, int KW, const char* what) {
  int rc;
  if ((rc = rg_as_cl5(one, O, what)) || (rc = rg_as_cl5(many, M, what))) return rc;
  if (KH < 1 || KW < 1 || O->C != 1 || M->C != KH * KW || O->N != M->N || O->D != M->D) {
    rg_set_error("%s: expects a 1-channel and a KH*KW-channel tensor with equal batch and depth", what);
    return RG_ERR_BAD_SHAPE;
  }
  return RG_OK;
}
extern "C" int rg_diag_spread(const rg_tensor* x, const rg_tensor* y, int kh, int kw, int sign, rg_stream s) {
  CL5 X, Y; int rc;
  if ((rc = diag_check(x, y, &X, &Y, kh, kw, "diag_spread"))) return rc;
  const int tiles_w = (Y.W + SP_TW - 1) / SP_TW, tiles_h = (Y.H + SP_TH - 1) / SP_TH;
  const size_t win = (size_t)(SP_TH + kh - 1) * (SP_TW + kw - 1);
  const size_t shm = ((win + 7) & ~(size_t)7) * 2 + (size_t)Y.lanes * 4;
  if (shm > 64 * 1024 || Y.lanes > 256) RG_FAIL(RG_ERR_UNSUPPORTED, "diag_spread: %dx%d taps need %zu bytes of LDS", kh, kw, shm);
  const long nblk = (long)Y.N * Y.D * tiles_h * tiles_w;
  if (nblk >= (1L << 31)) RG_FAIL(RG_ERR_UNSUPPORTED, "diag_spread: grid too large");
  if (nblk == 0) return RG_OK;
  hipLaunchKernelGGL(diag_spread_kernel, dim3((unsigned)nblk), dim3(256), shm, (hipStream_t)s, X, Y, kh, kw, sign < 0 ? -1 : 1, tiles_w, tiles_h);
  RG_CHECK_LAUNCH();
  return RG_OK;
}
extern "C" int rg_diag_sum(const rg_tensor* x, const rg_tensor* y, int kh, int kw, int sign, const float* bias, rg_stream s) {
  CL5 X, Y; int rc;
  if ((rc = diag_check(y, x, &Y, &X, kh, kw, "diag_sum"))) return rc;
  const size_t per_row = (size_t)(DS_TW + kw - 1) * (X.lanes + 4) * 2;
  const int TH = (8 + kh - 1) * per_row <= 64 * 1024 ? 8 : 4;
  const size_t shm = (size_t)(TH + kh - 1) * per_row;
  if (shm > 64 * 1024) RG_FAIL(RG_ERR_UNSUPPORTED, "diag_sum: %dx%d taps of %d lanes need %zu bytes of LDS", kh, kw, X.lanes, shm);
  const int tiles_w = (Y.W + DS_TW - 1) / DS_TW, tiles_h = (Y.H + TH - 1) / TH;
  const long nblk = (long)Y.N * Y.D * tiles_h * tiles_w;
  if (nblk >= (1L << 31)) RG_FAIL(RG_ERR_UNSUPPORTED, "diag_sum: grid too large");
  if (nblk == 0) return RG_OK;
  if (TH == 8) hipLaunchKernelGGL((diag_sum_kernel<8>), dim3((unsigned)nblk), dim3(8 * DS_TW), shm, (hipStream_t)s, X, Y, kh, kw, sign < 0 ? -1 : 1, bias, tiles_w, tiles_h);
  else hipLaunchKernelGGL((diag_sum_kernel<4>), dim3((unsigned)nblk), dim3(4 * DS_TW), shm, (hipStream_t)s, X, Y, kh, kw, sign < 0 ? -1 : 1, bias, tiles_w, tiles_h);
  RG_CHECK_LAUNCH();
  return RG_OK;
}

// ---------------------------------------------------------------------------- depth-to-space + add + activation (sub-pixel up-conv)
// out[n, o, c] = act( y[n, o, c] + z[n, (o + off) >> 1, parity(o + off) * C + c] )      (per dimension; parity = pd*4 + ph*2 + pw)
// z holds, on the low-resolution grid, the 8 parity classes of the upsampled operand's contribution (ops.upcat_conv3d).
// Backward: g = gout * act'(out) goes to y unchanged and to z through the inverse shuffle (cells the crop cut off get 0).
__global__ void d2s_add_act_kernel(CL5 y, CL5 z, CL5 out, int od, int oh, int ow, int act, float slope, int has_y) {
  const int G8 = out.lanes / 8;
  const long total = (long)out.N * out.D * out.H * out.W * G8;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int grp = (int)(i % G8); long t = i / G8;
    const int w = (int)(t % out.W); t /= out.W; const int h = (int)(t % out.H); t /= out.H; const int d = (int)(t % out.D); const int n = (int)(t / out.D);
    const int qd = d + od, qh = h + oh, qw = w + ow;
    const int par = ((qd & 1) << 2) | ((qh & 1) << 1) | (qw & 1);
    bf16x8 u = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
    if (has_y) u = *(const bf16x8*)(y.p + n * y.sn + d * y.sd + h * y.sh + w * y.sw + grp * 8);
    const bf16x8 v = *(const bf16x8*)(z.p + n * z.sn + (qd >> 1) * z.sd + (qh >> 1) * z.sh + (qw >> 1) * z.sw + (long)par * out.C + grp * 8);
    bf16x8 o;
#pragma unroll
    for (int k = 0; k < 8; ++k) o[k] = grp * 8 + k < out.C ? (short)f2bf(act_fwd(bf2f((bf16_t)u[k]) + bf2f((bf16_t)v[k]), act, slope)) : (short)0;
    *(bf16x8*)(out.p + n * out.sn + d * out.sd + h * out.sh + w * out.sw + grp * 8) = o;
  }
}
// one thread per (low-res cell, parity, 8 channels): writes gz everywhere (zeros where the crop cut) and g where in range
__global__ void d2s_add_act_bwd_kernel(CL5 gout, CL5 out, CL5 g, CL5 gz, int od, int oh, int ow, int act, float slope) {
  const int C = g.C, G8 = C / 8;
  const long total = (long)gz.N * gz.D * gz.H * gz.W * 8 * G8;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int grp = (int)(i % G8); long t = i / G8;
    const int par = (int)(t % 8); t /= 8;
    const int jw = (int)(t % gz.W); t /= gz.W; const int jh = (int)(t % gz.H); t /= gz.H; const int jd = (int)(t % gz.D); const int n = (int)(t / gz.D);
    const int d = 2 * jd + ((par >> 2) & 1) - od, h = 2 * jh + ((par >> 1) & 1) - oh, w = 2 * jw + (par & 1) - ow;
    bf16x8 o = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
    if ((unsigned)d < (unsigned)g.D && (unsigned)h < (unsigned)g.H && (unsigned)w < (unsigned)g.W) {
      const bf16x8 gv = *(const bf16x8*)(gout.p + n * gout.sn + d * gout.sd + h * gout.sh + w * gout.sw + grp * 8);
      bf16x8 yv = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
      if (act != RG_ACT_NONE) yv = *(const bf16x8*)(out.p + n * out.sn + d * out.sd + h * out.sh + w * out.sw + grp * 8);
#pragma unroll
      for (int k = 0; k < 8; ++k) o[k] = (short)f2bf(bf2f((bf16_t)gv[k]) * act_bwd_from_y(bf2f((bf16_t)yv[k]), act, slope));
      *(bf16x8*)(g.p + n * g.sn + d * g.sd + h * g.sh + w * g.sw + grp * 8) = o;
    }
    *(bf16x8*)(gz.p + n * gz.sn + jd * gz.sd + jh * gz.sh + jw * gz.sw + (long)par * C + grp * 8) = o;
  }
}
static int d2s_check(const CL5& Y, const CL5& Z, const int32_t off[3], const char* what) {
  if (Y.C % 8 || Z.C != 8 * Y.C || Z.N != Y.N) { rg_set_error("%s: z must have 8 x the channels of y (a multiple of 8)", what); return RG_ERR_BAD_SHAPE; }
  const int S[3] = {Y.D, Y.H, Y.W}, J[3] = {Z.D, Z.H, Z.W};
  for (int k = 0; k < 3; ++k)
    if (off[k] < 0 || off[k] > 1 || (S[k] - 1 + off[k]) / 2 >= J[k]) { rg_set_error("%s: output extent does not fit the low-resolution grid", what); return RG_ERR_BAD_SHAPE; }
  return RG_OK;
}
extern "C" int rg_d2s_add_act(const rg_tensor* y, const rg_tensor* z, const rg_tensor* out, const int32_t offset[3], int act, float slope, rg_stream s) {
  CL5 Y, Z, O; int rc;
  const int has_y = y && y->ptr;
  if ((rc = rg_as_cl5(z, &Z, "d2s.z")) || (rc = rg_as_cl5(out, &O, "d2s.out")) || (rc = d2s_check(O, Z, offset, "d2s_add_act"))) return rc;
  Y = O;
  if (has_y) {
    if ((rc = rg_as_cl5(y, &Y, "d2s.y"))) return rc;
    if (O.N != Y.N || O.C != Y.C || O.D != Y.D || O.H != Y.H || O.W != Y.W) RG_FAIL(RG_ERR_BAD_SHAPE, "d2s_add_act: y / out shape mismatch");
  }
  const long total = (long)O.N * O.D * O.H * O.W * (O.lanes / 8);
  hipLaunchKernelGGL(d2s_add_act_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)s, Y, Z, O, offset[0], offset[1], offset[2], act, slope, has_y);
  RG_CHECK_LAUNCH();
  return RG_OK;
}
extern "C" int rg_d2s_add_act_bwd(const rg_tensor* gout, const rg_tensor* out, const rg_tensor* g, const rg_tensor* gz, const int32_t offset[3],
                                  int act, float slope, rg_stream s) {
  CL5 GO, O, G, GZ; int rc;
  if ((rc = rg_as_cl5(gout, &GO, "d2s_bwd.gout")) || (rc = rg_as_cl5(g, &G, "d2s_bwd.g")) || (rc = rg_as_cl5(gz, &GZ, "d2s_bwd.gz")) ||
      (rc = d2s_check(G, GZ, offset, "d2s_add_act_bwd"))) return rc;
  memset(&O, 0, sizeof(O));
  if (act != RG_ACT_NONE && (rc = rg_as_cl5(out, &O, "d2s_bwd.out"))) return rc;
  if (GO.N != G.N || GO.C != G.C || GO.D != G.D || GO.H != G.H || GO.W != G.W) RG_FAIL(RG_ERR_BAD_SHAPE, "d2s_add_act_bwd: gout / g shape mismatch");
  const long total = (long)GZ.N * GZ.D * GZ.H * GZ.W * G.C;
  hipLaunchKernelGGL(d2s_add_act_bwd_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)s, GO, O, G, GZ, offset[0], offset[1], offset[2], act, slope);
  RG_CHECK_LAUNCH();
  return RG_OK;
}

// ---------------------------------------------------------------------------- sub-pixel weights of conv(nearest_upsample_x2(.))
// A 3^3 convolution applied to a nearest-x2 upsampled tensor is, per output parity p (one bit per dimension), a 2^3 convolution
// on the low-resolution tensor: taps that fall on the same low-resolution cell s are summed —
//   p = 0: s = 0 <- taps {0, 1}, s = 1 <- {2};   p = 1: s = 0 <- {0}, s = 1 <- {1, 2}      (per dimension)
// forward: w (O, C, 3,3,3) any dense strides -> wf (O, cf, 3,3,3) = w[:, :cf]  and  weff (8*O, C-cf, 2,2,2), both channels-last
// ([out][kd][kh][kw][in]); output channel of weff = ((pd*2+ph)*2+pw)*O + o.  backward: the adjoint, written in w's own strides.
struct SubpixW { long so, sc, sd, sh, sw; int O, C, cf; };
__device__ __forceinline__ void subpix_taps(int p, int s, int& t0, int& nt) { t0 = p == 0 ? (s == 0 ? 0 : 2) : (s == 0 ? 0 : 1); nt = (p == s) ? (p == 0 ? 2 : 2) : 1; }
__global__ void subpixel_w_fwd_kernel(const float* w, SubpixW q, float* wf, float* weff) {
  const int cg = q.C - q.cf;
  const long n_eff = (long)8 * q.O * 8 * cg, n_f = (long)q.O * 27 * q.cf;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n_eff + n_f; e += (long)gridDim.x * blockDim.x) {
    if (e < n_eff) {
      const int i = (int)(e % cg); long r = e / cg;
      const int s = (int)(r % 8); r /= 8;
      const int o = (int)(r % q.O), p = (int)(r / q.O);
      int d0, nd, h0, nh, w0, nw;
      subpix_taps((p >> 2) & 1, (s >> 2) & 1, d0, nd); subpix_taps((p >> 1) & 1, (s >> 1) & 1, h0, nh); subpix_taps(p & 1, s & 1, w0, nw);
      const float* base = w + o * q.so + (long)(q.cf + i) * q.sc;
      float acc = 0.f;
      for (int a = 0; a < nd; ++a) for (int b = 0; b < nh; ++b) for (int c = 0; c < nw; ++c)
        acc += base[(d0 + a) * q.sd + (h0 + b) * q.sh + (w0 + c) * q.sw];
      weff[e] = acc;
    } else {
      const long f = e - n_eff;
      const int i = (int)(f % q.cf); long r = f / q.cf;
      const int t = (int)(r % 27), o = (int)(r / 27);
      wf[f] = w[o * q.so + (long)i * q.sc + (t / 9) * q.sd + ((t / 3) % 3) * q.sh + (t % 3) * q.sw];
    }
  }
}
// the (p, s) pairs of a dimension that contain tap t: t = 0: (0,0), (1,0);  t = 1: (0,0), (1,1);  t = 2: (0,1), (1,1)
__global__ void subpixel_w_bwd_kernel(const float* gwf, const float* gweff, SubpixW q, float* dw) {
  const int cg = q.C - q.cf;
  const long n = (long)q.O * 27 * q.C;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) {
    const int c = (int)(e % q.C); long r = e / q.C;
    const int t = (int)(r % 27), o = (int)(r / 27);
    const int td = t / 9, th = (t / 3) % 3, tw = t % 3;
    float v = 0.f;
    if (c < q.cf) { if (gwf) v = gwf[((long)o * 27 + t) * q.cf + c]; }
    else if (gweff) {
      const int i = c - q.cf;
      for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int k = 0; k < 2; ++k) {
        // a-th pair of each dimension: p = a, s = (t == 0 ? 0 : t == 2 ? 1 : a)
        const int pd = a, sd = td == 0 ? 0 : td == 2 ? 1 : a, ph = b, sh = th == 0 ? 0 : th == 2 ? 1 : b, pw = k, sw = tw == 0 ? 0 : tw == 2 ? 1 : k;
        const int p = (pd << 2) | (ph << 1) | pw, s = (sd << 2) | (sh << 1) | sw;
        v += gweff[(((long)p * q.O + o) * 8 + s) * cg + i];
      }
    }
    dw[o * q.so + (long)c * q.sc + td * q.sd + th * q.sh + tw * q.sw] = v;
  }
}
// float4 forms for channels-last weights (input-channel stride 1, split and channel counts multiples of 4): four consecutive input
// channels per thread, 32-bit index arithmetic — the scalar kernels above spend their time in 64-bit divisions (0.33 / 0.17 ms for
// config U's 3024 -> 432 layer, 450 MB each way).
__device__ __forceinline__ float4 f4add(float4 a, const float4 b) { a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; return a; }
__global__ __launch_bounds__(256) void subpixel_w_fwd4_kernel(const float* w, SubpixW q, float* wf, float* weff) {
  const unsigned cg4 = (unsigned)(q.C - q.cf) / 4, cf4 = (unsigned)q.cf / 4;
  const unsigned n_eff = 64u * (unsigned)q.O * cg4, n_f = (unsigned)q.O * 27u * cf4;
  for (unsigned e = blockIdx.x * blockDim.x + threadIdx.x; e < n_eff + n_f; e += gridDim.x * blockDim.x) {
    if (e < n_eff) {
      const unsigned i4 = e % cg4; unsigned r = e / cg4;
      const unsigned s = r & 7; r >>= 3;
      const unsigned o = r % (unsigned)q.O, p = r / (unsigned)q.O;
      int d0, nd, h0, nh, w0, nw;
      subpix_taps((p >> 2) & 1, (s >> 2) & 1, d0, nd); subpix_taps((p >> 1) & 1, (s >> 1) & 1, h0, nh); subpix_taps(p & 1, s & 1, w0, nw);
      const float* base = w + (long)o * q.so + q.cf + 4 * i4;
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int a = 0; a < nd; ++a) for (int b = 0; b < nh; ++b) for (int c = 0; c < nw; ++c)
        acc = f4add(acc, *(const float4*)(base + (d0 + a) * q.sd + (h0 + b) * q.sh + (w0 + c) * q.sw));
      ((float4*)weff)[e] = acc;
    } else {
      const unsigned f = e - n_eff;
      const unsigned i4 = f % cf4; unsigned r = f / cf4;
      const unsigned t = r % 27u, o = r / 27u;
      ((float4*)wf)[f] = *(const float4*)(w + (long)o * q.so + 4 * i4 + (t / 9) * q.sd + ((t / 3) % 3) * q.sh + (t % 3) * q.sw);
    }
  }
}
__global__ __launch_bounds__(256) void subpixel_w_bwd4_kernel(const float* gwf, const float* gweff, SubpixW q, float* dw) {
  const unsigned C4 = (unsigned)q.C / 4, cf4 = (unsigned)q.cf / 4, cg4 = C4 - cf4;
  const unsigned n = (unsigned)q.O * 27u * C4;
  for (unsigned e = blockIdx.x * blockDim.x + threadIdx.x; e < n; e += gridDim.x * blockDim.x) {
    const unsigned c4 = e % C4; unsigned r = e / C4;
    const unsigned t = r % 27u, o = r / 27u;
    const int td = t / 9, th = (t / 3) % 3, tw = t % 3;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c4 < cf4) { if (gwf) v = ((const float4*)gwf)[(o * 27u + t) * cf4 + c4]; }
    else if (gweff) {
      const unsigned i4 = c4 - cf4;
      for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int k = 0; k < 2; ++k) {
        const int pd = a, sd = td == 0 ? 0 : td == 2 ? 1 : a, ph = b, sh = th == 0 ? 0 : th == 2 ? 1 : b, pw = k, sw = tw == 0 ? 0 : tw == 2 ? 1 : k;
        const unsigned p = (pd << 2) | (ph << 1) | pw, sx = (sd << 2) | (sh << 1) | sw;
        v = f4add(v, ((const float4*)gweff)[((p * (unsigned)q.O + o) * 8u + sx) * cg4 + i4]);
      }
    }
    *(float4*)(dw + (long)o * q.so + 4 * c4 + td * q.sd + th * q.sh + tw * q.sw) = v;
  }
}
static bool subpix_vec4(const SubpixW& q, const void* a, const void* b, const void* c) {
  return q.sc == 1 && q.cf % 4 == 0 && q.C % 4 == 0 && q.so % 4 == 0 && q.sd % 4 == 0 && q.sh % 4 == 0 && q.sw % 4 == 0 &&
         (((size_t)a | (size_t)b | (size_t)c) & 15) == 0 && (double)q.O * 64.0 * q.C < 4.0e9;
}
static int subpix_desc(const rg_tensor* w, int cf, SubpixW* q, const char* what) {
  if (!w || !w->ptr || w->dtype != RG_F32 || w->rank != 5 || w->sizes[2] != 3 || w->sizes[3] != 3 || w->sizes[4] != 3)
    { rg_set_error("%s: expected an fp32 (O, C, 3, 3, 3) weight", what); return RG_ERR_BAD_SHAPE; }
  if (cf < 0 || cf >= w->sizes[1]) { rg_set_error("%s: split %d outside (0, %ld)", what, cf, (long)w->sizes[1]); return RG_ERR_BAD_SHAPE; }
  q->so = w->strides[0]; q->sc = w->strides[1]; q->sd = w->strides[2]; q->sh = w->strides[3]; q->sw = w->strides[4];
  q->O = (int)w->sizes[0]; q->C = (int)w->sizes[1]; q->cf = cf;
  return RG_OK;
}
extern "C" int rg_subpixel_weights(const rg_tensor* w, int cf, float* wf, float* weff, rg_stream s) {
  SubpixW q; int rc;
  if ((rc = subpix_desc(w, cf, &q, "subpixel_weights"))) return rc;
  if (!weff || (cf > 0 && !wf)) RG_FAIL(RG_ERR_BAD_SHAPE, "subpixel_weights: null output");
  const long n = (long)64 * q.O * (q.C - q.cf) + (long)q.O * 27 * q.cf;
  long blocks = (n + 255) / 256; if (blocks > 65536) blocks = 65536;
  if (subpix_vec4(q, w->ptr, wf, weff)) {
    blocks = (n / 4 + 255) / 256; if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(subpixel_w_fwd4_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)s, (const float*)w->ptr, q, wf, weff);
  } else
  hipLaunchKernelGGL(subpixel_w_fwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)s, (const float*)w->ptr, q, wf, weff);
  RG_CHECK_LAUNCH();
  return RG_OK;
}
extern "C" int rg_subpixel_weights_bwd(const float* g_wf, const float* g_weff, const rg_tensor* dw, int cf, rg_stream s) {
  SubpixW q; int rc;
  if ((rc = subpix_desc(dw, cf, &q, "subpixel_weights_bwd"))) return rc;
  const long n = (long)q.O * 27 * q.C;
  long blocks = (n + 255) / 256; if (blocks > 65536) blocks = 65536;
  if (subpix_vec4(q, dw->ptr, g_wf, g_weff)) {
    blocks = (n / 4 + 255) / 256; if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(subpixel_w_bwd4_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)s, g_wf, g_weff, q, (float*)dw->ptr);
  } else
  hipLaunchKernelGGL(subpixel_w_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)s, g_wf, g_weff, q, (float*)dw->ptr);
  RG_CHECK_LAUNCH();
  return RG_OK;
}

// ---------------------------------------------------------------------------- fused Adam (torch.optim.Adam, no amsgrad)
__device__ __forceinline__ void adam_one(float& pi, float gi, float& mi, float& vi, float lr, float b1, float b2, float eps, float wd,
                                         float bc1, float bc2_sqrt, float gscale) {
  gi *= gscale;
  if (wd != 0.f) gi += wd * pi;
  mi = b1 * mi + (1.f - b1) * gi;
  vi = b2 * vi + (1.f - b2) * gi * gi;
  const float denom = sqrtf(vi) / bc2_sqrt + eps;
  pi = pi - (lr / bc1) * (mi / denom);
}
// 16 bytes per lane and stream (7 streams of 4 B/element: the step is pure HBM traffic); `vec` = all four pointers 16-B aligned
// `pk`: also refresh the bf16 forward operand of a channels-last conv weight while the updated values are in registers
// (storage order [row][tap][k]; packed [row][tap][k padded to red32] resp. FLAT rows of rowlen): pack_cl_rows_kernel's pass
// over the master weights disappears.  Pad lanes / rows of the packed buffer keep the zeros of its first full packing.
struct AdamPack { bf16_t* out; int red, red32, ntap; long rowlen; };
__device__ __forceinline__ void adam_four(float4& pp, const float4& gg, float4& mm, float4& vv, float lr, float b1, float b2, float eps, float wd,
                                          float bc1, float bc2_sqrt, float gscale) {
  adam_one(pp.x, gg.x, mm.x, vv.x, lr, b1, b2, eps, wd, bc1, bc2_sqrt, gscale);
  adam_one(pp.y, gg.y, mm.y, vv.y, lr, b1, b2, eps, wd, bc1, bc2_sqrt, gscale);
  adam_one(pp.z, gg.z, mm.z, vv.z, lr, b1, b2, eps, wd, bc1, bc2_sqrt, gscale);
  adam_one(pp.w, gg.w, mm.w, vv.w, lr, b1, b2, eps, wd, bc1, bc2_sqrt, gscale);
}
// address of element e = 4 i of the packed forward operand: 32-bit unsigned divisions (n < 2^32, host-checked)
__device__ __forceinline__ bf16_t* adam_pack_addr(const AdamPack& pk, unsigned e) {
  const unsigned rt = e / (unsigned)pk.red, k = e - rt * (unsigned)pk.red;
  const unsigned row = rt / (unsigned)pk.ntap, tap = rt - row * (unsigned)pk.ntap;
  return pk.out + (long)row * pk.rowlen + (long)tap * pk.red32 + k;
}
__global__ __launch_bounds__(256) void adam_kernel(float* p, const float* g, float* m, float* v, long n, float lr, float b1, float b2, float eps,
                                                  float wd, float bc1, float bc2_sqrt, float gscale, int vec, AdamPack pk) {
  const long tid = (long)blockIdx.x * blockDim.x + threadIdx.x, nthr = (long)gridDim.x * blockDim.x;
  const long n4 = vec ? n / 4 : 0;
  // one 16-byte piece per stream and thread and iteration: two in flight (8 loads) measured SLOWER (1.71 -> 1.91 ms per U-Net step)
  for (long i = tid; i < n4; i += nthr) {
    float4 pp = ((float4*)p)[i], mm = ((float4*)m)[i], vv = ((float4*)v)[i];
    const float4 gg = ((const float4*)g)[i];
    adam_four(pp, gg, mm, vv, lr, b1, b2, eps, wd, bc1, bc2_sqrt, gscale);
    ((float4*)p)[i] = pp; ((float4*)m)[i] = mm; ((float4*)v)[i] = vv;
    if (pk.out)   // red % 4 == 0 (host-checked): the four values share (row, tap)
      *(bf16x4*)adam_pack_addr(pk, (unsigned)(i * 4)) = (bf16x4){(short)f2bf(pp.x), (short)f2bf(pp.y), (short)f2bf(pp.z), (short)f2bf(pp.w)};
  }
  for (long q = n4 * 4 + tid; q < n; q += nthr) {
    float pi = p[q], mi = m[q], vi = v[q];
    adam_one(pi, g[q], mi, vi, lr, b1, b2, eps, wd, bc1, bc2_sqrt, gscale);
    p[q] = pi; m[q] = mi; v[q] = vi;
  }
}
// ---- multi-tensor form: ONE launch updates up to ADAM_MT tensors (the U-Net has 32 parameter tensors, the CycleGAN's four networks 96: a launch per tensor
// is ~5 us of dispatch per few-KB bias).  The item table travels by value in the kernel arguments; a workgroup finds its item by a scan of the
// (at most 24-entry) block prefix.  Same arithmetic, element by element, as adam_kernel.
#define ADAM_MT 24
struct AdamItem { float* p; const float* g; float* m; float* v; long n; AdamPack pk; int vec; int blk0; };
struct AdamTable { AdamItem it[ADAM_MT]; int n_items; int nblocks; };
__global__ __launch_bounds__(256) void adam_multi_kernel(AdamTable T, float lr, float b1, float b2, float eps, float wd, float bc1, float bc2_sqrt, float gscale, const float* hyp) {
  if (hyp) { lr = hyp[0]; bc1 = hyp[1]; bc2_sqrt = hyp[2]; }   // the step-dependent values from device memory: a launch captured in a HIP graph stays valid
  int k = 0;
  while (k + 1 < T.n_items && (int)blockIdx.x >= T.it[k + 1].blk0) ++k;
  const AdamItem& I = T.it[k];
  const int nb = (k + 1 < T.n_items ? T.it[k + 1].blk0 : T.nblocks) - I.blk0;
  const long tid = (long)((int)blockIdx.x - I.blk0) * 256 + threadIdx.x, nthr = (long)nb * 256;
  const long n4 = I.vec ? I.n / 4 : 0;
  for (long i = tid; i < n4; i += nthr) {
    float4 pp = ((float4*)I.p)[i], mm = ((float4*)I.m)[i], vv = ((float4*)I.v)[i];
    const float4 gg = ((const float4*)I.g)[i];
    adam_four(pp, gg, mm, vv, lr, b1, b2, eps, wd, bc1, bc2_sqrt, gscale);
    ((float4*)I.p)[i] = pp; ((float4*)I.m)[i] = mm; ((float4*)I.v)[i] = vv;
    if (I.pk.out) *(bf16x4*)adam_pack_addr(I.pk, (unsigned)(i * 4)) = (bf16x4){(short)f2bf(pp.x), (short)f2bf(pp.y), (short)f2bf(pp.z), (short)f2bf(pp.w)};
  }
  for (long q = n4 * 4 + tid; q < I.n; q += nthr) {
    float pi = I.p[q], mi = I.m[q], vi = I.v[q];
    adam_one(pi, I.g[q], mi, vi, lr, b1, b2, eps, wd, bc1, bc2_sqrt, gscale);
    I.p[q] = pi; I.m[q] = mi; I.v[q] = vi;
  }
}
static int adam_pack_of(void* packed, const rg_pack_params* pp, int64_t n, AdamPack* pk) {
  memset(pk, 0, sizeof(*pk));
  if (!packed) return RG_OK;
  int rows, red;
  if (!pp || pack_dims(pp, &rows, &red) || !pp->src_channels_last || pp->swap || (red & 3) || (((uintptr_t)packed) & 7))
    RG_FAIL(RG_ERR_BAD_SHAPE, "adam_pack: needs a channels-last source, swap = 0 and a reduced channel count that is a multiple of 4");
  for (int d = 0; d < 3; ++d) if (pp->ntap[d] != pp->K[d] || pp->k0[d] != 0 || pp->kstep[d] != 1) RG_FAIL(RG_ERR_BAD_SHAPE, "adam_pack: full tap set only");
  if ((int64_t)rows * pp->K[0] * pp->K[1] * pp->K[2] * red != n) RG_FAIL(RG_ERR_BAD_SHAPE, "adam_pack: packing parameters do not match n");
  pk->out = (bf16_t*)packed; pk->red = red; pk->red32 = round_up(red, pack_align(pp)); pk->ntap = pp->K[0] * pp->K[1] * pp->K[2];
  pk->rowlen = pack_rowlen(pp, red);
  return RG_OK;
}
// items: n_items x { p, g, m, v, packed (or NULL), pack params (or NULL) } with n[i] elements each; one set of hyper-parameters and ONE step count for all
extern "C" int rg_adam_step_multi(float* const* p, const float* const* g, float* const* m, float* const* v, const int64_t* n, void* const* packed,
                                  const rg_pack_params* const* pp, int32_t n_items, float lr, float beta1, float beta2, float eps, float weight_decay, int32_t step,
                                  float grad_scale, rg_stream s) {
  return rg_adam_step_multi_hyp(p, g, m, v, n, packed, pp, n_items, lr, beta1, beta2, eps, weight_decay, step, grad_scale, nullptr, s);
}
// hyp (device, 3 floats, or NULL): { lr, 1 - beta1^step, sqrt(1 - beta2^step) } read BY THE KERNEL instead of the by-value lr / step — the caller keeps them
// current on the device (FusedAdam(capturable=True)), so that the launch can be captured in a HIP graph and replayed step after step
extern "C" int rg_adam_step_multi_hyp(float* const* p, const float* const* g, float* const* m, float* const* v, const int64_t* n, void* const* packed,
                                      const rg_pack_params* const* pp, int32_t n_items, float lr, float beta1, float beta2, float eps, float weight_decay, int32_t step,
                                      float grad_scale, const float* hyp, rg_stream s) {
  if (n_items < 0 || (step < 1 && !hyp) || (n_items && (!p || !g || !m || !v || !n))) RG_FAIL(RG_ERR_BAD_SHAPE, "adam_multi: bad arguments");
  if (step < 1) step = 1;
  const float bc1 = 1.f - powf(beta1, (float)step), bc2 = 1.f - powf(beta2, (float)step);
  int done = 0;
  while (done < n_items) {
    AdamTable T; memset(&T, 0, sizeof(T));
    int blocks = 0;
    while (done < n_items && T.n_items < ADAM_MT) {
      const int i = done++;
      if (n[i] == 0) continue;
      if (!p[i] || !g[i] || !m[i] || !v[i] || n[i] < 0) RG_FAIL(RG_ERR_BAD_SHAPE, "adam_multi: null tensor %d", i);
      AdamItem& I = T.it[T.n_items];
      I.p = p[i]; I.g = g[i]; I.m = m[i]; I.v = v[i]; I.n = n[i];
      int rc = adam_pack_of(packed ? packed[i] : nullptr, pp ? pp[i] : nullptr, n[i], &I.pk);
      if (rc) return rc;
      I.vec = ((((uintptr_t)p[i]) | ((uintptr_t)g[i]) | ((uintptr_t)m[i]) | ((uintptr_t)v[i])) & 15) == 0;
      if (I.pk.out && (!I.vec || (n[i] & 3) || n[i] >= (1LL << 32))) RG_FAIL(RG_ERR_BAD_SHAPE, "adam_pack: 16-byte aligned tensors with a multiple of 4 (and fewer than 2^32) elements expected");
      I.blk0 = blocks;
      blocks += grid_for(I.vec ? (n[i] + 3) / 4 : n[i], 256, 8192);
      ++T.n_items;
    }
    if (!T.n_items) break;
    T.nblocks = blocks;
    hipLaunchKernelGGL(adam_multi_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)s, T, lr, beta1, beta2, eps, weight_decay, bc1, sqrtf(bc2), grad_scale, hyp);
    RG_CHECK_LAUNCH();
  }
  return RG_OK;
}
// any non-finite value in g -> *flag = 1 (never cleared here: one flag collects all the gradient tensors of a step).  Loss-scaled
// f16-storage training (raygun_amd/amp.py): an activation gradient that left half's range reaches the weight gradients as inf / NaN.
__global__ __launch_bounds__(256) void nonfinite_kernel(const float* g, long n, float* flag, int vec) {
  const long tid = (long)blockIdx.x * blockDim.x + threadIdx.x, nthr = (long)gridDim.x * blockDim.x;
  const long n4 = vec ? n / 4 : 0;
  bool bad = false;
  for (long i = tid; i < n4; i += nthr) {
    const float4 v = ((const float4*)g)[i];
    // (x - x) is 0 for every finite x and NaN for inf / NaN
    bad |= ((v.x - v.x) + (v.y - v.y) + (v.z - v.z) + (v.w - v.w)) != 0.f;
  }
  for (long q = n4 * 4 + tid; q < n; q += nthr) bad |= (g[q] - g[q]) != 0.f;
  if (__any(bad) && (threadIdx.x & 63) == 0) *flag = 1.f;
}
extern "C" int rg_nonfinite_check(const float* g, int64_t n, float* flag, rg_stream s) {
  if (!flag || (n > 0 && !g) || n < 0) RG_FAIL(RG_ERR_BAD_SHAPE, "nonfinite_check: bad arguments");
  if (n == 0) return RG_OK;
  const int vec = (((uintptr_t)g) & 15) == 0;
  hipLaunchKernelGGL(nonfinite_kernel, dim3(grid_for(vec ? (n + 3) / 4 : n, 256, 4096)), dim3(256), 0, (hipStream_t)s, g, (long)n, flag, vec);
  RG_CHECK_LAUNCH();
  return RG_OK;
}
extern "C" int rg_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                            float weight_decay, int32_t step, float grad_scale, rg_stream s) {
  return rg_adam_step_pack(p, g, m, v, n, lr, beta1, beta2, eps, weight_decay, step, grad_scale, nullptr, nullptr, s);
}
extern "C" int rg_adam_step_pack(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                                 float weight_decay, int32_t step, float grad_scale, void* packed, const rg_pack_params* pp, rg_stream s) {
  if (n == 0) return RG_OK;
  AdamPack pk; memset(&pk, 0, sizeof(pk));
  if (packed) {
    int rows, red;
    if (!pp || pack_dims(pp, &rows, &red) || !pp->src_channels_last || pp->swap || (red & 3) || (((uintptr_t)packed) & 7))
      RG_FAIL(RG_ERR_BAD_SHAPE, "adam_pack: needs a channels-last source, swap = 0 and a reduced channel count that is a multiple of 4");
    for (int d = 0; d < 3; ++d) if (pp->ntap[d] != pp->K[d] || pp->k0[d] != 0 || pp->kstep[d] != 1) RG_FAIL(RG_ERR_BAD_SHAPE, "adam_pack: full tap set only");
    if ((int64_t)rows * pp->K[0] * pp->K[1] * pp->K[2] * red != n) RG_FAIL(RG_ERR_BAD_SHAPE, "adam_pack: packing parameters do not match n");
    pk.out = (bf16_t*)packed; pk.red = red; pk.red32 = round_up(red, pack_align(pp)); pk.ntap = pp->K[0] * pp->K[1] * pp->K[2];
    pk.rowlen = pack_rowlen(pp, red);
  }
  if (!p || !g || !m || !v || n < 0 || step < 1) RG_FAIL(RG_ERR_BAD_SHAPE, "adam: bad arguments");
  const float bc1 = 1.f - powf(beta1, (float)step), bc2 = 1.f - powf(beta2, (float)step);
  const int vec = ((((uintptr_t)p) | ((uintptr_t)g) | ((uintptr_t)m) | ((uintptr_t)v)) & 15) == 0;
  if (pk.out && (!vec || (n & 3) || n >= (1LL << 32))) RG_FAIL(RG_ERR_BAD_SHAPE, "adam_pack: 16-byte aligned tensors with a multiple of 4 (and fewer than 2^32) elements expected");
  hipLaunchKernelGGL(adam_kernel, dim3(grid_for(vec ? (n + 3) / 4 : n, 256, 8192)), dim3(256), 0, (hipStream_t)s, p, g, m, v, (long)n, lr, beta1, beta2, eps,
                     weight_decay, bc1, sqrtf(bc2), grad_scale, vec, pk);
  RG_CHECK_LAUNCH();
  return RG_OK;
}
