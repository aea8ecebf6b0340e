// Shared device/host helpers for libraygun_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include "../../include/raygun_hip.h"

typedef __attribute__((ext_vector_type(8))) short bf16x8;   // 8 bf16 = 16 B
typedef __attribute__((ext_vector_type(4))) short bf16x4;   // 4 bf16 = 8 B
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef unsigned short bf16_t;

#define RG_WAVE 64

// ---- error plumbing -------------------------------------------------------
void rg_set_error(const char* fmt, ...);
void rg_set_kernel_name(const char* fmt, ...);  // name of the dominant kernel of the last call (profiling aid)
void rg_prof_mark(hipStream_t st);              // HIP event before / after the dominant kernel when rg_profile_enable(1)
// Diagnostic switches (rg_set_option / the RG_* environment variables read ONCE when the library is loaded — never in a launch path).
// 0 = production default everywhere.
enum { RG_OPT_FM = 0, RG_OPT_NO_HALO, RG_OPT_NO_THIN, RG_OPT_NO_THIN3, RG_OPT_NO_WLINES, RG_OPT_NO_KWP, RG_OPT_DEBUG_BOUNDS, RG_OPT_GEMM5, RG_OPT_NO_WGEMM, RG_OPT_MWS_WGS, RG_OPT_MWS_WMUL, RG_OPT_SK, RG_OPT_COUNT };
int rg_opt(int which);
// "once per device" latch for hipFuncSetAttribute (a function attribute belongs to the device that was current when it was set):
// `fn` runs exactly once per (latch, current device), under the latch lock: a second host thread (autograd's) launching the same
// kernel waits until the attribute is set.
struct rg_device_latch { unsigned long long done[4] = {0, 0, 0, 0}; };
void rg_once_per_device(rg_device_latch* l, void (*fn)());
#define RG_FAIL(code, ...) do { rg_set_error(__VA_ARGS__); return (code); } while (0)
#define RG_CHECK_LAUNCH() do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) { \
  rg_set_error("%s:%d HIP launch error: %s", __FILE__, __LINE__, hipGetErrorString(e_)); return RG_ERR_HIP; } } while (0)

// ---- the 16-bit storage type of activations and packed weights -------------------------------------------------------------
// One source, two builds: bf16 (libraygun_hip.so, the training default: fp32's exponent range) and, with -DRG_STORE_F16, IEEE half
// (libraygun_hip_f16.so: 11 significant bits instead of 8 — the block runner's precision mode).  The kernels never
// look inside an element except through these three definitions; `bf16_t` / `bf16x8` are the 16-bit containers in both builds,
// v_mfma_f32_16x16x32_f16 has the bf16 form's rate, tile shapes and lane maps (MI355X_MICROARCH.md, Matrix cores).
#ifdef RG_STORE_F16
typedef __attribute__((ext_vector_type(8))) _Float16 rg_h8;
#define RG_ACT_DTYPE RG_F16
#define RG_MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(rg_h8, (a)), __builtin_bit_cast(rg_h8, (b)), (c), 0, 0, 0)
__device__ __forceinline__ float bf2f(bf16_t v) { return (float)__builtin_bit_cast(_Float16, v); }
__device__ __forceinline__ bf16_t f2bf(float f) { return __builtin_bit_cast(bf16_t, (_Float16)f); }  // v_cvt_f16_f32: RNE, overflow -> inf, NaN stays NaN
__device__ __forceinline__ uint32_t f2bf_pk(float lo, float hi) { return (uint32_t)f2bf(lo) | ((uint32_t)f2bf(hi) << 16); }
#else
#define RG_ACT_DTYPE RG_BF16
#define RG_MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16((a), (b), (c), 0, 0, 0)
// bf16 <-> f32 (round to nearest even; NaN stays NaN)
__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
__device__ __forceinline__ bf16_t f2bf(float f) {
  uint32_t u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40);  // quiet NaN
  u += 0x7fffu + ((u >> 16) & 1u);
  return (bf16_t)(u >> 16);
}
// two values in one instruction (gfx950 v_cvt_pk_bf16_f32: round to nearest even like f2bf; NaN stays a quiet NaN) — for epilogues that convert whole tiles
typedef __bf16 rg_bf2 __attribute__((ext_vector_type(2)));
typedef float rg_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t f2bf_pk(float lo, float hi) { return __builtin_bit_cast(uint32_t, __builtin_convertvector((rg_f2){lo, hi}, rg_bf2)); }
#endif

__device__ __forceinline__ float act_fwd(float v, int act, float slope) {
  // none / ReLU / LeakyReLU without a branch (epilogues call this per element inside unrolled loops: the scalar branches of a
  // switch cost more than a thin layer's MFMAs, see conv5.hip); the two transcendental ones behind one wave-uniform branch
  if (act <= RG_ACT_LEAKY) {
    const float neg = act == RG_ACT_RELU ? 0.f : v * (act == RG_ACT_LEAKY ? slope : 1.f);
    return v < 0.f ? neg : v;   // (this form hands a NaN on, like torch.relu; `v > 0 ? v : neg` would turn it into 0 and hide an overflow)
  }
  return act == RG_ACT_SIGMOID ? 1.f / (1.f + __expf(-v)) : tanhf(v);
}
// Convolution epilogues convert whole accumulator tiles; a generic act_fwd per element under a per-lane `channel < cout` test makes the compiler
// predicate 100+ inlined tanh / exp bodies (gemm5.hip measured 17 us per 256 x 256 tile for that form, 2.5 us for this one).  They run one of two
// instantiations instead: LIN = the piecewise-linear activations (none / ReLU / LeakyReLU) as one multiply + select with nslope = rg_act_nslope(),
// or the transcendental ones through act_fwd.  One accumulator quad -> 4 stored channels: + bias, activation, zeros for channels >= live.
__device__ __forceinline__ float rg_act_nslope(int act, float slope) { return act == RG_ACT_RELU ? 0.f : (act == RG_ACT_LEAKY ? slope : 1.f); }
template <bool LIN>
__device__ __forceinline__ uint2 rg_epilogue_quad(const f32x4 acc, const float* bv, int live, float nslope, int act, float slope) {
  float v[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    float x = acc[r] + bv[r];
    if constexpr (LIN) x = x < 0.f ? x * nslope : x;      // (hands a NaN on, like act_fwd)
    else x = act_fwd(x, act, slope);
    v[r] = r < live ? x : 0.f;
  }
  uint2 o; o.x = f2bf_pk(v[0], v[1]); o.y = f2bf_pk(v[2], v[3]);
  return o;
}
// derivative expressed through the OUTPUT y of the activation
__device__ __forceinline__ float act_bwd_from_y(float y, int act, float slope) {
  switch (act) {
    case RG_ACT_RELU: return y > 0.f ? 1.f : 0.f;
    case RG_ACT_LEAKY: return y > 0.f ? 1.f : slope;
    case RG_ACT_SIGMOID: return y * (1.f - y);
    case RG_ACT_TANH: return 1.f - y * y;
    default: return 1.f;
  }
}

// ---- tensor views ----------------------------------------------------------
struct CL5 {  // channels-last 5-D activation view
  bf16_t* p;
  int N, C, D, H, W;
  int lanes;            // readable/writable channel lanes (multiple of 8, >= C)
  long sn, sd, sh, sw;  // element strides
};

static inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

// Validates the activation convention and fills a CL5.  `lanes` = C rounded up to 8.
static inline int rg_as_cl5(const rg_tensor* t, CL5* o, const char* name) {
  if (!t || !t->ptr) { rg_set_error("%s: null tensor", name); return RG_ERR_BAD_SHAPE; }
  if (t->dtype != RG_ACT_DTYPE || t->rank != 5) { rg_set_error("%s: expected a rank-5 (N,C,D,H,W) tensor of this build's 16-bit type (rg_dtype %d), got dtype %d", name, (int)RG_ACT_DTYPE, (int)t->dtype); return RG_ERR_BAD_SHAPE; }
  if (t->strides[1] != 1 && t->sizes[1] != 1) { rg_set_error("%s: channel stride must be 1 (channels-last)", name); return RG_ERR_BAD_SHAPE; }
  o->p = (bf16_t*)t->ptr;
  o->N = (int)t->sizes[0]; o->C = (int)t->sizes[1]; o->D = (int)t->sizes[2]; o->H = (int)t->sizes[3]; o->W = (int)t->sizes[4];
  o->sn = t->strides[0]; o->sd = t->strides[2]; o->sh = t->strides[3]; o->sw = t->strides[4];
  o->lanes = round_up(o->C, 8);
  if (o->sw < o->lanes || (o->sw % 8) || (o->sh % 8) || (o->sd % 8) || (o->sn % 8) || (((uintptr_t)t->ptr) & 15)) {
    rg_set_error("%s: strides must be multiples of 8 elements with W-stride >= padded lanes (C=%d sw=%ld) and base 16B aligned", name, o->C, o->sw);
    return RG_ERR_BAD_SHAPE;
  }
  if (o->N <= 0 || o->C <= 0 || o->D <= 0 || o->H <= 0 || o->W <= 0) { rg_set_error("%s: empty tensor", name); return RG_ERR_BAD_SHAPE; }
  return RG_OK;
}
