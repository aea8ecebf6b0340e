// igemm2 — the wide-channel implicit-GEMM convolution kernel (Cin lanes >= 64) for gfx950.
//
// Same GEMM view and row/tap/output maps as conv.hip's igemm_kernel, different machinery:
//   * tile BM (128|256) rows x 16*FN channels x BK = 64 channels of one tap per K-step; channel chunk OUTER,
//     tap INNER, so the 27-fold tap re-reads of a 64-channel slice hit the XCD's L2 (measured 94 % TCC hit rate);
//   * register-staged operand path: 16-byte global loads of stage k+2 are issued at the top of step k, rest in
//     VGPRs for one whole step and are written to the LDS double buffer at the top of step k+1 (ds_write_b128).
//     Measured on MI355X: a global_load_lds (LDS-DMA) piece costs ~95 issue cycles of the ONE wave per SIMD this
//     kernel runs, 13 pieces per step out-weighed the 72 MFMAs (1150 cycles); a load + ds_write pair costs ~25;
//   * the im2col gather (row @ tap) and the LDS bank swizzle (16-byte slot ^ ((row>>1)&7), conflict-free for every
//     ds_read_b128 lane group: SQ_LDS_BANK_CONFLICT = 0) are folded into the per-lane source address;
//   * fragment reads are hand-counted inline-asm ds_read_b128, two weight fragments ahead of the MFMAs that consume
//     them (s_waitcnt lgkmcnt(2)); the address arithmetic of the next stage is spread over the MFMA steps;
//   * out-of-volume taps (zero padding), rows past M, channel lanes past Cin and weight rows past Cout read a
//     128-byte page of zeros instead of being predicated: the loop body is branch-free;
//   * each wave loads exactly the activation rows it later multiplies (4 waves split the rows), the weight
//     tile is shared by the 4 waves.  One barrier per K-step.
// One workgroup per CU (up to 106 KB LDS, 1 wave per SIMD with the whole register file).
#include "common.h"
#include <type_traits>
#include <utility>
#include <stdlib.h>

template <class F, int... Is>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, Is...>) { (f(std::integral_constant<int, Is>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

struct Igemm2Args {
  const bf16_t* x; const bf16_t* w; const float* bias; bf16_t* y; const bf16_t* zeros;
  int N, ID, IH, IW;
  int RD, RH, RW;
  int TD, TH, TW;
  int rs[3], tb[3], ts[3], os[3], oo[3];
  long xsn, xsd, xsh, xsw;
  long ysn, ysd, ysh, ysw;
  int cink;       // K per tap in the packed weights (multiple of 64)
  int cin_lanes;  // readable lanes of x (multiple of 8)
  int cout, cout_lanes, wrows;
  int act; float slope;
  long M;
  int tiles_m, tiles_n;
  int check_bounds;  // 0 when every tap of every row lies inside the input volume (valid convolutions)
};

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// Hand-counted LDS fragment reads (hipcc would otherwise wait lgkmcnt(0) right in front of every MFMA group, exposing
// the full ds_read latency at one wave per SIMD).  The read is invisible to the compiler's waitcnt bookkeeping; the
// matching lds_wait<N>() names the fragment as read-write so that no consumer can be scheduled above it.
template <int OFF>
__device__ __forceinline__ void lds_read16(bf16x8& dst, unsigned addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF));
}
template <int N>
__device__ __forceinline__ void lds_wait(bf16x8& f) { asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(f) : "n"(N)); }
template <int BM, int FN, bool CHECK>
__global__ __launch_bounds__(256, (BM == 128 ? 2 : 1)) void igemm2_kernel(Igemm2Args a) {
  constexpr int BN = FN * 16;
  constexpr int BPIECES = ((BN / 8) + 3) / 4 * 4;  // 8-row pieces of the weight tile, padded to the 4 waves
  constexpr int BROWS = BPIECES * 8;
  constexpr int APW = BM / 32;                     // activation pieces per wave
  constexpr int BPW = BPIECES / 4;
  constexpr int FM = BM / 64;
  constexpr int STAGE = (BM + BROWS) * 64;         // bf16 elements per stage
  constexpr int LPS = APW + BPW;                   // LDS-DMA instructions per wave per stage
  __shared__ __attribute__((aligned(128))) bf16_t lds[2 * STAGE];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // XCD-aware tile order: consecutive workgroup ids go to different XCDs; give each XCD a contiguous run of
  // row tiles of one channel tile so that neighbouring row tiles (which share halo voxels) meet in one L2.
  const int nwg = a.tiles_m * a.tiles_n;
  int bid = blockIdx.x;
  {
    const int q = nwg / 8, r = nwg % 8, xcd = bid % 8, idx = bid / 8;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tile_n = bid / a.tiles_m, tile_m = bid % a.tiles_m;
  const long m0 = (long)tile_m * BM;
  const int n0 = tile_n * BN;

  const int lrow = lane >> 3, lslot = lane & 7;
  // 32-bit element offsets throughout (host guarantees every tensor < 2^31 elements): halves the address registers.
  // ---- activation rows of this lane: piece p = wave*APW + i covers tile rows 8p..8p+7
  int rd0[APW], rh0[APW], rw0[APW]; int rbase[APW];
  unsigned rvalid = 0;  // bit i: row of piece i is < M
#pragma unroll
  for (int i = 0; i < APW; ++i) {
    const int row = (wave * APW + i) * 8 + lrow;
    const long m = m0 + row;
    const bool v = m < a.M;
    rvalid |= (unsigned)v << i;
    unsigned t = v ? (unsigned)m : 0u;   // (M < 2^31, checked by the launcher: 32-bit divisions)
    const int rw = (int)(t % (unsigned)a.RW); t /= (unsigned)a.RW;
    const int rh = (int)(t % (unsigned)a.RH); t /= (unsigned)a.RH;
    const int rd = (int)(t % (unsigned)a.RD); t /= (unsigned)a.RD;
    rd0[i] = rd * a.rs[0] + a.tb[0]; rh0[i] = rh * a.rs[1] + a.tb[1]; rw0[i] = rw * a.rs[2] + a.tb[2];
    rbase[i] = (int)(t * a.xsn + rd0[i] * a.xsd + rh0[i] * a.xsh + rw0[i] * a.xsw);
  }
  // source channel offset inside the 64-channel chunk = (slot ^ ((tile_row>>1)&7))*8; tile_row = 8*piece + lrow
  const int sw_even = (lslot ^ (lrow >> 1)) * 8, sw_odd = (lslot ^ (4 + (lrow >> 1))) * 8;
  // ---- weight rows of this lane
  const int ntaps = a.TD * a.TH * a.TW;
  int wb[BPW]; unsigned wvalid = 0;
#pragma unroll
  for (int i = 0; i < BPW; ++i) {
    const int row = (wave * BPW + i) * 8 + lrow;
    const int n = n0 + row;
    const bool v = row < BN && n < a.wrows;
    wvalid |= (unsigned)v << i;
    wb[i] = v ? n * ntaps * a.cink : 0;
  }
  const int nchunk = a.cink >> 6;
  const int nk = ntaps * nchunk;

  // ---- producer state (the K-step being prepared).  Channel chunk OUTER, tap INNER.
  int ptap = 0, pchunk = 0, kd = 0, kh = 0, kw = 0;
  // out-of-range lanes switch their BASE pointer to the zero page (it lives in the code object, arbitrarily far from
  // the tensors, so it cannot be expressed as a 32-bit offset from them)
  const bf16_t* aptr[APW]; const bf16_t* bptr[BPW];
  // Address arithmetic of the stage being issued, one piece at a time so that it can be spread over the MFMA steps.
  // Branch-free: every lane always loads; out-of-range lanes read the zero page.
  auto prep_a = [&](int i) {
    const int c = pchunk * 64 + (((wave * APW + i) & 1) ? sw_odd : sw_even);
    const int od = kd * a.ts[0], oh = kh * a.ts[1], ow = kw * a.ts[2];
    const int tapoff = (int)(od * a.xsd + oh * a.xsh + ow * a.xsw);
    bool ok = ((rvalid >> i) & 1) & (c < a.cin_lanes);
    if (CHECK)
      ok &= ((unsigned)(rd0[i] + od) < (unsigned)a.ID) & ((unsigned)(rh0[i] + oh) < (unsigned)a.IH) &
            ((unsigned)(rw0[i] + ow) < (unsigned)a.IW);
    aptr[i] = (ok ? a.x : a.zeros) + (ok ? rbase[i] + tapoff + c : lslot * 8);
  };
  auto prep_b = [&](int i) {
    const bool ok = ((wvalid >> i) & 1) & (pchunk < nchunk);
    const int c = pchunk * 64 + (((wave * BPW + i) & 1) ? sw_odd : sw_even);
    bptr[i] = (ok ? a.w : a.zeros) + (ok ? wb[i] + ptap * a.cink + c : lslot * 8);
  };
  auto advance = [&]() {  // branch-free odometer (kw fastest, then kh, kd, then the channel chunk)
    ++ptap; ++kw;
    const int cw = kw == a.TW; kw = cw ? 0 : kw; kh += cw;
    const int ch = kh == a.TH; kh = ch ? 0 : kh; kd += ch;
    const int cd = kd == a.TD; kd = cd ? 0 : kd; ptap = cd ? 0 : ptap; pchunk += cd;
  };
  auto prepare = [&]() {
#pragma unroll
    for (int i = 0; i < APW; ++i) prep_a(i);
#pragma unroll
    for (int i = 0; i < BPW; ++i) prep_b(i);
    advance();
  };
  bf16x8 sreg[LPS];  // one stage of operands in flight through the registers
  auto gload = [&]() {
#pragma unroll
    for (int i = 0; i < APW; ++i) sreg[i] = *(const bf16x8*)aptr[i];
#pragma unroll
    for (int i = 0; i < BPW; ++i) sreg[APW + i] = *(const bf16x8*)bptr[i];
  };
  auto lstore = [&](int buf) {
    bf16_t* As = lds + buf * STAGE + lane * 8;
    bf16_t* Bs = As + BM * 64;
#pragma unroll
    for (int i = 0; i < APW; ++i) *(bf16x8*)(As + (wave * APW + i) * 512) = sreg[i];
#pragma unroll
    for (int i = 0; i < BPW; ++i) *(bf16x8*)(Bs + (wave * BPW + i) * 512) = sreg[APW + i];
  };

  f32x4 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  prepare(); gload(); lstore(0);   // stage 0 -> buffer 0
  prepare(); gload();              // stage 1 in registers
  prepare();                       // addresses of stage 2
  const int fr = lane & 15, fg = lane >> 4, fsw = fr >> 1;
  const unsigned lds0 = (unsigned)(size_t)(lptr_t)lds;  // LDS byte address of the ring
  const unsigned a_lane = lds0 + (wave * (BM / 4) + fr) * 128 + ((fg ^ fsw) << 4);
  const unsigned b_lane = lds0 + BM * 128 + fr * 128 + ((fg ^ fsw) << 4);
  constexpr int T2 = 2 * FN;
  int stage = 0;
  for (int ks = 0; ks < nk; ++ks) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // my LDS writes of stage ks (issued last step) are done
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    lstore(stage ^ 1);  // stage ks+1: registers -> the buffer whose readers finished before this barrier
    gload();   // stage ks+2 -> registers (addresses prepared during the previous step)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // keep the hand-counted fragment reads alone on the counter
    const unsigned sa = stage * (STAGE * 2);
    const unsigned a0 = a_lane + sa, a1 = a0 ^ 64u, b0 = b_lane + sa, b1 = b0 ^ 64u;  // second 32-k half: slot ^ 4
    bf16x8 xf0[FM], xf1[FM], wf[3];
    // issue order: xf0[*], xf1[*], w0, w1, then per step t: w(t+2); wait for w(t) (2 newer reads may fly); FM MFMAs
    static_for<FM>([&](auto I) { constexpr int i = decltype(I)::value; lds_read16<i * 2048>(xf0[i], a0); });
    static_for<FM>([&](auto I) { constexpr int i = decltype(I)::value; lds_read16<i * 2048>(xf1[i], a1); });
    lds_read16<0>(wf[0], b0);
    lds_read16<2048>(wf[1], b0);
    static_for<T2>([&](auto T) {
      constexpr int t = decltype(T)::value;
      constexpr int kk = t / FN, j = t % FN;
      if constexpr (t + 2 < T2) {
        constexpr int k2 = (t + 2) / FN, j2 = (t + 2) % FN;
        lds_read16<j2 * 2048>(wf[(t + 2) % 3], k2 ? b1 : b0);
      }
      constexpr int NW = (t + 1 < T2) + (t + 2 < T2);
      if constexpr (t == 0) {  // the first wait also covers every activation fragment (issued before w0)
        static_for<FM>([&](auto I) { constexpr int i = decltype(I)::value; lds_wait<NW>(xf0[i]); lds_wait<NW>(xf1[i]); });
      }
      lds_wait<NW>(wf[t % 3]);
#pragma unroll
      for (int i = 0; i < FM; ++i)
        acc[i][j] = RG_MFMA16(wf[t % 3], kk ? xf1[i] : xf0[i], acc[i][j]);
      // address math of stage ks+2, one DMA piece per MFMA step, in the shadow of the matrix pipe
      if constexpr (t < APW) prep_a(t); else if constexpr (t - APW < BPW) prep_b(t - APW);
      __builtin_amdgcn_sched_barrier(0);
    });
    static_for<(APW + BPW > T2 ? APW + BPW - T2 : 0)>([&](auto P) {  // pieces that did not fit a step (narrow tiles)
      constexpr int q = decltype(P)::value + T2;
      if constexpr (q < APW) prep_a(q); else prep_b(q - APW);
    });
    advance();
    asm volatile("" ::: "memory");
    stage ^= 1;
  }

  // ---- epilogue: lane holds channels c..c+3 of voxel (lane&15) for every fragment (two instantiations: common.h rg_epilogue_quad)
  const float nslope = rg_act_nslope(a.act, a.slope);
  auto finish = [&](auto LIN) {
#pragma unroll
    for (int i = 0; i < FM; ++i) {
      const long m = m0 + wave * (BM / 4) + i * 16 + fr;
      if (m >= a.M) continue;
      unsigned t = (unsigned)m;
      const int rw = (int)(t % (unsigned)a.RW); t /= (unsigned)a.RW;
      const int rh = (int)(t % (unsigned)a.RH); t /= (unsigned)a.RH;
      const int rd = (int)(t % (unsigned)a.RD); t /= (unsigned)a.RD;
      bf16_t* yrow = a.y + t * a.ysn + (long)(rd * a.os[0] + a.oo[0]) * a.ysd + (long)(rh * a.os[1] + a.oo[1]) * a.ysh +
                     (long)(rw * a.os[2] + a.oo[2]) * a.ysw;
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        const int c = n0 + j * 16 + fg * 4;
        if (c >= a.cout_lanes) continue;
        float bv[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) bv[r] = (a.bias && c + r < a.cout) ? a.bias[c + r] : 0.f;
        *(uint2*)(yrow + c) = rg_epilogue_quad<decltype(LIN)::value>(acc[i][j], bv, a.cout - c, nslope, a.act, a.slope);
      }
    }
  };
  if (a.act <= RG_ACT_LEAKY) finish(std::true_type{}); else finish(std::false_type{});
}

template <int BM, int FN>
static void launch_igemm2(Igemm2Args& a, hipStream_t s) {
  constexpr int BN = FN * 16;
  a.tiles_m = (int)((a.M + BM - 1) / BM);
  a.tiles_n = (a.cout_lanes + BN - 1) / BN;
  rg_set_kernel_name("igemm2_kernel<%d,%d,%d>", BM, FN, a.check_bounds);
  if (a.check_bounds) hipLaunchKernelGGL((igemm2_kernel<BM, FN, true>), dim3(a.tiles_m * a.tiles_n), dim3(256), 0, s, a);
  else hipLaunchKernelGGL((igemm2_kernel<BM, FN, false>), dim3(a.tiles_m * a.tiles_n), dim3(256), 0, s, a);
}

// K alignment (per tap) the engine wants for the packed weights of this problem: 64 -> igemm2/3/4, 1 -> FLAT (igemm4 / thin).
extern "C" int rg_conv3d_kalign(int cin, int cout) {
  const int lanes = round_up(cin, 8);
  // BK = 64 per tap wastes (round_up(lanes,64)-lanes)/lanes of the MFMA work: accept up to ~12 %
  if (lanes >= 64 && (round_up(lanes, 64) - lanes) * 8 <= lanes && round_up(cout, 16) >= 48) return 64;
  // everything else: K flattened across taps in 8-lane pieces (igemm4 FLAT)
  return 1;
}

int rg_launch_igemm2(const CL5& X, const CL5& Y, const void* w_packed, const float* bias, const rg_conv_params* p,
                     const void* zeros, hipStream_t st) {
  Igemm2Args a;
  a.x = X.p; a.w = (const bf16_t*)w_packed; a.bias = bias; a.y = Y.p; a.zeros = (const bf16_t*)zeros;
  a.N = X.N; a.ID = X.D; a.IH = X.H; a.IW = X.W;
  a.RD = p->R[0]; a.RH = p->R[1]; a.RW = p->R[2];
  a.TD = p->T[0]; a.TH = p->T[1]; a.TW = p->T[2];
  for (int d = 0; d < 3; ++d) { a.rs[d] = p->rstride[d]; a.tb[d] = p->tbase[d]; a.ts[d] = p->tstep[d]; a.os[d] = p->ostride[d]; a.oo[d] = p->ooffset[d]; }
  a.xsn = X.sn; a.xsd = X.sd; a.xsh = X.sh; a.xsw = X.sw;
  a.ysn = Y.sn; a.ysd = Y.sd; a.ysh = Y.sh; a.ysw = Y.sw;
  a.cink = round_up(X.C, 64); a.cin_lanes = X.lanes;
  a.cout = p->cout; a.cout_lanes = Y.lanes; a.wrows = round_up(p->cout, 16);
  a.act = p->act; a.slope = p->slope;
  a.M = (long)X.N * a.RD * a.RH * a.RW;
  a.check_bounds = 0;
  {
    const int I[3] = {X.D, X.H, X.W};
    for (int d = 0; d < 3; ++d) {
      const long lo0 = p->tbase[d], lo1 = p->tbase[d] + (long)p->tstep[d] * (p->T[d] - 1);
      const long rmax = (long)(p->R[d] - 1) * p->rstride[d];
      const long mn = (lo0 < lo1 ? lo0 : lo1), mx = (lo0 < lo1 ? lo1 : lo0) + rmax;
      if (mn < 0 || mx >= I[d]) a.check_bounds = 1;
    }
  }
  const int nf = a.wrows / 16;
  int fn;
  if (nf % 9 == 0) fn = 9; else if (nf % 8 == 0) fn = 8; else if (nf <= 5) fn = nf < 4 ? 4 : nf; else if (nf % 5 == 0) fn = 5;
  else fn = (round_up(nf, 9) - nf) * 8 <= (round_up(nf, 8) - nf) * 9 ? 9 : 8;
  // rows per tile: 256 unless that leaves the 256 CUs with fewer than ~2 rounds of workgroups
  // BM = 128 at two workgroups per CU (TLP hides the stage hand-over) measured faster than BM = 256 at one: 715-750
  // vs 460-540 TFLOP/s on the 432..3024-channel layers
#define RG_L2(FN_) launch_igemm2<128, FN_>(a, st)
  switch (fn) {
    case 4: RG_L2(4); break;
    case 5: RG_L2(5); break;
    case 8: RG_L2(8); break;
    default: RG_L2(9); break;
  }
#undef RG_L2
  return 0;
}
