// igemm4 — wide-channel implicit-GEMM convolution, activations straight into MFMA fragments (no LDS round trip).
//
// Measured on igemm2 (128x144 tile, 4 waves, both operands through LDS): with the global loads, the MFMAs and the address
// arithmetic all removed the loop still took 58 % of its time — the LDS itself.  ds_write_b128 moves 1 KiB per ~13 cycles
// (79 B/clk/CU), so the 36 stores + 88 fragment reads of one tile step cost ~820 LDS cycles against 576 MFMA cycles per
// wave.  But every wave of that kernel only ever multiplies the activation rows it loaded itself: the LDS trip of the
// activation operand is pure overhead.  The B operand of v_mfma_f32_16x16x32_bf16 wants, per lane (voxel n = l&15,
// k-group g = l>>4), 8 consecutive channels of voxel n — 16 contiguous bytes of a channels-last tensor.  So here
//   * each lane loads its fragment pieces of the activation tile directly (buffer_load_dwordx4, 16 rows x 64 B per
//     wave-instruction), half a K-step ahead in a single register set; out-of-volume taps, rows past M and lanes past Cin turn the
//     byte offset into an out-of-range one and the buffer descriptor returns zeros (branch-free zero padding);
//   * only the weight tile (shared by the 4 waves) goes through LDS, register-staged and double-buffered as in igemm2,
//     its per-step address a wave-uniform SGPR offset on a per-thread constant;
//   * same tile order (XCD-aware), K order (channel chunk outer, tap inner), swizzle, hand-counted fragment reads,
//     packed-weight format and epilogue as igemm2.
// LDS work per tile step drops to 18 stores + 72 reads, VALU address work from ~90 to ~25 instructions per wave-step.
#include "common.h"
#include <utility>
#include <stdlib.h>

template <class F, int... Is>
__device__ __forceinline__ void s4_impl(F&& f, std::integer_sequence<int, Is...>) { (f(std::integral_constant<int, Is>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void s4for(F&& f) { s4_impl(f, std::make_integer_sequence<int, N>{}); }

#include "igemm_args.h"

typedef __attribute__((address_space(3))) void* lptr4_t;
template <int OFF>
__device__ __forceinline__ void ld4_16(bf16x8& dst, unsigned addr) { asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF)); }
template <int N>
__device__ __forceinline__ void ld4_wait(bf16x8& f) { asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(f) : "n"(N)); }
#define IG4_OOB 0x80000000u
// Buffer loads hipcc does not count.  Its own bookkeeping merges the prologue's state into the loop header and ends up with
// s_waitcnt vmcnt(0) right after every step's barrier: the k-half 1 fragments issued at the end of a step were waited for
// at once.  The loads below are retired by hand-counted waits (vmcnt completes in issue order) that name the registers.
typedef int i32x4_t4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ i32x4_t4 ig4_rsrc(const void* base, unsigned nrec) {
  const unsigned long b = (unsigned long)base;
  return (i32x4_t4){(int)(unsigned)b, (int)((b >> 32) & 0xffffu), (int)nrec, 0x00020000};
}
__device__ __forceinline__ void ig4_load(u32x4_t& dst, i32x4_t4 rsrc, unsigned voff) {
  asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(dst) : "v"(voff), "s"(rsrc) : "memory");
}
__device__ __forceinline__ void ig4_load_s(u32x4_t& dst, i32x4_t4 rsrc, unsigned voff, int soff) {
  asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(dst) : "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}
template <int N>
__device__ __forceinline__ void ig4_vmwait(u32x4_t& first) { asm volatile("s_waitcnt vmcnt(%1)" : "+v"(first) : "n"(N) : "memory"); }
__device__ __forceinline__ void ig4_tie(u32x4_t& r) { asm volatile("" : "+v"(r)); }
#ifndef IG4_NS
#define IG4_NS 1   // activation register sets for tiles of <= 192 rows; 2 (loads 1.5 steps ahead) measured no faster on MI355X
#endif

// FLAT: K is walked in 8-lane pieces ACROSS taps (K = taps*lanes padded once to 64, not per tap): channel counts that are
// not multiples of 64 (12, 72, 84 ... lanes) waste no MFMA work.  A 64-wide K-step then mixes taps, so each lane looks its
// two pieces (k-half 0 and 1) up in an LDS table {byte offset of (tap, lane group), packed tap displacement}.
struct PieceInfo { int off; int disp; };  // disp < 0: piece past the end of K
template <int FM, int FN, bool CHECK, bool FLAT>
__global__ __launch_bounds__(256, (FM >= 3 ? 2 : (FM == 1 ? 4 : 3))) void igemm4_kernel(Igemm4Args a) {
  constexpr int BM = 64 * FM, BN = FN * 16;
  constexpr int BPIECES = ((BN / 8) + 3) / 4 * 4, BROWS = BPIECES * 8, BPW = BPIECES / 4;
  constexpr int BSTAGE = BROWS * 64;  // bf16 elements per weight stage
  __shared__ __attribute__((aligned(128))) bf16_t lds[2 * BSTAGE];
  extern __shared__ PieceInfo ptab[];  // FLAT only
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nwg = a.tiles_m * a.tiles_n * a.ksplit * a.nbatch;
  int bid = blockIdx.x;
  { const int q = nwg / 8, r = nwg % 8, xcd = bid % 8, idx = bid / 8; bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx; }
  const int batch = bid / (a.tiles_m * a.tiles_n * a.ksplit); bid -= batch * (a.tiles_m * a.tiles_n * a.ksplit);
  const bf16_t* const xbase = a.x + batch * a.xbatch;
  const bf16_t* const wbase = a.w + batch * a.wbatch;
  bf16_t* const ybase = a.y + batch * a.ybatch;
  const int split = bid / (a.tiles_m * a.tiles_n); bid -= split * (a.tiles_m * a.tiles_n);
  const int tile_n = bid / a.tiles_m, tile_m = bid % a.tiles_m;
  const long m0 = (long)tile_m * BM;
  const int n0 = tile_n * BN;
  const int fr = lane & 15, fg = lane >> 4, fsw = fr >> 1;
  const int lrow = lane >> 3, lslot = lane & 7;

  // ---- the FM activation rows of this lane (fragment i, voxel fr), as byte offsets into x
  int rd0[FM], rh0[FM], rw0[FM]; unsigned rbase[FM];
#pragma unroll
  for (int i = 0; i < FM; ++i) {
    const long m = m0 + wave * (16 * FM) + i * 16 + fr;
    const bool v = m < a.M;
    unsigned t = v ? (unsigned)m : 0u;   // (M < 2^31, checked by the launcher: 32-bit divisions)
    const int rw = (int)(t % (unsigned)a.RW); t /= (unsigned)a.RW;
    const int rh = (int)(t % (unsigned)a.RH); t /= (unsigned)a.RH;
    const int rd = (int)(t % (unsigned)a.RD); t /= (unsigned)a.RD;
    rd0[i] = rd * a.rs[0] + a.tb[0]; rh0[i] = rh * a.rs[1] + a.tb[1]; rw0[i] = rw * a.rs[2] + a.tb[2];
    // tap offsets are added later and may be negative: keep the sum inside [0, 2^31) for valid lanes, park invalid rows
    rbase[i] = v ? (unsigned)((t * a.xsn + rd0[i] * a.xsd + rh0[i] * a.xsh + rw0[i] * a.xsw + (FLAT ? 0 : fg * 8)) * 2) : IG4_OOB;
    if (!v) { rd0[i] = rh0[i] = rw0[i] = -(1 << 20); }
  }
  // ---- the BPW weight pieces of this lane: per-thread constant byte offset (row, swizzled slot); the tap/chunk part is
  // wave-uniform and rides in the instruction's SGPR offset (not range-checked, so it is only ever applied to valid steps)
  const int ntaps = a.TD * a.TH * a.TW;
  unsigned wv[BPW];
#pragma unroll
  for (int i = 0; i < BPW; ++i) {
    const int piece = wave * BPW + i, row = piece * 8 + lrow, n = n0 + row;
    const bool v = row < BN && n < a.wrows;
    const int slot = lslot ^ ((piece & 1) * 4 + (lrow >> 1));
    wv[i] = v ? (unsigned)(((long)n * (FLAT ? a.krow : ntaps * a.cink) + slot * 8) * 2) : IG4_OOB;
  }
  const int nchunk = a.cink >> 6;
  // K range of this workgroup in K-steps (non-FLAT: step = chunk * ntaps + tap): everything, or the split's share
  const int kunits = FLAT ? a.krow >> 6 : nchunk * ntaps, kper = (kunits + a.ksplit - 1) / a.ksplit;
  const int ku0 = split * kper, ku1 = ku0 + kper < kunits ? ku0 + kper : kunits;
  const int nk = ku1 > ku0 ? ku1 - ku0 : 0;
  if (FLAT) {
    for (int P = tid; P < (a.krow >> 6) * 8; P += 256) {
      const int tap = P / a.cp, c8 = P - tap * a.cp;
      PieceInfo e; e.off = 0; e.disp = -1;
      if (tap < ntaps) {
        const int kw_ = tap % a.TW, kh_ = (tap / a.TW) % a.TH, kd_ = tap / (a.TW * a.TH);
        const int od = kd_ * a.ts[0], oh = kh_ * a.ts[1], ow = kw_ * a.ts[2];
        e.off = ((int)(od * a.xsd + oh * a.xsh + ow * a.xsw) + c8 * 8) * 2;
        e.disp = (od + 512) | ((oh + 512) << 10) | ((ow + 512) << 20);
      }
      ptab[P] = e;
    }
    __syncthreads();
  }
  const i32x4_t4 rx = ig4_rsrc(xbase, a.xbytes), rw_ = ig4_rsrc(wbase, a.wbytes);

  // ---- producer odometers (the K-step whose loads are issued next): channel chunk OUTER, tap INNER.  The weights run
  // two steps ahead (registers, then LDS), the activation fragments one (registers only)
  int pchunk = ku0 / ntaps, kd, kh, kw;                    // activations: (chunk, tap) of the first step
  { const int t0 = ku0 - pchunk * ntaps; kw = t0 % a.TW; kh = (t0 / a.TW) % a.TH; kd = t0 / (a.TW * a.TH); }
  int btap = ku0 - pchunk * ntaps, bchunk = pchunk;        // weights
  int astep = ku0, bstep = ku0;                            // FLAT: plain K-step counters
  int bleft = nk;                                          // non-FLAT: weight steps still to load
  auto advance_a = [&]() {
    ++kw;
    const int cw = kw == a.TW; kw = cw ? 0 : kw; kh += cw;
    const int ch = kh == a.TH; kh = ch ? 0 : kh; kd += ch;
    const int cd = kd == a.TD; kd = cd ? 0 : kd; pchunk += cd;
  };
  auto advance_b = [&]() { ++btap; const int c = btap == ntaps; btap = c ? 0 : btap; bchunk += c; };
  u32x4_t sb[BPW];  // one stage of weights in flight through the registers
  auto load_b = [&]() {
    const bool ok = FLAT ? bstep < ku1 : bleft > 0;
    --bleft;
    const int so = ok ? (FLAT ? bstep * 128 : (btap * a.cink + bchunk * 64) * 2) : 0;
#pragma unroll
    for (int i = 0; i < BPW; ++i) ig4_load_s(sb[i], rw_, ok ? wv[i] : IG4_OOB, so);
    if (FLAT) ++bstep; else advance_b();
  };
  // activation fragments live in ONE register set: the k-half 0 pieces of step ks+1 are loaded once the k-half 0
  // MFMAs of step ks have issued (mid-step), the k-half 1 pieces at the end of the step — half a step of latency cover
  // each (>= 288 MFMA cycles per wave, times the waves sharing the SIMD) for half the registers of a double buffer
  constexpr int NS = FM <= 3 ? IG4_NS : 1;  // activation register sets: loads run NS - 1/2 steps ahead of their MFMAs
  u32x4_t xa[NS][FM][2];
  unsigned vo0[FM], vo1[FM];
  auto prep_a = [&]() {
    if constexpr (FLAT) {
      PieceInfo e0, e1;
      e0.off = 0; e0.disp = -1; e1 = e0;
      if (astep < ku1) { e0 = ptab[astep * 8 + fg]; e1 = ptab[astep * 8 + 4 + fg]; }
      ++astep;
      const int d0 = (e0.disp & 1023) - 512, h0 = ((e0.disp >> 10) & 1023) - 512, w0 = ((e0.disp >> 20) & 1023) - 512;
      const int d1 = (e1.disp & 1023) - 512, h1 = ((e1.disp >> 10) & 1023) - 512, w1 = ((e1.disp >> 20) & 1023) - 512;
#pragma unroll
      for (int i = 0; i < FM; ++i) {
        bool ok0 = e0.disp >= 0, ok1 = e1.disp >= 0;
        if (CHECK) {
          ok0 &= ((unsigned)(rd0[i] + d0) < (unsigned)a.ID) & ((unsigned)(rh0[i] + h0) < (unsigned)a.IH) & ((unsigned)(rw0[i] + w0) < (unsigned)a.IW);
          ok1 &= ((unsigned)(rd0[i] + d1) < (unsigned)a.ID) & ((unsigned)(rh0[i] + h1) < (unsigned)a.IH) & ((unsigned)(rw0[i] + w1) < (unsigned)a.IW);
        }
        vo0[i] = ok0 ? rbase[i] + (unsigned)e0.off : IG4_OOB;
        vo1[i] = ok1 ? rbase[i] + (unsigned)e1.off : IG4_OOB;
      }
    } else {
    const int od = kd * a.ts[0], oh = kh * a.ts[1], ow = kw * a.ts[2];
    const int tapoff = ((int)(od * a.xsd + oh * a.xsh + ow * a.xsw) + pchunk * 64) * 2;
    const int c = pchunk * 64 + fg * 8;
    const bool c0 = c < a.cin_lanes, c1 = c + 32 < a.cin_lanes;
#pragma unroll
    for (int i = 0; i < FM; ++i) {
      bool ok = true;  // rows past M were parked at IG4_OOB (no tap offset brings them back: |tapoff| < 2^30)
      if (CHECK) ok = ((unsigned)(rd0[i] + od) < (unsigned)a.ID) & ((unsigned)(rh0[i] + oh) < (unsigned)a.IH) & ((unsigned)(rw0[i] + ow) < (unsigned)a.IW);
      const unsigned vo = rbase[i] + (unsigned)tapoff;
      vo0[i] = (ok & c0) ? vo : IG4_OOB;
      vo1[i] = (ok & c1) ? vo + 64u : IG4_OOB;
    }
    advance_a();
    }
  };
  auto load_a0 = [&](auto S) {
#pragma unroll
    for (int i = 0; i < FM; ++i) ig4_load(xa[decltype(S)::value][i][0], rx, vo0[i]);
  };
  auto load_a1 = [&](auto S) {
#pragma unroll
    for (int i = 0; i < FM; ++i) ig4_load(xa[decltype(S)::value][i][1], rx, vo1[i]);
  };
  // vector-memory operations per step, in issue order: BPW weight pieces (top), FM k-half-0 fragments (mid-step), FM k-half-1
  // fragments (end); WAIT_B / WAIT_A = how many YOUNGER ones are in flight when the weight set / a fragment half is needed
  constexpr int WAIT_B = 2 * FM, WAIT_A = FM + BPW + (NS - 1) * (BPW + 2 * FM);
  static_assert(WAIT_A < 64 && NS <= 2, "vmcnt is a 6-bit counter; the prologue below orders its loads for NS <= 2");
  auto tie_b = [&]() {
#pragma unroll
    for (int i = 1; i < BPW; ++i) ig4_tie(sb[i]);
  };
  auto store_b = [&](int buf) {
    bf16_t* Bs = lds + buf * BSTAGE + lane * 8;
#pragma unroll
    for (int i = 0; i < BPW; ++i) *(u32x4_t*)(Bs + (wave * BPW + i) * 512) = sb[i];
  };

  f32x4 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // prologue: stage 0 weights -> LDS buffer 0, stage 1 weights in registers; activations of stage 0 in flight
  // (same issue order as the loop: the weight set of step 0 is followed by exactly 2*FM younger loads)
  load_b();
  if constexpr (NS == 2) { prep_a(); load_a0(std::integral_constant<int, 0>{}); load_a1(std::integral_constant<int, 0>{}); }
  ig4_vmwait<2 * FM * (NS - 1)>(sb[0]); tie_b();
  store_b(0);
  load_b();
  { prep_a(); load_a0(std::integral_constant<int, NS - 1>{}); load_a1(std::integral_constant<int, NS - 1>{}); }

  const unsigned lds0 = (unsigned)(size_t)(lptr4_t)lds;
  const unsigned b_lane = lds0 + fr * 128 + ((fg ^ fsw) << 4);
  constexpr int T2 = 2 * FN;
  int stage = 0;
  auto step = [&](auto S) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // my weight stores of this stage (issued last step) are done
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    ig4_vmwait<WAIT_B>(sb[0]); tie_b();
    store_b(stage ^ 1);           // weights of step ks+1: registers -> the buffer whose readers finished before the barrier
    load_b();                     // weights of step ks+2 -> registers
    // no wait for the ds_writes above: LDS operations of a wave complete in order, so the counted waits below (which
    // allow the 1-2 YOUNGEST reads to be outstanding) already cover them
    const unsigned b0 = b_lane + stage * (BSTAGE * 2), b1 = b0 ^ 64u;
    bf16x8 wf[3];
    ld4_16<0>(wf[0], b0);
    ld4_16<(1 % FN) * 2048>(wf[1], (1 / FN) ? b1 : b0);
    s4for<T2>([&](auto T) {
      constexpr int t = decltype(T)::value;
      constexpr int kk = t / FN, j = t % FN;
      if constexpr (t + 2 < T2) {
        constexpr int k2 = (t + 2) / FN, j2 = (t + 2) % FN;
        ld4_16<j2 * 2048>(wf[(t + 2) % 3], k2 ? b1 : b0);
      }
      constexpr int NW = (t + 1 < T2) + (t + 2 < T2);
      ld4_wait<NW>(wf[t % 3]);
      if constexpr (j == 0) {   // this k-half's fragments have landed
        ig4_vmwait<WAIT_A>(xa[decltype(S)::value][0][kk]);
#pragma unroll
        for (int i = 1; i < FM; ++i) ig4_tie(xa[decltype(S)::value][i][kk]);
      }
#pragma unroll
      for (int i = 0; i < FM; ++i) {
        bf16x8 xf;
        __builtin_memcpy(&xf, &xa[decltype(S)::value][i][kk], 16);
        acc[i][j] = RG_MFMA16(wf[t % 3], xf, acc[i][j]);
      }
      if constexpr (t == FN - 1) { prep_a(); load_a0(S); }  // k-half 0 fragments are free: fetch those of step ks+NS
    });
    load_a1(S);
    stage ^= 1;
  };
  if constexpr (NS == 1) {
    for (int ks = 0; ks < nk; ++ks) step(std::integral_constant<int, 0>{});
  } else {
    for (int ks = 0; ks < nk; ks += 2) {
      step(std::integral_constant<int, 0>{});
      if (ks + 1 < nk) step(std::integral_constant<int, 1>{});
    }
  }

  // the loads issued past the end of K still target xa / sb: retire them before the registers are reused
  ig4_vmwait<0>(sb[0]); tie_b();
  s4for<NS>([&](auto S) {
#pragma unroll
    for (int i = 0; i < FM; ++i) { ig4_tie(xa[decltype(S)::value][i][0]); ig4_tie(xa[decltype(S)::value][i][1]); }
  });
  // ---- epilogue: lane holds channels c..c+3 of voxel (lane&15) for every fragment
  if (a.ksplit > 1) {   // split-K: raw fp32 sums; bias, activation and the bf16 store happen in igemm4_combine_kernel
#pragma unroll
    for (int i = 0; i < FM; ++i) {
      const long m = m0 + wave * (16 * FM) + i * 16 + fr;
      if (m >= a.M) continue;
      float* prow = a.part + ((long)split * a.M + m) * a.cout_lanes;
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        const int c = n0 + j * 16 + fg * 4;
        if (c < a.cout_lanes) *(f32x4*)(prow + c) = acc[i][j];
      }
    }
    return;
  }
  // two instantiations: the piecewise-linear activations (none / ReLU / LeakyReLU: one multiply and one select per element) as straight-line code,
  // the transcendental ones apart — a generic act_fwd per element under the per-lane `channel < cout` test expands into thousands of predicated
  // instructions (gemm5.hip measured 17 us per 256 x 256 tile for that form)
  const float nslope = rg_act_nslope(a.act, a.slope);
  auto finish = [&](auto TR) {
    float bv[FN][4];
#pragma unroll
    for (int j = 0; j < FN; ++j) {
      const int c = n0 + j * 16 + fg * 4;
#pragma unroll
      for (int r = 0; r < 4; ++r) bv[j][r] = (a.bias && c + r < a.cout) ? a.bias[c + r] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < FM; ++i) {
      const long m = m0 + wave * (16 * FM) + i * 16 + fr;
      if (m >= a.M) continue;
      unsigned t = (unsigned)m;
      const int rw = (int)(t % (unsigned)a.RW); t /= (unsigned)a.RW;
      const int rh = (int)(t % (unsigned)a.RH); t /= (unsigned)a.RH;
      const int rd = (int)(t % (unsigned)a.RD); t /= (unsigned)a.RD;
      bf16_t* yrow = ybase + t * a.ysn + (long)(rd * a.os[0] + a.oo[0]) * a.ysd + (long)(rh * a.os[1] + a.oo[1]) * a.ysh +
                     (long)(rw * a.os[2] + a.oo[2]) * a.ysw;
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        const int c = n0 + j * 16 + fg * 4;
        if (c >= a.cout_lanes) continue;
        *(uint2*)(yrow + c) = rg_epilogue_quad<!decltype(TR)::value>(acc[i][j], bv[j], a.cout - c, nslope, a.act, a.slope);
      }
    }
  };
  if (a.act <= RG_ACT_LEAKY) finish(std::false_type{}); else finish(std::true_type{});
}

// y[row][c] = act(sum_s part[s][row][c] + bias[c]) for a split-K launch: one thread per (row, 4 channels), splits summed in order
__global__ __launch_bounds__(256) void igemm4_combine_kernel(Igemm4Args a) {
  const int q4 = a.cout_lanes >> 2;
  const long total = a.M * q4;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long m = i / q4; const int c = (int)(i - m * q4) * 4;
    f32x4 v = *(const f32x4*)(a.part + m * a.cout_lanes + c);
    for (int sp = 1; sp < a.ksplit; ++sp) {
      const f32x4 u = *(const f32x4*)(a.part + ((long)sp * a.M + m) * a.cout_lanes + c);
      v[0] += u[0]; v[1] += u[1]; v[2] += u[2]; v[3] += u[3];
    }
    unsigned t = (unsigned)m;
    const int rw = (int)(t % (unsigned)a.RW); t /= (unsigned)a.RW;
    const int rh = (int)(t % (unsigned)a.RH); t /= (unsigned)a.RH;
    const int rd = (int)(t % (unsigned)a.RD); t /= (unsigned)a.RD;
    bf16_t* yrow = a.y + t * a.ysn + (long)(rd * a.os[0] + a.oo[0]) * a.ysd + (long)(rh * a.os[1] + a.oo[1]) * a.ysh +
                   (long)(rw * a.os[2] + a.oo[2]) * a.ysw;
    float bv[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) bv[r] = (a.bias && c + r < a.cout) ? a.bias[c + r] : 0.f;
    *(uint2*)(yrow + c) = a.act <= RG_ACT_LEAKY ? rg_epilogue_quad<true>(v, bv, a.cout - c, rg_act_nslope(a.act, a.slope), a.act, a.slope)
                                                 : rg_epilogue_quad<false>(v, bv, a.cout - c, 0.f, a.act, a.slope);
  }
}
void rg_launch_igemm4_combine(const Igemm4Args& a, hipStream_t st) {
  long blocks = (a.M * (a.cout_lanes >> 2) + 255) / 256; if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(igemm4_combine_kernel, dim3((unsigned)blocks), dim3(256), 0, st, a);
}
// Split count for a grid of `tiles` workgroups on `slots` co-resident ones: split K when at most half the slots are used and
// the K range is long enough (>= 4 units per split); never beyond what the caller's workspace holds
static int ig4_ksplit(long tiles, long slots, int kunits) {
  if (tiles * 2 > slots || kunits < 16) return 1;
  long s = slots / tiles; if (s > 8) s = 8; if (s > kunits / 8) s = kunits / 8;
  return s < 2 ? 1 : (int)s;
}

template <int FM, int FN>
static void launch_igemm4(Igemm4Args& a, bool check, bool flat, hipStream_t s, float* ws, size_t ws_bytes) {
  constexpr int BM = 64 * FM, BN = FN * 16;
  a.tiles_m = (int)((a.M + BM - 1) / BM);
  a.tiles_n = (a.cout_lanes + BN - 1) / BN;
  a.ksplit = 1; a.part = nullptr;
  {
    const long slots = FM >= 3 ? 512 : (FM == 2 ? 768 : 1024);
    const int ks = ig4_ksplit((long)a.tiles_m * a.tiles_n, slots, flat ? a.krow >> 6 : (a.cink >> 6) * a.TD * a.TH * a.TW);
    if (ks > 1 && a.nbatch == 1 && ws && (size_t)ks * a.M * a.cout_lanes * 4 <= ws_bytes && (a.cout_lanes & 3) == 0) { a.ksplit = ks; a.part = ws; }
  }
  rg_set_kernel_name("igemm4_kernel<%d,%d,%d%s>%s", FM, FN, (int)check, flat ? ",flat" : "", a.ksplit > 1 ? "+splitK" : "");
  const dim3 grid(a.tiles_m * a.tiles_n * a.ksplit * a.nbatch);
  if (flat) {
    const size_t shm = (size_t)(a.krow / 8) * sizeof(PieceInfo);
    if (check) hipLaunchKernelGGL((igemm4_kernel<FM, FN, true, true>), grid, dim3(256), shm, s, a);
    else hipLaunchKernelGGL((igemm4_kernel<FM, FN, false, true>), grid, dim3(256), shm, s, a);
  } else {
    if constexpr (FN >= 4) {
      if (check) hipLaunchKernelGGL((igemm4_kernel<FM, FN, true, false>), grid, dim3(256), 0, s, a);
      else hipLaunchKernelGGL((igemm4_kernel<FM, FN, false, false>), grid, dim3(256), 0, s, a);
    }
  }
  if (a.ksplit > 1) {
    long blocks = (a.M * (a.cout_lanes >> 2) + 255) / 256; if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(igemm4_combine_kernel, dim3((unsigned)blocks), dim3(256), 0, s, a);
  }
}

// extent in elements of a channels-last view (last addressable lane + 1)
static long cl5_extent(const CL5& T) {
  return (long)(T.N - 1) * T.sn + (long)(T.D - 1) * T.sd + (long)(T.H - 1) * T.sh + (long)(T.W - 1) * T.sw + T.lanes;
}

static int ig4_fn(int nf) {
  if (nf % 9 == 0) return 9;
  if (nf % 8 == 0) return 8;
  if (nf <= 5) return nf < 4 ? 4 : nf;
  if (nf % 5 == 0) return 5;
  return (round_up(nf, 9) - nf) * 8 <= (round_up(nf, 8) - nf) * 9 ? 9 : 8;
}
// Wide output tiles only (128/144 channels): a 64/80-channel tile issues one 1-KiB activation load per 4-5 MFMAs and the
// texture path, not the LDS, becomes the limit (measured 570 vs 613-741 TFLOP/s for igemm2/igemm3 on 64-channel layers).
bool rg_igemm4_ok(const CL5& X, const rg_conv_params* p) {
  if (ig4_fn(round_up(p->cout, 16) / 16) < 8) return false;
  const long wel = (long)round_up(p->cout, 16) * p->T[0] * p->T[1] * p->T[2] * round_up(X.C, 64);
  // byte offsets are 32-bit and a parked lane sits at 2^31 -/+ one tap offset: everything addressable must stay below 2^30
  return cl5_extent(X) * 2 < (1L << 30) && wel * 2 < (1L << 30);
}
int rg_igemm4_tile_n(int cout) { return ig4_fn(round_up(cout, 16) / 16); }
// Column tile of rg_gemm_batched for n columns, in 16-column fragments (8 = 128, 9 = 144 columns), or 0 when n has no such
// tiling and the call would be refused (n = 145..160, 305..320, ...): callers pick another path BEFORE transforming operands.
extern "C" int32_t rg_gemm_batched_tile_n(int32_t n) {
  if (n <= 0) return 0;
  const int fn = ig4_fn(round_up(n, 16) / 16);
  return fn < 8 ? 0 : fn;
}

bool rg_igemm_thin_ok(const CL5& X, const rg_conv_params* p);
int rg_launch_igemm_thin(const Igemm4Args& a0, bool check, hipStream_t st);
bool rg_halo16_ok(const CL5& X, const CL5& Y, const rg_conv_params* p);   // conv5.hip: input halo tile staged in LDS
int rg_launch_halo16(const CL5& X, const CL5& Y, const void* w_packed, const float* bias, const rg_conv_params* p, int krow, hipStream_t st);
bool rg_igemm5_ok(const Igemm4Args& a, bool flat, size_t ws_bytes);
int rg_launch_igemm5(Igemm4Args& a, bool check, hipStream_t st, float* ws, size_t ws_bytes);
struct Igemm4Batch { int n; long xs, ws, ys; };
static int launch_igemm4_any(const CL5& X, const CL5& Y, const void* w_packed, const float* bias, const rg_conv_params* p, hipStream_t st, int flat,
                             void* workspace, size_t ws_bytes, const Igemm4Batch* bt, int d2s_c = 0);
int rg_launch_igemm4(const CL5& X, const CL5& Y, const void* w_packed, const float* bias, const rg_conv_params* p, hipStream_t st, int flat,
                     void* workspace, size_t ws_bytes) {
  return launch_igemm4_any(X, Y, w_packed, bias, p, st, flat, workspace, ws_bytes, nullptr);
}
// the sub-pixel form of a stride-2 transposed convolution with the depth-to-space shuffle in igemm5's epilogue (conv.hip: rg_conv3d_igemm_d2s);
// dry != 0: only answer whether this problem can run that way (RG_OK / RG_ERR_UNSUPPORTED)
int rg_launch_igemm5_d2s(const CL5& X, const CL5& Y, const void* w_packed, const float* bias, const rg_conv_params* p, hipStream_t st, int dry) {
  return launch_igemm4_any(X, Y, w_packed, bias, p, st, 0, nullptr, 0, nullptr, dry ? -Y.C : Y.C);
}
static int launch_igemm4_any(const CL5& X, const CL5& Y, const void* w_packed, const float* bias, const rg_conv_params* p, hipStream_t st, int flat,
                             void* workspace, size_t ws_bytes, const Igemm4Batch* bt, int d2s_c) {
  Igemm4Args a;
  a.d2s_c = 0;
  a.wld = 0; a.nbatch1 = 0; a.wbatch2 = 0; a.raw = nullptr; a.raw_ld = 0; a.raw_batch = 0;
  a.sk_grid = 0; a.sk_part = nullptr; a.sk_flags = nullptr; a.sk_epoch = 0;
  a.nbatch = bt ? bt->n : 1; a.xbatch = bt ? bt->xs : 0; a.wbatch = bt ? bt->ws : 0; a.ybatch = bt ? bt->ys : 0;
  a.x = X.p; a.w = (const bf16_t*)w_packed; a.bias = bias; a.y = Y.p;
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
  if (a.M >= (1L << 31)) RG_FAIL(RG_ERR_UNSUPPORTED, "conv: 2^31 output rows or more (the kernels decode rows with 32-bit divisions)");
  const int ntaps = a.TD * a.TH * a.TW;
  a.cp = X.lanes / 8;
  a.krow = round_up(ntaps * X.lanes, 64);
  const long wel = (long)a.wrows * (flat ? a.krow : ntaps * a.cink);
  if (cl5_extent(X) * 2 >= (1L << 30) || wel * 2 >= (1L << 30))
    RG_FAIL(RG_ERR_UNSUPPORTED, "conv: input view or packed weights of 1 GiB or more are not supported by the fragment-direct kernel");
  a.xbytes = (unsigned)(cl5_extent(X) * 2);
  a.wbytes = (unsigned)(wel * 2);
  bool check = false;
  const int I[3] = {X.D, X.H, X.W};
  for (int d = 0; d < 3; ++d) {
    const long lo0 = p->tbase[d], lo1 = p->tbase[d] + (long)p->tstep[d] * (p->T[d] - 1);
    const long mn = lo0 < lo1 ? lo0 : lo1, mx = (lo0 < lo1 ? lo1 : lo0) + (long)(p->R[d] - 1) * p->rstride[d];
    if (mn < 0 || mx >= I[d]) check = true;
    if (flat && (long)(p->T[d] - 1) * (p->tstep[d] < 0 ? -p->tstep[d] : p->tstep[d]) > 511)
      RG_FAIL(RG_ERR_UNSUPPORTED, "conv: tap displacement beyond 511 voxels along dim %d", d);
  }
  if (flat && a.krow > 48 * 1024) RG_FAIL(RG_ERR_UNSUPPORTED, "conv: K = %d per output row does not fit the piece table", a.krow);
  if (bt && flat) RG_FAIL(RG_ERR_UNSUPPORTED, "gemm_batched: K must be a multiple of 64 lanes");
  if (d2s_c) {   // columns = 8 parity classes x |d2s_c| channels, written interleaved (no split-K: the combine kernel does not shuffle)
    a.cout_lanes = round_up(p->cout, 8);
    a.os[0] = a.os[1] = a.os[2] = 2; a.oo[0] = a.oo[1] = a.oo[2] = 0;
    if (!rg_igemm5_ok(a, false, 0)) { rg_set_error("conv_d2s: no igemm5 tiling without split-K for this problem"); return RG_ERR_UNSUPPORTED; }
    if (d2s_c < 0) return RG_OK;
    a.d2s_c = d2s_c;
    return rg_launch_igemm5(a, check, st, nullptr, 0);
  }
  if (rg_igemm5_ok(a, flat != 0, ws_bytes)) return rg_launch_igemm5(a, check, st, (float*)workspace, ws_bytes);   // 256 x 288 / 256 tiles, both operands through LDS by LDS-DMA (gemm5.hip)
  if (flat && rg_halo16_ok(X, Y, p) && !rg_opt(RG_OPT_NO_HALO)) return rg_launch_halo16(X, Y, w_packed, bias, p, a.krow, st);
  if (flat && rg_igemm_thin_ok(X, p) && a.M < (1L << 31) && cl5_extent(Y) < (1L << 31) && !rg_opt(RG_OPT_NO_THIN)) return rg_launch_igemm_thin(a, check, st);
  const int nf = a.wrows / 16;
  const int fn = flat ? (nf <= 6 ? nf : (nf == 7 ? 8 : ig4_fn(nf))) : ig4_fn(nf);
  // rows per tile: 256 (FM = 4, two workgroups per CU, weight-fragment reads amortised over 4 row fragments: ~1000-1060
  // TFLOP/s when the grid fills the chip) or 128 (FM = 2, three per CU, ~820-900).  Few-tile problems lose whole CUs to
  // the coarser grid, so estimate both from the measured rates and the fill of the last round of workgroups
  const int ntn = (nf + fn - 1) / fn;
  const long b2 = ((a.M + 127) / 128) * ntn * a.nbatch, b4 = ((a.M + 255) / 256) * ntn * a.nbatch;
  const double e2 = b2 <= 768 ? b2 / 768.0 : 0.5 + 0.5 * b2 / (double)((b2 + 767) / 768 * 768);  // co-resident workgroups speed up when a CU is not full
  const double e4 = b4 <= 512 ? b4 / 512.0 : b4 / (double)((b4 + 511) / 512 * 512);
  const int e = rg_opt(RG_OPT_FM);   // forced tile height (diagnostics / tests), 0 = the heuristic
  int fm = e ? e : (1030.0 * e4 > 880.0 * e2 ? 4 : 2);
  // 192-row tiles (FM = 3, 144-channel tiles only) for grids that leave the last round of 256- and 128-row tiles half empty
  // (2592 -> 2592 at 17^3: 360 / 468 / 702 tiles -> 821 TFLOP/s with FM = 2, 887 with FM = 3); never for grids below one tile per CU
  if (!e && (fn == 9 || fn == 8) && a.rs[0] == 1 && a.rs[1] == 1 && a.rs[2] == 1) {  // (strided gathers measured slower with 192 rows)
    const long b3 = ((a.M + 191) / 192) * ntn * a.nbatch;
    const double e3 = b3 <= 512 ? b3 / 512.0 : b3 / (double)((b3 + 511) / 512 * 512);
    if (b3 >= 256 && 950.0 * e3 > 1.03 * (fm == 4 ? 1030.0 * e4 : 880.0 * e2)) fm = 3;
  }
  // small grids (PatchGAN's 15^3/16^3 layers, the parity classes of the transposed convs): 64-row tiles so that more CUs get a tile
  if (!e && fm == 2 && b2 < 192) fm = 1;
  float* wsf = (float*)workspace;
#define RG_L4(FN_) do { if (fm == 4) launch_igemm4<4, FN_>(a, check, flat != 0, st, wsf, ws_bytes); else if (fm == 3 && (FN_ == 9 || FN_ == 8)) launch_igemm4<3, (FN_ == 8 ? 8 : 9)>(a, check, flat != 0, st, wsf, ws_bytes); else if (fm == 1) launch_igemm4<1, FN_>(a, check, flat != 0, st, wsf, ws_bytes); \
                        else launch_igemm4<2, FN_>(a, check, flat != 0, st, wsf, ws_bytes); } while (0)
  switch (fn) {
    case 1: RG_L4(1); break;
    case 2: RG_L4(2); break;
    case 3: RG_L4(3); break;
    case 4: RG_L4(4); break;
    case 5: RG_L4(5); break;
    case 6: RG_L4(6); break;
    case 8: RG_L4(8); break;
    default: RG_L4(9); break;
  }
  return RG_OK;
}

// C[b][m][n] = sum_k A[b][m][k] * W[b][n][k]: `batches` independent GEMMs on the fragment-direct kernel (one launch; the
// transform-domain products of the Winograd path, winograd.hip).  A rows are a_rs elements apart (lanes past k must hold zeros up
// to the next multiple of 64, like W's rows), C rows c_rs; 16-bit storage type of this build, fp32 accumulation.
extern "C" int rg_gemm_batched(const void* A, const void* W, void* Cm, int64_t batches, int64_t m, int32_t k, int32_t n, int64_t a_rs,
                               int64_t a_bs, int64_t w_bs, int64_t c_rs, int64_t c_bs, rg_stream stream) {
  if (!A || !W || !Cm || batches <= 0 || m <= 0 || k <= 0 || n <= 0) RG_FAIL(RG_ERR_BAD_SHAPE, "gemm_batched: null pointer or empty problem");
  const int kp = round_up(k, 64);
  if (a_rs < kp || (a_rs & 7) || c_rs < round_up(n, 8) || (c_rs & 7) || m >= (1L << 31) || batches >= (1 << 16))
    RG_FAIL(RG_ERR_BAD_SHAPE, "gemm_batched: row strides must cover k rounded up to 64 / n rounded up to 8 and be multiples of 8");

  CL5 X, Y;
  X.p = (bf16_t*)A; X.N = 1; X.C = k; X.D = 1; X.H = 1; X.W = (int)m; X.lanes = kp; X.sn = m * a_rs; X.sd = m * a_rs; X.sh = m * a_rs; X.sw = a_rs;
  Y.p = (bf16_t*)Cm; Y.N = 1; Y.C = n; Y.D = 1; Y.H = 1; Y.W = (int)m; Y.lanes = round_up(n, 8); Y.sn = m * c_rs; Y.sd = m * c_rs; Y.sh = m * c_rs; Y.sw = c_rs;
  rg_conv_params p;
  memset(&p, 0, sizeof(p));
  p.R[0] = 1; p.R[1] = 1; p.R[2] = (int)m;
  for (int d = 0; d < 3; ++d) { p.T[d] = 1; p.rstride[d] = 1; p.tstep[d] = 1; p.ostride[d] = 1; }
  p.cout = n; p.act = RG_ACT_NONE;
  if (ig4_fn(round_up(n, 16) / 16) < 8) RG_FAIL(RG_ERR_UNSUPPORTED, "gemm_batched: n = %d has no 128/144-column tiling", n);
  Igemm4Batch bt; bt.n = (int)batches; bt.xs = a_bs; bt.ws = w_bs; bt.ys = c_bs;
  return launch_igemm4_any(X, Y, W, nullptr, &p, (hipStream_t)stream, 0, nullptr, 0, &bt);
}

// ---- thin outputs (Cout <= 16): weights resident in LDS ----------------------------------------------------------------
// A 16-channel output tile cannot amortise a per-K-step weight hand-over (8 MFMAs per barrier in igemm4<4,1>): these layers
// are bound by the activation reads (one 1-KiB fragment load per MFMA), so everything else is taken off the loop.  The
// whole packed weight matrix (16 rows x K, FLAT packing, <= 128 KB) is staged into LDS ONCE per workgroup, in 64-wide K
// chunks with the fragment-read swizzle; workgroups are persistent over row tiles; the K loop has no barrier — per step
// and lane two piece-table look-ups, 2*FM fragment loads (one step ahead, second register set), two ds_read_b128 and
// 2*FM MFMAs.
template <int FM, bool CHECK>
__global__ __launch_bounds__(256) void igemm_thin_kernel(Igemm4Args a) {
  extern __shared__ __attribute__((aligned(128))) unsigned char smem[];
  const int nk = a.krow >> 6;
  bf16_t* wl = (bf16_t*)smem;                                  // [nk][16][64] swizzled
  PieceInfo* ptab = (PieceInfo*)(smem + (size_t)nk * 2048);    // [nk*8]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ntaps = a.TD * a.TH * a.TW;
  const int fr = lane & 15, fg = lane >> 4;
  // stage the weights: 16-byte piece q = (row, k-piece); LDS slot = kslot ^ (row >> 1)
  for (int q = tid; q < 16 * (a.krow >> 3); q += 256) {
    const int row = q / (a.krow >> 3), kp = q - row * (a.krow >> 3);
    const u32x4_t v = *(const u32x4_t*)(a.w + (long)row * a.krow + kp * 8);
    *(u32x4_t*)(wl + (kp >> 3) * 1024 + row * 64 + (((kp & 7) ^ (row >> 1)) << 3)) = v;
  }
  for (int P = tid; P < nk * 8; P += 256) {
    const int tap = P / a.cp, c8 = P - tap * a.cp;
    PieceInfo e; e.off = 0; e.disp = -1;
    if (tap < ntaps) {
      const int kw_ = tap % a.TW, kh_ = (tap / a.TW) % a.TH, kd_ = tap / (a.TW * a.TH);
      const int od = kd_ * a.ts[0], oh = kh_ * a.ts[1], ow = kw_ * a.ts[2];
      e.off = ((int)(od * a.xsd + oh * a.xsh + ow * a.xsw) + c8 * 8) * 2;
      e.disp = (od + 512) | ((oh + 512) << 10) | ((ow + 512) << 20);
    }
    ptab[P] = e;
  }
  __syncthreads();
  const auto rx = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, (short)0, (int)a.xbytes, 0x00020000);
  const unsigned wl0 = (unsigned)(size_t)(lptr4_t)wl + fr * 128 + ((fg ^ (fr >> 1)) << 4);
  float bias4[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) bias4[r] = (a.bias && fg * 4 + r < a.cout) ? a.bias[fg * 4 + r] : 0.f;

  const int xsn = (int)a.xsn, xsd = (int)a.xsd, xsh = (int)a.xsh, xsw = (int)a.xsw, ysn = (int)a.ysn, ysd = (int)a.ysd, ysh = (int)a.ysh, ysw = (int)a.ysw;
  for (int tile = blockIdx.x; tile < a.tiles_m; tile += gridDim.x) {
    const long mw = (long)tile * (64 * FM) + wave * (16 * FM);
    // row decode: one 32-bit division chain for fragment 0 (host guarantees M < 2^31), the other fragments step 16 rows
    // on with carries — 64-bit divisions per fragment cost more than the whole K loop of a 3^3 x 16-lane layer
    int rd0[FM], rh0[FM], rw0[FM]; unsigned rbase[FM]; int yoff[FM];
    {
      unsigned t = (unsigned)(mw + fr);
      int rw = (int)(t % (unsigned)a.RW); t /= (unsigned)a.RW;
      int rh = (int)(t % (unsigned)a.RH); t /= (unsigned)a.RH;
      int rd = (int)(t % (unsigned)a.RD); int n = (int)(t / (unsigned)a.RD);
#pragma unroll
      for (int i = 0; i < FM; ++i) {
        const bool v = mw + i * 16 + fr < a.M;
        rd0[i] = rd * a.rs[0] + a.tb[0]; rh0[i] = rh * a.rs[1] + a.tb[1]; rw0[i] = rw * a.rs[2] + a.tb[2];
        // 32-bit offsets: the launcher admits only tensors whose extents fit (M, x and y extents < 2^31)
        rbase[i] = v ? (unsigned)(n * xsn + rd0[i] * xsd + rh0[i] * xsh + rw0[i] * xsw) * 2u : IG4_OOB;
        yoff[i] = v ? n * ysn + (rd * a.os[0] + a.oo[0]) * ysd + (rh * a.os[1] + a.oo[1]) * ysh + (rw * a.os[2] + a.oo[2]) * ysw : -1;
        rw += 16;
        while (rw >= a.RW) { rw -= a.RW; if (++rh == a.RH) { rh = 0; if (++rd == a.RD) { rd = 0; ++n; } } }
      }
    }
    f32x4 acc[FM];
#pragma unroll
    for (int i = 0; i < FM; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    u32x4_t xa[2][FM][2];
    auto fetch = [&](int ks, u32x4_t (&x)[FM][2]) {
      const PieceInfo e0 = ptab[ks * 8 + fg], e1 = ptab[ks * 8 + 4 + fg];
      const int d0 = (e0.disp & 1023) - 512, h0 = ((e0.disp >> 10) & 1023) - 512, w0 = ((e0.disp >> 20) & 1023) - 512;
      const int d1 = (e1.disp & 1023) - 512, h1 = ((e1.disp >> 10) & 1023) - 512, w1 = ((e1.disp >> 20) & 1023) - 512;
#pragma unroll
      for (int i = 0; i < FM; ++i) {
        bool ok0 = e0.disp >= 0, ok1 = e1.disp >= 0;
        if (CHECK) {
          ok0 &= ((unsigned)(rd0[i] + d0) < (unsigned)a.ID) & ((unsigned)(rh0[i] + h0) < (unsigned)a.IH) & ((unsigned)(rw0[i] + w0) < (unsigned)a.IW);
          ok1 &= ((unsigned)(rd0[i] + d1) < (unsigned)a.ID) & ((unsigned)(rh0[i] + h1) < (unsigned)a.IH) & ((unsigned)(rw0[i] + w1) < (unsigned)a.IW);
        }
        x[i][0] = __builtin_amdgcn_raw_buffer_load_b128(rx, (int)(ok0 ? rbase[i] + (unsigned)e0.off : IG4_OOB), 0, 0);
        x[i][1] = __builtin_amdgcn_raw_buffer_load_b128(rx, (int)(ok1 ? rbase[i] + (unsigned)e1.off : IG4_OOB), 0, 0);
      }
    };
    auto mma = [&](int ks, u32x4_t (&x)[FM][2]) {
      const bf16x8 w0 = *(const bf16x8*)(wl + ks * 1024 + fr * 64 + ((fg ^ (fr >> 1)) << 3));
      const bf16x8 w1 = *(const bf16x8*)(wl + ks * 1024 + fr * 64 + (((fg + 4) ^ (fr >> 1)) << 3));
#pragma unroll
      for (int i = 0; i < FM; ++i) {
        bf16x8 f0, f1;
        __builtin_memcpy(&f0, &x[i][0], 16); __builtin_memcpy(&f1, &x[i][1], 16);
        acc[i] = RG_MFMA16(w0, f0, acc[i]);
        acc[i] = RG_MFMA16(w1, f1, acc[i]);
      }
    };
    (void)wl0;
    fetch(0, xa[0]);
    for (int ks = 0; ks < nk; ks += 2) {
      if (ks + 1 < nk) fetch(ks + 1, xa[1]);
      mma(ks, xa[0]);
      if (ks + 1 < nk) {
        if (ks + 2 < nk) fetch(ks + 2, xa[0]);
        mma(ks + 1, xa[1]);
      }
    }
#pragma unroll
    for (int i = 0; i < FM; ++i) {
      if (yoff[i] < 0) continue;
      bf16_t* yrow = a.y + yoff[i];
      const int c = fg * 4;
      if (c >= a.cout_lanes) continue;
      *(uint2*)(yrow + c) = a.act <= RG_ACT_LEAKY ? rg_epilogue_quad<true>(acc[i], bias4, a.cout - c, rg_act_nslope(a.act, a.slope), a.act, a.slope)
                                                   : rg_epilogue_quad<false>(acc[i], bias4, a.cout - c, 0.f, a.act, a.slope);
    }
  }
}

// ---- thin kernel with kw reuse ---------------------------------------------------------------------------------------------
// igemm_thin re-reads every input voxel once per tap: 27 x 32 B per 24-B output row of a 12 -> 12 channel 3^3 layer, and the
// texture path (64 B/clk/CU) is what bounds it (55 % of that peak at 192^3).  Here a 16-row MFMA fragment is 16 consecutive
// voxels of ONE W line (fragments are line-aligned; the tail of a line is masked), so the kw = 1, 2 taps of row r are the
// kw = 0 voxels r+1, r+2 of the same line (r-1, r-2 when the taps run backwards): every wave parks the (kd, kh, 8-channel piece)
// operands of a fragment plus its two neighbour voxels in a private LDS strip and reads the shifted taps from there — one third
// of the global loads, no special cases.  Waves only read their own strips, in program order: no barrier.
// K order: base piece bp = (kd*TH + kh)*cp + c8, four per MFMA group (one per 16-lane row), three MFMAs (kw) per group; the
// weights stay in igemm_thin's LDS layout and are addressed by their piece index P = ((kd*TH + kh)*TW + kw)*cp + c8.
template <int FM, bool CHECK>
__global__ __launch_bounds__(256) void igemm_thin3_kernel(Igemm4Args a) {
  extern __shared__ __attribute__((aligned(128))) unsigned char smem[];
  const int nk = a.krow >> 6;
  bf16_t* wl = (bf16_t*)smem;                                  // [nk][16][64] swizzled
  constexpr int SE = 20;                                       // strip entries per (piece row, fragment): 16 + 2, padded
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fg = lane >> 4;
  const int BP = a.TD * a.TH * a.cp, NG = (BP + 3) >> 2;
  int4* gtab = (int4*)(smem + (size_t)nk * 2048);              // [(NG+2)*4] base pieces: {byte offset, od | oh << 16, P0, valid}; the tail is invalid
  unsigned char* strip = smem + (size_t)nk * 2048 + (size_t)(NG + 2) * 64 + (size_t)(wave * 4 + fg) * FM * SE * 16;
  for (int q = tid; q < 16 * (a.krow >> 3); q += 256) {
    const int row = q / (a.krow >> 3), kp = q - row * (a.krow >> 3);
    const u32x4_t v = *(const u32x4_t*)(a.w + (long)row * a.krow + kp * 8);
    *(u32x4_t*)(wl + (kp >> 3) * 1024 + row * 64 + (((kp & 7) ^ (row >> 1)) << 3)) = v;
  }
  for (int bp = tid; bp < (NG + 2) * 4; bp += 256) {
    const int q = bp / a.cp, c8 = bp - q * a.cp;
    const int kd = q / a.TH, kh = q - kd * a.TH;
    const int od = kd * a.ts[0], oh = kh * a.ts[1];
    const bool ok = bp < BP;
    gtab[bp] = make_int4(ok ? ((int)(od * a.xsd + oh * a.xsh) + c8 * 8) * 2 : 0, ok ? ((od + 512) | ((oh + 512) << 16)) : (512 | (512 << 16)),
                         ok ? q * a.TW * a.cp + c8 : 0, ok ? 1 : 0);
  }
  __syncthreads();
  const auto rx = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, (short)0, (int)a.xbytes, 0x00020000);
  const int dir = a.ts[2], shift0 = dir < 0 ? 2 : 0;
  const int fpl = (a.RW + 15) >> 4;                            // fragments per W line
  const long nfrag = (long)a.N * a.RD * a.RH * fpl;
  const int ei = (fr >> 1) & (FM - 1), ee = fr & 1;            // this lane's edge duty: fragment ei, neighbour voxel ee (lanes fr < 2*FM)
  const int eslot = (ei * SE + (fr < 2 * FM ? (dir > 0 ? 16 + ee : ee) : 18 + ee)) * 16;
  float bias4[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) bias4[r] = (a.bias && fg * 4 + r < a.cout) ? a.bias[fg * 4 + r] : 0.f;

  const int xsn = (int)a.xsn, xsd = (int)a.xsd, xsh = (int)a.xsh, xsw = (int)a.xsw, ysn = (int)a.ysn, ysd = (int)a.ysd, ysh = (int)a.ysh, ysw = (int)a.ysw;
  for (int tile = blockIdx.x; tile < a.tiles_m; tile += gridDim.x) {
    const long f0 = ((long)tile * 4 + wave) * FM;
    int rd0[FM], rh0[FM], rw0[FM]; unsigned rbase[FM]; int yoff[FM];
    int ed0 = 0, eh0 = 0, ew0 = 0; unsigned ebase = IG4_OOB;
    {
      // one division chain for the wave's first fragment, the others step one fragment on with carries
      unsigned t = (unsigned)f0;
      int wf = (int)(t % (unsigned)fpl); t /= (unsigned)fpl;
      int rh = (int)(t % (unsigned)a.RH); t /= (unsigned)a.RH;
      int rd = (int)(t % (unsigned)a.RD); int n = (int)(t / (unsigned)a.RD);
#pragma unroll
      for (int i = 0; i < FM; ++i) {
        const bool live = f0 + i < nfrag;
        const int rw = wf * 16 + fr;
        rd0[i] = rd * a.rs[0] + a.tb[0]; rh0[i] = rh * a.rs[1] + a.tb[1]; rw0[i] = rw + a.tb[2];
        const int lbase = n * xsn + rd0[i] * xsd + rh0[i] * xsh;   // 32-bit: extents < 2^31 (launcher)
        rbase[i] = live ? (unsigned)(lbase + rw0[i] * xsw) * 2u : IG4_OOB;   // rows beyond RW still feed their neighbours
        yoff[i] = live && rw < a.RW ? n * ysn + (rd * a.os[0] + a.oo[0]) * ysd + (rh * a.os[1] + a.oo[1]) * ysh + (rw * a.os[2] + a.oo[2]) * ysw : -1;
        // neighbour voxels of fragment i (lanes 2i, 2i+1 of every piece row): 16, 17 past the fragment start (taps forward) or 2, 1 before it
        const int w_e = wf * 16 + a.tb[2] + (dir > 0 ? 16 + ee : ee - 2);
        const bool mine = i == ei;
        ed0 = mine ? rd0[i] : ed0; eh0 = mine ? rh0[i] : eh0; ew0 = mine ? w_e : ew0;
        ebase = mine ? (live ? (unsigned)(lbase + w_e * xsw) * 2u : IG4_OOB) : ebase;
        if (++wf == fpl) { wf = 0; if (++rh == a.RH) { rh = 0; if (++rd == a.RD) { rd = 0; ++n; } } }
      }
      if (fr >= 2 * FM) ebase = IG4_OOB;
    }
    f32x4 acc[FM];
#pragma unroll
    for (int i = 0; i < FM; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    u32x4_t xb[2][FM], xe[2];
    auto issue = [&](int g, u32x4_t (&b)[FM], u32x4_t& e) {  // always FM + 1 loads (groups past the end read out of range)
      const int4 t = gtab[g * 4 + fg];
      const int off = t.x, od = (t.y & 0xffff) - 512, oh = (t.y >> 16) - 512;
      const bool ok = t.w != 0;
#pragma unroll
      for (int i = 0; i < FM; ++i) {
        bool k = ok;
        k &= ((unsigned)rw0[i] < (unsigned)a.IW);  // rows past the line end may point past the input line
        if (CHECK) k &= ((unsigned)(rd0[i] + od) < (unsigned)a.ID) & ((unsigned)(rh0[i] + oh) < (unsigned)a.IH);
        b[i] = __builtin_amdgcn_raw_buffer_load_b128(rx, (int)(k ? rbase[i] + (unsigned)off : IG4_OOB), 0, 0);
      }
      bool k = ok && ((unsigned)ew0 < (unsigned)a.IW);
      if (CHECK) k &= ((unsigned)(ed0 + od) < (unsigned)a.ID) & ((unsigned)(eh0 + oh) < (unsigned)a.IH);
      e = __builtin_amdgcn_raw_buffer_load_b128(rx, (int)(k ? ebase + (unsigned)off : IG4_OOB), 0, 0);
    };
    auto step = [&](int g, u32x4_t (&b)[FM], u32x4_t& e) {
      const int4 t = gtab[g * 4 + fg];
#pragma unroll
      for (int i = 0; i < FM; ++i) *(u32x4_t*)(strip + (i * SE + fr + shift0) * 16) = b[i];
      *(u32x4_t*)(strip + eslot) = e;  // lanes without edge duty write zeros to a pad entry
      __builtin_amdgcn_wave_barrier();
      u32x4_t x1[FM], x2[FM];
#pragma unroll
      for (int i = 0; i < FM; ++i) {
        x1[i] = *(const u32x4_t*)(strip + (i * SE + fr + shift0 + dir) * 16);
        x2[i] = *(const u32x4_t*)(strip + (i * SE + fr + shift0 + 2 * dir) * 16);
      }
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int P = t.z + (t.w ? kw * a.cp : 0);   // piece index of tap kw (pad lanes: piece 0, their operand is zero)
        const bf16x8 wv = *(const bf16x8*)(wl + (P >> 3) * 1024 + fr * 64 + (((P & 7) ^ (fr >> 1)) << 3));
#pragma unroll
        for (int i = 0; i < FM; ++i) {
          bf16x8 f;
          if (kw == 0) __builtin_memcpy(&f, &b[i], 16); else if (kw == 1) __builtin_memcpy(&f, &x1[i], 16); else __builtin_memcpy(&f, &x2[i], 16);
          acc[i] = RG_MFMA16(wv, f, acc[i]);
        }
      }
    };
    // Two statically indexed register sets, one K group of look-ahead.  Every iteration issues both its look-ahead groups
    // unconditionally (groups past the end are invalid table entries: out-of-range loads, no memory traffic), so the loop body is
    // straight-line as far as memory operations go and hipcc's vmcnt bookkeeping stays exact (with loads under `if (g + 1 < NG)`
    // it merged the paths and waited for the YOUNGER set too).  At most two dummy groups per tile: the kernel is used for
    // cp >= 4 (>= 9 groups per tile); 16-lane layers (5 groups) stay on igemm_thin.
    issue(0, xb[0], xe[0]);
    for (int g = 0; g < NG; g += 2) {
      issue(g + 1, xb[1], xe[1]);
      step(g, xb[0], xe[0]);
      issue(g + 2, xb[0], xe[0]);
      if (g + 1 < NG) step(g + 1, xb[1], xe[1]);
    }
#pragma unroll
    for (int i = 0; i < FM; ++i) {
      if (yoff[i] < 0) continue;
      bf16_t* yrow = a.y + yoff[i];
      const int c = fg * 4;
      if (c >= a.cout_lanes) continue;
      *(uint2*)(yrow + c) = a.act <= RG_ACT_LEAKY ? rg_epilogue_quad<true>(acc[i], bias4, a.cout - c, rg_act_nslope(a.act, a.slope), a.act, a.slope)
                                                   : rg_epilogue_quad<false>(acc[i], bias4, a.cout - c, 0.f, a.act, a.slope);
    }
  }
}

// weights-resident thin kernel: Cout <= 16 and the packed matrix (plus the piece table) fits 128 KB of LDS
bool rg_igemm_thin_ok(const CL5& X, const rg_conv_params* p) {
  if (p->cout > 16) return false;
  const int krow = round_up(p->T[0] * p->T[1] * p->T[2] * X.lanes, 64);
  return (size_t)krow * 32 + (size_t)(krow / 8) * sizeof(PieceInfo) <= 128 * 1024;
}
int rg_launch_igemm_thin(const Igemm4Args& a0, bool check, hipStream_t st) {
  Igemm4Args a = a0;
  constexpr int FM = 4;
  a.tiles_m = (int)((a.M + 64 * FM - 1) / (64 * FM));
  const size_t shm = (size_t)a.krow * 32 + (size_t)(a.krow / 8) * sizeof(PieceInfo);
  // persistent workgroups: as many as fit the CUs (LDS-limited), every one stages the weights once
  int per_cu = (int)((160 * 1024) / (shm + 1024)); if (per_cu > 8) per_cu = 8; if (per_cu < 1) per_cu = 1;
  int grid = 256 * per_cu; if (grid > a.tiles_m) grid = a.tiles_m;
  if (a.TW == 3 && (a.ts[2] == 1 || a.ts[2] == -1) && a.rs[2] == 1 && a.cp >= 4 && !rg_opt(RG_OPT_NO_THIN3)) {
    const size_t shm3 = (size_t)a.krow * 32 + (size_t)((a.TD * a.TH * a.cp + 3) / 4 + 2) * 64 + (size_t)16 * FM * 20 * 16;
    const long nfrag = (long)a.N * a.RD * a.RH * ((a.RW + 15) / 16);
    a.tiles_m = (int)((nfrag + 4 * FM - 1) / (4 * FM));
    int pc = (int)((160 * 1024) / (shm3 + 1024)); if (pc > 8) pc = 8; if (pc < 1) pc = 1;
    int grid3 = 256 * pc; if (grid3 > a.tiles_m) grid3 = a.tiles_m;
    rg_set_kernel_name("igemm_thin3_kernel<%d,%d>", FM, (int)check);
    static rg_device_latch attr3;
    rg_once_per_device(&attr3, [] {
      hipFuncSetAttribute((const void*)igemm_thin3_kernel<4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024);
      hipFuncSetAttribute((const void*)igemm_thin3_kernel<4, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024);
    });
    if (check) hipLaunchKernelGGL((igemm_thin3_kernel<FM, true>), dim3(grid3), dim3(256), shm3, st, a);
    else hipLaunchKernelGGL((igemm_thin3_kernel<FM, false>), dim3(grid3), dim3(256), shm3, st, a);
    return RG_OK;
  }
  rg_set_kernel_name("igemm_thin_kernel<%d,%d>", FM, (int)check);
  static rg_device_latch attr1;
  rg_once_per_device(&attr1, [] {
    hipFuncSetAttribute((const void*)igemm_thin_kernel<4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024);
    hipFuncSetAttribute((const void*)igemm_thin_kernel<4, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024);
  });
  if (check) hipLaunchKernelGGL((igemm_thin_kernel<FM, true>), dim3(grid), dim3(256), shm, st, a);
  else hipLaunchKernelGGL((igemm_thin_kernel<FM, false>), dim3(grid), dim3(256), shm, st, a);
  return RG_OK;
}
