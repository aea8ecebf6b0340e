// gemm5 — 256 x 288 x 64 tiles, 8 waves, BOTH operands through LDS by LDS-DMA in full 128-byte lines.
//
// igemm4 (conv4.hip) feeds the activation operand straight from global memory into MFMA fragments: 16 rows x 64 B per
// wave-instruction.  rocprofv3 on the U-Net step (profiles/r03a_unet_pmc_mfma.csv) shows what that costs: the MFMA pipe of
// igemm4<4,9> is 37-44 % busy while its waves spend 50 % of their cycles stalled at issue — the vector-memory path, not the
// matrix pipe, paces the loop (fragment-shaped loads keep the texture addresser busy twice as long per byte as full lines).
// Here every global byte moves as part of a whole 128-byte line, lands in LDS without passing a register
// (`buffer_load_dwordx4 ... offen lds`, 1 KiB per wave-instruction; out-of-range lanes store ZEROS — probed on gfx950,
// tools/kbench/ldsdma_probe.hip — so row tails and zero padding stay branch-free) and is shared by the waves that need it:
//
//   workgroup   8 waves = 4 (rows, 64 each) x 2 (columns, 144 each); wave tile 64 x 144 = 4 x 9 MFMA 16x16x32 accumulators
//   LDS         2 stages x (A 256 rows + B 320 rows) x 128 B = 144 KiB; 16-byte slot of piece p in row R: p ^ ((R >> 1) & 7)
//               (conflict-free ds_read_b128 for 16 consecutive rows); the DMA writes lane-linearly, so the swizzle is applied
//               to the SOURCE piece each lane fetches
//   K-tile      64 deep = 2 column groups x 2 k-halves = 4 phases of 20 / 20 / 16 / 16 MFMAs per wave; a phase is
//                 [fragment reads + DMA issue] lgkmcnt(0) s_barrier [MFMA cluster] s_barrier
//               and the two column groups (waves 0-3 / 4-7: one of each per SIMD) run ONE barrier apart, so on every SIMD one
//               wave issues MFMAs while the other reads fragments and feeds the DMA queue
//   prefetch    B(t+1) is issued in phases 0-1 of tile t into the stage tile t-1 was read from, A(t+2) in phases 2-3 into THIS
//               stage, whose A part is in registers by then (all those reads were retired by lgkmcnt(0) in front of a barrier
//               every wave has passed); every wave waits for its own pieces with ONE counted vmcnt per tile, in phase 3, in front
//               of a barrier; tile t+1 is first read one barrier later by the leading group, two by the trailing one
//
// C[b][m][n] = sum_k A[b][m][k] * W[b][n][k]  (+ bias[n], activation), 16-bit storage type of this build, fp32 accumulation.
#include "igemm_args.h"
#include <utility>
#include <atomic>
#include <mutex>

typedef int g5_i32x4 __attribute__((ext_vector_type(4)));

#define G5_BM 256
#define G5_BROWS 320                       // B rows a stage holds (FN = 9: 288 used, 5 DMA instructions; FN = 8: 256 used, 4 instructions)
#define G5_STAGE ((G5_BM + G5_BROWS) * 128) // bytes per stage
#define G5_OOB 0x80000000u
// dynamic LDS: the two stages, or the epilogue's 16-bit tile (256 rows of BN * 2 + 16 bytes) + row table, whichever is larger (FN = 9: 150 KiB of the CU's 160)
#define G5_LDS_BYTES(fn) ((2 * G5_STAGE) > (256 * (2 * (fn) * 16 * 2 + 16) + 2048) ? (2 * G5_STAGE) : (256 * (2 * (fn) * 16 * 2 + 16) + 2048))

__device__ __forceinline__ g5_i32x4 g5_rsrc(const void* base, unsigned nrec) {
  const unsigned long b = (unsigned long)base;
  return (g5_i32x4){(int)(unsigned)b, (int)((b >> 32) & 0xffffu), (int)nrec, 0x00020000};
}
// one LDS-DMA wave-instruction: lane l's 16 bytes at (rsrc + voff + soff) -> LDS[lds_base + 16 l]; zeros when out of range (voff +
// soff is range-checked as a whole on gfx950).  M0 (the destination base) is written in the statement that uses it (s_nop 0: SALU
// write of M0 -> LDS-DMA needs one wait state); nothing else in this kernel uses M0 (no movrel / GWS / sendmsg): not restored.
__device__ __forceinline__ void g5_dma(unsigned lds_base, g5_i32x4 rsrc, unsigned voff, int soff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" :: "s"(lds_base), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}
template <int OFF>
__device__ __forceinline__ void g5_ldr(bf16x8& dst, unsigned addr) { asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF)); }

template <class F, int... Is>
__device__ __forceinline__ void g5_impl(F&& f, std::integer_sequence<int, Is...>) { (f(std::integral_constant<int, Is>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void g5for(F&& f) { g5_impl(f, std::make_integer_sequence<int, N>{}); }

// The implicit-GEMM form (same row / tap / output maps and packed-weight format as igemm4, conv4.hip): A rows are output voxels, a
// K-tile is one 64-channel chunk of one tap (chunk outer, tap inner: consecutive tiles read neighbouring voxels); the tap's byte
// offset is added to each of the thread's 4 row offsets, rows / taps / channel lanes outside the tensor become out-of-range offsets
// (zeros).  rg_gemm_batched's problems are the 1x1x1-tap case with a batch index in the grid.
// FN = 16-column fragments per wave: 9 (144-column halves, 288-column tiles: the U-Net's 432 / 2592 / 3456-channel layers) or 8 (128-column
// halves, 256-column tiles: the CycleGAN's 256-channel ResNet blocks).
template <bool CHECK, int FN, bool RAW = false, bool SK = false>   // RAW: the weight-gradient form (raw fp32 tiles into the caller's matrix; its own instantiation = its own name in profiles)
__global__ __launch_bounds__(512, 1) void igemm5_kernel(Igemm4Args a) {
  constexpr int G5_BN = 2 * FN * 16, NB = (G5_BN + 63) / 64;   // columns per tile, B DMA instructions per thread and K-tile
  extern __shared__ __attribute__((aligned(128))) unsigned char g5_lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave & 3, wn = wave >> 2;     // row group (64 rows), column group (144 columns); column group = stagger group
  const int nwg = SK ? a.sk_grid : a.tiles_m * a.tiles_n * a.ksplit * a.nbatch;
  int bid = blockIdx.x;
  { const int q = nwg / 8, r = nwg % 8, xcd = bid % 8, idx = bid / 8; bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx; }
  const int wg = bid;
  const int ntaps = a.TD * a.TH * a.TW, nchunk = a.cink >> 6;
  const int kunits = nchunk * ntaps;
  // SK: this workgroup's run [u, u_end) of the flattened (tile, K-tile) space, one pass of the loop below per tile it touches.  Order: the piece of
  // the LAST tile first when the run ends inside that tile (a head piece: K-tiles 0 .., handed over as a partial), then the rest front to back (the
  // tail of the first tile — which this workgroup finishes — and whole tiles).  Head pieces start at K-tile 0 at time 0 everywhere and tail pieces
  // end together, so the workgroups of an XCD walk the weights in (two) lockstep fronts and B stays an L2 hit; natural order spreads the 32 of them
  // over 32 different K-tiles (measured 127 us against 95 for the ResNet-block forward with all hand-overs disabled).
  const long sk_units = SK ? (long)a.tiles_m * a.tiles_n * a.nbatch * kunits : 0;
  long u = SK ? sk_units * wg / nwg : 0;
  const long u_end = SK ? sk_units * (wg + 1) / nwg : 0;
  if (SK && u >= u_end) return;
  long sk_stop = u_end;                      // end of the front-to-back part
  bool head_first = false;
  if constexpr (SK) {
    const long ls = ((u_end - 1) / kunits) * kunits;
    if (ls > u && u_end < ls + kunits) { head_first = true; sk_stop = ls; }
  }
  for (;;) {
  int batch, split, tile_n, tile_m, ku0, ku1;
  long sk_tile = 0, sk_su = 0;
  if constexpr (SK) {
    long se;
    if (head_first) { sk_su = sk_stop; se = u_end; }
    else { sk_su = u; se = (u / kunits + 1) * kunits; if (se > sk_stop) se = sk_stop; }
    sk_tile = sk_su / kunits; ku0 = (int)(sk_su - sk_tile * kunits); ku1 = ku0 + (int)(se - sk_su);
    int tb = (int)sk_tile;
    batch = tb / (a.tiles_m * a.tiles_n); tb -= batch * (a.tiles_m * a.tiles_n);
    tile_n = tb / a.tiles_m; tile_m = tb - tile_n * a.tiles_m; split = 0;
  } else {
    batch = bid / (a.tiles_m * a.tiles_n * a.ksplit); bid -= batch * (a.tiles_m * a.tiles_n * a.ksplit);
    split = bid / (a.tiles_m * a.tiles_n); bid -= split * (a.tiles_m * a.tiles_n);
    tile_n = bid / a.tiles_m; tile_m = bid - tile_n * a.tiles_m;
    // K range of this workgroup in K-tiles (tile = chunk * ntaps + tap): everything, or the split's share
    const int kper = (kunits + a.ksplit - 1) / a.ksplit;
    ku0 = split * kper; ku1 = ku0 + kper < kunits ? ku0 + kper : kunits;
  }
  const int nk = ku1 > ku0 ? ku1 - ku0 : 0;
  const long m0 = (long)tile_m * G5_BM;
  const int n0 = tile_n * G5_BN;
  const bf16_t* const xbase = a.x + batch * a.xbatch;
  const bf16_t* const wbase = a.nbatch1 ? a.w + (batch % a.nbatch1) * a.wbatch + (batch / a.nbatch1) * a.wbatch2 : a.w + batch * a.wbatch;
  const g5_i32x4 ra = g5_rsrc(xbase, a.xbytes), rw = g5_rsrc(wbase, a.wbytes);
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) void*)g5_lds;

  // ---- DMA pieces of this thread: instruction i covers tile rows i*64 + wave*8 + (lane>>3); the lane fetches the piece that
  // belongs in ITS slot of the swizzled image
  const int drow = wave * 8 + (lane >> 3);                       // row within a 64-row instruction
  const int dpiece = (lane & 7) ^ ((drow >> 1) & 7);             // (i*64 does not change (row >> 1) & 7)
  int rd0[4], rh0[4], rw0[4]; unsigned va[4], vb[5];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const long m = m0 + i * 64 + drow;
    const bool v = m < a.M;
    unsigned t = v ? (unsigned)m : 0u;      // (M < 2^31: g5_plan; 32-bit divisions — the 64-bit ones were 3 of the 4 us this setup took)
    const int rw_ = (int)(t % (unsigned)a.RW); t /= (unsigned)a.RW;
    const int rh_ = (int)(t % (unsigned)a.RH); t /= (unsigned)a.RH;
    const int rd_ = (int)(t % (unsigned)a.RD); t /= (unsigned)a.RD;
    rd0[i] = rd_ * a.rs[0] + a.tb[0]; rh0[i] = rh_ * a.rs[1] + a.tb[1]; rw0[i] = rw_ * a.rs[2] + a.tb[2];
    // tap offsets are added later and may be negative: valid rows stay inside [0, 2^30), parked rows stay out of range
    va[i] = v ? (unsigned)(((long)t * a.xsn + rd0[i] * a.xsd + rh0[i] * a.xsh + rw0[i] * a.xsw + dpiece * 8) * 2) : G5_OOB;
    if (!v) { rd0[i] = rh0[i] = rw0[i] = -(1 << 20); }
  }
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    const int r = i * 64 + drow, n = n0 + r;
    vb[i] = (r < G5_BN && n < a.wrows) ? (unsigned)(((long)n * (a.wld ? a.wld : (long)ntaps * a.cink) + dpiece * 8) * 2) : G5_OOB;
  }
  const unsigned dma_lds = lds0 + wave * 1024;                   // + stage * G5_STAGE + (A: i * 8192 | B: 32768 + i * 8192)

  // ---- producer odometers: A runs two K-tiles ahead of the MFMAs, B one
  int akd, akh, akw, achunk, btap, bchunk;
  { const int c0 = ku0 / ntaps, t0 = ku0 - c0 * ntaps; achunk = bchunk = c0; btap = t0; akw = t0 % a.TW; akh = (t0 / a.TW) % a.TH; akd = t0 / (a.TW * a.TH); }
  int a_off = 0, a_od = 0, a_oh = 0, a_ow = 0; bool a_cok = true;      // the tile whose A pieces are being issued
  auto prep_a = [&]() {          // offsets of the next A tile; advances the odometer
    a_od = akd * a.ts[0]; a_oh = akh * a.ts[1]; a_ow = akw * a.ts[2];
    a_off = ((int)(a_od * a.xsd + a_oh * a.xsh + a_ow * a.xsw) + achunk * 64) * 2;
    a_cok = achunk * 64 + dpiece * 8 < a.cin_lanes;
    ++akw;
    const int cw = akw == a.TW; akw = cw ? 0 : akw; akh += cw;
    const int ch = akh == a.TH; akh = ch ? 0 : akh; akd += ch;
    const int cd = akd == a.TD; akd = cd ? 0 : akd; achunk += cd;
  };
  auto dma_a = [&](int stage, auto I) {
    constexpr int i = decltype(I)::value;
    bool ok = a_cok;
    if (CHECK) ok = ok & ((unsigned)(rd0[i] + a_od) < (unsigned)a.ID) & ((unsigned)(rh0[i] + a_oh) < (unsigned)a.IH) & ((unsigned)(rw0[i] + a_ow) < (unsigned)a.IW);
    g5_dma(__builtin_amdgcn_readfirstlane(dma_lds + stage * G5_STAGE + i * 8192), ra, ok ? va[i] + (unsigned)a_off : G5_OOB, 0);
  };
  int b_soff = 0;
  auto prep_b = [&]() { b_soff = (btap * a.cink + bchunk * 64) * 2; ++btap; const int c = btap == ntaps; btap = c ? 0 : btap; bchunk += c; };
  auto dma_b = [&](int stage, auto I) {
    constexpr int i = decltype(I)::value;
    g5_dma(__builtin_amdgcn_readfirstlane(dma_lds + stage * G5_STAGE + G5_BM * 128 + i * 8192), rw, vb[i], b_soff);
  };
  using C0 = std::integral_constant<int, 0>; using C1 = std::integral_constant<int, 1>; using C2 = std::integral_constant<int, 2>;
  using C3 = std::integral_constant<int, 3>; using C4 = std::integral_constant<int, 4>;

  // ---- fragment addresses: lane (fr, fg) reads piece h*4 + fg of row base + fr; slot = piece ^ (fr >> 1); k-half 1 = address ^ 64
  const int fr = lane & 15, fg = lane >> 4;
  const unsigned flane = fr * 128 + ((fg ^ (fr >> 1)) << 4);
  const unsigned fa0 = lds0 + wm * (64 * 128) + flane;                        // A fragments of this wave, stage 0, k-half 0
  const unsigned fb0 = lds0 + G5_BM * 128 + wn * (FN * 16 * 128) + flane;     // B fragments

  f32x4 acc[4][FN];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // ---- prologue: A(0), B(0) -> stage 0, A(1) -> stage 1 (stays in flight)
  const bool two = nk > 1;
  prep_a(); dma_a(0, C0{}); dma_a(0, C1{}); dma_a(0, C2{}); dma_a(0, C3{});
  prep_b(); dma_b(0, C0{}); dma_b(0, C1{}); dma_b(0, C2{}); dma_b(0, C3{}); if constexpr (NB > 4) dma_b(0, C4{});
  if (two) {
    prep_a(); dma_a(1, C0{}); dma_a(1, C1{}); dma_a(1, C2{}); dma_a(1, C3{});
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __builtin_amdgcn_s_barrier();
  if (wn) __builtin_amdgcn_s_barrier();      // the trailing group runs one barrier behind (matched after the loop)

  // Phase order within a K-tile: (columns 0-4, k-half 0) (columns 0-4, k-half 1) (columns 5-8, k-half 0) (columns 5-8, k-half 1).
  // The A fragments of BOTH k-halves are in registers after phase 1, so the A part of this stage is free two phases before the B
  // part: the DMA of A(t+2) is issued in phases 2-3 of tile t (a whole tile ahead of its first read), the DMA of B(t+1) in phases
  // 0-1 — 3 / 2 / 2 / 2 DMA instructions per phase instead of 5 / 4 / 0 / 0 (an LDS-DMA instruction costs its wave 60-180
  // cycles of issue; the read segment must stay shorter than the other group's MFMA cluster: 926 -> 1022 TFLOP/s on the Winograd
  // GEMMs).  One counted wait per tile, in phase 3: vmcnt(4) = everything but the 4 youngest pieces (A(t+2)) has landed, i.e.
  // A(t+1) and B(t+1).
  bf16x8 af[8], bf[5];
  for (int t = 0; t < nk; ++t) {
    const int stage = t & 1;
    const bool more1 = t + 1 < nk, more2 = t + 2 < nk;
    const unsigned sa = fa0 + stage * G5_STAGE, sb = fb0 + stage * G5_STAGE;
    g5for<4>([&](auto P) {
      constexpr int p = decltype(P)::value, h = p & 1, jg = p >> 1;
      constexpr int J0 = jg ? (FN + 1) / 2 : 0, NJ = jg ? FN / 2 : (FN + 1) / 2;
      // ---- read segment
      __builtin_amdgcn_sched_barrier(0);
      const unsigned aa = sa ^ (h * 64), bb = sb ^ (h * 64);
      if constexpr (jg == 0) g5for<4>([&](auto I) { g5_ldr<decltype(I)::value * 2048>(af[h * 4 + decltype(I)::value], aa); });
      g5for<NJ>([&](auto J) { g5_ldr<(J0 + decltype(J)::value) * 2048>(bf[decltype(J)::value], bb); });
      if constexpr (p == 0) { if (more1) { prep_b(); dma_b(stage ^ 1, C0{}); dma_b(stage ^ 1, C1{}); dma_b(stage ^ 1, C2{}); } }
      if constexpr (p == 1) { if (more1) { dma_b(stage ^ 1, C3{}); if constexpr (NB > 4) dma_b(stage ^ 1, C4{}); } }
      if constexpr (p == 2) { if (more2) { prep_a(); dma_a(stage, C0{}); dma_a(stage, C1{}); } }
      if constexpr (p == 3) {
        if (more2) { dma_a(stage, C2{}); dma_a(stage, C3{}); asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      // every fragment read of this phase has returned (names the registers: the MFMAs below depend on this statement)
      if constexpr (jg == 0 && NJ == 5)
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(af[h * 4]), "+v"(af[h * 4 + 1]), "+v"(af[h * 4 + 2]), "+v"(af[h * 4 + 3]), "+v"(bf[0]), "+v"(bf[1]), "+v"(bf[2]), "+v"(bf[3]), "+v"(bf[4]) :: "memory");
      else if constexpr (jg == 0)
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(af[h * 4]), "+v"(af[h * 4 + 1]), "+v"(af[h * 4 + 2]), "+v"(af[h * 4 + 3]), "+v"(bf[0]), "+v"(bf[1]), "+v"(bf[2]), "+v"(bf[3]) :: "memory");
      else
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bf[0]), "+v"(bf[1]), "+v"(bf[2]), "+v"(bf[3]) :: "memory");
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      // ---- MFMA cluster: column fragments J0 .. J0 + NJ - 1 x the 4 row fragments, k-half h
      __builtin_amdgcn_s_setprio(1);
      g5for<NJ>([&](auto J) {
        constexpr int j = decltype(J)::value;
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i][J0 + j] = RG_MFMA16(bf[j], af[h * 4 + i], acc[i][J0 + j]);
      });
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
    });
  }
  if (!wn) __builtin_amdgcn_s_barrier();     // matches the trailing group's extra barrier

  // ---- epilogue: lane holds columns c .. c+3 of row (lane & 15) of every fragment
  // (SK: the lane indices pass through an opaque move first, so that nothing of the epilogue's address arithmetic is loop-invariant — hoisted in front
  // of the run loop it would stay live across the K loop and spill)
  int efr = fr, efg = fg, etid = tid;
  if constexpr (SK) asm volatile("" : "+v"(efr), "+v"(efg), "+v"(etid));
  if constexpr (RAW) {  // raw fp32 tile straight into the caller's matrix (the weight gradient: row = output channel, column = input channel of one tap)
    float* const obase = a.raw + batch * a.raw_batch;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const long m = m0 + wm * 64 + i * 16 + efr;
      if (m >= a.M) continue;
      float* prow = obase + m * a.raw_ld;
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        const int c = n0 + wn * (FN * 16) + j * 16 + efg * 4;
        if (c + 3 < a.cout) *(f32x4*)(prow + c) = acc[i][j];
        else {
#pragma unroll
          for (int r = 0; r < 4; ++r) if (c + r < a.cout) prow[c + r] = acc[i][j][r];
        }
      }
    }
    return;
  }
  if (!SK && a.ksplit > 1) {   // split-K: raw fp32 sums; bias, activation and the 16-bit store happen in the combine kernel
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const long m = m0 + wm * 64 + i * 16 + efr;
      if (m >= a.M) continue;
      float* prow = a.part + ((long)split * a.M + m) * a.cout_lanes;
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        const int c = n0 + wn * (FN * 16) + j * 16 + efg * 4;
        if (c < a.cout_lanes) *(f32x4*)(prow + c) = acc[i][j];
      }
    }
    return;
  }
  bool finish = true;
  if constexpr (SK) {
    // Partials and flags move with device-scope (sc1) accesses: written through to / read from the memory side of the per-XCD L2s, so no cache-wide
    // write-back or invalidate is needed (the agent-scope fences that would make ordinary stores visible cost 35-50 us per launch: they flush and
    // drop the whole L2 under the workgroups still computing)
    constexpr long SLOT = 4L * FN * 512 * 4;   // floats per workgroup: accumulator (i, j) of thread t at ((i * FN + j) * 512 + t) * 4
    if (ku1 != kunits) {
      // the piece ends inside its tile: leave the sums for the workgroup that holds the tile's last K-tile
      float* slot = a.sk_part + (long)wg * SLOT + etid * 4;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(slot + (i * FN + j) * 2048), "v"(acc[i][j]) : "memory");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // my stores have been acknowledged
      __syncthreads();                                     // ... and everybody's
      if (etid == 0) __hip_atomic_store(a.sk_flags + wg, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      finish = false;
    } else if (ku0 != 0) {
      // the tail of a tile others started: every workgroup below this one whose run reaches into the tile left ONE partial there, the piece it
      // computes FIRST — before anything it might wait for itself, so this wait cannot deadlock whatever the dispatch order
      const long tile_start = sk_tile * kunits;
      long cov = sk_su;
      for (int w2 = wg - 1; cov > tile_start && w2 >= 0; --w2) {
        const long s2 = sk_units * w2 / nwg;
        if (s2 == cov) continue;                           // (an empty run)
        if (etid == 0) {
          while (__hip_atomic_load(a.sk_flags + w2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) __builtin_amdgcn_s_sleep(4);
          // every partial has exactly ONE reader: it lowers the flag again, so a launch leaves its row as it found it — zeros — and the same launch
          // can be replayed from a captured graph (an epoch in the kernel arguments would be frozen into the graph)
          __hip_atomic_store(a.sk_flags + w2, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        const float* slot = a.sk_part + (long)w2 * SLOT + etid * 4;
        // one accumulator row (FN loads in flight) at a time: the fragment registers are free here, more would spill
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          f32x4 v[FN];
#pragma unroll
          for (int j = 0; j < FN; ++j) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v[j]) : "v"(slot + (i * FN + j) * 2048) : "memory");
          if constexpr (FN == 9) asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]) :: "memory");
          else asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]) :: "memory");
#pragma unroll
          for (int j = 0; j < FN; ++j) { acc[i][j][0] += v[j][0]; acc[i][j][1] += v[j][1]; acc[i][j][2] += v[j][2]; acc[i][j][3] += v[j][3]; }
        }
        cov = s2;
      }
    }
  }
  if (finish) {
    // The 16-bit tile goes through LDS (free once every wave has left the K loop): each lane converts its 4 x FN accumulators (bias, activation,
    // rounding) into row-major [256][BN] with a 16-byte row pad (ds_write_b64, banks evenly loaded), thread r < 256 decodes output row r ONCE into a
    // table, and the tile leaves as 16-byte pieces — consecutive lanes = consecutive channels of one voxel, whole 128-byte lines — instead of 8-byte
    // pieces 16 rows apart with a row decode (three 64-bit divisions) per accumulator row.  Per tile: 17 us (conversion) + 3 us (stores) became 2.5 + 3 us.
    constexpr int EP = G5_BN * 2 + 16;                      // bytes per LDS row
    unsigned char* const etile = g5_lds;
    long* const erow = (long*)(g5_lds + 256 * EP);          // element offset of the row's voxel in y, -1: past M
    bf16_t* const ybase = a.y + batch * a.ybatch;
    if (etid < 256) {
      const long m = m0 + etid;
      long off = -1;
      if (m < a.M) {
        unsigned t = (unsigned)m;
        const int rw_ = (int)(t % (unsigned)a.RW); t /= (unsigned)a.RW;
        const int rh_ = (int)(t % (unsigned)a.RH); t /= (unsigned)a.RH;
        const int rd_ = (int)(t % (unsigned)a.RD); t /= (unsigned)a.RD;
        off = (long)t * a.ysn + (long)(rd_ * a.os[0] + a.oo[0]) * a.ysd + (long)(rh_ * a.os[1] + a.oo[1]) * a.ysh + (long)(rw_ * a.os[2] + a.oo[2]) * a.ysw;
      }
      erow[etid] = off;
    }
    const bool bias_v4 = a.bias && (((unsigned long)a.bias & 15) == 0);
    // two instantiations of the conversion: the piecewise-linear activations as straight-line code (none / ReLU / LeakyReLU = one multiply and one
    // select), the transcendental ones apart — one generic act_fwd per element under the per-lane `channel < cout` test expanded into 13 000
    // predicated instructions and took 17 of the 33 us a tile cost beyond its K loop
    const float nslope = a.act == RG_ACT_RELU ? 0.f : (a.act == RG_ACT_LEAKY ? a.slope : 1.f);
    auto convert = [&](auto TR) {
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        const int cl = wn * (FN * 16) + j * 16 + efg * 4;     // column within the tile
        const int c = n0 + cl;
        int cc = c;
        if (a.d2s_c) cc = c % a.d2s_c;                        // bias per channel of the parity block
        float bv[4] = {0.f, 0.f, 0.f, 0.f};
        if (a.bias) {
          if (bias_v4 && c + 3 < a.cout) { const f32x4 b4 = *(const f32x4*)(a.bias + cc); bv[0] = b4[0]; bv[1] = b4[1]; bv[2] = b4[2]; bv[3] = b4[3]; }
          else {
#pragma unroll
            for (int r = 0; r < 4; ++r) if (c + r < a.cout) bv[r] = a.bias[cc + r];
          }
        }
        const int live = a.cout - c;                          // channels c .. c + live - 1 exist; the pad lanes behind them are written as zeros
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float v[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float x = acc[i][j][r] + bv[r];
            if constexpr (decltype(TR)::value) x = act_fwd(x, a.act, a.slope);
            else x = x < 0.f ? x * nslope : x;                // (hands a NaN on, like act_fwd)
            v[r] = r < live ? x : 0.f;
          }
          uint2 o; o.x = f2bf_pk(v[0], v[1]); o.y = f2bf_pk(v[2], v[3]);
          *(uint2*)(etile + (wm * 64 + i * 16 + efr) * EP + cl * 2) = o;
        }
      }
    };
    if (a.act <= RG_ACT_LEAKY) convert(std::false_type{}); else convert(std::true_type{});
    __syncthreads();
    constexpr int PPR = G5_BN / 8;                          // 16-byte pieces per row
#pragma unroll 2
    for (int pidx = etid; pidx < 256 * PPR; pidx += 512) {
      const int row = pidx / PPR, q8 = pidx - row * PPR;
      const int c = n0 + q8 * 8;
      const long off = erow[row];
      if (off < 0 || c >= a.cout_lanes) continue;
      int cc = c; long yo = 0;
      if (a.d2s_c) {   // depth-to-space: column block q = c / C is parity (qd, qh, qw) of output voxel 2 * row + parity (a piece never straddles blocks: C % 16 == 0)
        const int q = c / a.d2s_c; cc = c - q * a.d2s_c;
        yo = (long)((q >> 2) & 1) * a.ysd + (long)((q >> 1) & 1) * a.ysh + (long)(q & 1) * a.ysw;
      }
      const u32x4_t v = *(const u32x4_t*)(etile + row * EP + q8 * 16);
      bf16_t* dst = ybase + off + yo + cc;
      if (c + 8 <= a.cout_lanes && (((unsigned long)dst & 15) == 0)) *(u32x4_t*)dst = v;
      else {   // a view whose voxel pitch or channel offset is not a multiple of 8 lanes, or the last lanes of a row: 8-byte halves
        *(unsigned long*)dst = ((unsigned long)v[1] << 32) | v[0];
        if (c + 4 < a.cout_lanes) *(unsigned long*)(dst + 4) = ((unsigned long)v[3] << 32) | v[2];
      }
    }
    if constexpr (SK) __syncthreads();                      // the next piece's DMA overwrites this LDS
  }
  if constexpr (!SK) return;
  if (head_first) head_first = false; else u += nk;
  if (u >= sk_stop) return;
  }   // next piece of this workgroup's run (SK)
}

// Which igemm5 configuration serves this problem (0: none): FN = 9 for column counts in 144-column halves, FN = 8 for 128-column halves;
// the tile must be (nearly) full, the grid must give most CUs a workgroup (ONE 8-wave workgroup per CU) — small grids with a long K are
// cut into up to 8 K-slices whose fp32 partial tiles the caller's workspace holds ([slice][row][column], summed by igemm4's combine
// kernel) — and K must be long enough to amortise the two-tile prologue.  opt = rg_opt(RG_OPT_GEMM5): 0 heuristic, 1 wherever the
// tiling allows, 2 also the 432-column layers (one of whose two column tiles is half empty), -1 off.
struct G5Plan { int fn; int ksplit; int sk; };
// Stream-K (sk = workgroups, one per CU): the grid's tiles x K-tiles are cut into 256 equal contiguous runs instead of whole tiles, for grids whose last
// round of tiles would leave much of the chip idle (CycleGAN: 256 -> 256 at 32^3 is 128 tiles forward, 154 backward — half a round and 0.6 of one).  A
// run that starts inside a tile parks its fp32 partial in the caller's workspace (one slot per workgroup, accumulator-register order: fully coalesced)
// and the workgroup that started that tile adds the 1-2 partials before the ordinary epilogue: no combine pass, ~0.7 partials per tile instead of
// `ksplit` whole copies.  rg_opt(RG_OPT_SK): 0 heuristic, 1 wherever the tiling allows, -1 off.
#define G5_SK_GRID 256
#define G5_SK_ROWS 64       // flag rows: consecutive launches take consecutive rows, so launches in flight on different streams do not share one
                            // (a row comes round again 64 stream-K launches later; launches of one stream are ordered anyway)
static size_t g5_sk_bytes(int fn) { return (size_t)G5_SK_GRID * 4 * fn * 512 * 16; }
static G5Plan g5_plan(long M, int wrows, int kunits, int nbatch, bool flat, size_t ws_bytes, int cout_lanes, bool sk_ok = true) {
  G5Plan pl = {0, 1, 0};
  const int opt = rg_opt(RG_OPT_GEMM5);
  if (opt < 0 || flat || (wrows % 16) || cout_lanes < 8 || M >= (1L << 31)) return pl;   // (the kernel decodes rows with 32-bit divisions)
  const int nf = wrows / 16;
  const int fn = nf % 9 == 0 ? 9 : (nf % 8 == 0 ? 8 : 0);
  if (!fn) return pl;
  const int bn = 2 * fn * 16;
  const long tiles_m = (M + G5_BM - 1) / G5_BM, tiles_n = (wrows + bn - 1) / bn;
  const long tiles = tiles_m * tiles_n * nbatch;
  const int sko = rg_opt(RG_OPT_SK);
  const bool sk_room = sk_ok && sko >= 0 && ws_bytes >= g5_sk_bytes(fn) && tiles * kunits >= G5_SK_GRID && tiles < (1L << 20);
  if (opt == 1) { pl.fn = fn; if (sk_room && sko >= 1) pl.sk = G5_SK_GRID; return pl; }
  const double fill_n = (double)wrows / (double)(tiles_n * bn);
  // (batched GEMMs — the Winograd products — take a 75 %-full column tiling too: 64 x [1331 x 2592] x [2592 x 432] measured 0.295 ms here against 0.328 on igemm4)
  if (fill_n < ((opt == 2 || nbatch > 1) ? 0.70 : 0.88) || kunits < 8) return pl;
  auto round_fill = [](long t) { const double r = (double)t / 256.0; return r / (double)(long)(r + 0.999999); };
  // runs of >= 16 K-tiles (a run pays a two-tile prologue) and tiles long enough that a partial is small next to the work it saves
  const bool sk_can = sk_room && (sko >= 1 || (kunits >= 16 && tiles * kunits >= (long)G5_SK_GRID * 16));
  // Measured on 256 -> 256 at 32^3 (MI355X, us): backward, 154 tiles — whole tiles 182, igemm4 (192-row tiles) 200, stream-K 150; forward, 128 tiles —
  // whole tiles 168, 2 K-slices + combine 121 -> 113 and stream-K 137 -> 106 once the epilogues were straight-line code (gemm5.hip's LDS-staged tile): with
  // exactly two pieces per tile the hand-over is the combine pass run by half the chip, but it saves the combine launch and the slices' second fp32 copy.
  // So: stream-K for every grid that is not (nearly) whole rounds; K-slices only for callers whose workspace holds nothing else.
  if (tiles >= 150 && round_fill(tiles) >= 0.70) {
    pl.fn = fn;
    if (sk_can && (sko >= 1 || (round_fill(tiles) < 0.75 && tiles <= 3 * 256))) pl.sk = G5_SK_GRID;
    return pl;
  }
  if (sk_can) { pl.fn = fn; pl.sk = G5_SK_GRID; return pl; }
  if (nbatch == 1 && tiles < 150 && kunits >= 32 && (cout_lanes & 3) == 0) {   // split K: whole rounds of one workgroup per CU
    long ks = 256 / tiles; if (ks > 8) ks = 8; if (ks > kunits / 16) ks = kunits / 16;
    if (ks >= 2 && tiles * ks >= 150 && (size_t)ks * M * cout_lanes * 4 <= ws_bytes) { pl.fn = fn; pl.ksplit = (int)ks; }
  }
  return pl;
}
bool rg_igemm5_ok(const Igemm4Args& a, bool flat, size_t ws_bytes) {
  return g5_plan(a.M, a.wrows, (a.cink >> 6) * a.TD * a.TH * a.TW, a.nbatch, flat, ws_bytes, a.cout_lanes).fn != 0;
}
// the same decision from the tensor views (conv.hip routes 128-channel-tile layers to igemm3 unless igemm5 takes them)
bool rg_igemm5_wanted(const CL5& X, const CL5& Y, const rg_conv_params* p, size_t ws_bytes) {
  const long M = (long)X.N * p->R[0] * p->R[1] * p->R[2];
  return g5_plan(M, round_up(p->cout, 16), (round_up(X.C, 64) >> 6) * p->T[0] * p->T[1] * p->T[2], 1, false, ws_bytes, Y.lanes).fn != 0;
}
// workspace a convolution needs for the stream-K schedule (0: that problem would not run stream-K even with the room)
size_t rg_igemm5_sk_bytes(const CL5& X, const CL5& Y, const rg_conv_params* p) {
  const long M = (long)X.N * p->R[0] * p->R[1] * p->R[2];
  const G5Plan pl = g5_plan(M, round_up(p->cout, 16), (round_up(X.C, 64) >> 6) * p->T[0] * p->T[1] * p->T[2], 1, false, (size_t)1 << 40, Y.lanes);
  return pl.sk ? g5_sk_bytes(pl.fn) : 0;
}
void rg_launch_igemm4_combine(const Igemm4Args& a, hipStream_t st);   // conv4.hip: y = act(sum of the K-slices' partial tiles + bias)

// per-device flag rows of the stream-K schedule, zeroed once; every launch lowers the flags it raised (the reader of a partial resets its flag)
static int* g5_sk_flag_rows() {
  static std::mutex mu; static int* cache[64];
  int dev = 0; if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
  std::lock_guard<std::mutex> lk(mu);
  if (!cache[dev]) {
    void* p = nullptr;
    if (hipMalloc(&p, (size_t)G5_SK_ROWS * G5_SK_GRID * sizeof(int)) != hipSuccess) return nullptr;
    if (hipMemset(p, 0, (size_t)G5_SK_ROWS * G5_SK_GRID * sizeof(int)) != hipSuccess || hipDeviceSynchronize() != hipSuccess) { hipFree(p); return nullptr; }
    cache[dev] = (int*)p;
  }
  return cache[dev];
}

int rg_launch_igemm5(Igemm4Args& a, bool check, hipStream_t st, float* ws, size_t ws_bytes) {
  const G5Plan pl = g5_plan(a.M, a.wrows, (a.cink >> 6) * a.TD * a.TH * a.TW, a.nbatch, false, ws_bytes, a.cout_lanes);
  const int bn = 2 * pl.fn * 16;
  a.tiles_m = (int)((a.M + G5_BM - 1) / G5_BM);
  a.tiles_n = (a.wrows + bn - 1) / bn;
  a.ksplit = pl.ksplit; a.part = pl.ksplit > 1 ? ws : nullptr;
  static rg_device_latch attr;
  rg_once_per_device(&attr, [] {
    hipFuncSetAttribute((const void*)igemm5_kernel<false, 9>, hipFuncAttributeMaxDynamicSharedMemorySize, G5_LDS_BYTES(9));
    hipFuncSetAttribute((const void*)igemm5_kernel<true, 9>, hipFuncAttributeMaxDynamicSharedMemorySize, G5_LDS_BYTES(9));
    hipFuncSetAttribute((const void*)igemm5_kernel<false, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, G5_LDS_BYTES(8));
    hipFuncSetAttribute((const void*)igemm5_kernel<true, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, G5_LDS_BYTES(8));
    hipFuncSetAttribute((const void*)igemm5_kernel<false, 9, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, G5_LDS_BYTES(9));
    hipFuncSetAttribute((const void*)igemm5_kernel<true, 9, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, G5_LDS_BYTES(9));
    hipFuncSetAttribute((const void*)igemm5_kernel<false, 8, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, G5_LDS_BYTES(8));
    hipFuncSetAttribute((const void*)igemm5_kernel<true, 8, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, G5_LDS_BYTES(8));
  });
  if (a.d2s_c && pl.ksplit > 1) { rg_set_error("igemm5: depth-to-space epilogue with split-K"); return RG_ERR_UNSUPPORTED; }
  if (pl.sk) {
    static std::atomic<unsigned> launches{0};
    int* rows = g5_sk_flag_rows();
    if (!rows) { rg_set_error("igemm5: cannot allocate the stream-K flags"); return RG_ERR_HIP; }
    const unsigned e = launches.fetch_add(1, std::memory_order_relaxed);
    a.sk_grid = pl.sk; a.sk_part = ws; a.sk_epoch = 0; a.sk_flags = rows + (size_t)(e % G5_SK_ROWS) * G5_SK_GRID;
    rg_set_kernel_name("igemm5_kernel<%d,%d>+streamK%s", (int)check, pl.fn, a.d2s_c ? "+d2s" : "");
    const dim3 grid((unsigned)pl.sk);
    if (pl.fn == 9) {
      if (check) hipLaunchKernelGGL((igemm5_kernel<true, 9, false, true>), grid, dim3(512), G5_LDS_BYTES(9), st, a);
      else hipLaunchKernelGGL((igemm5_kernel<false, 9, false, true>), grid, dim3(512), G5_LDS_BYTES(9), st, a);
    } else {
      if (check) hipLaunchKernelGGL((igemm5_kernel<true, 8, false, true>), grid, dim3(512), G5_LDS_BYTES(8), st, a);
      else hipLaunchKernelGGL((igemm5_kernel<false, 8, false, true>), grid, dim3(512), G5_LDS_BYTES(8), st, a);
    }
    return RG_OK;
  }
  rg_set_kernel_name("igemm5_kernel<%d,%d>%s", (int)check, pl.fn, pl.ksplit > 1 ? "+splitK" : (a.d2s_c ? "+d2s" : ""));
  const dim3 grid((unsigned)(a.tiles_m * a.tiles_n * a.ksplit * a.nbatch));
  if (pl.fn == 9) {
    if (check) hipLaunchKernelGGL((igemm5_kernel<true, 9>), grid, dim3(512), G5_LDS_BYTES(9), st, a);
    else hipLaunchKernelGGL((igemm5_kernel<false, 9>), grid, dim3(512), G5_LDS_BYTES(9), st, a);
  } else {
    if (check) hipLaunchKernelGGL((igemm5_kernel<true, 8>), grid, dim3(512), G5_LDS_BYTES(8), st, a);
    else hipLaunchKernelGGL((igemm5_kernel<false, 8>), grid, dim3(512), G5_LDS_BYTES(8), st, a);
  }
  if (a.ksplit > 1) rg_launch_igemm4_combine(a, st);
  return RG_OK;
}

// ---- the weight gradient as a batched GEMM --------------------------------------------------------------------------------------
// dW[a][tap][b] = sum_rows G[row][a] X[row @ tap][b] reduces over voxels — the SLOW dimension of both channels-last operands, which is why
// wgrad_kw* (wgrad.hip) read every fragment with the transposing ds_read_b64_tr_b16: 24 of them per 27 MFMAs, and the kernel sits at ~850 TFLOP/s
// (34 % of peak, MFMA pipe 39 % busy) whichever way its tiles reach LDS (round 3: producer waves and LDS-DMA measured level).  For the layers
// whose operands are SMALL next to their FLOPs (config U's 2592 -> 2592 at 17^3: 2 x 36 MB for 1.8 TFLOP) it pays to transpose the operands
// once, in HBM, and run the reduction as what it then is — 27 plain GEMMs with K-contiguous rows on igemm5 (1.2 PFLOP/s on this shape class):
//   GT[a][od][P]                  = G[od][oh][ow][a]                p = oh * OW + ow, plane padded to P = 64-element lines with ZEROS
//   XT[(kh, kw)][b][d][P]         = X[d][oh + kh][ow + kw][b]       one copy per in-plane tap: the kd taps are a K offset of kd * P
//   dW[a][(kd, kh, kw)][b]        = sum_k GT[a][k] * XT[(kh, kw)][b][kd * P + k]        k = od * P + p
// (wgrad_transpose_kernel below writes both; igemm5 runs batch (b1, b2) = ((kh, kw), kd) with B offset b1 * copy + b2 * P, B rows ID * P apart, raw fp32
// output at column offset tap * B: the channels-last gradient itself, no split-K, no reduction pass).
struct WgradTrArgs { const bf16_t* src; bf16_t* dst; int C, lanes, D, OH, OW, P, KH, KW; long ssd, ssh, ssw; long copy_stride, row_stride; };
__global__ __launch_bounds__(256) void wgrad_transpose_kernel(WgradTrArgs t) {
  // block = (64 positions of one plane d) x (64 channels) of copy j: read channel-fastest 16-byte pieces, write position-fastest 16-byte pieces
  __shared__ bf16_t tile[64][72];
  const int j = blockIdx.z % (t.KH * t.KW), d = blockIdx.z / (t.KH * t.KW);
  const int kh = j / t.KW, kw = j % t.KW;
  const int p0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
  const int tid = threadIdx.x;
#pragma unroll
  for (int q = tid; q < 64 * 8; q += 256) {          // 64 positions x 8 pieces of 8 channels
    const int pp = q >> 3, pc = (q & 7) * 8, pos = p0 + pp, c = c0 + pc;
    u32x4_t v = {0u, 0u, 0u, 0u};
    if (pos < t.OH * t.OW && c < t.lanes) {
      const int oh = pos / t.OW, ow = pos - oh * t.OW;
      v = *(const u32x4_t*)(t.src + (long)d * t.ssd + (long)(oh + kh) * t.ssh + (long)(ow + kw) * t.ssw + c);
    }
    *(u32x4_t*)&tile[pp][pc] = v;
  }
  __syncthreads();
#pragma unroll
  for (int q = tid; q < 64 * 8; q += 256) {          // 64 channels x 8 pieces of 8 positions
    const int cc = q >> 3, pq = (q & 7) * 8, c = c0 + cc;
    if (c >= t.C) continue;
    bf16_t o[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) o[k] = tile[pq + k][cc];
    *(u32x4_t*)(t.dst + (long)j * t.copy_stride + (long)c * t.row_stride + (long)d * t.P + p0 + pq) = *(const u32x4_t*)o;
  }
}

struct WgradGemmPlan { int ok; int P, OD, planes, K, fn; size_t gt_el, xt_el; };
static WgradGemmPlan wgrad_gemm_plan(const CL5& G, const CL5& X, const rg_conv_params* p) {
  WgradGemmPlan pl; memset(&pl, 0, sizeof(pl));
  if (rg_opt(RG_OPT_GEMM5) < 0 || G.N != 1) return pl;
  for (int d = 0; d < 3; ++d)
    if (p->rstride[d] != 1 || p->tstep[d] != 1 || p->tbase[d] < 0) return pl;
  const int I[3] = {X.D, X.H, X.W};
  for (int d = 0; d < 3; ++d) if (p->tbase[d] + p->T[d] - 1 + p->R[d] - 1 >= I[d]) return pl;     // valid region only: no zero padding
  const int nf = round_up(X.C, 16) / 16;
  const int fn = nf % 9 == 0 ? 9 : (nf % 8 == 0 ? 8 : 0);
  if (!fn || (X.C & 7) || G.C < 1024 || X.C < 1024) return pl;       // the operands must be small next to the FLOPs: wide layers only
  const int P = round_up(p->R[1] * p->R[2], 64), K = p->R[0] * P;
  const long tiles_m = (G.C + G5_BM - 1) / G5_BM, tiles_n = (round_up(X.C, 16) + 2 * fn * 16 - 1) / (2 * fn * 16);
  const double fill = (double)G.C / (tiles_m * G5_BM) * (double)X.C / (tiles_n * 2 * fn * 16) * (double)(p->R[1] * p->R[2]) / P;
  const long wgs = tiles_m * tiles_n * p->T[0] * p->T[1] * p->T[2];
  const double rounds = (double)wgs / 256.0, round_fill = rounds / (double)(long)(rounds + 0.999999);   // ONE workgroup per CU
  if (fill < 0.70 || K < 32 * 64 || wgs < 256 || round_fill < 0.70) return pl;
  const int planes = p->R[0] + p->T[0] - 1;
  pl.P = P; pl.OD = p->R[0]; pl.planes = planes; pl.K = K; pl.fn = fn;
  pl.gt_el = (size_t)G.C * K;
  pl.xt_el = (size_t)p->T[1] * p->T[2] * round_up(X.C, 16) * planes * P;      // (rows up to the 16-row fragment: igemm5 reads them, their columns are never stored)
  if (pl.gt_el * 2 >= (1ul << 30) || pl.xt_el * 2 >= (1ul << 31)) return pl;   // 32-bit buffer offsets
  pl.ok = 1;
  return pl;
}
// workspace bytes of the GEMM form, 0 when the problem does not take it
size_t rg_wgrad_gemm_bytes(const CL5& G, const CL5& X, const rg_conv_params* p) {
  const WgradGemmPlan pl = wgrad_gemm_plan(G, X, p);
  return pl.ok ? (pl.gt_el + pl.xt_el) * 2 + 256 : 0;
}
int rg_launch_wgrad_gemm(const CL5& G, const CL5& X, const rg_conv_params* p, float* dw, void* workspace, size_t ws_bytes, hipStream_t st) {
  const WgradGemmPlan pl = wgrad_gemm_plan(G, X, p);
  if (!pl.ok) return RG_ERR_UNSUPPORTED;
  if (ws_bytes < (pl.gt_el + pl.xt_el) * 2 + 256) RG_FAIL(RG_ERR_WORKSPACE, "wgrad (GEMM form): workspace %zu bytes too small", ws_bytes);
  bf16_t* gt = (bf16_t*)(((uintptr_t)workspace + 127) & ~(uintptr_t)127);
  bf16_t* xt = gt + pl.gt_el;
  const int KH = p->T[1], KW = p->T[2], KD = p->T[0];
  WgradTrArgs tg; tg.src = G.p; tg.dst = gt; tg.C = G.C; tg.lanes = G.lanes; tg.D = pl.OD; tg.OH = p->R[1]; tg.OW = p->R[2]; tg.P = pl.P; tg.KH = 1; tg.KW = 1;
  tg.ssd = G.sd; tg.ssh = G.sh; tg.ssw = G.sw; tg.copy_stride = 0; tg.row_stride = pl.K;
  rg_prof_mark(st);      // (the two transposing passes are part of this path's cost: inside the bracket)
  hipLaunchKernelGGL(wgrad_transpose_kernel, dim3(pl.P / 64, (G.C + 63) / 64, pl.OD), dim3(256), 0, st, tg);
  WgradTrArgs tx; tx.src = X.p + (long)p->tbase[0] * X.sd + (long)p->tbase[1] * X.sh + (long)p->tbase[2] * X.sw;
  tx.dst = xt; tx.C = X.C; tx.lanes = X.lanes; tx.D = pl.planes; tx.OH = p->R[1]; tx.OW = p->R[2]; tx.P = pl.P; tx.KH = KH; tx.KW = KW;
  tx.ssd = X.sd; tx.ssh = X.sh; tx.ssw = X.sw; tx.row_stride = (long)pl.planes * pl.P; tx.copy_stride = (long)round_up(X.C, 16) * tx.row_stride;
  hipLaunchKernelGGL(wgrad_transpose_kernel, dim3(pl.P / 64, (X.C + 63) / 64, pl.planes * KH * KW), dim3(256), 0, st, tx);
  Igemm4Args a; memset(&a, 0, sizeof(a));
  a.x = gt; a.w = xt; a.bias = nullptr; a.y = nullptr;
  a.N = 1; a.ID = 1; a.IH = 1; a.IW = G.C;
  a.RD = 1; a.RH = 1; a.RW = G.C; a.TD = a.TH = a.TW = 1;
  for (int d = 0; d < 3; ++d) { a.rs[d] = 1; a.tb[d] = 0; a.ts[d] = 1; a.os[d] = 1; a.oo[d] = 0; }
  a.xsn = a.xsd = a.xsh = (long)G.C * pl.K; a.xsw = pl.K;
  a.cink = pl.K; a.cin_lanes = pl.K; a.cout = X.C; a.cout_lanes = round_up(X.C, 8); a.wrows = round_up(X.C, 16);
  a.act = RG_ACT_NONE; a.slope = 0.f;
  a.M = G.C;
  a.xbytes = (unsigned)(pl.gt_el * 2);
  a.wbytes = (unsigned)(pl.xt_el * 2);
  a.krow = 0; a.cp = 0; a.ksplit = 1; a.part = nullptr;
  a.nbatch = KD * KH * KW; a.xbatch = 0; a.ybatch = 0;
  a.nbatch1 = KH * KW; a.wbatch = tx.copy_stride; a.wbatch2 = pl.P;      // batch = kd * (KH * KW) + (kh * KW + kw) = the tap index
  a.wld = tx.row_stride;
  a.raw = dw; a.raw_ld = (long)KD * KH * KW * X.C; a.raw_batch = X.C;
  a.d2s_c = 0;
  const int bn = 2 * pl.fn * 16;
  a.tiles_m = (int)((a.M + G5_BM - 1) / G5_BM);
  a.tiles_n = (a.wrows + bn - 1) / bn;
  static rg_device_latch attr;
  rg_once_per_device(&attr, [] {
    hipFuncSetAttribute((const void*)igemm5_kernel<false, 9, true>, hipFuncAttributeMaxDynamicSharedMemorySize, G5_LDS_BYTES(9));
    hipFuncSetAttribute((const void*)igemm5_kernel<false, 8, true>, hipFuncAttributeMaxDynamicSharedMemorySize, G5_LDS_BYTES(8));
  });
  rg_set_kernel_name("wgrad_gemm_igemm5_kernel<0,%d,1>", pl.fn);
  const dim3 grid((unsigned)(a.tiles_m * a.tiles_n * a.nbatch));
  if (pl.fn == 9) hipLaunchKernelGGL((igemm5_kernel<false, 9, true>), grid, dim3(512), G5_LDS_BYTES(9), st, a);
  else hipLaunchKernelGGL((igemm5_kernel<false, 8, true>), grid, dim3(512), G5_LDS_BYTES(8), st, a);
  rg_prof_mark(st);
  return RG_OK;
}
