// conv6 — "line" convolution: taps along W only, narrow on both sides (<= 64 channel lanes in, <= 64 out).
//
// The CycleGAN generators' 7^3 stem and composed tail arrive here after their in-plane taps were folded into channels (DESIGN 3a,
// rg_diag_spread / rg_diag_sum): (1,1,7)-tap convolutions between 49 (56 lanes) and 64 channels over 2.1 - 2.4 M voxels, 16 forward /
// backward-data launches per train step.  igemm4's FLAT path ran them at 180 - 235 TFLOP/s (0.43 - 0.50 ms each): with 64 output
// columns every activation byte feeds only 64 MACs per tap, and each of the 7 taps fetched it again as 16-row x 64-byte fragments.
// Both rooflines (578 MB, 120 GFLOP) say ~0.12 ms.  This kernel reaches 0.37 - 0.43 ms (250 - 265 TFLOP/s) whether the lines are contiguous in
// memory or a 2-MB plane apart per record: it is bound by its own loop — one wave per SIMD (150 KB of LDS per workgroup), 7 dependent
// ds_read_b128 per 12 MFMAs, two fragment sets in flight — not by memory.
// STATUS: parity-green (tests/test_ops_gpu.py::test_line_convolution_kernel) but NOT the default (rg_set_option "RG_LINEW" 1): since ops.py folds
// (kd, kh) into channels and leaves the KW taps on the memory-fastest axis, igemm4's FLAT path runs these layers at 0.36 ms and the CycleGAN step is
// 1.1 ms slower with this kernel (same-box A/B, 84.7 against 83.5 ms).
//
// Staged in LDS with the records of a line back to back, the K index of an output voxel is a CONTIGUOUS window: 7 taps along W x L lanes =
// TW * L elements starting at the voxel's record, so
//   * a wave stages ONE W-line (RW + TW - 1 records, <= 20 KB) in its own LDS region with plain 16-byte loads, record by record (the
//     records of a line may lie a whole plane apart in memory) — every input byte is read from memory once per launch;
//   * an MFMA A-fragment (16 voxels x 32 flat-K elements) is one ds_read_b128 per lane at  row * P + k * 2 : record pitch P = 112 B
//     for 56 lanes (records packed back to back — K-steps of 32 straddle records — and 28 dwords apart: 16 rows hit 16 distinct
//     bank quads), 144 B for 64 lanes (128-byte records would put 16 rows on two bank quads);
//   * the weights (<= 64 rows x 448 K, tap order reversed for the backward-data direction) live in LDS once per workgroup, rows 912 B
//     apart (same bank argument); three row fragments share every weight fragment (7 reads per 12 MFMAs).
// Waves are independent after the weight load (a wave owns its line buffer): no workgroup barrier in the loop.
#include "igemm_args.h"

#define LW_ROWS 144            // rows (output voxels of one line) a wave handles: 3 groups x 3 fragments x 16
#define LW_RECS 160            // records of a wave's line buffer (rows + taps + the K padding's read-ahead)
#define LW_KROW 448            // K elements per weight row (TW * L padded to a multiple of 64)
#define LW_BP 912              // weight row pitch in LDS (bytes)
#define LW_APMAX 144           // largest record pitch
#define LW_LDS (64 * LW_BP + 4 * LW_RECS * LW_APMAX)

__global__ __launch_bounds__(256, 1) void linew_kernel(Igemm4Args a, int L, int P, int nlines) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lw_lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fg = lane >> 4;
  const int TW = a.TW, rev = a.ts[2] < 0;
  const int G = L >> 3;                          // 16-byte pieces per record
  const int nrec = a.RW + TW - 1;                // records of the window of one line
  unsigned char* const Bl = lw_lds;
  unsigned char* const Al = lw_lds + 64 * LW_BP + wave * (LW_RECS * LW_APMAX);
  // ---- weights -> LDS, in WINDOW order (ascending memory): for the backward-data direction window position u holds tap TW-1-u
  for (int q = tid; q < 64 * (LW_KROW / 8); q += 256) {
    const int row = q / (LW_KROW / 8), pc = q - row * (LW_KROW / 8);
    int spc = pc;
    const int k = pc * 8, u = k / L;
    if (rev && u < TW) spc = ((TW - 1 - u) * L + (k - u * L)) >> 3;
    bf16x8 v = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
    if (row < a.wrows && k < TW * L) v = *(const bf16x8*)(a.w + (long)row * a.krow + spc * 8);
    *(bf16x8*)(Bl + row * LW_BP + pc * 16) = v;
  }
  __syncthreads();
  const int npieces = nrec * G;                   // 16-byte pieces of a line window (<= 150 * 8 = 1200)
  const int wfirst = rev ? a.tb[2] - (TW - 1) : a.tb[2];   // input voxel of window record 0 (row stride along W is 1)
  // lines are numbered with the memory-FASTER of the two outer axes innermost: the four waves of a workgroup — and neighbouring workgroups —
  // then stage records that lie next to each other (the D<->W-exchanged views put the tap axis on the slowest memory axis: the records of one
  // line are a whole plane apart)
  const bool dfast = a.xsd < a.xsh;
  const int inner = dfast ? a.RD : a.RH;
  constexpr int NP = (LW_RECS * 8 + 63) / 64;     // pieces per lane, at most
  bf16x8 pre[NP];
  // all of a lane's loads of a line are issued together (one load per loop trip, waited for at once, is ~16 serialised memory latencies per line)
  auto fetch = [&](int line) {
    const int n = (int)(line / ((long)a.RD * a.RH)); const int rem = (int)(line - (long)n * a.RD * a.RH);
    const int o = rem / inner, in_ = rem - o * inner;
    const int rd = dfast ? in_ : o, rh = dfast ? o : in_;
    const int id = rd * a.rs[0] + a.tb[0], ih = rh * a.rs[1] + a.tb[1];
    const bool lv = (unsigned)id < (unsigned)a.ID && (unsigned)ih < (unsigned)a.IH;
    const bf16_t* src = a.x + (long)n * a.xsn + (long)id * a.xsd + (long)ih * a.xsh;
#pragma unroll
    for (int u = 0; u < NP; ++u) {
      const int q = lane + u * 64;
      const int r = q / G, g = q - r * G, wv = wfirst + r;
      pre[u] = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
      if (lv && q < npieces && (unsigned)wv < (unsigned)a.IW) pre[u] = *(const bf16x8*)(src + (long)wv * a.xsw + g * 8);
    }
  };
  int line = blockIdx.x * 4 + wave;
  if (line < nlines) fetch(line);
  for (; line < nlines; line += gridDim.x * 4) {
    const int n = (int)(line / ((long)a.RD * a.RH)); const int rem = (int)(line - (long)n * a.RD * a.RH);
    const int o = rem / inner, in_ = rem - o * inner;
    const int rd = dfast ? in_ : o, rh = dfast ? o : in_;
    // ---- stage the line: piece q = (record r, piece g); two zero records behind the window (the K padding reads one record ahead)
#pragma unroll
    for (int u = 0; u < NP; ++u) {
      const int q = lane + u * 64;
      if (q < npieces + 2 * G) { const int r = q / G, g = q - r * G; *(bf16x8*)(Al + r * P + g * 16) = pre[u]; }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (the wave's own LDS writes; a wave runs in lockstep)
    if (line + (int)gridDim.x * 4 < nlines) fetch(line + gridDim.x * 4);   // the next line's loads fly while this one is multiplied
    bf16_t* const ybase = a.y + (long)n * a.ysn + (long)(rd * a.os[0] + a.oo[0]) * a.ysd + (long)(rh * a.os[1] + a.oo[1]) * a.ysh;
    for (int grp = 0; grp * 48 < a.RW; ++grp) {
      f32x4 acc[3][4];
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
      const unsigned char* ar = Al + (grp * 48 + fr) * P;
      const unsigned char* br = Bl + fr * LW_BP + fg * 16;
      // two fragment sets: step s+1 is read while step s is multiplied (one wave per SIMD: nothing else hides the LDS latency); the scheduler
      // must not hoist more than that — fully unrolled and free, it kept all 14 steps' fragments live and spilled 87 registers
      bf16x8 af[2][3], bf[2][4];
      auto ld = [&](int buf, int st) {
        const int k0 = st * 32 + fg * 8, t = k0 / L;            // flat k0 -> record k0 / L, lane k0 % L
        const int ao = t * P + (k0 - t * L) * 2;
#pragma unroll
        for (int i = 0; i < 3; ++i) af[buf][i] = *(const bf16x8*)(ar + i * 16 * P + ao);
#pragma unroll
        for (int j = 0; j < 4; ++j) bf[buf][j] = *(const bf16x8*)(br + j * 16 * LW_BP + st * 64);
      };
      ld(0, 0);
#pragma unroll
      for (int st = 0; st < LW_KROW / 32; ++st) {
        __builtin_amdgcn_sched_barrier(0);
        if (st + 1 < LW_KROW / 32) ld((st + 1) & 1, st + 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j] = RG_MFMA16(bf[st & 1][j], af[st & 1][i], acc[i][j]);
      }
      __builtin_amdgcn_sched_barrier(0);
      // ---- epilogue: lane holds columns c .. c+3 of row (lane & 15) of every fragment
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const int row = grp * 48 + i * 16 + fr;
        if (row >= a.RW) continue;
        bf16_t* yrow = ybase + (long)(row * a.os[2] + a.oo[2]) * a.ysw;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int c = j * 16 + fg * 4;
          if (c >= a.cout_lanes) continue;
          bf16x4 o;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float v = 0.f;
            if (c + r < a.cout) v = act_fwd(acc[i][j][r] + (a.bias ? a.bias[c + r] : 0.f), a.act, a.slope);
            o[r] = (short)f2bf(v);
          }
          *(bf16x4*)(yrow + c) = o;
        }
      }
    }
  }
}

// taps along W only, unit row strides, 56- or 64-lane records (any voxel pitch: a line is staged record by record — the D<->W-exchanged views of
// ops.conv_kd_as_kw put the tap axis on the SLOWEST memory axis), <= 64 output lanes, enough lines to fill the chip
bool rg_linew_ok(const Igemm4Args& a, bool flat) {
  if (!rg_opt(RG_OPT_LINEW)) return false;   // opt-in (rg_set_option "RG_LINEW" 1): level with the kernels it would replace at step level, see the header
  const int L = a.cin_lanes;
  if (a.TD != 1 || a.TH != 1 || a.TW < 2 || a.nbatch != 1 || (L != 56 && L != 64) || a.TW * L > LW_KROW) return false;
  if (a.rs[0] < 1 || a.rs[1] < 1 || a.rs[2] != 1 || (a.ts[2] != 1 && a.ts[2] != -1)) return false;
  if ((a.xsw & 7) || a.xsw < L || a.RW > LW_ROWS || a.RW + a.TW - 1 + 2 > LW_RECS || a.wrows > 64 || a.cout_lanes > 64 || (a.cout_lanes & 3)) return false;
  if ((flat ? a.krow : a.TW * a.cink) != ((a.TW * L + 63) / 64) * 64) return false;   // the packed row must be [tap][L lanes] padded once
  if ((((unsigned long)a.x) & 15) || (a.xsn & 7) || (a.xsd & 7) || (a.xsh & 7)) return false;
  // measured on the CycleGAN 128^3 step (tools/shapes_cg.py, ms per launch, this kernel / igemm4-FLAT or igemm3): 56 lanes backward-data 0.43 / 0.50,
  // 64 lanes forward 0.38 / 0.44, 64 lanes backward-data 0.37 / 0.43 — but 56 lanes forward (128-row lines) 0.56 / 0.52: left where it was
  if (L == 56 && a.ts[2] > 0) return false;
  const long nlines = (long)a.N * a.RD * a.RH;
  return nlines >= 1024 && nlines < (1L << 31);
}
int rg_launch_linew(Igemm4Args& a, bool flat, hipStream_t st) {
  const int L = a.cin_lanes, P = L == 56 ? 112 : LW_APMAX;
  if (!flat) a.krow = a.TW * a.cink;
  const int nlines = (int)((long)a.N * a.RD * a.RH);
  static rg_device_latch attr;
  rg_once_per_device(&attr, [] { hipFuncSetAttribute((const void*)linew_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LW_LDS); });
  int grid = (nlines + 3) / 4; if (grid > 256) grid = 256;
  rg_set_kernel_name("linew_kernel<%d,%d>", L, a.ts[2] < 0 ? 1 : 0);
  hipLaunchKernelGGL(linew_kernel, dim3(grid), dim3(256), LW_LDS, st, a, L, P, nlines);
  return RG_OK;
}
// the same decision from the tensor views (conv.hip routes 64-lane problems to igemm3 / igemm2 unless this kernel takes them)
bool rg_linew_wanted(const CL5& X, const CL5& Y, const rg_conv_params* p) {
  Igemm4Args a;
  memset(&a, 0, sizeof(a));
  a.x = X.p; a.N = X.N; a.nbatch = 1;
  a.RD = p->R[0]; a.RH = p->R[1]; a.RW = p->R[2]; a.TD = p->T[0]; a.TH = p->T[1]; a.TW = p->T[2];
  for (int d = 0; d < 3; ++d) { a.rs[d] = p->rstride[d]; a.ts[d] = p->tstep[d]; }
  a.xsn = X.sn; a.xsd = X.sd; a.xsh = X.sh; a.xsw = X.sw;
  a.cink = round_up(X.C, 64); a.cin_lanes = X.lanes; a.cout_lanes = Y.lanes; a.wrows = round_up(p->cout, 16);
  return rg_linew_ok(a, false);
}
