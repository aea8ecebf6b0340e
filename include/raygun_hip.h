/* libraygun_hip.so — C-ABI of the MI355X (gfx950) hot path of raygun.
 *
 * Every entry point replaces a torch/cuDNN (or affogato) call the reference issues on its hot
 * path; the reference file:line each one stands in for is cited next to it (paths relative to
 * /root/reference/src/raygun/).  Conventions:
 *   - plain pointers + sizes only, no torch types; the caller owns every buffer (workspace too);
 *   - all work is enqueued asynchronously on the hipStream_t passed in; no hidden syncs;
 *   - return 0 on success, a negative rg_status otherwise; rg_last_error_string() explains it;
 *   - never throws, never aborts; re-entrant across streams/devices (no global mutable state
 *     except the thread-local error string).
 *
 * Two builds of the same sources export the same entry points: libraygun_hip.so stores activations and packed weights as
 * bf16 (RG_BF16), libraygun_hip_f16.so (-DRG_STORE_F16) as IEEE half (RG_F16; rg_version() has bit 16 set).  Wherever "bf16" is
 * written below, read "the build's 16-bit storage type"; a tensor of the other type is rejected with RG_ERR_BAD_SHAPE.
 *
 * Tensor convention: activations are channels-last bf16, "NDHWC with padded lanes": logical shape
 * (N,C,D,H,W), channel stride 1, every other stride a multiple of 8 elements, base 16-byte aligned.
 * Lanes [C, lanes) of every voxel are zero (lanes = C rounded up to 8).  Strides are in elements.
 */
#ifndef RAYGUN_HIP_H
#define RAYGUN_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ihipStream_t* rg_stream; /* == hipStream_t */

enum rg_status {
  RG_OK = 0,
  RG_ERR_BAD_SHAPE = -1,
  RG_ERR_UNSUPPORTED = -2,
  RG_ERR_HIP = -3,
  RG_ERR_WORKSPACE = -4
};
enum rg_dtype { RG_BF16 = 0, RG_F32 = 1, RG_U8 = 2, RG_I32 = 3, RG_U64 = 4, RG_F64 = 5, RG_F16 = 6 };
enum rg_act { RG_ACT_NONE = 0, RG_ACT_RELU = 1, RG_ACT_LEAKY = 2, RG_ACT_SIGMOID = 3, RG_ACT_TANH = 4 };

typedef struct {
  void* ptr;
  int32_t dtype; /* rg_dtype */
  int32_t rank;  /* 5 for activations: (N,C,D,H,W) logical order */
  int64_t sizes[6];
  int64_t strides[6]; /* in elements */
} rg_tensor;

int rg_version(void);
/* measurement aid: with rg_profile_enable(1) the conv / weight-gradient entry points record a HIP event pair around their dominant
 * kernel on the launch stream; rg_profile_drain returns the elapsed milliseconds in call order (synchronises) and resets. */
void rg_profile_enable(int on);
int rg_profile_drain(float* ms, int cap);
const char* rg_last_error_string(void);
/* name (template instantiation) of the main kernel launched by this thread's last conv/wgrad call — profiling aid */
const char* rg_last_kernel_name(void);
/* Diagnostic switches (0 = production default): "RG_FM" (forced igemm4 tile height 1..4), "RG_NO_HALO", "RG_NO_THIN", "RG_NO_THIN3",
 * "RG_NO_WLINES", "RG_NO_KWP" (kernel families switched off: the launch falls through to the next candidate), "RG_GEMM5" (the LDS-DMA
 * GEMM of gemm5.hip: -1 off, 1 wherever its 144-column tiling allows, 0 the heuristic), "RG_NO_WGEMM" (1: the wide weight gradients stay on wgrad_kwp instead of
 * the transposed-operand GEMM form), "RG_DEBUG_BOUNDS"
 * (every entry point re-derives the byte extent its kernels may touch from the descriptors and fails with RG_ERR_BAD_SHAPE instead
 * of launching when a workspace / packed operand / view is smaller).  The environment variables of the same names are read ONCE,
 * when the library is loaded; no launch path reads the environment.  Process-wide. */
int rg_hip_status(void);   /* 0, or the hipError_t pending on this thread (a failed kernel launch is otherwise silent); sets the error string */
int rg_set_option(const char* name, int32_t value);
int32_t rg_get_option(const char* name);   /* -1: unknown name */

/* ---- layout conversion (torch .to(memory_format)/.float() around the nets: torch/predict/worker.py:101-110,146-149)
 * src: any dtype in {bf16,f32,u8}, any strides.  dst: bf16 or f32, any strides.  dst_lanes >= C: channel lanes
 * [C,dst_lanes) of dst are zero-filled (dst channel stride must be 1 when dst_lanes > C).
 * y = x*scale + shift fuses the block-runner normalisation (worker.py:109-114). */
int rg_copy_convert(const rg_tensor* src, const rg_tensor* dst, int dst_lanes, float scale, float shift, rg_stream s);
/* the block runner's write-back (worker.py:133-149) in one pass: the network's channels-last output straight into the ROI of the destination array —
 * dst may also be u8: y = (uint8) clamp(x * scale + shift, lo, hi), truncated like the reference's `(out * iinfo.max).clip(...).astype(dtype)` —
 * and *nonfinite_flag (optional, float, never cleared here) = 1 when a source value is inf / NaN (IEEE-half storage overflows at 65504: such a
 * block is predicted again with bf16 storage). */
int rg_copy_convert_store(const rg_tensor* src, const rg_tensor* dst, int dst_lanes, float scale, float shift, float lo, float hi,
                          float* nonfinite_flag, rg_stream s);

/* ---- weight packing.  w is the fp32 master in torch layout:
 *   transposed == 0 : Conv3d weight (Cout,Cin,KD,KH,KW)        [torch.nn.Conv3d, networks/UNet.py:58-69]
 *   transposed == 1 : ConvTranspose3d weight (Cin,Cout,KD,KH,KW) [networks/ResNet.py:325-333]
 * mode 0 (forward operand of a conv / dgrad operand of a convT): rows = dim0 ... see conv.hip.
 * Packed layout: bf16 [rows16][ntaps][redK], rows16 = rows rounded up to 16, redK = reduced channels rounded
 * up to red_align (8, 32 or 64), zero filled; red_align 1 = FLAT: bf16 [rows16][(tap, reduced channels rounded up to 8), the row padded
 * once to a multiple of 64] (no per-tap padding).  `swap` selects which of the two channel dims is the row dim (0: dim0 rows, 1: dim1 rows).
 * The tap subset is {k0 + kstep*j, j < ntap} per dim (parity classes of strided transposed convs). */
typedef struct {
  int32_t dim0, dim1, K[3];
  int32_t swap;
  int32_t k0[3], kstep[3], ntap[3];
  int32_t red_align; /* = rg_conv3d_kalign(reduced, rows): 64/32/8 = reduced-channel padding per tap, 1 = FLAT rows; 0 means 32 */
  int32_t src_channels_last; /* 0: w is (dim0, dim1, KD,KH,KW) contiguous; 1: torch.channels_last_3d memory, i.e. [dim0][KD][KH][KW][dim1] */
} rg_pack_params;
/* Packing mode (rg_pack_params.red_align) that rg_conv3d_igemm expects for a problem with `cin` reduced and `cout` produced channels */
int rg_conv3d_kalign(int cin, int cout);
int64_t rg_pack_weights_bytes(const rg_pack_params* p);
int rg_pack_weights(const float* w, void* packed_bf16, const rg_pack_params* p, rg_stream s);

/* ---- implicit-GEMM convolution engine (Conv3d fwd, Conv3d dgrad, ConvTranspose3d fwd/dgrad).
 * Replaces torch.nn.Conv3d/ConvTranspose3d forward+backward-data as issued by ConvPass (networks/UNet.py:58-69),
 * ResnetGenerator3D (networks/ResNet.py:283,291-298,325-338), ResnetBlock3D (:407,416) and
 * NLayerDiscriminator3D (networks/NLayerDiscriminator.py:136-176).
 * rows form a grid (N,RD,RH,RW); row r and tap k read x at  r*rstride + tbase + tstep*k  (zero outside the volume)
 * and write y at r*ostride + ooffset.  y = act(sum + bias). */
typedef struct {
  int32_t R[3];       /* row grid per sample */
  int32_t T[3];       /* taps per dim */
  int32_t rstride[3]; /* row -> input coordinate stride */
  int32_t tbase[3], tstep[3];
  int32_t ostride[3], ooffset[3];
  int32_t cout;       /* real output channels */
  int32_t act;        /* rg_act, fused epilogue */
  float slope;        /* LeakyReLU slope */
} rg_conv_params;
int rg_conv3d_igemm(const rg_tensor* x, const void* w_packed, const float* bias, const rg_tensor* y,
                    const rg_conv_params* p, rg_stream s);
/* The same with a caller-owned workspace: launches whose tile grid would leave most CUs idle (few output rows, long K: the
 * backward-data pass of the 432 <- 2592 layer at 21^3, the PatchGAN's 15^3 layers) split K over up to 8 workgroups per tile, fp32
 * partial sums go through the workspace and a second pass applies bias + activation.  rg_conv3d_igemm_workspace_bytes returns an
 * upper bound (0: never splits); a NULL / too small workspace simply runs unsplit. */
/* A stride-2 / pad-1 / 3^3 ConvTranspose3d (networks/ResNet.py:325-333; equally the backward-data pass of the stride-2 Conv3d of :291-298) as ONE
 * convolution on the input grid — 8 parity classes x C channels = 8 C columns, 2^3 taps (weights: the sub-pixel rearrangement, packed like any
 * forward operand) — whose epilogue writes column block q = (qd, qh, qw) of row r to channel c of output voxel 2 r + q: no intermediate tensor,
 * no shuffle pass.  y: (N, C, 2 R0, 2 R1, 2 R2), C % 16 == 0; bias: C floats or NULL; p->cout = 8 C, p->T = (2, 2, 2).  dry_run != 0 only
 * answers whether the problem can run this way (RG_OK) — otherwise use rg_conv3d_igemm2 + rg_d2s_add_act. */
int rg_conv3d_igemm_d2s(const rg_tensor* x, const void* w_packed, const float* bias, const rg_tensor* y, const rg_conv_params* p,
                        int32_t dry_run, rg_stream s);
int64_t rg_conv3d_igemm_workspace_bytes(const rg_tensor* x, const rg_tensor* y, const rg_conv_params* p);
int rg_conv3d_igemm2(const rg_tensor* x, const void* w_packed, const float* bias, const rg_tensor* y,
                     const rg_conv_params* p, void* workspace, size_t ws_bytes, rg_stream s);

/* ---- Winograd F(2x2x2, 3x3x3) path for wide stride-1 3x3x3 convolutions (same torch.nn.Conv3d call sites as above, networks/UNet.py:58-69:
 * config U's 2592 -> 2592 level).  y = A^T [ sum_c U(k,c) . V(c) ] A per 2x2x2 output tile; the 64 transform positions are 64
 * independent GEMMs.  Layouts (16-bit storage type of this build):
 *   V, M  [64][tiles][row stride]      tiles = N * tiles[0] * tiles[1] * tiles[2], tile t covers outputs 2t .. 2t+1 per axis
 *   U     [64][rg_wino_weights_rows(cout)][rg_wino_weights_red(cin)]   forward operand (rows = output channels)
 *   Ut    [64][rg_wino_weights_rows(cin)][rg_wino_weights_red(cout)]   backward-data operand (the flipped kernel: positions mirrored)
 * rg_wino_input reads x at origin + 2*tile + (0..3) per axis, zeros outside the tensor (origin 0: valid convolution; -2: the
 * full correlation of the backward-data pass) and zero-fills the lanes up to v_cstride.  rg_wino_output adds the bias, applies
 * the activation and crops the last odd tile.  rg_gemm_batched: C[b][m][n] = sum_k A[b][m][k] * W[b][n][k] on the
 * fragment-direct implicit-GEMM kernel, fp32 accumulation; n must tile by 128 or 144 columns, A's lanes k .. round_up(k, 64)
 * must hold zeros (rg_wino_input writes them). */
int64_t rg_wino_weights_rows(int32_t channels);
int64_t rg_wino_weights_red(int32_t channels);
int rg_wino_weights(const void* w_f32, int32_t channels_last, int32_t cout, int32_t cin, void* U, void* Ut, rg_stream s);
int rg_wino_input(const rg_tensor* x, void* V, int64_t v_cstride, const int32_t tiles[3], const int32_t origin[3], rg_stream s);
int rg_wino_output(const void* M, int64_t m_cstride, const int32_t tiles[3], rg_tensor* y, const float* bias, int32_t cout,
                   int32_t act, float slope, rg_stream s);
int32_t rg_gemm_batched_tile_n(int32_t n);   /* 8 | 9 (x 16 columns) or 0: rg_gemm_batched refuses this n — host logic, no GPU call */
int rg_gemm_batched(const void* A, const void* W, void* C, int64_t batches, int64_t m, int32_t k, int32_t n, int64_t a_row_stride,
                    int64_t a_batch_stride, int64_t w_batch_stride, int64_t c_row_stride, int64_t c_batch_stride, rg_stream s);

/* ---- weight gradient.  dW[a][b][tap] = sum_rows g[row][a] * x[row*rstride + tbase + tstep*tap][b]
 * (torch's conv backward-weight for the layers above).  Two steps: partial sums over row splits into the
 * caller's fp32 workspace, then a deterministic reduction that also transposes into torch's (a,b,KD,KH,KW) layout.
 * `accumulate` != 0 adds into dw instead of overwriting. */
int64_t rg_conv3d_wgrad_workspace_bytes(const rg_tensor* g, const rg_tensor* x, const rg_conv_params* p, int32_t* nsplit);
int rg_conv3d_wgrad(const rg_tensor* g, const rg_tensor* x, float* dw, const rg_conv_params* p, int accumulate,
                    void* workspace, size_t ws_bytes, rg_stream s);
/* flags bit 0: accumulate; bit 1: dw is channels-last memory, [a][KD][KH][KW][b] (torch.channels_last_3d of an (a,b,KD,KH,KW)
 * tensor) — the layout the kernels accumulate in: with one row split the partial sums ARE the result (no transpose pass). */
int rg_conv3d_wgrad2(const rg_tensor* g, const rg_tensor* x, float* dw, const rg_conv_params* p, int flags,
                     void* workspace, size_t ws_bytes, rg_stream s);

/* ---- activation backward + bias gradient: g = gy * act'(y);  db[c] = sum_voxels g  (fp32, may be NULL).
 * Replaces autograd of ReLU/LeakyReLU/Sigmoid after a conv (UNet.py:90-91, NLayerDiscriminator.py:138,156,173). */
int rg_act_bwd_bias(const rg_tensor* gy, const rg_tensor* y, const rg_tensor* g, float* db, int act, float slope,
                    void* workspace, size_t ws_bytes, rg_stream s);
int64_t rg_act_bwd_bias_workspace_bytes(const rg_tensor* gy);

/* ---- MaxPool3d k=s=factor, floor mode (MaxDownsample, networks/UNet.py:209-228). bwd recomputes the arg-max
 * (first maximum in d,h,w scan order, torch's tie rule) and writes the full gx (zeros elsewhere). */
int rg_maxpool3d_fwd(const rg_tensor* x, const rg_tensor* y, const int32_t factor[3], rg_stream s);
int rg_maxpool3d_bwd(const rg_tensor* x, const rg_tensor* gy, const rg_tensor* gx, const int32_t factor[3], rg_stream s);

/* ---- nearest upsample + centre crop + channel concat (Upsample.forward, networks/UNet.py:338-351):
 * out[:, :Cf] = crop(f, off_f) ; out[:, Cf:] = crop(upsample(g, factor), off_g).
 * bwd: gf = zero-embedded crop of gout[:, :Cf] ; gg = factor^3 sum-pool of gout[:, Cf:] over the cropped window. */
int rg_upcat_fwd(const rg_tensor* f, const rg_tensor* g, const rg_tensor* out, const int32_t factor[3],
                 const int32_t off_f[3], const int32_t off_g[3], rg_stream s);
int rg_upcat_bwd(const rg_tensor* gout, const rg_tensor* gf, const rg_tensor* gg, const int32_t factor[3],
                 const int32_t off_f[3], const int32_t off_g[3], rg_stream s);

/* ---- InstanceNorm3d(affine=False, track_running_stats=False, eps) (+activation, +residual add)
 * (networks/ResNet.py:284-285,299-300,408-409,417; NLayerDiscriminator.py:154-155; systems/BaseSystem.py:193-199).
 * stats: float2 (mean, rstd) per (n,c).  apply: y = act((x-mean)*rstd) [+ res].  pad > 0 writes y reflect-padded
 * by `pad` voxels per side (fuses the following ReflectionPad3d, ResNet.py:270,392). */
int64_t rg_instnorm_workspace_bytes(const rg_tensor* x);
int rg_instnorm_stats(const rg_tensor* x, float* mean_rstd, float eps, void* workspace, size_t ws_bytes, rg_stream s);
int rg_instnorm_apply(const rg_tensor* x, const float* mean_rstd, const rg_tensor* res, const rg_tensor* y,
                      int act, float slope, rg_stream s);
/* gx = rstd * (g - mean(g) - xhat*mean(g*xhat)),  g = gy*act'(y).  y may be NULL when act == none. */
int rg_instnorm_bwd(const rg_tensor* x, const float* mean_rstd, const rg_tensor* y, const rg_tensor* gy,
                    const rg_tensor* gx, int act, float slope, void* workspace, size_t ws_bytes, rg_stream s);

/* Affine norms — InstanceNormNd(affine=True) and BatchNormNd (NLayerDiscriminator.py:107-115's default norm_layer; a user's norm_layer argument,
 * networks/ResNet.py:231-262): the same two kernels with a per-channel (gamma, beta) table `gamma_beta` (float2 per channel, NULL = none):
 *   apply: y = act((x - mean) * rstd * gamma + beta) [+ res]     with whatever (mean, rstd) table the caller built (instance statistics from
 *          rg_instnorm_stats, batch statistics = their combination over n, or running statistics);
 *   bwd:   gx = gamma * rstd * (g - m0 - xhat * m1), g = gy * act'(y), (m0, m1) = means of g and g * xhat over the instance, or — batch_stats != 0 —
 *          over the whole batch.  The (m0, m1) table is left in the workspace behind the partial sums (offset rg_instnorm_workspace_bytes(x),
 *          float2 per (n, c)): d beta[c] = voxels * sum_n m0, d gamma[c] = voxels * sum_n m1. */
int rg_norm_apply_affine(const rg_tensor* x, const float* mean_rstd, const float* gamma_beta, const rg_tensor* res, const rg_tensor* y,
                         int act, float slope, rg_stream s);
int rg_norm_bwd_affine(const rg_tensor* x, const float* mean_rstd, const float* gamma_beta, int batch_stats, const rg_tensor* y,
                       const rg_tensor* gy, const rg_tensor* gx, int act, float slope, void* workspace, size_t ws_bytes, rg_stream s);

/* ---- ReflectionPad3d / zero pad (ResNet.py:270,337,392): y = pad(x); bwd folds gy back onto gx. mode 0 zero, 1 reflect. */
int rg_pad3d_fwd(const rg_tensor* x, const rg_tensor* y, const int32_t pad[3], int mode, rg_stream s);
int rg_pad3d_bwd(const rg_tensor* gy, const rg_tensor* gx, const int32_t pad[3], int mode, rg_stream s);

/* ---- residual add  y = a + b  (ResnetBlock3D.forward, ResNet.py:436-443) */
int rg_add(const rg_tensor* a, const rg_tensor* b, const rg_tensor* y, rg_stream s);
/* y = a + zero_pad(b), b placed at `offset` inside a's extent (a may be NULL): the gradient of a tensor consumed whole by one op
 * and centre-cropped by another (UNet skip connections, networks/UNet.py:97-106,327-336) in one pass. */
int rg_add_pad(const rg_tensor* a, const rg_tensor* b, const rg_tensor* y, const int32_t offset[3], rg_stream s);

/* ---- in-plane taps <-> channels of a one-channel volume (build-owned helpers of the 7^3 stem / composed tail,
 * networks/ResNet.py:283,338): spread  y[n,d,h,w][kh*KW+kw] = x[n,d,h+s*kh,w+s*kw][0] (0 outside x);
 * sum  y[n,d,h,w][0] = bias[0] + sum_{kh,kw} x[n,d,h+s*kh,w+s*kw][kh*KW+kw] (terms outside x dropped); s = sign (+1/-1).
 * x/y: bf16 channels-last, one of them 1 channel, the other KH*KW channels, equal batch and depth. */
int rg_diag_spread(const rg_tensor* x, const rg_tensor* y, int kh, int kw, int sign, rg_stream s);
int rg_diag_sum(const rg_tensor* x, const rg_tensor* y, int kh, int kw, int sign, const float* bias, rg_stream s);
/* Weights of `conv3d(cat([skip, nearest_upsample_x2(low)]), w)` (networks/UNet.py:338-351 followed by ConvPass, :70-95) split for
 * the sub-pixel evaluation: wf = w[:, :cf] (O, cf, 3,3,3) and weff (8*O, C-cf, 2,2,2) = the 2^3 kernels the upsampled channels
 * reduce to on the low-resolution tensor, one per output parity; both channels-last ([out][kd][kh][kw][in]) fp32.  w: fp32
 * (O, C, 3,3,3), any strides.  _bwd: the adjoint — dw (strides as given) from the gradients of wf / weff (either may be NULL). */
int rg_subpixel_weights(const rg_tensor* w, int cf, float* wf, float* weff, rg_stream s);
/* out = act(y + depth_to_space(z)) (y may be NULL: a plain shuffle): z (N, 8*C, J, J, J) holds the 8 output parities of the upsampled operand's partial sums on the
 * low-resolution grid (channel = parity*C + c); out[o] takes cell (o + offset) >> 1, parity (o + offset) & 1 per dimension
 * (offset in {0,1}).  _bwd: g = gout * act'(out) (the gradient of y) and gz = its inverse shuffle (0 where the crop cut). */
int rg_d2s_add_act(const rg_tensor* y, const rg_tensor* z, const rg_tensor* out, const int32_t offset[3], int act, float slope, rg_stream s);
int rg_d2s_add_act_bwd(const rg_tensor* gout, const rg_tensor* out, const rg_tensor* g, const rg_tensor* gz, const int32_t offset[3],
                       int act, float slope, rg_stream s);
int rg_subpixel_weights_bwd(const float* g_wf, const float* g_weff, const rg_tensor* dw, int cf, rg_stream s);

/* ---- losses: value (fp32 scalar in out[0]) and gradient wrt pred in one pass.
 * kind 0: mean((p-t)^2) vs constant target  (GANLoss lsgan, losses/GANLoss.py:24-25,56-58)
 * kind 1: SmoothL1(beta=1) mean vs tensor t  (LinkCycleLoss.py:22,99-106)
 * kind 2: weighted MSE of WeightedMSELoss_MTLSD._calc_loss (losses/WeightedMSELoss_MTLSD.py:11-25):
 *         out[0]=sum_{w>0} w*(p-t)^2, out[1]=count(w>0), out[2]=count(scaled!=0); grad = grad_scale*2*w*(p-t) (w>0).
 * kind 3: BCEWithLogitsLoss against the constant const_target (GANLoss "vanilla", losses/GANLoss.py:26-27): out[0]=sum(max(p,0) - p*ct + log(1+exp(-|p|))),
 *         grad = grad_scale*(sigmoid(p) - ct).   kind 4: out[0]=sum(const_target*p), grad = grad_scale*const_target ("wgangp", GANLoss.py:52-57: ct = -1 real, +1 fake).
 * pred/target/weight/grad: rank-1 contiguous f32 of equal length.  `out` must hold 3 + 3*512 floats (the tail is the
 * per-workgroup partials; the three results are reduced from them in a fixed order). */
int rg_loss_fwd_bwd(int kind, const rg_tensor* pred, const rg_tensor* target, const rg_tensor* weight, float const_target,
                    float grad_scale, const rg_tensor* grad, float* out, rg_stream s);

/* ---- fused Adam over one flat fp32 parameter (torch.optim.Adam semantics, optimizers/utils.py:4-9) */
int rg_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                 float weight_decay, int32_t step, float grad_scale, rg_stream s);
/* same, and refreshes the packed bf16 forward operand (rg_pack_weights of the same channels-last master, swap = 0, all taps) in the
 * same pass; `packed` must have been produced by rg_pack_weights with `pp` before (its pad entries are kept). */
int rg_adam_step_pack(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                      float weight_decay, int32_t step, float grad_scale, void* packed, const rg_pack_params* pp, rg_stream s);
/* the same update for n_items tensors in ONE launch per 24 tensors (torch.optim.Adam(params) over a whole network: systems/CycleGAN.py:167-184 builds
 * optimisers over four networks = 96 tensors): arrays of n_items pointers / element counts; packed[i] / pp[i] may be NULL per tensor (or the arrays NULL);
 * one set of hyper-parameters and one step count for all. */
int rg_adam_step_multi(float* const* p, const float* const* g, float* const* m, float* const* v, const int64_t* n, void* const* packed,
                       const rg_pack_params* const* pp, int32_t n_items, float lr, float beta1, float beta2, float eps, float weight_decay, int32_t step,
                       float grad_scale, rg_stream s);
/* the same with the step-dependent values in DEVICE memory: hyp = { lr, 1 - beta1^step, sqrt(1 - beta2^step) } (3 floats; NULL = rg_adam_step_multi) is
 * read by the kernel, `lr` and `step` are ignored — a launch captured in a HIP graph (raygun_amd/graphs.py) stays correct as the step count advances
 * (torch.optim.Adam(capturable=True) keeps its step on the device for the same reason). */
int rg_adam_step_multi_hyp(float* const* p, const float* const* g, float* const* m, float* const* v, const int64_t* n, void* const* packed,
                           const rg_pack_params* const* pp, int32_t n_items, float lr, float beta1, float beta2, float eps, float weight_decay, int32_t step,
                           float grad_scale, const float* hyp, rg_stream s);

/* ---- loss-scaled f16-storage training (no reference counterpart: the reference trains in fp32; torch.cuda.amp.GradScaler's
 * "unscale and look for inf" step without the unscale pass — rg_adam_step takes 1/scale as grad_scale).
 * *flag = 1.0f if any of the n values of g is inf or NaN; the flag is never cleared here (clear it once, check every gradient). */
int rg_nonfinite_check(const float* g, int64_t n, float* flag, rg_stream s);

/* ---- mutex watershed (affogato.segmentation.compute_mws_segmentation, called at segment.py:197-199).
 * affogato's Python wrapper (not vendored; signature as published: compute_mws_segmentation(weights, offsets,
 * number_of_attractive_channels, strides=None, randomize_strides=False, invert_repulsive_weights=True, bias_cut=0., mask=None,
 * algorithm='kruskal')) transforms the weights before sorting, in float32:  w[n_attractive:] = w[n_attractive:] * -1 + 1  when
 * invert_repulsive_weights, then  w[:n_attractive] += bias_cut.  Both are applied here while the sort keys are formed (same
 * float32 roundings), the caller's array is not modified.  The reference passes neither argument, so it gets invert = 1, bias 0. */
typedef struct {
  int32_t shape[3];
  int32_t n_offsets, n_attractive;
  int32_t strides[3];     /* sub-sampling stride of repulsive edges */
  int32_t randomize_strides; /* must be 0 */
  int32_t invert_repulsive_weights; /* affogato default: 1 */
  float bias_cut;                   /* affogato default: 0 */
  int32_t origin[3];      /* global coordinate of voxel (0,0,0): the stride grid is (coordinate + origin) % stride == 0 (slabs of a larger volume) */
} rg_mws_params;
int64_t rg_mws_workspace_bytes(const rg_mws_params* p);
/* affs: float32 (n_offsets,Z,Y,X) already in "edge weight" form (segment.py:191-194); offsets int32 (n_offsets,3);
 * mask: uint8 (Z,Y,X) or NULL; labels: uint64 (Z,Y,X).  * The call waits for the stream once (the edge count sizes the radix sort).
 */
int rg_mws_segment(const float* affs, const int32_t* offsets, const uint8_t* mask, uint64_t* labels,
                   const rg_mws_params* p, void* workspace, size_t ws_bytes, rg_stream s);
/* The same pass reading the block runner's uint8 affinities (worker.py:133-144: value * 255) with the preprocessing of segment.py:188-194 applied on
 * the fly, bit for bit: a = (float)u8 * scale, w = 1 - a, channels below `prepare_sep` flipped back (w * -1 + 1) — i.e. rg_mws_segment on the output
 * of rg_mws_prepare_u8 without the float32 weight array (4 bytes per value: 87 GB for config 5's 2048^3 / 8 part).  `channel_stride`: elements
 * between channels (>= voxels of `p->shape`: a Z-slab of a larger volume is addressed in place).  `mask` as before (rg_mws_prepare_u8 with weights =
 * NULL computes it alone). */
int rg_mws_segment_u8(const uint8_t* affs8, float scale, int32_t prepare_sep, int64_t channel_stride, const int32_t* offsets, const uint8_t* mask,
                      uint64_t* labels, const rg_mws_params* p, void* workspace, size_t ws_bytes, rg_stream s);
/* Diagnostics of the last rg_mws_segment run that used `workspace`: out = {edges, rounds, window visits, 100-MHz ticks in phase
 * A, B, C of the round kernel, mutex-list nodes, 0}.  Synchronises the stream. */
int rg_mws_stats(const void* workspace, const rg_mws_params* p, uint64_t out[8], rg_stream s);
/* segment.py:188-194 on the device (mask from the channel mean, weights = 1 - affs with the attractive channels flipped back),
 * rounded exactly as numpy does in float32.  affs / weights: (n_offsets, n_voxels) float32 (may not alias), mask: n_voxels bytes. */
int rg_mws_prepare(const float* affs, float* weights, uint8_t* mask, int64_t n_voxels, int32_t n_offsets, int32_t n_attractive,
                   float thresh, rg_stream s);
/* the same from uint8 affinities (what the block runner stores for integer destinations, worker.py:133-144): a = (float)u8 * scale in
 * float32, i.e. bit for bit rg_mws_prepare(affs8.astype(float32) * scale) without the float array.  weights == NULL: the mask only. */
int rg_mws_prepare_u8(const uint8_t* affs, float scale, float* weights, uint8_t* mask, int64_t n_voxels, int32_t n_offsets,
                      int32_t n_attractive, float thresh, rg_stream s);


/* ---- slab decomposition of the mutex watershed (no reference counterpart: affogato segments one in-memory array; SURVEY.md 8e,
 * DESIGN.md 6).  First level: rg_mws_segment per Z-slab (origin[0] = the slab's first global plane).  Second level: a mutex
 * watershed over first-level clusters — rg_mws_cross_edges collects the grid edges from this slab into the neighbour above
 * (dir +1) / below (dir -1) given the neighbour's H nearest planes of GLOBAL cluster ids, rg_mws_slab_mutexes the mutexes the
 * first level left inside the slab between second-level nodes (is_node[local label]; pairs packed lo << 32 | hi, duplicates
 * possible), rg_mws_graph runs the Kruskal pass on the merged edge list (ascending wkey, ties by ascending gid; wkey 0 = mutexes
 * that exist before every edge) and returns rep[i] = smallest node of i's cluster, rg_mws_relabel maps labels through a table.
 * Record arrays are caller-allocated with capacity `cap`; *count (zeroed by the caller) ends as the number of records produced
 * (a value > cap means the arrays were too small and the surplus was dropped). */
int rg_mws_cross_edges(const float* weights, const int64_t* labels, int64_t base_own, const int64_t* nzone, int32_t H, int32_t dir,
                       int64_t n_global_voxels, const int32_t* offsets, const rg_mws_params* p, uint32_t* wkey, uint64_t* gid,
                       int64_t* gu, int64_t* gv, uint8_t* attractive, uint32_t* count, uint32_t cap, rg_stream s);
/* rg_mws_cross_edges reading uint8 affinities as rg_mws_segment_u8 does (same scale / prepare_sep / channel_stride). */
int rg_mws_cross_edges_u8(const uint8_t* affs8, float scale, int32_t prepare_sep, int64_t channel_stride, const int64_t* labels, int64_t base_own,
                          const int64_t* nzone, int32_t H, int32_t dir, int64_t n_global_voxels, const int32_t* offsets, const rg_mws_params* p,
                          uint32_t* wkey, uint64_t* gid, int64_t* gu, int64_t* gv, uint8_t* attractive, uint32_t* count, uint32_t cap, rg_stream s);
int rg_mws_slab_mutexes(const int64_t* labels, const uint8_t* is_node, const int32_t* offsets, const rg_mws_params* p, uint64_t* pairs,
                        uint32_t* count, uint32_t cap, rg_stream s);
int64_t rg_mws_graph_workspace_bytes(int64_t n_nodes, int64_t n_edges);
int rg_mws_graph(const uint32_t* wkey, const uint64_t* gid, const uint32_t* eu, const uint32_t* ev, const uint8_t* attractive,
                 int64_t n_edges, int64_t n_nodes, int32_t* rep, void* workspace, size_t ws_bytes, rg_stream s);
int rg_mws_relabel(int64_t* labels, const int64_t* lut, int64_t n, rg_stream s);

/* ---- the DEFAULT raygun-segment route: seeded watershed on the boundary-distance transform + waterz-style agglomeration
 * (reference: src/raygun/segment.py:15-136 `watershed_from_boundary_distance` / `watershed_from_affinities` / `get_segmentation`, dispatched from
 * `segment` :233-296; replaces scipy.ndimage.{distance_transform_edt, maximum_filter, label}, skimage.segmentation.watershed and waterz.agglomerate
 * on that path).  All volumes are dense C-order (Z, Y, X) device arrays, `shape` = host int32[3], < 2^31 voxels, extents < 16384.
 *   rg_ws_mask      segment.py:93    mask = mean over the n_channels float32 affinity channels > thresh, in numpy's float32 arithmetic
 *   rg_ws_edt       segment.py:94    EXACT squared Euclidean distance (int32) to the nearest mask == 0 voxel inside the volume
 *   rg_ws_maxima    segment.py:25-26 maxima[v] = (maximum filter of `size` with scipy's window / reflected borders) == d2[v]
 *   rg_ws_label     segment.py:27    6-connected components numbered 1.. by their first voxel in raster order (scipy.ndimage.label); *n_out (device)
 *   rg_ws_flood     segment.py:36-38 seeded watershed of (max distance - distance) within the mask: the build-defined minimum-spanning-forest order
 *                                    (see csrc/watershed.hip; parity unpinned against skimage); synchronises; *rounds_out (host, optional)
 *   rg_ws_rag_*     waterz region graph: one 64-bit record (u << 36 | v << 8 | bin) per (voxel, axis) whose ends carry different non-zero fragment ids
 *                   u < v < 2^28, bin = (int)(affinity * 255) of affs[axis][voxel]; count first, then emit into `keys` (*cursor_dev ends as the count)
 *   rg_ws_agglomerate (HOST arrays, no device work) waterz's iterative region merging with OneMinus<HistogramQuantileAffinity<.., 75, .., 256, false>>
 *                   over sorted unique records + multiplicities; lut_out[t * (n_fragments + 1) + id] = surviving fragment id at thresholds[t] (ascending)
 *   rg_ws_relabel   out[v] = lut[fragments[v]] */
int rg_ws_mask(const float* affs, int32_t n_channels, int64_t n_voxels, float thresh, uint8_t* mask, rg_stream s);
int64_t rg_ws_edt_workspace_bytes(const int32_t* shape);
int rg_ws_edt(const uint8_t* mask, const int32_t* shape, int32_t* d2, void* workspace, size_t ws_bytes, rg_stream s);
int64_t rg_ws_maxima_workspace_bytes(const int32_t* shape);
int rg_ws_maxima(const int32_t* d2, const int32_t* shape, int32_t size, uint8_t* maxima, void* workspace, size_t ws_bytes, rg_stream s);
int64_t rg_ws_label_workspace_bytes(const int32_t* shape);
int rg_ws_label(const uint8_t* fg, const int32_t* shape, int32_t* labels, int32_t* n_out, void* workspace, size_t ws_bytes, rg_stream s);
int64_t rg_ws_flood_workspace_bytes(const int32_t* shape);
int rg_ws_flood(const int32_t* d2, const int32_t* markers, const uint8_t* mask, const int32_t* shape, int32_t* out, int32_t* rounds_out,
                void* workspace, size_t ws_bytes, rg_stream s);
int rg_ws_rag_count(const int32_t* fragments, const int32_t* shape, uint64_t* count_dev, rg_stream s);
int rg_ws_rag_emit(const int32_t* fragments, const float* affs, const int32_t* shape, uint64_t* keys, uint64_t capacity, uint64_t* cursor_dev, rg_stream s);
int rg_ws_agglomerate(const uint64_t* keys, const int64_t* counts, int64_t n_records, int64_t n_fragments, const float* thresholds, int32_t n_thresholds,
                      int64_t* lut_out);
int rg_ws_relabel(const int32_t* fragments, const int64_t* lut, int64_t* out, int64_t n_voxels, rg_stream s);

/* ---- host-side codec for the chunk reader (no device work): one LZ4 block (the stream format inside zarr's default Blosc
 * compressor) -> dst; returns the number of bytes written or -1 on malformed input / a too small buffer. */
int64_t rg_lz4_decompress_block(const uint8_t* src, int64_t src_len, uint8_t* dst, int64_t dst_cap);
/* the writer's side (Blosc / lz4 frames for predict's zarr destinations, src/raygun/predict.py:111-125 -> daisy.prepare_ds's default compressor): one
 * LZ4 block (greedy matcher, 64 KiB window); returns the compressed size or -1 when dst_cap < what the input needs (rg_lz4_compress_bound(n) always fits). */
int64_t rg_lz4_compress_bound(int64_t n);
int64_t rg_lz4_compress_block(const uint8_t* src, int64_t src_len, uint8_t* dst, int64_t dst_cap);
/* whole Blosc (c-blosc 1.x) frames, one call per chunk (the Python side releases the GIL around it): compress = lz4 inner codec, byte shuffle for
 * typesize > 1 when shuffle != 0, unsplit blocks (blocksize 0: 256 KiB), stored form when nothing is gained; dst_cap >= rg_blosc_compress_bound(n).
 * decompress = lz4 / lz4hc and stored frames with byte shuffle, split or unsplit: returns nbytes, -1 malformed, -2 a frame for the Python reader
 * (bit shuffle, other inner codecs). */
int64_t rg_blosc_compress_bound(int64_t n);
int64_t rg_blosc_compress(const uint8_t* src, int64_t n, int32_t typesize, int32_t shuffle, int32_t blocksize, uint8_t* dst, int64_t dst_cap);
int64_t rg_blosc_nbytes(const uint8_t* src, int64_t src_len);
int64_t rg_blosc_decompress(const uint8_t* src, int64_t src_len, uint8_t* dst, int64_t dst_cap);

/* ---- train-time target generation of the MTLSD system (reference: src/raygun/io/MTLSDDataPipe.py:99-131; gunpowder's AddAffinities /
 * BalanceLabels and lsd's AddLocalShapeDescriptor restated — parity unpinned against those packages, oracle/targets.py).  Dense device
 * arrays: labels int32 (Z,Y,X), label mask uint8, outputs float32 (channels, oz, oy, ox) over the ROI [out_offset, out_offset + out_shape).
 * rg_add_affinities: aff[e][p] = labels[p] == labels[p + nhood[e]] and both > 0 (0 where p + nhood[e] leaves the array); affs_mask (nullable)
 *   is the same graph of labels_mask.  nhood: n_offsets x 3 int32 on the device.
 * rg_balance_labels: scale = mask * w[label], w[c] = 1 / 2 / clip(share of class c among the masked-in voxels, clipmin, clipmax); labels are
 *   the 0 / 1 affinities, mask nullable; workspace: rg_balance_labels_workspace_bytes() bytes.
 * rg_lsd_targets: 10 channels (mean offset z y x, variance z y x, Pearson zy zx yx, mass) of lsd's Gaussian mode on the grid subsampled by
 *   `downsample`; wz / wy / wx: the normalised 1-D Gaussian weights (2 radius + 1 floats each, device) of sigma / (voxel size * downsample),
 *   sub_voxel_size and sigma in world units (host arrays). */
int rg_add_affinities(const int32_t* labels, const uint8_t* labels_mask, const int32_t* nhood, int32_t n_offsets, const int32_t shape[3],
                      const int32_t out_offset[3], const int32_t out_shape[3], float* affs, float* affs_mask, rg_stream s);
int64_t rg_balance_labels_workspace_bytes(void);
int rg_balance_labels(const float* labels, const float* mask, int64_t n, float clipmin, float clipmax, float* scale, void* workspace,
                      size_t ws_bytes, rg_stream s);
int rg_lsd_targets(const int32_t* labels, const int32_t shape[3], const int32_t out_offset[3], const int32_t out_shape[3], int32_t downsample,
                   const float* wz, const float* wy, const float* wx, const int32_t radius[3], const float sub_voxel_size[3],
                   const float sigma[3], float* out, rg_stream s);

#ifdef __cplusplus
}
#endif
#endif
