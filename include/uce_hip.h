/*
 * uce_hip.h - C ABI of libuce_hip.so: the MI355X (gfx950) kernels behind the UCE hot path.
 *
 * The reference (rohitgandikota/unified-concept-editing) is pure Python on torch/diffusers and
 * has no FFI of its own; this ABI sits exactly where its inlined torch ops sit.  Each entry
 * point cites the reference lines it replaces (paths relative to the reference root).
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no torch types.  Every function returns int:
 *     0 = OK, negative = error (UCE_E* below, or -(hipError_t) - 1000 for HIP failures).
 *   - All data pointers are DEVICE pointers owned by the caller (e.g. tensor.data_ptr()).
 *     The library owns only an opaque per-handle workspace (uce_create / uce_reserve).
 *   - All work is enqueued on the caller's stream (pass torch.cuda.current_stream().cuda_stream
 *     as a void*); no call synchronises the device except uce_create, uce_reserve, uce_reserve_rows, uce_destroy
 *     (allocation) and uce_status (reads one int back).
 *   - One handle per GPU per thread; calls on one handle must not overlap in time - neither from two threads nor on two
 *     streams that run concurrently: the handle's hand-off words, split-contraction slabs and GroupNorm partials exist ONCE
 *     (launches that share them are ordered by being on one stream; a second stream needs an event edge or its own handle).
 *   - Matrices are row-major and dense unless stated.  d (the text-embedding width) must be
 *     a multiple of 64 (768 SD-1.x, 1024 SD-2.x, 2048 SDXL).
 *
 * Math: with C [N,d] the concept embeddings (edit rows first, then
 * preserve rows), G [N_edit,d] the targets of the edit rows (a preserve row's target is itself),
 * s [N] the per-row scales and lambda the regulariser, the reference's per-module
 *     W_new = (lambda W_old + sum_i s_i v*_i c_i^T)(lambda I + sum_i s_i c_i c_i^T)^-1 ,  v*_i = W_old g_i
 * collapses (to_k/to_v have no bias) to one module-independent update
 *     W_new = W_old + W_old Delta ,   Delta = (G-C)_e^T S_e C_e A^-1 ,   A = lambda I + C^T S C .
 */
#ifndef UCE_HIP_H
#define UCE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct uce_ctx* uce_handle_t;
typedef void* uce_stream_t; /* a hipStream_t */

enum {
  UCE_OK = 0,
  UCE_EINVAL = -22,   /* bad argument (null pointer, d not a multiple of 64, N <= 0, ...) */
  UCE_ENOMEM = -12,   /* workspace allocation failed */
  UCE_EDOM = -33,     /* system not positive definite (lambda <= 0 with rank-deficient C, s_i < 0) */
  UCE_ENOSYS = -38,   /* feature not built / not available in this process */
  UCE_ECOMM = -70,    /* the collective library reported an error (uce_bcast) */
  UCE_ETIMEDOUT = -110, /* a bounded in-launch wait between cooperating workgroups gave up (uce_status; the hand-off words are re-armed) */
  UCE_EHIP = -1000    /* -1000 - hipError_t */
};

enum { UCE_ALGO_AUTO = 0, UCE_ALGO_PRIMAL = 1, UCE_ALGO_DUAL = 2 };
enum { UCE_DTYPE_BF16 = 0, UCE_DTYPE_F16 = 1, UCE_DTYPE_F32 = 2 };

int uce_version(void);
const char* uce_strerror(int code);

/* Lifetime.  `device` is a HIP device ordinal.  uce_reserve pre-allocates the workspace for
 * systems up to n_max x n_max with embedding width up to d_max so that later calls never
 * allocate (the compute calls grow the workspace on demand otherwise, which synchronises). */
int uce_create(uce_handle_t* h, int device);
int uce_destroy(uce_handle_t h);
int uce_reserve(uce_handle_t h, int d_max, int n_max);

/* a4 - accumulate (uce_sd_erase.py:56-79, uce_sd_debias.py:114-138; once instead of per module):
 *   A  [d,d] f64 = lambda I + C^T S C                     (symmetric, both triangles written)
 *   Bt [d,d] f64 = C_e^T S_e (G - C_e)   ( = B^T, B = (G-C)_e^T S_e C_e )
 * fp32 inputs, products and sums in f64 (v_mfma_f64_16x16x4_f64). */
int uce_gram(uce_handle_t h, const float* C, const float* G, const float* s, int N, int N_edit, int d,
             float lamb, double* A, double* Bt, uce_stream_t stream);

/* a5 - solve (replaces torch.inverse(mat2.float()) of uce_sd_erase.py:82, 32x/140x identical):
 *   DeltaT [d,d] f32 = A^-1 Bt = Delta^T  via blocked f64 Cholesky + two triangular solves.
 * A is destroyed.  A non-positive pivot is recorded in the handle (see uce_status). */
int uce_solve_delta(uce_handle_t h, double* A, const double* Bt, int d, float* DeltaT, uce_stream_t stream);

/* The same solve for a NARROW right-hand side: X [d, m] f32 = A^-1 B, B [d, m] f64, m a multiple of 64 (the bias
 * direction u = A^-1 sum_i s_i c_i of the biased-Linear variants, trainscripts/uce_flux_edit.py:86-113, rides in one
 * column).  A is destroyed; pivot failures as uce_solve_delta. */
int uce_solve_rhs(uce_handle_t h, double* A, const double* B, int d, int m, float* X, uce_stream_t stream);

/* The EDGE path of a5: a symmetric INDEFINITE system (negative scales or lambda <= 0 - the reference's `torch.inverse(mat2)` of
 * uce_sd_erase.py:82 is an LU inverse and accepts them, a Cholesky does not):  X [n, m] f32 = A^-1 B  by Gaussian elimination with
 * partial pivoting in f64 on [A | B] (csrc/uce_lu.hip: one pivot launch + one update launch per column, then the back
 * substitution - latency-bound, ~14 ms at n = 768).  A and B are destroyed.  Synchronises the stream once (the singularity
 * threshold n eps max|A|); a pivot below it is recorded in the handle like a failed Cholesky pivot (uce_status: column + 1). */
int uce_solve_general(uce_handle_t h, double* A, double* B, int n, int m, float* X, uce_stream_t stream);

/* a5 - apply (replaces `mat1 @ inverse` of uce_sd_erase.py:82 for ALL modules in one launch):
 *   W_new [rows,d] = W_old + W_old Delta ; rows = sum of the modules' out_features (the host
 *   keeps every attn2.to_k/to_v weight in one [rows,d] slab).  Products carry fp32 accuracy: each row of W_old
 *   and of (I + Delta)^T is scaled by a power of two and split exactly into two f16 terms, the three significant
 *   partial products run on the f16 matrix cores with fp32 accumulation, the scales leave in the epilogue (each
 *   operand carries 22 significand bits - 2^-22 relative - for elements within 2^16 of their row's maximum, <= 2^-29
 *   of that maximum below; 3.2e-7 rel. Frobenius against fp64 on the SD-1.4 slab; bf16 split 4.3e-7, f32-MFMA kernel 6.8e-7).  UCE_APPLY_VARIANT (read at uce_create) = 1: the three-way bf16 split, six products, no scales - also
 *   what a slab beyond 2 GB of planes takes; 0: the f32-MFMA kernel whose products are bit-exact fmaf chains.
 *   Uses the handle's row workspace (uce_reserve_rows(h, rows, d + 64) pre-sizes it). */
int uce_apply(uce_handle_t h, const float* W_old, const float* DeltaT, float* W_new, long rows, int d,
              uce_stream_t stream);

/* Dual (N < d) form of a4+a5: with K = lambda S^-1 + C C^T (N x N, SPD),
 *   Dm [N_edit,d] f32 = (G - C_e),   R [N_edit,d] f32 = rows 0..N_edit-1 of K^-1 C,
 *   Delta = Dm^T R.  Rows with s_i == 0 are rejected (UCE_EINVAL). */
int uce_dual_factors(uce_handle_t h, const float* C, const float* G, const float* s, int N, int N_edit,
                     int d, float lamb, float* Dm, float* R, uce_stream_t stream);

/* Low-rank apply: W_new = W_old + (W_old Dm^T) R (N_edit <= 256).  For d in {768, 1024, 2048} and >= 1024 rows this
 * is uce_lowrank_project (into the handle's T) + uce_lowrank_update; otherwise one fused 16-row-tile pass. */
int uce_apply_lowrank(uce_handle_t h, const float* W_old, const float* Dm, const float* R, float* W_new,
                      long rows, int d, int N_edit, uce_stream_t stream);

/* The same update as two kernels (what uce_edit uses for d in {768, 1024, 2048} and slabs of >= 1024 rows):
 *   uce_lowrank_project : T [rows, NEP] f32 = W_old Dm^T,  NEP = roundup(N_edit, 64) is T's row stride.
 *                         Needs only Dm, not the solve: uce_edit runs the 64x64 Cholesky of the dual
 *                         system inside the same launch (block 0), hidden under this GEMM.
 *   uce_lowrank_update  : W_new = W_old + T R   (one HBM-bound pass over the weights).
 * uce_reserve_rows pre-allocates the handle's own T for uce_edit (rows x roundup(n_edit_max, 64)). */
int uce_lowrank_project(uce_handle_t h, const float* W_old, const float* Dm, float* T, long rows, int d,
                        int N_edit, uce_stream_t stream);
int uce_lowrank_update(uce_handle_t h, const float* W_old, const float* T, const float* R, float* W_new,
                       long rows, int d, int N_edit, uce_stream_t stream);
int uce_reserve_rows(uce_handle_t h, long rows_max, int n_edit_max);

/* DeltaT [d,d] f32 from the dual factors (used when N_edit is too large for the low-rank apply). */
int uce_delta_from_factors(uce_handle_t h, const float* Dm, const float* R, int N_edit, int d,
                           float* DeltaT, uce_stream_t stream);

/* The whole of uce_sd_erase.py:45-82 for every module: picks dual/primal and low-rank/full by
 * (N, N_edit, d) when algo == UCE_ALGO_AUTO.  W_new may not alias W_old.
 * Stream capture: once the workspace is reserved (uce_reserve / uce_reserve_rows, or one eager call of the same shape)
 * the call allocates nothing and keeps no per-launch value on the host - the hand-off words of its cooperating
 * workgroups live on the device and are re-armed by the launch itself - so it may be captured into a hipGraph and
 * replayed (tests/test_stress_gpu.py). */
int uce_edit(uce_handle_t h, const float* C, const float* G, const float* s, int N, int N_edit, int d,
             float lamb, const float* W_old, float* W_new, long rows, int algo, uce_stream_t stream);

/* Synchronises `stream` and returns the status word of the last solve on this handle:
 * *info = 0 OK, k > 0: leading minor k not positive definite (function then returns UCE_EDOM); k < 0: a bounded wait between
 * cooperating workgroups of ONE launch expired (the edit's rider chain, the persistent Cholesky, the one-launch GroupNorm of
 * uce_groupnorm_nhwc_fwd - e.g. on a device that could not keep the launch's grid resident): UCE_ETIMEDOUT, every hand-off word
 * of the handle is re-armed, the outputs of that launch are not valid. */
int uce_status(uce_handle_t h, int* info, uce_stream_t stream);

/* Measurement aid for bench.py (per-kernel time measured live with HIP events on the launch
 * stream): between uce_profile_begin and uce_profile_end every kernel launch (or launch chain) that uce_edit and
 * the calls it is composed of issue on this handle is bracketed by two HIP events.  uce_profile_end synchronises
 * `stream` and writes one line per kernel, "<name> <total ms> <launches>\n", into `report` (UCE_EINVAL when `cap`
 * is too small).  The brackets serialise nothing but add two event records per launch: never inside a timed region. */
int uce_profile_begin(uce_handle_t h);
int uce_profile_end(uce_handle_t h, uce_stream_t stream, char* report, size_t cap);

/* a6 - debias drift (uce_sd_debias.py:122-127, cumulative over iterations):
 *   G [N_edit,d] f32 = C_edit + Dsum [N_edit,N_debias] (f64, = sum_t direction_scale_t) @ C_debias */
int uce_debias_targets(uce_handle_t h, const float* C_edit, const float* C_debias, const double* Dsum,
                       int N_edit, int N_debias, int d, float* G, uce_stream_t stream);

/* a2 / SURVEY 8f row 1 - last-token gather of the BATCHED concept-embedding extraction (the reference runs one
 * text-encoder call per string and slices t_emb[0][0, attention_mask.sum() - 2, :], uce_sd_erase.py:25-42):
 *   out [B, d] f32 = hidden[i, idx[i], :]  for hidden [B, L, d] in bf16 / f16 / f32 (dtype), idx [B] int32 on the device
 *   (clamped to [0, L)), d % 8 == 0.  Reads only the B rows that are needed and widens them to fp32 (C / G rows). */
int uce_gather_last_token(uce_handle_t h, const void* hidden, const int* idx, float* out, int B, int L, int d, int dtype,
                          uce_stream_t stream);

/* a8 - weight patch (generate-images-sd.py:17-19, uce_sd_debias.py:15-19): f32 -> bf16
 * round-to-nearest-even cast of the edited slab into the U-Net's parameter storage. */
int uce_cast_bf16(uce_handle_t h, const float* src, void* dst_bf16, long n, uce_stream_t stream);

/* a10 - cross-attention at inference (diffusers AttnProcessor2_0 -> F.scaled_dot_product_attention,
 * reached from generate-images-sd.py:37-42):  O = softmax(Q K^T * scale) V per (batch, head).
 *   q,o: [B, Lq, H*dh]   k,v: [B, Lk, H*dh]   (diffusers' [B, L, C] layout, heads are column
 *   slices), Lk <= 128 (77 for CLIP), dh in {40, 64, 80, 96, 128, 160}, bf16 or f16 I/O,
 *   f32 softmax and accumulation. */
int uce_xattn_fwd(uce_handle_t h, const void* q, const void* k, const void* v, void* o, int B, int H,
                  int Lq, int Lk, int dh, float scale, int dtype, uce_stream_t stream);

/* SURVEY section 8(f) row 3 - the U-Net's SELF-attention (attn1) at inference, same call site and layout as
 * uce_xattn_fwd but any Lk (flash-style streaming over key tiles, online softmax):
 *   q,o: [B, Lq, H*dh]   k,v: [B, Lk, H*dh],  dh a multiple of 8, <= 160, B*H <= 65535, bf16 or f16 I/O.
 * Uses a V^T scratch in the handle (grown on demand: the first call at a larger shape allocates). */
int uce_sattn_fwd(uce_handle_t h, const void* q, const void* k, const void* v, void* o, int B, int H,
                  int Lq, int Lk, int dh, float scale, int dtype, uce_stream_t stream);

/* The same self-attention reading q, k and v from ONE packed projection qkv [B, L, 3 * H * dh] (columns [0, C) = q, [C, 2C) = k,
 * [2C, 3C) = v with C = H * dh): what the fused to_q | to_k | to_v linear of an attn1 layer writes (uce_linear_fwd on the three
 * weights concatenated along their rows) - the three [B, L, C] tensors never exist.  o [B, L, C]. */
int uce_sattn_packed_fwd(uce_handle_t h, const void* qkv, void* o, int B, int H, int L, int dh, float scale, int dtype,
                         uce_stream_t stream);

/* The packed self-attention in the exp2 domain: the q columns of qkv already hold q * scale * log2(e) - written that way by
 * uce_linear_colscale_fwd (the scale is applied to the f32 product, so q is rounded ONCE, as in the unscaled projection) - and
 * the softmax numerator is exp2(q' . k - max).  Where uce_sattn_exp2_form(h, B, H, L, dh) returns 1 (dh = 40 on the streaming
 * two-tile kernel: SD-1.4's 64 x 64 level) the running maximum rides in the head's padding dim and a score leaves the matrix pipe
 * as the argument of exp2 - no multiply-add per element in a loop that is bound by vector-instruction issue; every other shape
 * runs its usual kernel with a unit factor.  Same result as uce_sattn_packed_fwd on the unscaled projection up to the rounding
 * of q' in place of q (diffusers Attention / AttnProcessor2_0 under evalscripts/generate-images-sd.py:37-42). */
int uce_sattn_exp2_form(uce_handle_t h, int B, int H, int L, int dh);
int uce_sattn_packed_exp2_fwd(uce_handle_t h, const void* qkv, void* o, int B, int H, int L, int dh, int dtype,
                              uce_stream_t stream);

/* SURVEY section 8(f) row 3 - GroupNorm (+ SiLU) of the U-Net / VAE at inference (diffusers ResnetBlock2D:
 * conv(silu(group_norm(x)))) for channels-last activations:  x, y [N, HW, C] (an NCHW tensor in
 * torch.channels_last memory format), gamma, beta [C], all bf16 or f16; G <= 64 groups of C/G consecutive
 * channels, C % 8 == 0, C <= 4096; statistics in f32/f64.  ws: N * uce_groupnorm_chunks(HW) * G * 2 floats of
 * caller-owned scratch. */
int uce_groupnorm_chunks(int HW);
int uce_groupnorm_nhwc_fwd(uce_handle_t h, const void* x, const void* addend, const void* gamma, const void* beta, void* y,
                           float* ws, int N, int HW, int C, int G, float eps, int silu, int dtype, long addend_ld,
                           uce_stream_t stream);

/* uce_groupnorm_nhwc_fwd of the channel-wise concatenation of TWO tensors that is never written: channels [0, C1) of a pixel come
 * from x [N, HW, C1], channels [C1, C) from x2 [N, HW, C - C1] - `res(torch.cat([x, skip], dim=1))` of the U-Net's up blocks
 * (diffusers UpBlock2D / CrossAttnUpBlock2D).  C1 % 8 == 0, 0 < C1 < C; y [N, HW, C]. */
int uce_groupnorm_cat_nhwc_fwd(uce_handle_t h, const void* x, const void* x2, int C1, const void* addend, const void* gamma,
                               const void* beta, void* y, float* ws, int N, int HW, int C, int G, float eps, int silu, int dtype,
                               long addend_ld, uce_stream_t stream);
/* `addend` (may be NULL): [N, C] with row stride addend_ld elements (0: C; a multiple of 8 - a column slice of the U-Net's one
 * hoisted time-projection GEMM), same dtype; x + addend[n][c] is what gets normalised - the time-embedding add of
 * ResnetBlock2D and the bias of the convolution that produced x, folded into the normalisation.
 * uce_add_bias_nhwc_fwd: y = a + b + bias[c] over [pixels, C] (b and bias may be NULL): the residual joins. */
int uce_add_bias_nhwc_fwd(uce_handle_t h, const void* a, const void* b, const void* bias, void* y, long pixels, int C,
                          int dtype, uce_stream_t stream);

/* Classifier-free-guidance combine + PNDM (PLMS) scheduler step in one pass (diffusers' `noise_pred_uncond + g *
 * (noise_pred_text - noise_pred_uncond)` and `PNDMScheduler.step`, reached from generate-images-sd.py:37-42):
 *   e = eps[0:n] + guidance * (eps[n:2n] - eps[0:n])  (cfg != 0; else e = eps[0:n])          -> eps_out [n]
 *   prev = cs * sample - ce * (w[0] e + w[1] h1 + w[2] h2 + w[3] h3)                           -> prev_out [n]
 * h1..h3: earlier model outputs (NULL = absent), w: the multistep weights (host floats), bf16 or f16, n % 8 == 0. */
int uce_cfg_pndm_step(uce_handle_t h, const void* eps, int cfg, float guidance, const void* h1, const void* h2,
                      const void* h3, const float* w, const void* sample, float cs, float ce, void* eps_out, void* prev_out,
                      long n, int dtype, uce_stream_t stream);

/* GEGLU of the transformer feed-forward (diffusers GEGLU, exact erf GELU): x [rows, 2*inner] -> y [rows, inner] =
 * x[:, :inner] * gelu(x[:, inner:]), bf16 or f16, inner % 8 == 0. */
int uce_geglu_fwd(uce_handle_t h, const void* x, void* y, long rows, int inner, int dtype, uce_stream_t stream);

/* Row softmax ahead of a P V product: p [rows, L] (bf16 / f16) = softmax(scale * s [rows, L]) with s in f32 - the middle step of
 * the single-head, 512-dim attention of the VAE decoder's mid block (diffusers AutoencoderKL; too wide for the register-resident
 * attention kernels), whose two products run on uce_linear_fwd.  L % 8 == 0, L <= 16384. */
int uce_softmax_rows(uce_handle_t h, const float* s, void* p, long rows, int L, float scale, int dtype, uce_stream_t stream);

/* LayerNorm over the last dim of [rows, C] (bf16 or f16, C % 8 == 0, C <= 2560; gamma / beta [C] in the same dtype;
 * f32 statistics) - the norm1/2/3 of diffusers' BasicTransformerBlock - optionally fused with the residual join in
 * front of it: when `residual` is given, s = x + residual is rounded to the element type, written to `sum_out`
 * ([rows, C]) and y = LN(s).  residual and sum_out are both NULL or both set.  UCE_ENOSYS for wider rows. */
int uce_layernorm_fwd(uce_handle_t h, const void* x, const void* residual, const void* gamma, const void* beta, void* y,
                      void* sum_out, long rows, int C, float eps, int dtype, uce_stream_t stream);

/* Patch matrix of a 3x3 / stride 1 / pad 1 convolution: x [N, H, W, C] (channels-last, 16-bit elements, C % 8 == 0)
 * -> cols [N*H*W, 9*C], column (ky*3 + kx)*C + c, zero outside the image; the convolution is then ONE library GEMM
 * with the channels-last weight viewed as [Cout, 9*C] (1.0-1.3 PF/s in hipBLASLt vs MIOpen's 0.3-0.55 PF/s).
 * upsample = 1: x is [N, H/2, W/2, C] and the patches are those of its 2x nearest-neighbour upsampling (H, W even;
 * diffusers' Upsample2D = F.interpolate(scale 2, "nearest") + conv) - the upsampled tensor is never written. */
int uce_im2col3x3_nhwc(uce_handle_t h, const void* x, void* cols, int N, int H, int W, int C, int upsample,
                       uce_stream_t stream);

/* Patch matrix of a 3x3 / pad 1 convolution with FOUR input channels (conv_in on the latents, U-Net and VAE decoder):
 * x [N, H, W, 4] -> cols [N*H*W, 64] (16-bit elements): column (ky*3 + kx)*4 + c, columns 36..63 zero - the convolution is then
 * uce_linear_fwd against the weight laid out the same way ([Cout, 64]). */
int uce_im2col3x3_c4(uce_handle_t h, const void* x, void* cols, int N, int H, int W, uce_stream_t stream);

/* The same convolution as ONE implicit-GEMM kernel (csrc/uce_conv_dma.hip, uce_conv_igemm.hip): the nine taps are gathered on the
 * way into LDS, no patch matrix exists.  x [N, H, W, Cin] (with upsample = 1: [N, H/2, W/2, Cin], the convolution of its 2x
 * nearest-neighbour upsampling; with stride = 2: [N, 2H, 2W, Cin], diffusers' Downsample2D), w [Cout, 3, 3, Cin] (a channels-last
 * Conv2d weight), bias [Cout] or NULL, residual [N, H, W, Cout] or NULL (added in the epilogue: the `x + conv2(h) + b` join of
 * ResnetBlock2D), y [N, H, W, Cout]; bf16 or f16, f32 accumulation; Cin % 32 == 0, Cout % 8 == 0.  stride = 2 and residual need
 * Cout % 128 == 0 or Cout % 320 == 0 (UCE_ENOSYS otherwise). */
int uce_conv3x3_nhwc_fwd(uce_handle_t h, const void* x, const void* w, const void* bias, void* y, int N, int H, int W,
                         int Cin, int Cout, int upsample, int stride, const void* residual, int dtype, uce_stream_t stream);

/* Linear layer with a fused epilogue (csrc/uce_gemm.hip) - every nn.Linear / 1x1 convolution under `pipe(...)` of
 * evalscripts/generate-images-sd.py:37-42 (diffusers Attention.to_q/to_k/to_v/to_out, FeedForward, Transformer2DModel.proj_in /
 * proj_out, ResnetBlock2D.conv_shortcut; the reference gets them from torch's GEMM library plus separate element-wise passes):
 *   UCE_EPILOGUE_NONE :  y [M, N]   = x [M, K] w [N, K]^T (+ bias [N]) (+ residual [M, N])
 *   UCE_EPILOGUE_GEGLU:  y [M, N/2] = (hidden + b_h) * gelu(gate + b_g), erf form (diffusers GEGLU); w / bias hold the hidden
 *                        and gate rows INTERLEAVED per 32: row 32 t + r = hidden row 16 t + r (r < 16), gate row 16 t + r - 16
 *                        (r >= 16); no residual.
 *   UCE_EPILOGUE_F32  :  as NONE with y [M, N] in f32 (16-byte aligned): attention scores ahead of uce_softmax_rows.
 * bf16 or f16 elements, f32 accumulation; ldx / ldr / ldy = row strides in elements (operands may be column slices of wider
 * tensors).  K % 32 == 0, N % 4 == 0 (GEGLU: N % 32 == 0), ldx % 8 == 0, ldy % 4 == 0, ldr % 4 == 0; x, w 16-byte aligned, the
 * others 8-byte aligned; bias / residual may be NULL. */
enum { UCE_EPILOGUE_NONE = 0, UCE_EPILOGUE_GEGLU = 1, UCE_EPILOGUE_F32 = 2 };
int uce_linear_fwd(uce_handle_t h, const void* x, long ldx, const void* w, const void* bias, const void* residual, long ldr,
                   void* y, long ldy, long M, int N, int K, int epilogue, int dtype, uce_stream_t stream);

/* uce_linear_fwd (no bias, no residual, 16-bit output) whose columns [0, scale_cols) leave multiplied by `scale`:
 *   y[:, :scale_cols] = scale * (x w^T)[:, :scale_cols],  y[:, scale_cols:] = (x w^T)[:, scale_cols:]
 * - the f32 accumulator is scaled before the one rounding to the element type.  scale_cols % 32 == 0.  The packed q | k | v
 * projection of an attn1 layer with scale_cols = H * dh, scale = dh^-0.5 * log2(e) feeds uce_sattn_packed_exp2_fwd. */
int uce_linear_colscale_fwd(uce_handle_t h, const void* x, long ldx, const void* w, void* y, long ldy, long M, int N, int K,
                            int scale_cols, float scale, int dtype, uce_stream_t stream);

/* uce_linear_fwd over a two-source contraction: columns [0, K1) of a row of the input come from x (row stride ldx), columns
 * [K1, K) from x2 (row stride ldx2) - the 1x1 conv_shortcut of an up block's ResnetBlock2D reading x and the skip connection in
 * place instead of their torch.cat.  K1 % 32 == 0, 0 < K1 < K, ldx2 % 8 == 0, x2 16-byte aligned. */
int uce_linear_cat_fwd(uce_handle_t h, const void* x, long ldx, const void* x2, long ldx2, int K1, const void* w, const void* bias,
                       const void* residual, long ldr, void* y, long ldy, long M, int N, int K, int epilogue, int dtype,
                       uce_stream_t stream);

/* e - the one exchange step of the multi-GPU generation path (broadcast of the edited weights from rank 0 over RCCL / xGMI;
 * generate-images-sd.py:17-19 loads the artifact on every process - here rank 0 loads it and the others receive it).
 *   buf [bytes] on the device, in place on every rank; `comm` is the caller's ncclComm_t (RCCL), passed as void*.
 * The Python host issues this step through torch.distributed (backend "nccl" = RCCL, uce_amd/generate.py); this entry point gives
 * a host WITHOUT torch the same collective.  The library does not link RCCL: ncclBroadcast is resolved at the first call from the
 * library named by UCE_RCCL_LIB, else from the RCCL already visible in the process, else from librccl.so (UCE_ENOSYS when none is
 * there) - `comm` must come from that same RCCL. */
int uce_bcast(uce_handle_t h, void* buf, size_t bytes, int root, void* comm, uce_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* UCE_HIP_H */
