/*
 * sdn.h -- C ABI of libsdn.so, the MI355X (gfx950) engine behind the safe-denoiser hot path.
 *
 * Conventions (all entry points):
 *   - every pointer is a DEVICE pointer to row-major contiguous memory owned by the caller,
 *     unless the parameter name ends in _host;
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream); calls are asynchronous
 *     and stream-ordered, never synchronise, never allocate (workspaces are caller-provided and
 *     sized by the *_workspace_bytes queries) -> every call is hipGraph-capturable;
 *   - return value: 0 = launched, <0 = SDN_E_* (invalid argument / launch error); no exceptions;
 *   - no global mutable state; one host thread per GPU.
 *
 * The reference (MingyuKim87/Safe_Denoiser) is pure Python with no FFI; each entry point cites
 * the reference interface (file:line under the reference root) whose arithmetic it replaces.
 * The reference-side binding a maintainer would add is shown in INTEGRATION.md.
 */
#ifndef SDN_H_
#define SDN_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SDN_OK            0
#define SDN_E_INVALID    -1   /* bad shape / null pointer / unsupported size            */
#define SDN_E_LAUNCH     -2   /* hipLaunchKernel reported an error                       */
#define SDN_E_WORKSPACE  -3   /* workspace too small                                     */
#define SDN_E_ARCH       -4   /* device is not gfx950                                    */

/* ABI version; bumped on any signature change. */
int sdn_abi_version(void);
/* Name of the device the library sees (host string, e.g. "gfx950"); NULL if no device. */
const char* sdn_device_arch_host(void);

/* ===================================================================================== *
 *  Repellency projection  
 * ===================================================================================== */

/* weight functions */
#define SDN_REPEL_RBF     0   /* w = exp(-||x-r|| / (2 sigma^2)),  den = sum w + eps      */
#define SDN_REPEL_SPARSE  1   /* w = relu(radius/||x-r|| - 1) for ||x-r|| < radius        */
/* query pre-processing */
#define SDN_QNORM_NONE    0
#define SDN_QNORM_CHANNEL 1   /* x / ||x||_2 over the channel axis per pixel (SD-v3)      */
/* what is written to out_x */
#define SDN_OUT_REPELLED  0   /* x - scale*neg   (RBF)  |  x + scale*force  (SPARSE)      */
#define SDN_OUT_NEG       1   /* the negative score itself (threshold module's conditioning_1,
                                 fast module's conditioning_2)                           */

typedef struct sdn_repel_params {
  int32_t n_query;        /* N: queries (prompts in flight); reference is fixed at 1      */
  int32_t n_ref;          /* M: rows of proj_ref                                          */
  int32_t channels;       /* C                                                            */
  int32_t hw;             /* H*W;  D = C*H*W must be a multiple of 4                      */
  int32_t weight_fn;      /* SDN_REPEL_*                                                  */
  int32_t qnorm;          /* SDN_QNORM_*                                                  */
  float   sigma;          /* RBF bandwidth (ignored for SPARSE)                           */
  float   radius;         /* SPARSE radius (ignored for RBF)                              */
  float   scale;          /* step size of the in-place update                             */
  float   epsilon;        /* additive epsilon of the RBF denominator                      */
  float   gate;           /* is_negation = den > gate (RBF);  sum w != 0 (SPARSE)         */
} sdn_repel_params;

/* Bytes of scratch sdn_repel_apply needs for these sizes. */
size_t sdn_repel_workspace_bytes(int32_t n_query, int32_t n_ref, int32_t channels, int32_t hw);

/*
 * One repellency projection for N queries against proj_ref [M, C, H, W] (fp32, NCHW flattened).
 *
 *   x        [N, D] fp32, updated IN PLACE:  x <- x - scale*neg  (RBF)  |  x + scale*force (SPARSE)
 *                                            (the reference mutates pred_original_sample in place)
 *   out_neg  [N, D] fp32 or NULL: the negative score (RBF) / the force (SPARSE)
 *   out_den  [N]    fp32 or NULL: RBF denominator incl. epsilon / SPARSE sum of weights
 *   out_isneg[N]    int32 or NULL: device-side gate, no host sync
 *
 * Replaces: RBFKernelRepellency.empirical_denoiser + conditioning_threshold / conditioning_1 / conditioning_2
 *   repellency/repellency_methods_threshold.py:171-193,309-349
 *   repellency/repellency_methods_fast.py:120-137,223-262
 *   repellency/repellency_methods_fast_sdv3.py:126-143,229-271 (SDN_QNORM_CHANNEL)
 * and SparseRepellency.repellency_force / conditioning_1
 *   repellency/repellency_methods_threshold.py:415-454, repellency_methods_fast.py:306-340.
 */
int sdn_repel_apply(const sdn_repel_params* p_host, float* x, const float* proj_ref,
                    float* out_neg, float* out_den, int32_t* out_isneg,
                    void* workspace, size_t workspace_bytes, void* stream);

/*
 * beta[n] = sum_m exp(-||q_n - r_m|| / (2 sigma^2)) + eps   for calibration queries q [N, D].
 * Replaces the per-timestep body of empirical_beta, repellency_methods_threshold.py:361-378
 * (the quantile over n stays on the host side).  With weight_fn = SDN_REPEL_SPARSE writes the
 * N*M pairwise distances to out [N, M] instead (empirical_radius, :472-487).
 */
int sdn_repel_calibrate(const sdn_repel_params* p_host, const float* queries, const float* proj_ref,
                        float* out, void* workspace, size_t workspace_bytes, void* stream);

/* ===================================================================================== *
 *  Guidance + scheduler step math  (rows P2, P3, S1, S1', S3) -- fp32, elementwise
 * ===================================================================================== */

/*
 * eps[p] = eps_u[p] + g * (eps_t[p] - eps_u[p]);  model_out is [n_branch * P, D] laid out as
 * chunk(n_branch): [P uncond | P text | (P extra, discarded when n_branch == 3)].
 * Replaces the CFG combine, models/textuals_visual/modified_safree_diffusion_pipeline_threshold_time.py:542-548.
 */
int sdn_cfg_combine(const float* model_out, int32_t n_prompt, int32_t n_branch, int64_t d,
                    float guidance_scale, float* eps, void* stream);

/*
 * Safe-latent-diffusion guidance (SLD eq. 3-8) for the 3-branch model output [P uncond | P text | P safety concept]:
 *   scale = min(|e_t - e_c| * sld_guidance_scale, 1), zeroed where (e_t - e_c) >= sld_threshold
 *   gs    = (e_c - e_u) * scale + sld_momentum_scale * momentum ;  momentum <- beta * momentum + (1 - beta) * gs
 *   eps   = e_u + g * ((e_t - e_u) - (apply_safety ? gs : 0))        apply_safety = (step index >= sld_warmup_steps)
 * `momentum` [P, D] fp32 is the caller's state (zero before the first step), updated in place.
 * Replaces models/textuals_visual/modified_sld_pipeline_threshold_time.py:467-503.
 */
int sdn_sld_guidance(const float* model_out, int32_t n_prompt, int64_t d, float guidance_scale,
                     float sld_guidance_scale, float sld_threshold, float sld_momentum_scale, float sld_mom_beta,
                     int32_t apply_safety, float* momentum, float* eps, void* stream);

/*
 * The two guidance kernels above with ONE guidance scale PER PROMPT: `guidance_rows` is a device array float[n_prompt].
 * The reference's drivers read the scale row by row from the prompt table (`guidance = data.guidance if hasattr(data,
 * 'guidance') else args.guidance_scale`, run_nudity.py:390-396) and call the pipeline once per prompt; the batched engine
 * keeps prompts with different scales in one batch instead of splitting the batch.  Same arithmetic per element.
 */
int sdn_cfg_combine_rows(const float* model_out, int32_t n_prompt, int32_t n_branch, int64_t d,
                         const float* guidance_rows, float* eps, void* stream);
int sdn_sld_guidance_rows(const float* model_out, int32_t n_prompt, int64_t d, const float* guidance_rows,
                          float sld_guidance_scale, float sld_threshold, float sld_momentum_scale, float sld_mom_beta,
                          int32_t apply_safety, float* momentum, float* eps, void* stream);

/* x0 = (x - sqrt_one_minus_ac * eps) / sqrt_ac, clamped to [-clip, clip] when clip > 0
 *   (epsilon prediction; DDPM/DDIM pred_original_sample).
 * Replaces DDPMScheduler.step(...).pred_original_sample at ...threshold_time.py:554 (diffusers 0.29.0). */
int sdn_pred_x0(const float* x, const float* eps, int64_t n, float sqrt_ac, float sqrt_one_minus_ac,
                float clip, float* x0, void* stream);

/* prev = c_x0 * x0(x, eps) + c_x * x + sigma * noise, with x0 as above (optionally clamped to
 * [-clip, clip] when clip > 0).  One kernel serves DDPM (ancestral; c_x0/c_x = posterior mean
 * coefficients, sigma = sqrt(max(var,1e-20)), noise may be NULL when sigma == 0) and DDIM eta=0
 * (c_x0 = sqrt(ac_prev) - c_eps*sqrt_ac_ratio ... folded by the host into the same three numbers).
 * Replaces scheduler.step(...).prev_sample at ...threshold_time.py:576 (diffusers 0.29.0 DDPM/DDIM). */
int sdn_sched_step(const float* x, const float* eps, const float* noise, int64_t n,
                   float sqrt_ac, float sqrt_one_minus_ac, float c_x0, float c_x, float c_eps, float sigma,
                   float clip, float* prev, void* stream);

/* noisy = sqrt_ac * x0 + sqrt_one_minus_ac * noise   (scheduler.add_noise, ...threshold_time.py:569). */
int sdn_add_noise(const float* x0, const float* noise, int64_t n, float sqrt_ac, float sqrt_one_minus_ac,
                  float* out, void* stream);

/* Device-side re-noise select for the repellency window, per prompt p (row of d elements):
 *   latents[p] <- isneg[p] ? sqrt_ac * x0r[p] + sqrt_one_minus_ac * noise[p] : latents[p]
 * Replaces the host branch `if repellency_dict.get("is_negation")` + add_noise, ...threshold_time.py:558-569. */
int sdn_renoise_select(float* latents, const float* x0r, const float* noise, const int32_t* isneg,
                       int32_t n_prompt, int64_t d, float sqrt_ac, float sqrt_one_minus_ac, void* stream);

/* Flow-matching Euler step (SD-v3): prev = x + (sigma_next - sigma) * v, computed in fp32.
 * Replaces FlowMatchEulerDiscreteScheduler.step at models/sdv3/safe_denoiser_pipeline.py:1165. */
int sdn_flow_euler_step(const float* x, const float* v, int64_t n, float sigma, float sigma_next,
                        float* prev, void* stream);

/* Flow-matching repellency re-noise (models/sdv3/safe_denoiser_pipeline.py:1142-1161):
 *   x0 = x - sigma*v ; x1 = x + (1-sigma)*v        (sdn_flow_endpoints)
 *   noise = sqrt(sigma_next)*x1 + sqrt(1-sigma_next)*z ; out = x0r + sigma_next*(noise - x0r)   (sdn_flow_renoise)
 * (sigma - delta == sigma_next with delta = sigma - sigma_next.) */
int sdn_flow_endpoints(const float* x, const float* v, int64_t n, float sigma, float* x0, float* x1, void* stream);
int sdn_flow_renoise(const float* x0r, const float* x1, const float* z, int64_t n, float sigma_next,
                     float* out, void* stream);

/* ===================================================================================== *
 *  Denoiser-network operators (rows U1-U6) -- bf16 storage, fp32 accumulation.
 *  Activations are NHWC bf16: a [B,H,W,C] feature map is the [B*H*W, C] token matrix.
 *  The reference runs these through diffusers 0.29.0 (third-party; wiring spec vendored at
 *  models/unet.py:683-932, models/unet_2d_blocks.py, models/transformer_2d.py:239-359).
 * ===================================================================================== */

#define SDN_A_PLAIN    0   /* A is [M, K] row-major (optionally split in two sources at K1)   */
#define SDN_A_CONV3X3  1   /* A is the implicit im2col of an NHWC map: 3x3, pad 1             */
#define SDN_ACT_NONE   0
#define SDN_ACT_SILU   1
#define SDN_ACT_GEGLU  2   /* W rows interleaved value/gate in blocks of 16; out = v*gelu(g)  */
#define SDN_ACT_GELU_TANH 3 /* GELU(approximate="tanh") -- MMDiT feed-forward                   */
#define SDN_ACT_QUICK_GELU 4 /* x * sigmoid(1.702 x) -- CLIP text encoder MLP                     */
#define SDN_OUT_BF16      0   /* [M, ldc] bf16                                                */
#define SDN_OUT_F32       1   /* [M, ldc] f32                                                 */
#define SDN_OUT_F32_NCHW  2   /* [B, n_valid, rows_per_batch] f32 (conv_out -> latent layout) */

typedef struct sdn_gemm_desc {
  int32_t M, N, K;          /* C[M,N] = A[M,K] . W[N,K]^T ; K % 64 == 0 ; N % 32 == 0           */
  int32_t a_mode;           /* SDN_A_*                                                         */
  int32_t K1;               /* PLAIN: columns [0,K1) come from `a` (ld K1), [K1,K) from `a2`
                               (ld K-K1): the skip-concat of the up blocks, never materialised;
                               0 or K = single source                                         */
  int32_t Hs, Ws, Cin;      /* CONV3X3: stored input map [B,Hs,Ws,Cin], K = 9*Cin             */
  int32_t Ho, Wo, stride;   /*          output map, stride 1|2 (Downsample2D)                  */
  int32_t upsample;         /*          1 = nearest-2x of the stored map first (Upsample2D)    */
  int32_t act;              /* SDN_ACT_*                                                       */
  int32_t out_kind;         /* SDN_OUT_*                                                       */
  int32_t rows_per_batch;   /* rows of one sample (H*W): row -> sample for rowbias / NCHW      */
  int32_t ld_rowbias;       /* leading dimension of rowbias [B, ld_rowbias]                    */
  int32_t ld_rowgate;       /* leading dimension of rowgate [B, ld_rowgate]                    */
  int32_t residual_bcast;   /* 1 = residual is [rows_per_batch, N], shared by every sample     */
  int32_t n_valid;          /* columns actually stored (0 = N); W is zero-padded to N rows     */
  int32_t ldc;              /* leading dimension of out/residual (0 = natural)                 */
  int32_t asym_pad;         /* CONV3X3, stride 2 only: 1 = zero padding (0,1,0,1) (right/bottom only: the VAE encoder's
                               Downsample2D(padding=0) + F.pad) instead of 1 on every side             */
  int32_t split_k;          /* number of k-loop slices for sdn_gemm_splitk_* (0 / 1 = none); plain sdn_gemm_* rejects > 1 */
  int32_t res_pre;          /* 1 = add the 16-bit residual into the accumulators BEFORE the k loop instead of in the epilogue
                               (plain 16-bit output, no gate / activation / split-K): the sum is formed in a different order
                               (bias + residual first), so the last bit of an output may differ from res_pre = 0           */
  int32_t x3_out;           /* sdn_gemm_bf16 inside the bf16x3 plan (see "bf16x3 by operand expansion" below): 0 = off;
                               1 = out is F32 [M, ldc]; 2 = GEGLU, out is a bf16 triple [M, 3 ldc]; 3 = out is a bf16 triple;
                               4 = out is a bf16 hi | lo PAIR [M, 2 ldc] (the operand form of sdn_attention_x3_pairs).
                               With any of them `residual` is F32 [M, ldc] and out_kind is ignored.
                               5 = EXPERIMENTAL (round 5, sdn_gemm_f16 only, plain A, no activation): the "h8" operand form -- rows of
                               4 bytes per element, A' = [fp16(a) | e4m3(2^11 (a - fp16(a))) | e4m3(a)], W' = [fp16(w) | e4m3(w) |
                               e4m3(2^11 (w - fp16(w)))] (OCP e4m3), K = 2 x the logical K (the row length in 16-bit units; logical
                               K % 128 == 0), out = F32 [M, ldc]: an fp16 main term plus two correction products on the scaled fp8
                               matrix instruction, f32 accumulation throughout (1.2e-5 ... 2.2e-5 from float64 per
                               GEMM against fp16's 2.9e-4 and bf16x3's 4.5e-6).  256-row tiles only (N % 256 == 0 or N % 320 == 0
                               with >= 192 tiles); not used by any plan yet                                                   */
} sdn_gemm_desc;

/* out = act((A.W^T + bias[n] + rowbias[b(m), n]) * rowgate[b(m), n] + residual[m, n])   (rowgate NULL = 1).
 * Replaces F.linear / conv2d(3x3 | 1x1) + the adds around them: ResnetBlock2D conv1 (+ time_emb_proj
 * broadcast), conv2 (+ shortcut), Downsample2D, Upsample2D, Transformer2DModel proj_in/proj_out
 * (models/transformer_2d.py:810-858), Attention to_q/k/v/out, FeedForward GEGLU (models/transformer_2d.py:335-355). */
int sdn_gemm_bf16(const sdn_gemm_desc* d_host, const void* a, const void* a2, const void* w,
                  const float* bias, const float* rowbias, const float* rowgate, const void* residual, void* out,
                  void* stream);
/* Same operator with IEEE fp16 storage (the reference's SD-v3 dtype; 8x tighter parity than bf16). */
int sdn_gemm_f16(const sdn_gemm_desc* d_host, const void* a, const void* a2, const void* w,
                 const float* bias, const float* rowbias, const float* rowgate, const void* residual, void* out,
                 void* stream);

/* GroupNorm (+ optional SiLU) over an NHWC bf16 map, optionally over the channel-concat of two maps
 * (x [B,HW,C1] ++ x2 [B,HW,C2]) written as ONE normalised map [B,HW,C1+C2].
 * stats_ws: B*129*groups*2 floats of scratch (row-tile partials + final mean/rstd; deterministic reduction).  Replaces GroupNorm+SiLU of ResnetBlock2D (eps 1e-5) and the
 * GroupNorm of Transformer2DModel (eps 1e-6, no SiLU; models/transformer_2d.py:506-512). */
int sdn_groupnorm_bf16(const void* x, const void* x2, int32_t batch, int32_t hw, int32_t c1, int32_t c2,
                       int32_t groups, float eps, int32_t silu, const float* gamma, const float* beta,
                       void* out, float* stats_ws, void* stream);
int sdn_groupnorm_f16(const void* x, const void* x2, int32_t batch, int32_t hw, int32_t c1, int32_t c2,
                      int32_t groups, float eps, int32_t silu, const float* gamma, const float* beta,
                      void* out, float* stats_ws, void* stream);

/* LayerNorm over the last axis of [rows, C] bf16 (eps 1e-5, affine) -- BasicTransformerBlock norm1/2/3
 * (models/transformer_2d.py:265,305,335). */
int sdn_layernorm_bf16(const void* x, int64_t rows, int32_t c, float eps, const float* gamma,
                       const float* beta, void* out, void* stream);
int sdn_layernorm_f16(const void* x, int64_t rows, int32_t c, float eps, const float* gamma,
                      const float* beta, void* out, void* stream);

/* softmax(Q K^T * scale) V per (batch, head); flash-style, never materialises the score matrix.
 *   q [B, Nq, ldq] (head h at columns h*d .. h*d+d), k/v [B, Nk, ldk/ldv] likewise, out [B, Nq, ldo].
 * d in {40, 80, 160} (SD-v1.4: 8 heads at C = 320/640/1280) or 64 (MMDiT).
 * Replaces F.scaled_dot_product_attention in diffusers' AttnProcessor2_0 (imported at
 * models/unet_2d_blocks.py:24; called from BasicTransformerBlock, models/transformer_2d.py:284-328). */
int sdn_attention_bf16(const void* q, const void* k, const void* v, void* out, int32_t batch, int32_t heads,
                       int32_t nq, int32_t nk, int32_t head_dim, int32_t ldq, int32_t ldk, int32_t ldv,
                       int32_t ldo, float scale, void* stream);
int sdn_attention_f16(const void* q, const void* k, const void* v, void* out, int32_t batch, int32_t heads,
                      int32_t nq, int32_t nk, int32_t head_dim, int32_t ldq, int32_t ldk, int32_t ldv,
                      int32_t ldo, float scale, void* stream);

/* Joint attention over TWO token streams kept in separate buffers (MMDiT: image tokens then text tokens):
 * rows [0, n1) of the sequence come from q/k/v/out ([B, n1, ld*]), rows [n1, n_total) from the *2 pointers
 * ([B, n_total - n1, ld*2]); the concatenated sequence is never materialised.  dtype 0 = bf16, 1 = f16.
 * Replaces JointAttnProcessor2_0 (diffusers 0.29.0; reached through self.transformer(...),
 * models/sdv3/safe_denoiser_pipeline.py:1120-1127). */
typedef struct sdn_attn_segment2 {
  const void* q2; const void* k2; const void* v2; void* out2;
  int32_t n1, ldq2, ldk2, ldv2, ldo2;
} sdn_attn_segment2;
int sdn_joint_attention(int32_t dtype, const void* q, const void* k, const void* v, void* out,
                        const sdn_attn_segment2* seg2_host, int32_t batch, int32_t heads, int32_t n_total,
                        int32_t head_dim, int32_t ldq, int32_t ldk, int32_t ldv, int32_t ldo, float scale,
                        void* stream);

/* adaLN layer norm: out = LN(x) * (1 + scale[b]) + shift[b]  (no affine; scale/shift are per-sample fp32 rows of
 * leading dimension ld_mod; sample of a row = row / rows_per_batch).  AdaLayerNormZero / AdaLayerNormContinuous of
 * the MMDiT blocks (diffusers 0.29.0). */
int sdn_layernorm_mod_bf16(const void* x, int64_t rows, int32_t c, float eps, const float* scale, const float* shift,
                           int32_t ld_mod, int32_t rows_per_batch, void* out, void* stream);
int sdn_layernorm_mod_f16(const void* x, int64_t rows, int32_t c, float eps, const float* scale, const float* shift,
                          int32_t ld_mod, int32_t rows_per_batch, void* out, void* stream);

/* MMDiT PatchEmbed front end: fp32 NCHW latent -> 16-bit [B*(H/p)*(W/p), C*p*p] (column order (c,py,px)), so the
 * patch embedding conv (k = s = p) is one GEMM; and the back end: fp32 tokens [.., p*p*C] -> fp32 NCHW latent. */
int sdn_patchify_bf16(const float* latents, int32_t batch, int32_t c, int32_t h, int32_t w, int32_t p, void* out, void* stream);
int sdn_patchify_f16(const float* latents, int32_t batch, int32_t c, int32_t h, int32_t w, int32_t p, void* out, void* stream);
int sdn_unpatchify_f32(const float* tokens, int32_t batch, int32_t c, int32_t h, int32_t w, int32_t p, float* out, void* stream);

/* conv_in: 3x3 conv of the fp32 NCHW latent [B,Cin<=16,H,W] into an NHWC bf16 map [B,H,W,Cout]
 * (models/unet.py:840).  w is [Cout][3][3][Cin] bf16, bias f32. */
int sdn_conv_in_bf16(const float* latents_nchw, const void* w, const float* bias, int32_t batch, int32_t cin,
                     int32_t h, int32_t wd, int32_t cout, void* out_nhwc, void* stream);
int sdn_conv_in_f16(const float* latents_nchw, const void* w, const float* bias, int32_t batch, int32_t cin,
                    int32_t h, int32_t wd, int32_t cout, void* out_nhwc, void* stream);

/* Sinusoidal timestep features, flip_sin_to_cos, shift 0: out[b] = [cos(t f_k) | sin(t f_k)], f_k =
 * exp(-ln(1e4) k / half) -> bf16 [B, dim]  (Timesteps(320), models/unet.py:764-786). */
int sdn_timestep_embed_bf16(float timestep, int32_t batch, int32_t dim, void* out, void* stream);
int sdn_timestep_embed_f16(float timestep, int32_t batch, int32_t dim, void* out, void* stream);

/* ---- fp32 storage forms of the operators above (dtype 2 of sdn_unet_config: the plan's PRECISION mode) ----
 * Same argument meaning; every activation / weight pointer is f32 and SDN_OUT_BF16 means "f32 [M, ldc]".  Contractions
 * run on the f32-input matrix cores (v_mfma_f32_16x16x4_f32: exact f32 products and sums), 1/16 of the 16-bit rate.
 * They exist so the SAME launch plan can be compared with the reference's fp32 arithmetic (run_nudity.py:277 loads the
 * pipeline with torch_dtype=float32) to ~1e-6 per forward at full size; 16-bit storage cannot do better than 1e-3
 * (fp16) / 1e-2 (bf16) in any implementation.  The LayerNorm-folded, column-statistics and split-K forms have no f32
 * counterpart: an f32 plan uses the plain operator chain.
 * sdn_groupnorm_f32's `stats_ws` is NOT optional scratch any more (round 4): a non-null, 8-byte-aligned pointer selects the
 * row-major two-pass form, which WRITES batch * nchunk * groups * 2 DOUBLES there (nchunk <= 64): the full
 * batch * 129 * groups * 2 floats documented for sdn_groupnorm_bf16 must be available.  Pass NULL for the per-group kernels,
 * which need none. */
int sdn_gemm_f32(const sdn_gemm_desc* d_host, const void* a, const void* a2, const void* w,
                 const float* bias, const float* rowbias, const float* rowgate, const void* residual, void* out,
                 void* stream);
int sdn_groupnorm_f32(const void* x, const void* x2, int32_t batch, int32_t hw, int32_t c1, int32_t c2,
                      int32_t groups, float eps, int32_t silu, const float* gamma, const float* beta,
                      void* out, float* stats_ws, void* stream);
int sdn_layernorm_f32(const void* x, int64_t rows, int32_t c, float eps, const float* gamma,
                      const float* beta, void* out, void* stream);
int sdn_attention_f32(const void* q, const void* k, const void* v, void* out, int32_t batch, int32_t heads,
                      int32_t nq, int32_t nk, int32_t head_dim, int32_t ldq, int32_t ldk, int32_t ldv,
                      int32_t ldo, float scale, void* stream);
int sdn_conv_in_f32(const float* latents_nchw, const void* w, const float* bias, int32_t batch, int32_t cin,
                    int32_t h, int32_t wd, int32_t cout, void* out_nhwc, void* stream);
int sdn_timestep_embed_f32(float timestep, int32_t batch, int32_t dim, void* out, void* stream);

/* ---- bf16x3 contractions on fp32 storage (dtype 3 of sdn_unet_config: the ACCURATE THROUGHPUT mode) ----
 * Same operators, arguments and f32 storage as sdn_gemm_f32 / sdn_attention_f32; every operand element is split on the
 * fly into bf16 hi + lo (16 mantissa bits) and each product runs as hi.hi + hi.lo + lo.hi on the bf16 matrix cores
 * (3 x v_mfma_f32_16x16x32_bf16 per 16x16x32 block, f32 accumulation) instead of 8 f32-input MFMAs.  Norms, softmax,
 * epilogues and storage stay f32.  Distance from the reference's fp32 arithmetic (run_nudity.py:277): ~1.5e-5 per UNet
 * forward (1.9e-5 measured in the engine), 5.3e-5 / 5.5e-5 over the 10- / 50-step loop (engine, profiles/round3_parity.json; the
 * emulation gave 3.4e-5) -- inside the north star's 1e-3, which a single 16-bit rounding of the MFMA
 * operands cannot meet (2.96e-3 fp16 / 2.3e-2 bf16: profiles/round3_precision_ablation.md). */
int sdn_gemm_x3(const sdn_gemm_desc* d_host, const void* a, const void* a2, const void* w,
                const float* bias, const float* rowbias, const float* rowgate, const void* residual, void* out,
                void* stream);
int sdn_attention_x3(const void* q, const void* k, const void* v, void* out, int32_t batch, int32_t heads,
                     int32_t nq, int32_t nk, int32_t head_dim, int32_t ldq, int32_t ldk, int32_t ldv,
                     int32_t ldo, float scale, void* stream);

/* ---- bf16x3 by operand expansion (round 4): the same three-term products as sdn_gemm_x3, on the LDS-DMA tiles of sdn_gemm_bf16.
 * A tensor x [rows, C] that a GEMM will read is kept as the bf16 TRIPLE [rows, 3C] = [hi(C) | lo(C) | hi(C)], hi = bf16(x),
 * lo = bf16(x - hi) (x = hi + lo to 2^-17); its weight W [N, K] as [N, 3K] with every K-group g (K itself, or the Cin of one
 * conv tap) expanded to [hi(g) | hi(g) | lo(g)].  Then  A' . W'^T = a_hi w_hi + a_lo w_hi + a_hi w_lo  is ONE bf16 GEMM with three
 * times the k loop -- both operands arrive by LDS-DMA, no staging registers, no split arithmetic inside the loop (sdn_gemm_x3
 * splits f32 operands on the fly and is bound by moving them through registers into LDS: DESIGN.md).  sdn_gemm_bf16 with
 * sdn_gemm_desc.x3_out != 0 (K, Cin = the EXPANDED sizes) adds an F32 residual and writes F32 rows or the next GEMM's triple.
 *   sdn_split3: f32 [rows, c1] (++ f32 [rows, c2]) -> triple [rows, 3 (c1 + c2)]  (raw residual-stream tensors, text states)
 *   sdn_expand3_weights: f32 W [rows, cols] -> bf16 [rows, 3 cols], group = cols, or the Cin of a [O][ky][kx][Cin] conv weight
 *   sdn_groupnorm_f32 / sdn_layernorm_f32 / sdn_attention_x3 with `triple_out` write their result in that form directly. */
int sdn_split3(const float* x, const float* x2, int64_t rows, int32_t c1, int32_t c2, void* out_triple, void* stream);
int sdn_expand3_weights(const float* w, int64_t rows, int32_t cols, int32_t group, void* out_bf16, void* stream);
int sdn_groupnorm_f32_triple(const void* x, const void* x2, int32_t batch, int32_t hw, int32_t c1, int32_t c2, int32_t groups,
                             float eps, int32_t silu, const float* gamma, const float* beta, void* out_triple, float* stats_ws,
                             void* stream);
int sdn_layernorm_f32_triple(const void* x, int64_t rows, int32_t c, float eps, const float* gamma, const float* beta,
                             void* out_triple, void* stream);
int sdn_attention_x3_triple(const void* q, const void* k, const void* v, void* out_triple, int32_t batch, int32_t heads, int32_t nq,
                            int32_t nk, int32_t head_dim, int32_t ldq, int32_t ldk, int32_t ldv, int32_t ldo, float scale,
                            void* stream);
/* Attention of the bf16x3 plan on PRE-SPLIT operands (replaces Attention.forward's softmax(Q K^T) V of models/transformer_2d.py
 * in that mode, self- and cross-attention): q / k / v point at the hi planes of hi | lo PAIR rows written by a projection run with
 * x3_out = 4; the lo plane of a Q row lies lo_offset_q elements on, that of a K / V row lo_offset_kv elements on
 * (self-attention over the qkv projection [hi(3C) | lo(3C)]: ld = 6 C, both offsets 3 C; cross-attention: q from [hi(C) | lo(C)],
 * k / v from the text projection [hi(2C) | lo(2C)]).  Every product is the three-term bf16 sum of sdn_attention_x3, but K / V
 * reach LDS by LDS-DMA instead of being split per tile per workgroup.  head_dim 40, 80 or 160.
 * triple_out = 0: out is F32 [batch, nq, ldo]; 1: out is the bf16 triple [batch, nq, 3 ldo]. */
int sdn_attention_x3_pairs(const void* q, const void* k, const void* v, int32_t lo_offset_q, int32_t lo_offset_kv, void* out,
                           int32_t batch, int32_t heads, int32_t nq, int32_t nk, int32_t head_dim, int32_t ldq, int32_t ldk,
                           int32_t ldv, int32_t ldo, float scale, int32_t triple_out, void* stream);

/* ---- random draws (row S2): P per-prompt generators in one launch ------------------------------------------------
 * Replaces the per-prompt `torch.randn(latents_shape, generator=gen)` calls behind prepare_latents / scheduler.step /
 * add_noise (...threshold_time.py:494-503,554,565-567,576; run_nudity.py:142,448) bit for bit: Philox4x32-10 seeded
 * (seed, subsequence = thread index, offset = the generator's philox offset), Box-Muller normal4, torch's launch geometry.
 * out[rows[p] or p, 0..numel) is what `torch.randn(numel, generator=g_p)` would return; afterwards advance generator p by
 * the increment sdn_randn_philox_plan reports (torch.Generator.set_offset).  seeds / offsets / rows: device arrays. */
int sdn_randn_philox_plan(int64_t numel, int32_t* grid_out, int64_t* offset_increment_out);
int sdn_randn_philox(const uint64_t* seeds, const uint64_t* offsets, const int32_t* rows, int32_t n_gen, int64_t numel,
                     float* out, void* stream);
/* The same draws with the (seed, offset) pairs kept ON THE DEVICE for the whole call: row p is drawn when flags == NULL or
 * flags[p] != 0 (the loop's device-side is_negation vector selects the conditional re-noise draws of ...threshold_time.py:
 * 565-567 without a host-built index list), then offsets[p] += increment on the device.  out == NULL advances only (the
 * discarded variance draw of the x0 probe's scheduler.step, :554).  No host <-> device copy per draw. */
int sdn_randn_philox_state(const uint64_t* seeds, uint64_t* offsets, const int32_t* flags, int32_t n_gen, int64_t numel,
                           float* out, void* stream);

/* ---- whole-network entry: SD-v1.4-family UNet2DConditionModel forward -------------------------- */
typedef struct sdn_unet_config {
  int32_t in_channels, out_channels, sample_size;      /* 4, 4, 64                                  */
  int32_t n_levels;                                    /* 4                                         */
  int32_t block_out_channels[4];                       /* 320, 640, 1280, 1280                      */
  int32_t level_has_attn[4];                           /* 1,1,1,0 (CrossAttnDown x3, DownBlock2D)   */
  int32_t layers_per_block;                            /* 2                                         */
  int32_t n_heads;                                     /* 8 (config key attention_head_dim, legacy) */
  int32_t cross_dim, text_len;                         /* 768, 77                                   */
  int32_t norm_groups;                                 /* 32                                        */
  int32_t dtype;                                       /* 0 = bf16 storage, 1 = fp16 storage, 2 = fp32
                                                          storage (precision mode: weights, text and every
                                                          activation f32; see sdn_gemm_f32), 3 = fp32
                                                          storage with bf16x3 contractions (sdn_gemm_x3) */
  int32_t latent_repeat;                               /* 0/1 = off.  r > 1: the batch is r guidance branches of the
                                                          SAME latents (`torch.cat([latents] * r)`, ...threshold_time.py
                                                          :535): sdn_unet_forward then takes latents [B / r, ...] and
                                                          repeats them itself, computing everything up to the first
                                                          cross-attention (conv_in, resnet 0, the first self-attention)
                                                          ONCE per latent instead of r times.  Text rows stay [B]:
                                                          branch-major, row b and row b + B/r share a latent.  Results
                                                          are bit-identical to the plain plan on repeated latents.   */
} sdn_unet_config;

typedef struct sdn_unet sdn_unet;   /* opaque: op plan + parameter manifest (host memory only) */

/* parameter kinds = how a diffusers state_dict tensor is laid out in the packed weight buffer */
#define SDN_P_VEC_F32     0   /* 1-D -> f32                                                    */
#define SDN_P_MAT         1   /* [out,in] or [out,in,1,1] -> bf16 [out][in]                    */
#define SDN_P_CONV3X3     2   /* [out,in,3,3] -> bf16 [out][ky][kx][in], rows zero-padded      */
#define SDN_P_GEGLU_MAT   3   /* [2F,in] -> bf16, rows interleaved value/gate in blocks of 16  */
#define SDN_P_GEGLU_VEC   4   /* [2F] -> f32, interleaved the same way                         */
#define SDN_P_POS_CROP    5   /* [1, max*max, C] -> 16-bit [h*w, C], centre crop (MMDiT pos_embed) */
#define SDN_P_DERIVED     6   /* not a state_dict tensor: `rows` BYTES the engine fills itself from other entries
                                 (LayerNorm-folded weights); loaders skip it and call sdn_unet_prepare afterwards */

typedef struct sdn_param_info {
  char     name[128];       /* diffusers state_dict key                                          */
  int32_t  kind;            /* SDN_P_*                                                            */
  int32_t  rows, cols;      /* source logical matrix (rows = out features; cols = in*kh*kw)      */
  int32_t  rows_padded;     /* rows reserved in the packed buffer (zero filled beyond `rows`)    */
  int64_t  offset;          /* byte offset in the packed weight buffer                           */
} sdn_param_info;

int    sdn_unet_create(const sdn_unet_config* cfg_host, sdn_unet** out_host);
void   sdn_unet_destroy(sdn_unet* u);
int    sdn_unet_param_count(const sdn_unet* u);
int    sdn_unet_param_info(const sdn_unet* u, int32_t index, sdn_param_info* info_host);
size_t sdn_unet_weight_bytes(const sdn_unet* u);
size_t sdn_unet_workspace_bytes(sdn_unet* u, int32_t batch);
/* Total algorithmic FLOPs of one forward at this batch (2*M*N*K of every GEMM + attention cores), and the
 * attention-core share -- the numerators of the MFMA roofline. */
double sdn_unet_flops(sdn_unet* u, int32_t batch, double* attention_core_flops_host);

/* eps = UNet(latents, t, text):  latents [B, in_ch, S, S] fp32 NCHW, text [B, text_len, cross_dim] in the plan's 16-bit dtype,
 * out [B, out_ch, S, S] fp32 NCHW.  One timestep for the whole batch (the reference passes a scalar t,
 * ...threshold_time.py:538).  Replaces self.unet(latent_model_input, t, encoder_hidden_states=E).sample. */
int sdn_unet_forward(sdn_unet* u, const void* weights, const float* latents, float timestep, const void* text,
                     float* out, int32_t batch, void* workspace, size_t workspace_bytes, void* stream);

/* ---- SD-v3 MMDiT (SD3Transformer2DModel, diffusers 0.29.0) -- same opaque handle type and the same
 * param / workspace / flops / profile / destroy entry points as the UNet --------------------------------- */
typedef struct sdn_mmdit_config {
  int32_t in_channels, out_channels;     /* 16, 16                                                     */
  int32_t sample_size, patch_size;       /* latent side (64 for 512x512 images, 128 for 1024x1024), 2  */
  int32_t num_layers;                    /* 24                                                         */
  int32_t num_heads, head_dim;           /* 24, 64                                                     */
  int32_t joint_dim, pooled_dim;         /* 4096, 2048                                                 */
  int32_t text_len;                      /* 333 = 77 CLIP + 256 T5 tokens                              */
  int32_t time_dim;                      /* 256 sinusoidal features                                    */
  int32_t dtype;                         /* 0 = bf16, 1 = fp16 (the reference runs SD-v3 in fp16)      */
} sdn_mmdit_config;
int sdn_mmdit_create(const sdn_mmdit_config* cfg_host, sdn_unet** out_host);
/* v = transformer(latents [B,16,S,S] fp32, t, text [B,text_len,joint_dim] 16-bit, pooled [B,pooled_dim] 16-bit)
 * -> out [B,16,S,S] fp32.  Replaces self.transformer(...)[0], models/sdv3/safe_denoiser_pipeline.py:1120-1127. */
int sdn_mmdit_forward(sdn_unet* m, const void* weights, const float* latents, float timestep, const void* text,
                      const void* pooled, float* out, int32_t batch, void* workspace, size_t workspace_bytes,
                      void* stream);

/* LayerNorm folded into the GEMM that consumes it (BasicTransformerBlock: norm1 -> to_q/k/v, norm2 -> attn2.to_q, norm3 ->
 * GEGLU proj; models/transformer_2d.py:239-359): out = LayerNorm_eps(A; gamma, beta) . W^T + bias [GEGLU], computed as
 * rstd[m] * (A . W'^T - mean[m] * c[n]) + d[n] with W' = W * gamma (per input channel, 16-bit), c[n] = sum_k W'[n,k],
 * d[n] = sum_k beta[k] W[n,k] + bias[n] -- the three produced ONCE per weight set by sdn_ln_fold (bias may be NULL; GEGLU
 * weights / biases in their interleaved layout).  The kernel takes the row sums from the A fragments it feeds the MFMAs
 * anyway (row_stats NULL; meant for narrow N -- every n-tile repeats that work), or reads (mean, rstd) per row from
 * row_stats [M][2], written by the read-only pre-pass sdn_row_stats_* (wide N); the normalised activation never exists in
 * HBM.  16-bit output, act NONE or GEGLU; N a multiple of 160 or 64 without row_stats, of 160 / 128 / 64 with. */
int sdn_ln_fold(int32_t dtype, const void* w, const float* gamma, const float* beta, const float* bias, int32_t rows,
                int32_t cols, void* w_folded, float* c, float* d, void* stream);
int sdn_gemm_ln_bf16(const sdn_gemm_desc* desc, const void* a, const void* w_folded, const float* c, const float* d, float eps,
                     const float* row_stats, void* out, void* stream);
int sdn_gemm_ln_f16(const sdn_gemm_desc* desc, const void* a, const void* w_folded, const float* c, const float* d, float eps,
                    const float* row_stats, void* out, void* stream);
/* out[m] = (mean, 1 / sqrt(var + eps)) of row m of the 16-bit [rows, c] matrix x */
int sdn_row_stats_bf16(const void* x, int64_t rows, int32_t c, float eps, float* out, void* stream);
int sdn_row_stats_f16(const void* x, int64_t rows, int32_t c, float eps, float* out, void* stream);

/* GEGLU feed-forward of a BasicTransformerBlock contracted with the block's proj_out, in ONE launch (models/transformer_2d.py:
 * 335-355 norm3 -> ff -> residual, then :845-858 proj_out + the Transformer2DModel residual):
 *   out = residual + [ GEGLU(LayerNorm(x) W1^T + b1) | x ] . w_cat^T + b_cat
 * x [M, C] 16-bit (the block's hidden state after cross-attention), row_stats [M][2] from sdn_row_stats_* -- or NULL: the kernel
 * then takes the statistics from the operand fragments of its first chunk (the arithmetic of sdn_gemm_ln_* with row_stats = NULL,
 * eps = 1e-5) and no pass over x is needed --, w1_folded [8C, C] /
 * c1 / d1 [8C] from sdn_ln_fold over the value / gate-interleaved GEGLU weight, w_cat [C, 5C] = [Wpo W2 | Wpo] and
 * b_cat [C] = Wpo b2 + bpo (formed once per weight set by sdn_unet_prepare), residual / out [M, C] 16-bit; col_stats nullable,
 * [ceil(M / 128)][C][2] as sdn_gemm_stats_*.  The [M, 4C] hidden activation stays in LDS.  Bit-identical to sdn_gemm_ln_*
 * (act GEGLU, the same row_stats argument) followed by the two-source sdn_gemm_* it replaces.  dtype 0 = bf16, 1 = fp16; only C == 320 is instantiated
 * (SDN_E_INVALID otherwise: the caller keeps the two-launch form). */
int sdn_ffn_geglu_fused(int32_t dtype, int64_t M, int32_t C, const void* x, const float* row_stats, const void* w1_folded,
                        const float* c1, const float* d1, const void* w_cat, const float* b_cat, const void* residual, void* out,
                        float* col_stats, void* stream);

/* GroupNorm statistics without a pass over the tensor: the GEMM that PRODUCES a GroupNorm input also emits, per block of
 * 128 output rows and per column, the (sum, sum of squares) of the 16-bit values it stores -- col_stats [ceil(M/128)][N][2]
 * f32 (sdn_gemm_stats_*: 16-bit output, n_valid == N, no GEGLU, no rowgate) -- and sdn_groupnorm_cols_* reduces those
 * instead of reading x (hw % 128 == 0; x2 / cols2 = the second tensor of a channel concat, each with its own partials). */
int sdn_gemm_stats_bf16(const sdn_gemm_desc* desc, const void* a, const void* a2, const void* w, const float* bias,
                        const float* rowbias, const void* residual, void* out, float* col_stats, void* stream);
int sdn_gemm_stats_f16(const sdn_gemm_desc* desc, const void* a, const void* a2, const void* w, const float* bias,
                       const float* rowbias, const void* residual, void* out, float* col_stats, void* stream);
int sdn_groupnorm_cols_bf16(const void* x, const void* x2, int32_t batch, int32_t hw, int32_t c1, int32_t c2, int32_t groups,
                            float eps, int32_t silu, const float* gamma, const float* beta, void* out, float* stats_ws,
                            const float* cols1, const float* cols2, void* stream);
int sdn_groupnorm_cols_f16(const void* x, const void* x2, int32_t batch, int32_t hw, int32_t c1, int32_t c2, int32_t groups,
                           float eps, int32_t silu, const float* gamma, const float* beta, void* out, float* stats_ws,
                           const float* cols1, const float* cols2, void* stream);

/* Split-K form for small M / long K (one-prompt batches: 8-20 tiles for 256 CUs): the k loop is cut into desc->split_k
 * slices, each writes an fp32 partial [M, N] into `partials` (>= split_k * M * N * 4 bytes), and a second kernel sums
 * them in a fixed order and applies the epilogue.  16-bit outputs, any activation but GEGLU, n_valid == N. */
int sdn_gemm_splitk_bf16(const sdn_gemm_desc* d, const void* a, const void* a2, const void* w, const float* bias,
                         const float* rowbias, const float* rowgate, const void* residual, void* out, void* partials,
                         size_t partial_bytes, void* stream);
int sdn_gemm_splitk_f16(const sdn_gemm_desc* d, const void* a, const void* a2, const void* w, const float* bias,
                        const float* rowbias, const float* rowgate, const void* residual, void* out, void* partials,
                        size_t partial_bytes, void* stream);

/* out[k * bytes + i] = in[i], k < rep (device-side `torch.cat([x] * rep)`; bytes % 16 == 0) */
int sdn_repeat(const void* in, size_t bytes, int32_t rep, void* out, void* stream);

/* Fills the SDN_P_DERIVED regions of a freshly uploaded weight buffer (LayerNorm-folded projections: sdn_ln_fold over the
 * packed tensors).  Call once after every upload / change of the weights, on the stream that will run the forwards. */
int sdn_unet_prepare(sdn_unet* u, void* weights, void* stream);

/* ---- AutoencoderKL decoder (the "next" row after the denoising loop) ------------------------------
 * Replaces `self.vae.decode(latents / scaling_factor)` inside StableDiffusionPipeline.decode_latents, called at
 * models/textuals_visual/modified_safree_diffusion_pipeline_threshold_time.py:589 (and :539-545 of the SLD
 * pipeline).  The handle is an sdn_unet: manifest, weight and workspace queries are the sdn_unet_* ones; the
 * manifest uses the diffusers AutoencoderKL keys `post_quant_conv.*` and `decoder.*` (attention: to_q/to_k/to_v/
 * to_out.0).  The encoder half (proj_ref builder) is not part of this row yet. */
typedef struct sdn_vae_config {
  int32_t latent_channels, out_channels; /* 4, 3                                                       */
  int32_t sample_size;                   /* LATENT side: 64 -> 512 x 512 images                        */
  int32_t n_levels;                      /* 4                                                          */
  int32_t block_out_channels[4];         /* 128, 256, 512, 512 (multiples of 64)                       */
  int32_t layers_per_block;              /* 2 (the decoder's up blocks run layers_per_block + 1 resnets) */
  int32_t norm_groups;                   /* 32                                                         */
  int32_t dtype;                         /* 0 = bf16, 1 = fp16 activations / weights                   */
} sdn_vae_config;
int sdn_vae_decoder_create(const sdn_vae_config* cfg_host, sdn_unet** out_host);
/* image [B, out_channels, 8S.., 8S..] fp32 NCHW (the decoder's raw output, nominally in [-1, 1]) =
 * decoder(post_quant_conv(latent_scale * latents [B, latent_channels, S, S] fp32)).  latent_scale = 1 / scaling_factor
 * folds the first line of decode_latents.  Any batch: byte offsets inside one activation are 32-bit, so the entry
 * point replays the plan on chunks of at most 8 images (fewer at 1024 x 1024) against the same workspace, stream-ordered;
 * sdn_unet_workspace_bytes(vae, batch) returns the size of the largest chunk's workspace. */
int sdn_vae_decode(sdn_unet* vae, const void* weights, const float* latents, float latent_scale, float* image,
                   int32_t batch, void* workspace, size_t workspace_bytes, void* stream);
/* Encoder half: the embed_fn of the proj_ref builder, `pipe.vae.encode(x).latent_dist.sample() * scaling_factor`
 * (run_nudity.py:308, consumed by RepellencyMethod.project, repellency/repellency_methods_threshold.py:54-72).
 * Same config struct (out_channels = image channels); manifest keys `encoder.*`, `quant_conv.*`.
 * moments [B, 2*latent_channels, S, S] fp32 NCHW = quant_conv(encoder(image [B, 3, H, W] fp32 NCHW)) = (mean | logvar). */
int sdn_vae_encoder_create(const sdn_vae_config* cfg_host, sdn_unet** out_host);
int sdn_vae_encode(sdn_unet* vae_encoder, const void* weights, const float* image, float* moments, int32_t batch,
                   void* workspace, size_t workspace_bytes, void* stream);
/* DiagonalGaussianDistribution: out [B, L, hw] = scale * (mean + exp(0.5 * clamp(logvar, -30, 20)) * noise);
 * noise == NULL gives scale * mean (`.mode()`). */
int sdn_gaussian_sample(const float* moments, const float* noise, int32_t batch, int32_t latent_channels, int32_t hw,
                        float scale, float* out, void* stream);
/* decode_latents' tail + numpy_to_pil: v = clamp(x / 2 + 0.5, 0, 1); out_nhwc01 [B,H,W,C] fp32 = v (nullable);
 * out_nhwc_u8 [B,H,W,C] = round_half_even(255 v) (nullable; at least one output). */
int sdn_image_postprocess(const float* image_nchw, int32_t batch, int32_t channels, int32_t height, int32_t width,
                          float* out_nhwc01, uint8_t* out_nhwc_u8, void* stream);
/* building blocks of the decoder's single-head d = 512 attention (also usable alone):
 * out[b, co, p] = bias[co] + sum_ci w[co, ci] * in_scale * z[b, ci, p]   (1x1 conv on an fp32 NCHW map, C <= 16) */
int sdn_latent_mix(const float* z, const float* w, const float* bias, int32_t batch, int32_t channels, int32_t hw,
                   float in_scale, float* out, void* stream);
/* out[r, :] = softmax(scale * scores[r, :n]) as 16-bit (dtype 0 bf16 / 1 fp16); n % 4 == 0, n <= 16384 */
int sdn_softmax_rows(int32_t dtype, const float* scores, int64_t ld_scores, int64_t rows, int32_t n, float scale,
                     void* out, int64_t ld_out, void* stream);
/* out[c, r] = in[r, c] for a 16-bit [rows, cols] matrix */
int sdn_transpose16(const void* in, int32_t rows, int32_t cols, int64_t ld_in, void* out, int64_t ld_out, void* stream);

/* ---- CLIP text encoder ------------------------------------------------------------------------------
 * Replaces `self.text_encoder(input_ids, attention_mask=...)[0]` (transformers CLIPTextModel, third party), called at
 * models/textuals_visual/modified_safree_diffusion_pipeline_threshold_time.py:197,225,287,333.  Handle = sdn_unet
 * (manifest / weights / workspace through the sdn_unet_* queries); manifest keys are CLIPTextModel's, without the
 * `text_model.` prefix: embeddings.{token,position}_embedding.weight, encoder.layers.N.{layer_norm1,self_attn.{q,k,v,out}_proj,
 * layer_norm2,mlp.fc1,mlp.fc2}.*, final_layer_norm.*.  The tokenizer (vocabulary files) stays with the caller. */
typedef struct sdn_clip_config {
  int32_t vocab_size;                    /* 49408                                                      */
  int32_t hidden_size, intermediate_size;/* 768, 3072                                                  */
  int32_t num_layers, num_heads;         /* 12, 12 (head dim must be 64)                               */
  int32_t max_position_embeddings;       /* 77 = the sequence length every call uses                   */
  int32_t dtype;                         /* 0 = bf16, 1 = fp16 storage; 2 = fp32 storage on the f32-input matrix cores; 3 = fp32
                                          * storage with bf16x3 split-operand GEMMs (sdn_gemm_x3) -- the reference loads the text
                                          * encoder in fp32 like the rest of the pipeline (run_nudity.py:277 -> load_sd(...,
                                          * torch.float32)); modes 2 / 3 take f32 weights and return f32 hidden states */
} sdn_clip_config;
int sdn_clip_create(const sdn_clip_config* cfg_host, sdn_unet** out_host);
/* last_hidden_state [B, 77, hidden] (16-bit, or f32 for dtype 2 / 3; after final_layer_norm) = text_model(input_ids [B, 77] int32,
 * attention_mask [B, 77] int32 with 1 = attend / 0 = padding, or NULL).  Attention is causal, as in CLIP. */
int sdn_clip_forward(sdn_unet* clip, const void* weights, const int32_t* input_ids, const int32_t* attention_mask,
                     void* last_hidden_state, int32_t batch, void* workspace, size_t workspace_bytes, void* stream);
/* its building blocks: embedding lookup, and attention with a causal and / or key-padding mask (head dim 64 only) */
int sdn_clip_embed(int32_t dtype, const int32_t* input_ids, const void* token_embedding, const void* position_embedding,
                   int64_t rows, int32_t seq_len, int32_t hidden, int32_t vocab, void* out, void* stream);
int sdn_masked_attention(int32_t dtype, const void* q, const void* k, const void* v, void* out, const int32_t* key_mask,
                         int32_t causal, int32_t batch, int32_t heads, int32_t n, int32_t head_dim, int32_t ldq,
                         int32_t ldk, int32_t ldv, int32_t ldo, float scale, void* stream);
/* the same two blocks on f32 storage (dtype 2 / 3 plans): exact f32 products on the f32-input matrix cores */
int sdn_clip_embed_f32(const int32_t* input_ids, const void* token_embedding, const void* position_embedding, int64_t rows,
                       int32_t seq_len, int32_t hidden, int32_t vocab, void* out, void* stream);
int sdn_masked_attention_f32(const void* q, const void* k, const void* v, void* out, const int32_t* key_mask, int32_t causal,
                             int32_t batch, int32_t heads, int32_t n, int32_t head_dim, int32_t ldq, int32_t ldk, int32_t ldv,
                             int32_t ldo, float scale, void* stream);

/* Text K / V reuse across the steps of a denoising loop.  The cross-attention key / value projections (16 per forward) read only
 * `text`, which the reference's loop feeds unchanged for all but a few of its 50 steps (the SAFREE-projected embeddings for the
 * first `beta_adjusted` steps, the plain ones afterwards: ...threshold_time.py:525-532).  Declare the CONTENTS of the text operand
 * with a non-zero version number; while consecutive sdn_unet_forward calls carry the same version, batch and weights / text /
 * workspace addresses, those projections are not recomputed (their outputs are kept in workspace slots no other tensor uses).
 * 0 (the default) = undeclared: always computed, and setting 0 drops the cached K / V at once.  Version numbers must be unique per
 * CONTENTS for the lifetime of the handle (the host side draws them from one process-wide counter): two callers that re-used a
 * number for different text at the same addresses would be served each other's K / V.  Bit-identical either way.  Ignored in graph
 * mode and by profiled forwards. */
void sdn_unet_set_text_version(sdn_unet* u, uint64_t version);

/* Graph mode for launch-bound (small) batches: sdn_unet_forward / sdn_mmdit_forward capture their ~850 launches into a
 * hipGraph once per (batch, operand addresses) and replay it afterwards -- one launch per forward plus a one-float store
 * of the timestep.  Results are identical.  Off by default; a forward issued while the caller's stream is itself being
 * captured, or a profiled forward, always launches op by op.  Keeps up to 16 instantiated graphs per handle. */
void sdn_unet_set_graph_mode(sdn_unet* u, int32_t on);
/* Small-batch option: GEMMs of the plan whose tile grid would leave most CUs idle (M <= ~2048 with a long k loop: the
 * 8x8 / 16x16-level convs of a one-prompt batch) run as sdn_gemm_splitk_*.  Off by default because the fp32 summation
 * order then depends on the batch size (without it a sample's output is bit-identical at every batch size).  Changing it
 * rebuilds the plans: query sdn_unet_workspace_bytes again. */
void sdn_unet_set_split_k(sdn_unet* u, int32_t on);

/* ---- opt-in measurement: HIP events around every launch of ONE forward, on the forward's own stream ---- */
typedef struct sdn_profile_row {
  char    kernel[24];     /* kernel symbol, e.g. "k_gemm<5>", "k_attn<40>"                              */
  int32_t launches;       /* launches of it in the profiled forward                                     */
  double  ms;             /* summed event-to-event time of those launches                               */
  double  flops;          /* summed algorithmic FLOPs (2MNK; 4 B H Nq Nk d)                             */
  double  bytes;          /* summed algorithmic HBM bytes (operands once + result once)                 */
} sdn_profile_row;
/* Arms profiling for the NEXT sdn_unet_forward on this handle (that forward records 2 events per launch). */
void sdn_unet_profile_next(sdn_unet* u);
/* Waits for the profiled forward's events and aggregates them by kernel.  Returns the number of rows (<0 = error). */
int  sdn_unet_profile_read(sdn_unet* u, sdn_profile_row* rows_host, int32_t max_rows);

#ifdef __cplusplus
}
#endif
#endif /* SDN_H_ */
