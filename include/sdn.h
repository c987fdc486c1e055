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

#ifdef __cplusplus
}
#endif
#endif /* SDN_H_ */
