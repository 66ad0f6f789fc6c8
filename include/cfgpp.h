/* libcfgpp_hip.so - C ABI of the MI355X-native CFG++ sampling hot path.
 *
 * The reference (CFGpp-diffusion/CFGpp) has no FFI of its own: its hot path is
 * the pure-Python seam
 *     self.unet(z_in, t_in, encoder_hidden_states=..., added_cond_kwargs=...)['sample']
 *         latent_diffusion.py:146,149,155      latent_sdxl.py:170,174,181
 * plus the per-step elementwise sampler arithmetic
 *         latent_diffusion.py:660-666 (and 179-180, 283-286, 479-490, 708-710, 855-866, 907-908)
 *         latent_sdxl.py:738-744 (and 317-318, 453-456, 904-919, 972-973).
 * This header declares what a binding for that seam calls instead.  Each entry
 * point cites the reference lines it replaces.
 *
 * Conventions: plain C types only; every data pointer is a DEVICE pointer owned
 * by the caller (PyTorch-ROCm tensor storage) unless the name says `host`; all
 * work is enqueued asynchronously on the hipStream_t passed as `void* stream`
 * and nothing synchronises (one exception: the tile-tuning passes of the first
 * forward at a batch size, see cfgpp_igemm_set_autotune); return 0 on success, < 0 on error with a message in
 * cfgpp_last_error() (thread-local).  Not thread-safe.  ONE DEVICE PER PROCESS (the torchrun layout, one rank per
 * GPU): launcher state (K-split workspace, tuning switches) is process-global, and cfgpp_unet_create /
 * cfgpp_vae_create refuse a device other than the first one the process used.
 */
#ifndef CFGPP_H
#define CFGPP_H
#ifdef __cplusplus
extern "C" {
#endif

const char* cfgpp_last_error(void);

/* ---- fused sampler step (K12) ------------------------------------------- */

/* Generalised DDIM update on n fp32 latent elements, in place:
 *     eps_hat = eps_uc + lam*(eps_c - eps_uc)
 *     z0t = (z - c1*A)/c2 ;  z = c3*z0t + c4*B ;  A,B in {eps_hat, eps_uc}
 * replaces latent_diffusion.py:660-666 (CFG++: tweedie_uc=0, renoise_uc=1),
 * :280-286 (CFG: 0,0), :177-180 (inversion CFG: 0,0 with coefficients swapped),
 * :905-908 (inversion CFG++: 1,0) and latent_sdxl.py:738-744, 450-456, 315-318, 970-973.
 * eps_is_half=1: eps are fp16 (the autocast UNet output) and every eps product is
 * rounded to fp16 exactly as torch promotion does.  n must be a multiple of 4.
 * Divisors (c2 here; s of cfgpp_kdiff_input mode 0; sigma_item and two_r of cfgpp_step_kdiff) are positive on this
 * path, and their SIGN selects how the quotient is formed: c > 0 is an IEEE division (torch-CPU); c < 0 means the
 * caller passes c = -fl32(1/divisor) and the kernel multiplies by -c - torch's GPU `div` with a CPU-scalar divisor
 * (`/ at.sqrt()`, `/ sigma.item()`, `/ (2*r)`), which is `a * (1/b)` with the reciprocal taken once in fp32. */
int cfgpp_step_ddim(void* z, void* z0t_out, const void* eps_uc, const void* eps_c, int eps_is_half,
                    float lam, float c1, float c2, float c3, float c4,
                    int tweedie_uc, int renoise_uc, long n, void* stream);

/* The same update on n fp16 latent elements (z, z0t_out and both eps fp16), every op rounded to
 * fp16: the reference's inversion / edit paths, whose latent starts as the fp16 `vae.encode(...)`
 * sample and therefore stays fp16 through `inversion()` and the regeneration loop
 * (latent_diffusion.py:168-180,527-541,901-908; latent_sdxl.py:307-318,966-1011). */
int cfgpp_step_ddim_h(void* z, void* z0t_out, const void* eps_uc, const void* eps_c,
                      float lam, float c1, float c2, float c3, float c4,
                      int tweedie_uc, int renoise_uc, long n, void* stream);

/* k-diffusion UNet input scaling on fp16 latents: mode 0: xc = x / s
 * (latent_diffusion.py:229-230, s = sqrt(sigma^2+1)); mode 1: xc = x * s (latent_sdxl.py:901). */
int cfgpp_kdiff_input(const void* x, void* xc, float s, int mode, long n, void* stream);

/* Euler / DPM-Solver++(2M) update on fp16 latents, in place.
 * coef[9] = {lam, sigma, c_out_h, sigma_item, sigma_next, neg_exp_mh_h, expm1_mh_h, two_r, exp_mh_h}
 * variant 0 = CFG (latent_diffusion.py:477-490, 329-333), 1 = CFG++ SD1.5 (:853-866, 706-710),
 * 2 = CFG++ SDXL (latent_sdxl.py:901-919).  xl_form: denoised = x + c_out*eps instead of x - eps*sigma.
 * euler_branch=1 selects x' = den + ((x - d_from)/sigma)*sigma_next.  den_out receives `denoised`. */
int cfgpp_step_kdiff(void* x, void* den_out, void* old, const void* eps_uc, const void* eps_c,
                     const float* coef_host, int variant, int xl_form, int euler_branch, int write_old,
                     long n, void* stream);

/* denoised = x - eps_hat*sigma and uncond_denoised = x - eps_uc*sigma on fp16 latents
 * (kdiffusion_x_to_denoised, latent_diffusion.py:235-241); used by the 2-stage DPM++(2S) samplers. */
int cfgpp_kdiff_denoise(const void* x, const void* eps_uc, const void* eps_c, float lam, float sigma,
                        void* den, void* uden, long n, void* stream);
/* fp16 linear combinations with torch's rounding order (out may alias x):
 * mode 0: out = x*a - y*b            (latent_diffusion.py:428,435,804)
 * mode 1: out = (y - z*b) + x*a      (latent_diffusion.py:811)
 * mode 2: out = x + y*a              (ancestral noise, latent_diffusion.py:379,438,755,814) */
int cfgpp_lincomb(void* out, const void* x, const void* y, const void* z, float a, float b, int mode, long n, void* stream);

/* ---- UNet engine (replaces `self.unet(...)`) ------------------------------ */

typedef struct cfgpp_unet_config {
    int in_channels, out_channels;
    int num_levels;               /* 4 for SD1.5, 3 for SDXL */
    int block_out_channels[4];
    int layers_per_block;         /* 2 */
    int level_has_attn[4];        /* CrossAttn{Down,Up}Block2D per level (down order) */
    int transformer_depth[4];     /* transformer layers per attention, per level */
    int num_heads[4];             /* heads per level (SD1.5: 8 everywhere; SDXL: C/64) */
    int cross_attention_dim;      /* 768 / 2048 */
    int addition_embed;           /* 0 none, 1 = "text_time" (SDXL) */
    int addition_time_embed_dim;  /* 256 */
    int addition_pooled_dim;      /* 1280 */
    int norm_groups;              /* 32 */
    int sample_h, sample_w;       /* latent size this context is built for */
    int max_rows;                 /* max UNet batch rows (2 * chains) */
} cfgpp_unet_config;

typedef struct cfgpp_unet cfgpp_unet;

/* `pipe.unet` construction (latent_diffusion.py:63-67, latent_sdxl.py:40,50). */
cfgpp_unet* cfgpp_unet_create(const cfgpp_unet_config* cfg, int device_id);
void cfgpp_unet_destroy(cfgpp_unet* u);

/* Load one state-dict entry by its diffusers key (e.g.
 * "down_blocks.0.resnets.0.conv1.weight").  `host` points to HOST memory,
 * dtype 0 = fp32, 1 = fp16; conv weights OIHW, linear weights [out,in].
 * The library copies; returns -3 for an unknown key. */
int cfgpp_unet_load_tensor(cfgpp_unet* u, const char* key, const void* host, int dtype,
                           const long* shape, int ndim);
/* Number of parameters still missing (0 = complete), names via cfgpp_last_error(). */
int cfgpp_unet_missing(cfgpp_unet* u);
/* Repack all weights into MFMA-friendly device layouts and build the launch plan. */
int cfgpp_unet_finalize(cfgpp_unet* u);

/* Conditioning for the next forwards: ehs [rows][77][cross_dim] fp16 (uc rows first,
 * then c rows: the `torch.cat([uc, c])` of latent_diffusion.py:152); SDXL:
 * text_embeds [cond_rows][1280] fp16, time_ids [cond_rows][6] fp32, cond_rows = rows or 1
 * (1 = broadcast, the lambda==1.0 Lightning case of latent_sdxl.py:249-252).
 * Precomputes the step-invariant cross-attention K/V of every block. */
int cfgpp_unet_set_context(cfgpp_unet* u, const void* ehs, int rows, int tokens,
                           const void* text_embeds, const void* time_ids, int cond_rows, void* stream);

/* eps[rows][out_ch][H][W] (fp16) = UNet(z[(row % z_rows)], t).  z: [z_rows][in_ch][H][W],
 * fp32 (z_is_half=0) or fp16.  rows = 2*z_rows reproduces cat([zt]*2) / chunk(2) of
 * latent_diffusion.py:153-156: eps_uc = eps[0:z_rows], eps_c = eps[z_rows:]. */
int cfgpp_unet_forward(cfgpp_unet* u, const void* z, int z_is_half, int z_rows, float t,
                       void* eps_out, int rows, void* stream);

/* Same forward with a HIP event between every launch (on `stream`): per kernel family
 * k = 0 implicit-GEMM conv/linear, 1 attention, 2 GroupNorm/LayerNorm, 3 small ops:
 * elapsed ms, algorithmic FLOPs and launch counts.  Used by bench.py's roofline block. */
int cfgpp_unet_profile(cfgpp_unet* u, const void* z, int z_is_half, int z_rows, float t, void* eps_out, int rows,
                       void* stream, double* out_ms, double* out_flops, int* out_launches,
                       char* detail /* optional: one text line per launch */, long detail_cap);

/* Export (set = 0) / import (set = 1) the per-launch tile configs pinned by the in-situ tuning at batch `rows`
 * (one int per implicit-GEMM launch, plan order); returns the number of slots.  A profiled run imports what the
 * un-profiled run chose, so PMC passes see the same kernels without timing passes of their own. */
int cfgpp_unet_tuning(cfgpp_unet* u, int rows, int* hints, int cap, int set);

/* Algorithmic FLOPs (2*MAC over conv/linear/attention matmuls) of one forward at `rows`. */
double cfgpp_unet_flops(cfgpp_unet* u, int rows);
/* Bytes of device memory held (weights + activations). */
double cfgpp_unet_device_bytes(cfgpp_unet* u);

/* ---- VAE engine: decoder (replaces `self.vae.decode(z / scale).sample`) and encoder ------
 * latent_diffusion.py:123-129 (scale 0.18215), latent_sdxl.py:155-164 (vae.config.scaling_factor).
 * Weights by diffusers AutoencoderKL keys (post_quant_conv.*, decoder.*).  img fp32 [B][3][8h][8w]. */
typedef struct cfgpp_vae cfgpp_vae;
cfgpp_vae* cfgpp_vae_create(int latent_h, int latent_w, int max_batch, float scaling_factor, int device_id);
void cfgpp_vae_destroy(cfgpp_vae* v);
int cfgpp_vae_load_tensor(cfgpp_vae* v, const char* key, const void* host, int dtype, const long* shape, int ndim);
int cfgpp_vae_finalize(cfgpp_vae* v);
int cfgpp_vae_decode(cfgpp_vae* v, const void* z, void* img, int B, void* stream);
/* decode + `(img / 2 + 0.5).clamp(0, 1)` of the solvers' sample() (latent_diffusion.py:676-677,
 * latent_sdxl.py:274-275) in the decoder's last kernel: img[B][3][8h][8w] fp32 in [0, 1]. */
int cfgpp_vae_decode_image(cfgpp_vae* v, const void* z, void* img, int B, void* stream);
/* Encoder (replaces `self.vae.encode(x).latent_dist.sample() * scale`, latent_diffusion.py:117-121,
 * latent_sdxl.py:150-153); available when every encoder.* and quant_conv.* tensor was loaded.
 * img [B][3][8h][8w] f32, noise [B][4][h][w] f32 or NULL (posterior mean), z [B][4][h][w] f32,
 * moments [B][8][h][w] f32 or NULL (mean | logvar clamped to [-30, 20]). */
int cfgpp_vae_encode(cfgpp_vae* v, const void* img, const void* noise, void* z, void* moments, int B, void* stream);
/* one decode (image post-processing included) with a HIP event between every launch of the decoder plan; `detail` receives one
 * line per launch: index \t family (0 igemm, 1 attention GEMMs, 2 norm / softmax, 3 small) \t description \t us \t GFLOP */
int cfgpp_vae_profile(cfgpp_vae* v, const void* z, void* img, int B, void* stream, char* detail, long detail_cap);
double cfgpp_vae_flops(cfgpp_vae* v, int B);
double cfgpp_vae_encode_flops(cfgpp_vae* v, int B);
double cfgpp_vae_device_bytes(cfgpp_vae* v);

/* ---- CLIP text transformer -------------------------------------------------------------------------
 * Replaces `self.text_encoder(tokens)` of the reference (latent_diffusion.py:105-113: last_hidden_state;
 * latent_sdxl.py:76-93: hidden_states[-2] / [-(clip_skip+2)] and the second tower's projected pooled output).  Once per
 * prompt, off the per-step path.  Geometry: hidden = heads * 64 (CLIP-L 768 / 12, OpenCLIP-bigG 1280 / 20), 77 tokens,
 * act 0 = quick_gelu, 1 = gelu, proj_dim 0 = no text_projection.  Keys of cfgpp_text_load_tensor = the `transformers`
 * CLIPTextModel(WithProjection) state dict ("text_model.embeddings.token_embedding.weight", ...,
 * "text_model.encoder.layers.<i>.self_attn.q_proj.weight", ..., "text_model.final_layer_norm.bias", "text_projection.weight").
 * First contact with hardware: tests/test_gpu_text.py (opt-in); nothing on the default path calls these yet. */
typedef struct cfgpp_text cfgpp_text;
cfgpp_text* cfgpp_text_create(int vocab, int hidden, int layers, int heads, int intermediate, int act, int proj_dim,
                              int max_batch, int device_id);
void cfgpp_text_destroy(cfgpp_text* t);
int cfgpp_text_load_tensor(cfgpp_text* t, const char* key, const void* host, int dtype, const long* shape, int ndim);
int cfgpp_text_finalize(cfgpp_text* t);
/* ids: HOST int32 [B][77]; eos_pos: HOST int32 [B] (row that is pooled) or NULL; layer: -1 = last_hidden_state (final
 * LayerNorm applied), k in [0, layers] = hidden_states[k]; hidden_out: DEVICE fp16 [B][77][hidden]; pooled_out: DEVICE fp32
 * [B][proj_dim] or NULL. */
int cfgpp_text_encode(cfgpp_text* t, const int* ids, const int* eos_pos, int B, int layer, void* hidden_out, float* pooled_out,
                      void* stream);
double cfgpp_text_device_bytes(cfgpp_text* t);

/* ---- single ops, exposed for parity tests and micro-benchmarks ------------- */
int cfgpp_op_softmax_rows(void* s, long rows, int ncols, void* stream);
int cfgpp_op_conv_in_ex(const void* z, int z_is_half, void* out, const float* w, const float* bias,
                        int R, int zB, int Cin, int H, int W, int Cout, const float* pre_w, const float* pre_b,
                        float in_scale, void* stream);
/* stats: scratch of N*(1024*G*2 + G*2) floats (per-block partials + mean/rstd); deterministic, no atomics */
int cfgpp_op_groupnorm(const void* src0, const void* src1, void* dst, const float* gamma, const float* beta,
                       float* stats, int N, int H, int W, int C0, int C1, int G, float eps, int silu,
                       int dst_padded, void* stream);
/* development / A-B switch of the GroupNorm form: 0 auto, 1 always the two-launch form, 2 the one-launch
 * slab-in-registers kernel whenever the slab fits (csrc/norm_kernels.hip). */
void cfgpp_groupnorm_set_mode(int mode);
/* LayerNorm statistics only: stats[row] = (mean, rstd) fp32, exact two-pass variance; the projection that consumes the
 * LayerNorm applies it in its epilogue (cfgpp_op_linear_ln / cfgpp_op_igemm_heads_ln, and the UNet's transformer blocks) */
int cfgpp_op_ln_stats(const void* x, float* stats, long rows, int C, float eps, void* stream);
int cfgpp_op_layernorm(const void* x, void* y, const float* gamma, const float* beta, long rows, int C,
                       float eps, void* stream);
/* development / A-B switch: token rows each wave of the LayerNorm kernel keeps in flight (0 = by row count, 1 / 2 / 4);
 * the result does not depend on it. */
void cfgpp_layernorm_set_rows_per_wave(int rpw);
/* V^T contract of cfgpp_op_attention: vt is [B*heads][dp][tok_pad] with the keys of every 32-key block
 * permuted - key k lives in column (k & ~12) | ((k & 4) << 1) | ((k & 8) >> 1) (bits 2 and 3 swapped), which is
 * how the QKV projection (cfgpp_op_igemm_heads) writes it; and when d % 32 != 0, row d of every matrix holds
 * ones (softmax denominator through the PV MFMA): call prepare_vt once on the zero-initialised buffer. */
int cfgpp_op_attention_prepare_vt(void* vt, int BH, int d, int tok_pad, void* stream);
int cfgpp_op_attention(const void* q, const void* k, const void* vt, void* o, int B, int heads, int d,
                       int nq, int nk, int q_tok_pad, int k_tok_pad, void* stream);
/* A/B switch for head dims padded to 64: 1 (default) the LDS-DMA kernel, 0 the register-staged kernel */
void cfgpp_attention_set_dma(int mode);
/* A/B knob of the LDS-DMA attention kernel: the workgroups sharing a CU start `sleeps` x 64 cycles apart per dispatch slot
 * (0 = together, the default) so that their QK^T / softmax / PV phases interleave instead of coinciding */
void cfgpp_attention_set_stagger(int sleeps);
/* A/B switch: 1 (default) attention with <= 128 keys and head dims padded to 64 (the 77-token cross-attention) runs the
 * resident-K/V single-pass kernel, 0 the flash loop */
void cfgpp_attention_set_cross(int on);
int cfgpp_op_conv_in(const void* z, int z_is_half, void* out, const float* w, const float* bias,
                     int R, int zB, int Cin, int H, int W, int Cout, void* stream);
/* quant_conv (1x1, 8->8) + DiagonalGaussian posterior on the encoder's 8-channel conv_out (fp32 NCHW). */
int cfgpp_op_vae_posterior(const float* conv_out, const float* qw, const float* qb, const float* noise, float* z,
                           float* moments, int B, int HW, float scale, void* stream);
int cfgpp_op_conv_out(const void* x, void* out, int out_is_half, const void* w, const float* bias,
                      int R, int H, int W, int C, int Cout, void* stream);
/* the same with `* post_scale + post_shift` and an optional clamp to [0, 1] applied to the fp32 result (the VAE
 * decoder's conv_out with the sampler's `(img / 2 + 0.5).clamp(0, 1)` folded in) */
int cfgpp_op_conv_out_ex(const void* x, void* out, int out_is_half, const void* w, const float* bias,
                         int R, int H, int W, int C, int Cout, float post_scale, float post_shift, int clamp01, void* stream);
int cfgpp_op_sinusoid(const float* vals, float scalar, float* out, int count, int dim, int out_ld, int out_off,
                      void* stream);
int cfgpp_op_skinny_gemm(const float* x, int ldx, const void* w, const float* bias, const float* addend, int add_ld,
                         float* out, int ldo, int M, int N, int K, int silu_in, int silu_out, void* stream);
int cfgpp_op_f16_to_f32_rows(const void* in, float* out, int rows, int cols, int out_ld, int out_off, void* stream);

/* Generic implicit GEMM (conv3x3 / conv1x1 / linear), see cfgpp_amd/csrc/igemm.h.
 * a0/a1: activation sources (C0/C1 channels), amode 0 linear rows, 1 halo-padded NHWC,
 * 2 padded stride-2, 3 padded nearest-2x upsample; w [N][taps*(C0+C1)] fp16 with K order channel-block major,
 * tap minor: k = (cb*taps + tap)*64 + c, cb = 64-channel block of the concatenated input;
 * epi 0 store (+bias +temb +resid), 1 GEGLU (packed weights).  omode/rmode: 0 linear, 1 padded. */
int cfgpp_op_igemm(const void* a0, const void* a1, int C0, int C1, int taps, int amode, int H, int W,
                   const void* w, int M, int N, const float* bias, const float* temb, int temb_ld,
                   const void* resid, int rmode, int rld, void* out, int omode, int old_, int epi,
                   void* stream);
/* QKV / KV projection with head-major scatter (EPI_HEADS) */
int cfgpp_op_igemm_heads(const void* a, int K, const void* w, int M, int N, const float* bias, int rows_per_batch,
                         void* hq, void* hk, void* hvt, int part0, int part_width, int head_dim, int heads,
                         int q_tok_pad, int tok_pad, void* stream);
/* the head-major projection / the GEGLU projection with a LayerNorm of the input rows folded in: a = UN-normalised rows,
 * w = W * gamma (per input channel), bias = W beta (+ the layer's bias), ln_c[n] = sum_k w[n][k], ln_stats = (mean, rstd) per
 * row from cfgpp_op_ln_stats, or NULL: the kernel accumulates sum / sum of squares of its rows from the activation fragments of
 * its K loop (eps 1e-5); the epilogue forms rstd * (acc - mean * ln_c) + bias.  Same output contracts as
 * cfgpp_op_igemm_heads / cfgpp_op_igemm with epi = 1 (w and bias in the packed GEGLU order). */
int cfgpp_op_igemm_heads_ln(const void* a, int K, const void* w, int M, int N, const float* bias, const float* ln_stats,
                            const float* ln_c, int rows_per_batch, void* hq, void* hk, void* hvt, int part0, int part_width,
                            int head_dim, int heads, int q_tok_pad, int tok_pad, void* stream);
int cfgpp_op_geglu_ln(const void* a, int K, const void* w, int M, int N, const float* bias, const float* ln_stats,
                      const float* ln_c, void* out, void* stream);
/* 1: UNet engines finalized after this call fold the transformer blocks' LayerNorms into the projections that consume
 * them, (mean, rstd) per row from a statistics pass; 2: folded in, and the consuming kernel takes (mean, rstd) from its own
 * operand fragments in the K loop - no statistics launch (same function as 0, different fp16 rounding points);
 * 0 (default): separate layernorm launches */
void cfgpp_unet_set_fuse_ln(int on);
/* 0 = heuristic, 1 = 128x128, 2 = 256x64, 3 = 64x64, 4 = 256x256, 5 = 256x320, 6 = 256x128, 7 = 128x160, 8 = 128x320,
 * 10 = 256x320 (waves along M); 9 / 11 = 128x160 on a 3- / 4-stage LDS ring, 12 = 128x128 and 14 = 256x128 on 3 stages;
 * 18 / 19 = 128x160 as 8 waves of 32x80 on the 16x16x32 MFMA, 3 / 4 stages (plain-store launches with N % 160 == 0);
 * 21..23 = register-staged 1..3 */
void cfgpp_igemm_force_config(int cfg);
void cfgpp_igemm_set_tail_split(int on);  /* 1 = K-split tiny grids with long K into fp32 partials + reduce (default 1); 2 = the
                                           * round-1 slice count (rounded up: a second partial round of workgroups), for A/B */
/* 8-wave 16x16x32-MFMA 128x160 tile for plain-store launches whose 128x160 grid is 200..256 tiles: 0 = off, 3 / 4 (default 4) =
 * on with that many LDS stages.  Rule-based (the tile sums k in a different order than the others, so the tuner never picks it). */
void cfgpp_igemm_set_mf16(int mode);
/* 1: QKV / Q / KV projections (head-major epilogue) may use that tile too; default 0 until validated on hardware */
void cfgpp_igemm_set_mf16_heads(int on);
/* A/B knob of that rule: also take grids of exactly 2 .. n full rounds of 256 tiles (default 1 = one round only) */
void cfgpp_igemm_set_mf16_rounds(int n);
/* tile of the rule-based K-split launches: 14 (default) = 256x128 on 3 stages, 1 = 128x128 on 2 stages, 12 = 128x128 on 3 stages */
void cfgpp_igemm_set_split_tile(int cfg);
/* diagnostics: with a forced config, K-split every tile of a plain-store launch this many ways (0 = off) */
void cfgpp_igemm_force_split(int s);
/* big-tile K-split rule (M x N too small for 8-wave tiles to fill the chip, K long): least K-tiles (of 64) per slice for the
 * rule to fire; 0 = rule off (default - see csrc/igemm_kernel.hip).  Rule-based, so results never depend on tile tuning. */
void cfgpp_igemm_set_big_split(int min_kt);
/* tile walk of the implicit GEMM: -1 (default) by operand bytes / the tuner's pin, 0 always M-major, 1 always N-major; the
 * result does not depend on it */
void cfgpp_igemm_set_n_major(int mode);
/* in-situ tuning candidates: bit c set = tile config c may be pinned, bit 31 = the tile-walk stage runs (default: all) */
void cfgpp_igemm_set_tune_mask(unsigned mask);
/* 1 (default): on the first cfgpp_unet_forward / cfgpp_vae_decode at a batch size the engine times every igemm
 * launch of its plan in place (HIP events, a few extra forwards, one host sync) per candidate tile config and pins
 * the fastest; results are bit-identical across candidates, K-split launches stay rule-based.  0: fixed heuristic. */
void cfgpp_igemm_set_autotune(int on);
void cfgpp_igemm_set_big_tiles(int on);  /* 1 = allow the 8-wave 256x256 / 256x320 tiles (default) */
void cfgpp_igemm_set_staged_epilogue(int on); /* 1 = LDS-transposed row-coalesced store epilogue (default) */
void cfgpp_igemm_set_staging(int glds);   /* 1 = global_load_lds tiles (default), 0 = register staging */
/* diagnostics: per-workgroup timeline of ONE implicit-GEMM launch.  Arms the `target`-th launch (0-based) after this call:
 * every workgroup writes 16 x uint64 into buf[grid][16] (device memory, cap_blocks records): s_memtime at {entry, first
 * K-tile landed, k-loop done, stores done}, s_memrealtime (100 MHz) at {entry, exit}, HW_ID | XCC_ID << 32, s_memtime after
 * the first K-tile, s_memtime {before the first LDS-DMA is issued, after the prologue's DMAs are issued}.  buf = NULL disarms.  cfgpp_igemm_timeline_info: {tile id, grid, threads, BM, BN, LDS stages, K-split,
 * N-major walk, M, N, K, epilogue} of the recorded launch (scripts/igemm_timeline.py). */
void cfgpp_igemm_timeline(void* buf, long cap_blocks, int target);
void cfgpp_igemm_timeline_info(int* out12);

#ifdef __cplusplus
}
#endif
#endif /* CFGPP_H */
