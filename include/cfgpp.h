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
 * forward at a batch size: the engine times every implicit-GEMM launch of its plan in place once per candidate tile and
 * pins the fastest - a few extra forwards and host syncs, results bit-identical across candidates; environment
 * CFGPP_AUTOTUNE=0 of the Python binding, or cfgpp_unet_tuning(set = 1), skips them); return 0 on success, < 0 on error
 * with a message in cfgpp_last_error() (thread-local).  Not thread-safe: one host thread drives the library; several
 * engines may run concurrently on different streams (the K-split workspace is per stream).  ONE DEVICE PER PROCESS (the
 * torchrun layout, one rank per GPU): cfgpp_unet_create / cfgpp_vae_create refuse a device other than the first one the
 * process used.
 *
 * This header is the whole drop-in boundary.  Single-kernel test hooks (cfgpp_op_*) and the development / A-B switches of
 * the launcher live in cfgpp_amd/csrc/cfgpp_debug.h: exported by the same library, bound only by tests/ and scripts/.
 */
#ifndef CFGPP_H
#define CFGPP_H
#ifdef __cplusplus
extern "C" {
#endif

const char* cfgpp_last_error(void);

/* Provenance of this binary (no reference counterpart): "cfgpp-build:<digest of the kernel sources + flags it was compiled
 * from>:<git HEAD when it was linked>[+local]".  cfgpp_amd/build.py names every object file by the same digests, so a stale
 * object cannot be linked; bench.py, smoke() and the real-size GPU tests print this string next to their numbers. */
const char* cfgpp_build_id(void);

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

/* Whole-loop hipGraph replay: the reference's DDIM loops when callback_fn is None (latent_diffusion.py:653-674, 272-294,
 * 160-182, 888-910; latent_sdxl.py:730-752, 838-858) as ONE captured step - the UNet forward at `rows` plus the fused
 * generalised DDIM update of cfgpp_step_ddim / cfgpp_step_ddim_h - replayed n_steps times on `stream`.
 * host_steps[n_steps][5] = {t, c1, c2, c3, c4} per step: exactly the values the eager loop would pass to
 * cfgpp_unet_forward and cfgpp_step_ddim (HOST memory, copied before the call returns); they live in a device table
 * indexed by a device step counter, so one graph serves every step and every later call with the same buffers (the
 * engine keeps the most recent graph; other buffers / flags / batch re-capture).  z, z0t: [z_rows][in_ch][H][W] fp32 or
 * fp16 (z updated in place, z0t written every step); eps: [rows][out_ch][H][W] fp16 scratch for the UNet output;
 * eps_uc / eps_c point into it (the same pointer for the lambda == 1 Lightning form).  The first call at a batch runs one
 * eager forward (tile tuning, see above) and captures on an engine-owned stream - `stream` may be the legacy default
 * stream.  One host sync per call (the table upload), none per step.  Latents are bit-identical to the eager loop. */
int cfgpp_sample_graph_ddim(cfgpp_unet* u, void* z, void* z0t, int z_is_half, int z_rows, void* eps, const void* eps_uc,
                            const void* eps_c, int rows, const float* host_steps, int n_steps, float lam, int tweedie_uc,
                            int renoise_uc, void* stream);

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
 * Pinned against `transformers` on the MI355X (tests/test_gpu_text.py); the path `--model_dir` checkpoints take. */
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

#ifdef __cplusplus
}
#endif
#endif /* CFGPP_H */
