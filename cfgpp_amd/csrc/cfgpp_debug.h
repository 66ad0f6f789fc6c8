/* libcfgpp_hip.so - test hooks and development switches.  NOT part of the drop-in boundary (include/cfgpp.h is):
 *   * cfgpp_op_*  : single kernels behind the C ABI so that tests/hip_ops.py can compare each one with an fp32 reference
 *                   and scripts/ can time / profile it alone;
 *   * cfgpp_*_set_* / cfgpp_igemm_force_* / cfgpp_igemm_timeline* : process-global A/B switches and diagnostics of the
 *                   launcher (measurement runs only: the defaults are what ships, nothing on the product path changes them).
 * Same conventions as include/cfgpp.h (device pointers, `void* stream`, int return + cfgpp_last_error()). */
#ifndef CFGPP_DEBUG_H
#define CFGPP_DEBUG_H
#ifdef __cplusplus
extern "C" {
#endif

/* ---- single ops, exposed for parity tests and micro-benchmarks ------------- */
int cfgpp_op_softmax_rows(void* s, long rows, int ncols, void* stream);
int cfgpp_op_conv_in_ex(const void* z, int z_is_half, void* out, const float* w, const float* bias,
                        int R, int zB, int Cin, int H, int W, int Cout, const float* pre_w, const float* pre_b,
                        float in_scale, void* stream);
/* stats: scratch of N*(1024*G*2 + G*2) floats (per-block partials + mean/rstd); deterministic, no atomics */
int cfgpp_op_groupnorm(const void* src0, const void* src1, void* dst, const float* gamma, const float* beta,
                       float* stats, int N, int H, int W, int C0, int C1, int G, float eps, int silu,
                       int dst_padded, void* stream);
/* GroupNorm with the statistics taken from the PRODUCERS of src0 / src1 (round 5): gst0 / gst1 = [N*H*W / 32][C0 or C1][2] fp32
 * {mean, M2} per 32-pixel block and channel, written by the LDS-staged store epilogues of the implicit GEMM (IGemmArgs::gstat);
 * a finalize launch (N x G workgroups, Chan's parallel variance in a fixed order) + the apply launch.  stats: >= N*G*2 floats. */
int cfgpp_op_groupnorm_pre(const void* src0, const void* src1, void* dst, const float* gamma, const float* beta,
                           const float* gst0, const float* gst1, float* stats, int N, int H, int W, int C0, int C1, int G,
                           float eps, int silu, int dst_padded, void* stream);
/* test hook: the next cfgpp_op_igemm launches write those statistics of their output into buf (NULL = off);
 * cfgpp_op_igemm_gstat_written(): did the last one (0 for K-split launches / generic epilogues)? */
void cfgpp_op_igemm_set_gstat(void* buf);
int cfgpp_op_igemm_gstat_written(void);
/* 1 (default): the engines' GroupNorms take the producers' statistics whenever the producer wrote them; 0: always their own pass (A/B) */
void cfgpp_groupnorm_set_prestats(int on);
int cfgpp_groupnorm_prestats_enabled(void);
/* development / A-B switch of the GroupNorm form: 0 auto, 1 always the two-launch form, 2 the one-launch
 * slab-in-registers kernel whenever the slab fits (csrc/norm_kernels.hip). */
void cfgpp_groupnorm_set_mode(int mode);
/* LayerNorm over the last axis: C % 8 == 0 and C <= 2048 (16-byte loads, the rows of a wave held in registers) */
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
/* A/B switch: 1 (default) conv_out on maps of >= 128 x 128 pixels runs the LDS-tiled kernel, 0 the wave-per-pixel kernel */
void cfgpp_conv_out_set_tiled(int on);
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
/* 0 = heuristic, 1 = 128x128, 2 = 256x64, 3 = 64x64, 4 = 256x256, 5 = 256x320, 6 = 256x128, 7 = 128x160, 8 = 128x320,
 * 10 = 256x320 (waves along M); 9 / 11 = 128x160 on a 3- / 4-stage LDS ring, 12 = 128x128 and 14 = 256x128 on 3 stages;
 * 15 / 16 / 17 = 256x128 (8 waves) / 128x128 / 256x64 (4 waves) on 32-deep K-tiles and three stages, 2 - 3 workgroups per CU;
 * 13 / 20 = 256x256 / 256x320 on 32-deep K-tiles and FOUR stages (tile32_kernel);
 * 18 / 19 = 128x160 as 8 waves of 32x80 on the 16x16x32 MFMA, 3 / 4 stages (plain-store launches with N % 160 == 0);
 * 21..23 = register-staged 1..3 */
void cfgpp_igemm_force_config(int cfg);
void cfgpp_igemm_set_tail_split(int on);  /* 1 = K-split tiny grids with long K into fp32 partials + reduce (default 1); 2 = the
                                           * round-1 slice count (rounded up: a second partial round of workgroups), for A/B */
/* 8-wave 16x16x32-MFMA 128x160 tile for plain-store launches whose 128x160 grid is 200..256 tiles: 0 = off, 3 / 4 (default 4) =
 * on with that many LDS stages.  Rule-based (the tile sums k in a different order than the others, so the tuner never picks it). */
void cfgpp_igemm_set_mf16(int mode);
/* 1 (default): the rule also takes token-major linears; 0: convolutions only - the in-situ tuner then picks the linears' tile (A/B) */
void cfgpp_igemm_set_mf16_linear(int on);
/* 1 (default since round 3, validated on hardware): QKV / Q / KV projections (head-major epilogue) may use that tile too; 0: off */
void cfgpp_igemm_set_mf16_heads(int on);
/* A/B knob of that rule: also take grids of exactly 2 .. n full rounds of 256 tiles (default 2: the M = 16384 x N = 640 class;
 * 1 = one round of 200 .. 256 tiles only) */
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
/* 1 (default): launches whose round-by-round byte count says so take the XCD-blocked 2-D tile walk (IGemmArgs::walk_bn,
 * csrc/igemm_kernel.hip walk_plan); 0: the 1-D M- / N-major walks only (A/B).  Results do not depend on it. */
void cfgpp_igemm_set_blocked_walk(int on);
/* host-only probe of that choice (tests; no GPU): token GEMM M x N x K on BM x BN tiles, smem bytes of LDS per workgroup, the 1-D
 * default (n_major); out5 = {blocks along M (0: the 1-D walk stays), along N, tiles per block along M, along N, inner order} */
void cfgpp_igemm_walk_plan_probe(int M, int N, int K, int BM, int BN, int smem, int n_major, int* out5);
/* in-situ tuning candidates: bit c set = tile config c may be pinned (c = 1 .. 27; 24 - 26 = the one-wave-per-SIMD tiles of
 * big4_kernel.hip), bit 31 = the tile-walk stage runs.  Default 0xf1ffffff: everything but 25 / 26 / 27, which lose in situ
 * (profiles/r05/ab/) */
void cfgpp_igemm_set_tune_mask(unsigned mask);
/* all switches that change what the tuner measures or may pin (mask, big tiles, tail split, walk, staging, forced config),
 * folded into one word: pins are persisted only by a process whose word is still the default (cfgpp_amd/tune_cache.py) */
unsigned cfgpp_igemm_tuner_state(void);
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
#endif /* CFGPP_DEBUG_H */
