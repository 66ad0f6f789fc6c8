// Argument block of the implicit-GEMM kernel (K1/K2/K3/K4/K5 of SURVEY.md section 2.1).
#pragma once
#include "common.h"

enum IGemmEpi : int {
    EPI_STORE = 0,    // out = acc (+bias) (+temb[batch]) (+resid)          fp16, row-mapped
    EPI_GEGLU = 1,    // packed (value|gate) column pairs -> out[M][N/2] = v * gelu_erf(g)
    EPI_HEADS = 2,    // scatter into head-major Q / K / V^T buffers for attention
};

struct IGemmArgs {
    // ---- A operand (activations), up to two channel-concatenated sources ----
    const half_t* a0;
    const half_t* a1;
    int C0, C1;        // channels of each source, multiples of 64 (C1 may be 0)
    int taps;          // 1 (linear / 1x1) or 9 (3x3, pad 1)
    int amode;         // 0 linear rows, 1 padded NHWC, 2 padded stride-2, 3 padded nearest-2x upsample
    int H, W;          // OUTPUT spatial size (amode >= 1): rows m enumerate (n, y, x)
    int ashift;        // amode 2 only: 0 = pad 1 (UNet downsample), 1 = pad (0,1,0,1) (VAE encoder downsample)
    // ---- B operand: weights [N][K] fp16, K = taps*(C0+C1).  K order is CHANNEL-BLOCK major, tap minor:
    //      k = (cb*taps + tap)*64 + c  (cb = 64-channel block of the concatenated input).  The 9 taps of one
    //      channel block are consecutive K-tiles, so their shifted re-reads of the same pixels hit L2.
    const half_t* w;
    int M, N, K;
    // ---- epilogue ----
    int epi;
    const float* bias;        // [N] fp32 or null (packed order for GEGLU)
    const float* temb;        // [batch][temb_ld] fp32 or null; batch = m / (H*W)
    int temb_ld;
    int rows_per_batch;       // H*W of the OUTPUT (for temb / padded maps / heads)
    const half_t* resid;      // residual or null
    int rmode, rld;           // residual row map: 0 linear, 1 padded (H, W as above)
    half_t* out;
    int omode, old;           // output row map: 0 linear, 1 padded
    float out_scale;          // multiplies acc before bias (1.0 normally)
    // GroupNorm statistics of the OUTPUT, written by the LDS-staged plain-store epilogues (round 5): per 32-row block and output
    // column the pair (mean, sum of squared deviations) of the fp16 values actually stored (after bias / time embedding /
    // residual), gstat[(m / 32) * N + n] = {mean, M2} - fixed arithmetic per block and column, so the pairs do not depend on the
    // tile config.  A consumer GroupNorm combines them (cfgpp_op_groupnorm_pre) instead of re-reading the tensor.  Null: off.
    // *stat_flag (HOST memory) is set by igemm_launch to 1 when the launch it issued writes them, 0 when it does not (K-split
    // launches, generic epilogues): the consumer falls back to its own statistics pass then.
    float* gstat;
    int* stat_flag;
    // EPI_HEADS
    half_t* hq; half_t* hk; half_t* hvt;
    int part0;                // which part column 0 belongs to: 0=Q, 1=K (K,V projection)
    int part_width;           // C: columns per part
    int head_dim, head_dim_pad, heads;
    int vt_linear;            // 0: V^T columns in the attention kernel's permuted key order (cfgpp_vt_pos); 1: natural order
                              // (the VAE consumes V^T as the weight operand of a plain GEMM)
    int tok_pad;              // padded token count of K rows / V^T columns (>= rows_per_batch)
    int q_tok_pad;            // padded token count of Q rows
    // ---- tile scheduling (filled by igemm_launch) ----
    int n_main;               // tiles [0, n_main) are computed whole by one block each
    int ksplit;               // tiles [n_main, T) are K-split ksplit ways into fp32 partials ...
    float* ws;                // ... in this workspace, finished by igemm_reduce_kernel
    float* ws_buf;            // the CALLER's K-split workspace (engines own one: allocated at plan-build time, counted in
    long ws_bytes;            // *_device_bytes(), freed with the engine); null: a process-global per-stream buffer (test wrappers)
    int cfg_hint;             // pinned by the engine's in-situ tuning pass: bits 0-5 tile config (0 = heuristic; 1, 4 .. 12, 14),
                              // bits 6-7 tile walk (0 = by operand bytes, 1 = M-major, 2 = N-major)
    int allow_split;          // 0: never K-split this launch (autotuned launches: keeps results independent of the tile choice)
    int staged_epi;           // 1: EPI_STORE goes through the LDS-transposed, row-coalesced epilogue
    int n_major;              // 1: N-major tile walk (weight slabs stay L2-resident): weight-heavy launches
    int walk_div;             // tiles along the minor axis of the walk (filled by igemm_launch)
    int walk_hint;            // decoded from cfg_hint by igemm_launch
    // XCD-blocked 2-D walk (round 5; 0 = off): the T tiles (T % 8 == 0) are cut into walk_bm x walk_bn = 8 rectangular blocks of
    // walk_tmb x walk_tnb tiles, one per XCD (workgroup b runs on XCD b % 8 and the numbering hands an XCD a contiguous range of
    // walk_per = T / 8 indices); inside a block the order is M- or N-major as n_major says.  The workgroups an XCD runs at the same
    // time then share activation row-blocks AND weight slabs in its 4-MiB L2.  Chosen by the launcher from a count of the bytes
    // each round of resident workgroups pulls into the L2 (walk_plan); results do not depend on it.
    int walk_bn, walk_per, walk_tmb, walk_tnb;
    int split;                // >= 2: K-split every tile this many ways (igemm_launch's big-tile rule / diagnostics); 0: launcher's rule
    // ---- diagnostics (cfgpp_igemm_timeline): per-workgroup time stamps of ONE chosen launch, null otherwise ----
    int par_nb;               // time-embedding rows (batches) a tile stages in LDS (set by the launcher: covers every batch a tile's rows touch)
    unsigned long long* tl;   // [grid][16]: s_memtime at {entry, first tile landed, k-loop done, stores done}, s_memrealtime at
                              // {entry, exit}, HW_ID | XCC_ID << 32, s_memtime after the first K-tile, s_memtime {before the
                              // first LDS-DMA is issued, after the prologue's DMAs are issued}
};

int igemm_launch(const IGemmArgs& a, hipStream_t stream);
int igemm_autotune_enabled();
int igemm_last_hint_applied();     // did the last igemm_launch run the tile its cfg_hint named?
unsigned igemm_tune_mask();
