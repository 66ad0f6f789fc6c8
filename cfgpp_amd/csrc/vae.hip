// AutoencoderKL engine.
// DECODER: `vae.decode(z / scaling_factor).sample` (latent_diffusion.py:123-129, latent_sdxl.py:155-164);
// ENCODER (optional - built when the encoder.* / quant_conv.* weights are loaded):
// `vae.encode(x).latent_dist.sample() * scaling_factor` (latent_diffusion.py:117-121, latent_sdxl.py:150-153)
// with the posterior noise supplied by the caller (null = posterior mean).
// Both run on the same hand-written HIP kernels as the UNet - implicit-GEMM conv3x3 (incl. fused nearest-2x upsample), GroupNorm(+SiLU), 1x1
// shortcut - plus the mid-block attention (ONE 512-wide head over H*W tokens) as
//   S = (Q K^T)/sqrt(512) [igemm, per image]  ->  row softmax in place  ->  O = P V [igemm]
// with Q / K / V^T written head-major by the QKV GEMM epilogue.  `z / scaling_factor` and the
// 1x1 post_quant_conv are folded into conv_in's input gather.  Weights by diffusers
// AutoencoderKL state-dict keys (decoder.*, post_quant_conv.*).  Output fp32 NCHW.
#include "engine_base.h"

extern "C" int cfgpp_op_softmax_rows(void* s, long rows, int ncols, void* stream);
extern "C" int cfgpp_op_vae_posterior(const float* conv_out, const float* qw, const float* qb, const float* noise, float* z,
                                      float* moments, int B, int HW, float scale, void* stream);
extern "C" int cfgpp_op_conv_in_ex(const void* z, int z_is_half, void* out, const float* w, const float* bias,
                                   int R, int zB, int Cin, int H, int W, int Cout, const float* pre_w, const float* pre_b,
                                   float in_scale, void* stream);

struct cfgpp_vae : EngineBase {
    int h = 0, w = 0, device = 0;
    float scaling = 1.f;
    bool finalized = false, has_encoder = false;
    int ch[4] = {128, 256, 512, 512};
    const void* in_z = nullptr; void* out_img = nullptr;
    // encoder
    std::vector<Op> enc_plan;
    const void* in_img = nullptr; const float* in_noise = nullptr; float* out_z = nullptr; float* out_moments = nullptr;
    float* enc_co = nullptr;       // conv_out result [R][8][h][w] fp32
    float post_scale = 1.0f, post_shift = 0.0f; int post_clamp = 0;     // folded into the decoder's conv_out (set per call)
    double dec_macs = 0, enc_macs = 0;
    half_t *tok_a = nullptr, *tok_o = nullptr, *hq = nullptr, *hk = nullptr, *hvt = nullptr, *smat = nullptr;
};

namespace {

void vae_expect_res(cfgpp_vae* v, const std::string& p, long i, long o) {
    expect_norm(v, p + ".norm1", i); expect_conv(v, p + ".conv1", o, i, 3);
    expect_norm(v, p + ".norm2", o); expect_conv(v, p + ".conv2", o, o, 3);
    if (i != o) expect_conv(v, p + ".conv_shortcut", o, i, 1);
}

void vae_param_table(cfgpp_vae* v) {
    const int* ch = v->ch;
    expect_conv(v, "post_quant_conv", 4, 4, 1);
    expect_conv(v, "decoder.conv_in", ch[3], 4, 3);
    long c = ch[3];
    vae_expect_res(v, "decoder.mid_block.resnets.0", c, c);
    expect_norm(v, "decoder.mid_block.attentions.0.group_norm", c);
    for (const char* n : {"to_q", "to_k", "to_v", "to_out.0"}) expect_linear(v, std::string("decoder.mid_block.attentions.0.") + n, c, c, true);
    vae_expect_res(v, "decoder.mid_block.resnets.1", c, c);
    for (int i = 0; i < 4; ++i) {
        const long co = ch[3 - i];
        for (int j = 0; j < 3; ++j) { vae_expect_res(v, "decoder.up_blocks." + std::to_string(i) + ".resnets." + std::to_string(j), c, co); c = co; }
        if (i != 3) expect_conv(v, "decoder.up_blocks." + std::to_string(i) + ".upsamplers.0.conv", co, co, 3);
    }
    expect_norm(v, "decoder.conv_norm_out", c);
    expect_conv(v, "decoder.conv_out", 3, c, 3);
    // encoder (optional as a whole)
    expect_conv(v, "encoder.conv_in", ch[0], 3, 3);
    c = ch[0];
    for (int i = 0; i < 4; ++i) {
        for (int j = 0; j < 2; ++j) { vae_expect_res(v, "encoder.down_blocks." + std::to_string(i) + ".resnets." + std::to_string(j), c, ch[i]); c = ch[i]; }
        if (i != 3) expect_conv(v, "encoder.down_blocks." + std::to_string(i) + ".downsamplers.0.conv", c, c, 3);
    }
    vae_expect_res(v, "encoder.mid_block.resnets.0", c, c);
    expect_norm(v, "encoder.mid_block.attentions.0.group_norm", c);
    for (const char* n : {"to_q", "to_k", "to_v", "to_out.0"}) expect_linear(v, std::string("encoder.mid_block.attentions.0.") + n, c, c, true);
    vae_expect_res(v, "encoder.mid_block.resnets.1", c, c);
    expect_norm(v, "encoder.conv_norm_out", c);
    expect_conv(v, "encoder.conv_out", 8, c, 3);
    expect_conv(v, "quant_conv", 8, 8, 1);
}

bool is_encoder_key(const std::string& k) { return k.rfind("encoder.", 0) == 0 || k.rfind("quant_conv.", 0) == 0; }

}  // namespace

extern "C" {

cfgpp_vae* cfgpp_vae_create(int latent_h, int latent_w, int max_batch, float scaling_factor, int device_id) {
    if (latent_h <= 0 || latent_w <= 0 || max_batch <= 0 || (latent_h * latent_w) % 128 != 0 || latent_h * latent_w > 16384) {
        cfgpp_set_error("vae_create: latent %dx%d unsupported (H*W must be a multiple of 128 and <= 16384)", latent_h, latent_w);
        return nullptr;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= device_id || hipSetDevice(device_id) != hipSuccess) {
        cfgpp_set_error("vae_create: no HIP device %d - the HIP path has no CPU fallback", device_id);
        return nullptr;
    }
    if (cfgpp_claim_device(device_id)) return nullptr;
    cfgpp_vae* v = new cfgpp_vae();
    v->h = latent_h; v->w = latent_w; v->max_rows = max_batch; v->norm_groups = 32; v->scaling = scaling_factor; v->device = device_id;
    // conv_in weight lives in the fp32 table like the UNet's
    vae_param_table(v);
    return v;
}

void cfgpp_vae_destroy(cfgpp_vae* v) { delete v; }

int cfgpp_vae_load_tensor(cfgpp_vae* v, const char* key, const void* host, int dtype, const long* shape, int ndim) {
    CFGPP_REQUIRE(v && key && host && shape && !v->finalized, "vae_load_tensor: bad state/args");
    auto it = v->params.find(key);
    if (it == v->params.end()) { cfgpp_set_error("vae_load_tensor: unknown key %s", key); return -3; }
    HostParam& p = it->second;
    long n = 1; for (int i = 0; i < ndim; ++i) n *= shape[i];
    CFGPP_REQUIRE(n == p.numel(), "vae_load_tensor: %s has %ld elements, expected %ld", key, n, p.numel());
    const std::string k(key);
    const bool as_f32 = !p.is_matrix || k == "decoder.conv_in.weight" || k == "post_quant_conv.weight" ||
                        k == "encoder.conv_in.weight" || k == "quant_conv.weight";
    if (!as_f32) {
        p.h.resize(n);
        if (dtype == 0) { const float* s = (const float*)host; for (long i = 0; i < n; ++i) p.h[i] = (half_t)s[i]; }
        else std::memcpy(p.h.data(), host, n * sizeof(half_t));
    } else {
        p.f.resize(n);
        if (dtype == 0) std::memcpy(p.f.data(), host, n * sizeof(float));
        else { const half_t* s = (const half_t*)host; for (long i = 0; i < n; ++i) p.f[i] = (float)s[i]; }
    }
    p.loaded = true;
    return 0;
}

int cfgpp_vae_finalize(cfgpp_vae* v) {
    CFGPP_REQUIRE(v && !v->finalized, "vae_finalize: bad state");
    {
        int n = 0, enc_loaded = 0, enc_total = 0; std::string names;
        for (auto& kv : v->params) {
            if (is_encoder_key(kv.first)) { ++enc_total; enc_loaded += kv.second.loaded ? 1 : 0; continue; }
            if (!kv.second.loaded) { if (n < 6) names += kv.first + " "; ++n; }
        }
        CFGPP_REQUIRE(n == 0, "vae_finalize: missing %d parameters: %s...", n, names.c_str());
        CFGPP_REQUIRE(enc_loaded == 0 || enc_loaded == enc_total, "vae_finalize: encoder partially loaded (%d of %d tensors)", enc_loaded, enc_total);
        v->has_encoder = enc_loaded == enc_total;
    }
    CFGPP_HIP_CHECK(hipSetDevice(v->device));
    Builder B{v};
    Plan PD{v, &B, &v->plan};
    Plan PE{v, &B, &v->enc_plan};
    const int R = v->max_rows, C = v->ch[3];
    int H = v->h, W = v->w;
    const int T = H * W;
    v->d_gn_stats = (float*)v->dmalloc((size_t)R * (1024 * 64 * 2 + 64 * 2) * sizeof(float));
    v->tok_a = (half_t*)v->dmalloc((size_t)R * T * C * 2);
    v->tok_o = (half_t*)v->dmalloc((size_t)R * T * C * 2);
    v->hq = (half_t*)v->dmalloc((size_t)R * T * C * 2);
    v->hk = (half_t*)v->dmalloc((size_t)R * T * C * 2);
    v->hvt = (half_t*)v->dmalloc((size_t)R * T * C * 2);
    v->smat = (half_t*)v->dmalloc((size_t)R * T * T * 2);
    CFGPP_REQUIRE(v->d_gn_stats && v->tok_a && v->tok_o && v->hq && v->hk && v->hvt && v->smat, "vae_finalize: hipMalloc failed");

    auto resblock = [&](Plan& P, const std::string& p, const Tensor& x, int cout) {
        const int cin = x.C;
        float* n1g = B.f32(p + ".norm1.weight"); float* n1b = B.f32(p + ".norm1.bias");
        half_t* w1 = B.conv3(p + ".conv1.weight"); float* b1 = B.f32(p + ".conv1.bias");
        float* n2g = B.f32(p + ".norm2.weight"); float* n2b = B.f32(p + ".norm2.bias");
        half_t* w2 = B.conv3(p + ".conv2.weight"); float* b2 = B.f32(p + ".conv2.bias");
        Tensor g1 = v->acq(x.H, x.W, cin);
        P.groupnorm(x, nullptr, g1.p, true, n1g, n1b, 1e-6f, true);
        Tensor h1 = v->acq(x.H, x.W, cout);
        P.conv3x3(g1, h1, w1, b1, 1, nullptr, 0, nullptr);
        v->rel(g1);
        Tensor g2 = v->acq(x.H, x.W, cout);
        P.groupnorm(h1, nullptr, g2.p, true, n2g, n2b, 1e-6f, true);
        v->rel(h1);
        Tensor out = v->acq(x.H, x.W, cout);
        if (cin != cout) {
            half_t* wsc = B.linear(p + ".conv_shortcut.weight"); float* bsc = B.f32(p + ".conv_shortcut.bias");
            Tensor sc = v->acq(x.H, x.W, cout);
            P.conv1x1(x, nullptr, sc, wsc, bsc);
            P.conv3x3(g2, out, w2, b2, 1, nullptr, 0, &sc);
            v->rel(sc);
        } else {
            P.conv3x3(g2, out, w2, b2, 1, nullptr, 0, &x);
        }
        v->rel(g2);
        return out;
    };

    // mid block: Res, GroupNorm -> one 512-wide attention head over H*W tokens -> +x, Res   (decoder and encoder)
    auto mid_block = [&](Plan& P, const std::string& pfx, Tensor x) -> Tensor {
        const bool tagged = P.ops == &v->plan;
        Tensor y = resblock(P, pfx + ".resnets.0", x, C);
        v->rel(x);
        const std::string ap = pfx + ".attentions.0";
        float* ng = B.f32(ap + ".group_norm.weight"); float* nb = B.f32(ap + ".group_norm.bias");
        half_t* wqkv = B.concat({ap + ".to_q.weight", ap + ".to_k.weight", ap + ".to_v.weight"});
        std::vector<float> bb;
        for (const char* n : {".to_q.bias", ".to_k.bias", ".to_v.bias"}) {
            HostParam* hp = B.get(ap + n);
            if (hp) bb.insert(bb.end(), hp->f.begin(), hp->f.end());
        }
        float* bqkv = B.upload(bb);
        half_t* wo = B.linear(ap + ".to_out.0.weight"); float* bo = B.f32(ap + ".to_out.0.bias");
        P.groupnorm(y, nullptr, v->tok_a, false, ng, nb, 1e-6f, false);
        {   // QKV projection -> head-major (1 head, d = 512)
            IGemmArgs a = base_args(v);
            a.a0 = v->tok_a; a.C0 = C; a.amode = 0; a.w = wqkv; a.N = 3 * C; a.K = C; a.bias = bqkv; a.epi = EPI_HEADS;
            a.rows_per_batch = T; a.hq = v->hq; a.hk = v->hk; a.hvt = v->hvt; a.part0 = 0; a.part_width = C;
            a.head_dim = C; a.head_dim_pad = C; a.heads = 1; a.tok_pad = T; a.q_tok_pad = T;
            a.vt_linear = 1;        // V^T is the weight operand of the P V GEMM below: natural key order
            v->macs_per_row += (double)T * 3 * C * C;
            P.ops->push_back([a, T](hipStream_t s, int rows) mutable { IGemmArgs b = a; b.M = rows * T; return igemm_launch(b, s); });
            if (tagged) v->tag(0, (double)T * 3 * C * C, "vae qkv");
        }
        {   // S = Q K^T / sqrt(C), softmax, O = P V   (per image)
            cfgpp_vae* vv = v; const float scale = 1.0f / sqrtf((float)C);
            P.ops->push_back([=](hipStream_t s, int rows) {
                for (int b = 0; b < rows; ++b) {
                    IGemmArgs a = base_args(vv);      // (the engine's K-split workspace exists since the plan was built)
                    a.a0 = vv->hq + (size_t)b * T * C; a.C0 = C; a.amode = 0; a.w = vv->hk + (size_t)b * T * C; a.M = T; a.N = T; a.K = C;
                    a.out = vv->smat + (size_t)b * T * T; a.omode = 0; a.old = T; a.epi = EPI_STORE; a.out_scale = scale; a.rows_per_batch = T;
                    int e = igemm_launch(a, s); if (e) return e;
                }
                return 0;
            });
            if (tagged) v->tag(1, (double)T * T * C, "vae attn QK^T");
            P.ops->push_back([=](hipStream_t s, int rows) { return cfgpp_op_softmax_rows(vv->smat, (long)rows * T, T, s); });
            if (tagged) v->tag(2, 0.0, "vae softmax");
            P.ops->push_back([=](hipStream_t s, int rows) {
                for (int b = 0; b < rows; ++b) {
                    IGemmArgs a = base_args(vv);
                    a.a0 = vv->smat + (size_t)b * T * T; a.C0 = T; a.amode = 0; a.w = vv->hvt + (size_t)b * C * T; a.M = T; a.N = C; a.K = T;
                    a.out = vv->tok_o + (size_t)b * T * C; a.omode = 0; a.old = C; a.epi = EPI_STORE; a.rows_per_batch = T;
                    int e = igemm_launch(a, s); if (e) return e;
                }
                return 0;
            });
            if (tagged) v->tag(1, (double)T * T * C, "vae attn PV");
            v->attn_macs_per_row += 2.0 * T * (double)T * C;
        }
        Tensor z2 = v->acq(y.H, y.W, C);
        P.linear_to_padded(v->tok_o, C, z2, wo, bo, y);
        v->rel(y);
        Tensor out = resblock(P, pfx + ".resnets.1", z2, C);
        v->rel(z2);
        return out;
    };

    // ======================= decoder plan =======================
    Tensor x = v->acq(H, W, C);
    {   // conv_in with z/scale and post_quant_conv folded into the gather
        HostParam* pw = B.get("decoder.conv_in.weight"); HostParam* pb = B.get("decoder.conv_in.bias");
        HostParam* qw = B.get("post_quant_conv.weight"); HostParam* qb = B.get("post_quant_conv.bias");
        CFGPP_REQUIRE(pw && pb && qw && qb, "vae_finalize: %s", B.err.c_str());
        std::vector<float> r((size_t)36 * C);
        for (int o = 0; o < C; ++o) for (int i = 0; i < 4; ++i) for (int t = 0; t < 9; ++t)
            r[(size_t)(t * 4 + i) * C + o] = pw->f[((size_t)o * 4 + i) * 9 + t];
        float* dw = B.upload(r); float* db = B.upload(pb->f);
        float* dqw = B.upload(qw->f); float* dqb = B.upload(qb->f);
        cfgpp_vae* vv = v; half_t* xp = x.p; const int HH = H, WW = W; const float inv = 1.0f / v->scaling;
        v->plan.push_back([=](hipStream_t s, int rows) {
            return cfgpp_op_conv_in_ex(vv->in_z, 0, xp, dw, db, rows, rows, 4, HH, WW, C, dqw, dqb, inv, s);
        });
        v->tag(3, 0.0, "vae conv_in");
        v->macs_per_row += (double)H * W * C * 36.0;
    }
    x = mid_block(PD, "decoder.mid_block", x);
    for (int i = 0; i < 4; ++i) {
        const int co = v->ch[3 - i];
        const std::string p = "decoder.up_blocks." + std::to_string(i);
        for (int j = 0; j < 3; ++j) {
            Tensor y = resblock(PD, p + ".resnets." + std::to_string(j), x, co);
            v->rel(x); x = y;
        }
        if (i != 3) {
            half_t* wu = B.conv3(p + ".upsamplers.0.conv.weight"); float* bu = B.f32(p + ".upsamplers.0.conv.bias");
            H *= 2; W *= 2;
            Tensor y = v->acq(H, W, co);
            PD.conv3x3(x, y, wu, bu, 3, nullptr, 0, nullptr);
            v->rel(x); x = y;
        }
    }
    {   // GroupNorm + SiLU -> conv_out (3 channels, fp32 NCHW)
        float* g = B.f32("decoder.conv_norm_out.weight"); float* b = B.f32("decoder.conv_norm_out.bias");
        Tensor gn = v->acq(H, W, x.C);
        PD.groupnorm(x, nullptr, gn.p, true, g, b, 1e-6f, true);
        HostParam* pw = B.get("decoder.conv_out.weight"); float* bo = B.f32("decoder.conv_out.bias");
        CFGPP_REQUIRE(pw, "vae_finalize: %s", B.err.c_str());
        const int Cc = x.C;
        std::vector<half_t> r((size_t)4 * 9 * Cc, (half_t)0.f);      // padded to 4 output rows
        for (int o = 0; o < 3; ++o) for (int i = 0; i < Cc; ++i) for (int t = 0; t < 9; ++t)
            r[((size_t)o * 9 + t) * Cc + i] = pw->h[((size_t)o * Cc + i) * 9 + t];
        half_t* dw = B.upload(r);
        cfgpp_vae* vv = v; half_t* gp = gn.p; const int HH = H, WW = W;
        v->macs_per_row += (double)H * W * 3 * 9.0 * Cc;
        v->plan.push_back([=](hipStream_t s, int rows) {
            return cfgpp_op_conv_out_ex(gp, vv->out_img, 0, dw, bo, rows, HH, WW, Cc, 3, vv->post_scale, vv->post_shift, vv->post_clamp, s);
        });
        v->tag(3, 0.0, "vae conv_out");
        v->rel(gn); v->rel(x);
    }
    v->dec_macs = v->macs_per_row + v->attn_macs_per_row;

    // ======================= encoder plan =======================
    if (v->has_encoder) {
        int EH = 8 * v->h, EW = 8 * v->w;
        const int c0 = v->ch[0];
        Tensor e = v->acq(EH, EW, c0);
        {   // conv_in: fp32 NCHW image [B,3,8h,8w] -> padded NHWC fp16 (input rounded to fp16 like the fp16 pipe)
            HostParam* pw = B.get("encoder.conv_in.weight"); HostParam* pb = B.get("encoder.conv_in.bias");
            CFGPP_REQUIRE(pw && pb, "vae_finalize: %s", B.err.c_str());
            std::vector<float> r((size_t)27 * c0);
            for (int o = 0; o < c0; ++o) for (int i = 0; i < 3; ++i) for (int t = 0; t < 9; ++t)
                r[(size_t)(t * 3 + i) * c0 + o] = pw->f[((size_t)o * 3 + i) * 9 + t];
            float* dw = B.upload(r); float* db = B.upload(pb->f);
            cfgpp_vae* vv = v; half_t* ep = e.p; const int HH = EH, WW = EW;
            v->enc_plan.push_back([=](hipStream_t s, int rows) {
                return cfgpp_op_conv_in_ex(vv->in_img, 0, ep, dw, db, rows, rows, 3, HH, WW, c0, nullptr, nullptr, 1.0f, s);
            });
            v->macs_per_row += (double)EH * EW * c0 * 27.0;
        }
        for (int i = 0; i < 4; ++i) {
            const int co = v->ch[i];
            const std::string p = "encoder.down_blocks." + std::to_string(i);
            for (int j = 0; j < 2; ++j) {
                Tensor y = resblock(PE, p + ".resnets." + std::to_string(j), e, co);
                v->rel(e); e = y;
            }
            if (i != 3) {   // F.pad(x, (0,1,0,1)) + conv3x3 stride 2 pad 0  ==  stride-2 gather shifted by one pixel
                half_t* wd = B.conv3(p + ".downsamplers.0.conv.weight"); float* bd = B.f32(p + ".downsamplers.0.conv.bias");
                EH /= 2; EW /= 2;
                Tensor y = v->acq(EH, EW, co);
                PE.conv3x3(e, y, wd, bd, 2, nullptr, 0, nullptr, /*ashift=*/1);
                v->rel(e); e = y;
            }
        }
        e = mid_block(PE, "encoder.mid_block", e);
        {   // GroupNorm + SiLU -> conv_out (8 moments channels, fp32) -> quant_conv + posterior
            float* g = B.f32("encoder.conv_norm_out.weight"); float* b = B.f32("encoder.conv_norm_out.bias");
            Tensor gn = v->acq(EH, EW, C);
            PE.groupnorm(e, nullptr, gn.p, true, g, b, 1e-6f, true);
            HostParam* pw = B.get("encoder.conv_out.weight"); float* bo = B.f32("encoder.conv_out.bias");
            HostParam* qw = B.get("quant_conv.weight"); HostParam* qb = B.get("quant_conv.bias");
            CFGPP_REQUIRE(pw && qw && qb, "vae_finalize: %s", B.err.c_str());
            std::vector<half_t> r((size_t)8 * 9 * C);
            for (int o = 0; o < 8; ++o) for (int i = 0; i < C; ++i) for (int t = 0; t < 9; ++t)
                r[((size_t)o * 9 + t) * C + i] = pw->h[((size_t)o * C + i) * 9 + t];
            half_t* dw = B.upload(r);
            std::vector<float> qwr(qw->f);
            for (auto& f : qwr) f = (float)(half_t)f;            // the fp16 pipe holds quant_conv in fp16
            float* dqw = B.upload(qwr); float* dqb = B.upload(qb->f);
            v->enc_co = (float*)v->dmalloc((size_t)R * 8 * T * sizeof(float));
            CFGPP_REQUIRE(v->enc_co, "vae_finalize: hipMalloc failed");
            cfgpp_vae* vv = v; half_t* gp = gn.p; const int HH = EH, WW = EW; const float sc = v->scaling;
            v->macs_per_row += (double)T * 8 * 9.0 * C;
            v->enc_plan.push_back([=](hipStream_t s, int rows) { return cfgpp_op_conv_out(gp, vv->enc_co, 0, dw, bo, rows, HH, WW, C, 8, s); });
            v->enc_plan.push_back([=](hipStream_t s, int rows) {
                return cfgpp_op_vae_posterior(vv->enc_co, dqw, dqb, vv->in_noise, vv->out_z, vv->out_moments, rows, T, sc, s);
            });
            v->rel(gn); v->rel(e);
        }
        v->enc_macs = v->macs_per_row + v->attn_macs_per_row - v->dec_macs;
    }
    CFGPP_REQUIRE(B.ok, "vae_finalize: %s", B.err.c_str());
    CFGPP_HIP_CHECK(hipDeviceSynchronize());
    v->plan_kind.resize(v->plan.size(), 3); v->plan_macs.resize(v->plan.size(), 0.0); v->plan_desc.resize(v->plan.size());
    v->finalized = true;
    return 0;
}

// img[B][3][8h][8w] fp32 = decoder(post_quant_conv(z / scaling_factor)),  z [B][4][h][w] fp32
static int vae_decode_impl(cfgpp_vae* v, const void* z, void* img, int B, void* stream);

int cfgpp_vae_decode(cfgpp_vae* v, const void* z, void* img, int B, void* stream) {
    if (v) { v->post_scale = 1.0f; v->post_shift = 0.0f; v->post_clamp = 0; }
    return vae_decode_impl(v, z, img, B, stream);
}

// decode + the sampler's `(img / 2 + 0.5).clamp(0, 1)` (latent_diffusion.py:676-677, latent_sdxl.py:274-275) folded
// into the last kernel: the image leaves the GPU ready for `.cpu()`
int cfgpp_vae_decode_image(cfgpp_vae* v, const void* z, void* img, int B, void* stream) {
    if (v) { v->post_scale = 0.5f; v->post_shift = 0.5f; v->post_clamp = 1; }
    return vae_decode_impl(v, z, img, B, stream);
}

static int vae_decode_impl(cfgpp_vae* v, const void* z, void* img, int B, void* stream) {
    CFGPP_REQUIRE(v && v->finalized && z && img && B > 0 && B <= v->max_rows, "vae_decode: bad args (B=%d, max %d)", B, v ? v->max_rows : 0);
    v->in_z = z; v->out_img = img;
    if (v->tuned_rows != B && igemm_autotune_enabled()) { int e = v->tune_plan((hipStream_t)stream, B); if (e) return e; }
    for (auto& op : v->plan) { int e = op((hipStream_t)stream, B); if (e) return e; }
    return 0;
}

// z[B][4][h][w] fp32 = (mean + std * noise) * scaling_factor of the posterior of img[B][3][8h][8w] fp32;
// noise[B][4][h][w] fp32 (null = posterior mean); moments[B][8][h][w] fp32 (mean | clamped logvar) optional.
int cfgpp_vae_encode(cfgpp_vae* v, const void* img, const void* noise, void* z, void* moments, int B, void* stream) {
    CFGPP_REQUIRE(v && v->finalized && img && z && B > 0 && B <= v->max_rows, "vae_encode: bad args (B=%d, max %d)", B, v ? v->max_rows : 0);
    CFGPP_REQUIRE(v->has_encoder, "vae_encode: the encoder.* / quant_conv.* weights were not loaded");
    v->in_img = img; v->in_noise = (const float*)noise; v->out_z = (float*)z; v->out_moments = (float*)moments;
    for (auto& op : v->enc_plan) { int e = op((hipStream_t)stream, B); if (e) return e; }
    return 0;
}

// One decode with a HIP event between every launch of the decoder plan (the VAE's twin of cfgpp_unet_profile): one line per
// launch in `detail` - index, family (0 igemm, 1 attention GEMMs, 2 norm / softmax, 3 small), description, us, GFLOP.
int cfgpp_vae_profile(cfgpp_vae* v, const void* z, void* img, int B, void* stream, char* detail, long detail_cap) {
    CFGPP_REQUIRE(v && v->finalized && z && img && detail && detail_cap > 0 && B > 0 && B <= v->max_rows, "vae_profile: bad args");
    v->in_z = z; v->out_img = img; v->post_scale = 0.5f; v->post_shift = 0.5f; v->post_clamp = 1;
    hipStream_t s = (hipStream_t)stream;
    if (v->tuned_rows != B && igemm_autotune_enabled()) { int e = v->tune_plan(s, B); if (e) return e; }
    const size_t n = v->plan.size();
    std::vector<hipEvent_t> ev(n + 1);
    for (auto& e : ev) CFGPP_HIP_CHECK(hipEventCreate(&e));
    CFGPP_HIP_CHECK(hipEventRecord(ev[0], s));
    int rc = 0;
    for (size_t i = 0; i < n && rc == 0; ++i) { rc = v->plan[i](s, B); if (rc == 0 && hipEventRecord(ev[i + 1], s) != hipSuccess) rc = -1; }
    if (rc == 0 && hipStreamSynchronize(s) != hipSuccess) rc = -1;
    if (rc == 0) {
        std::string txt;
        for (size_t i = 0; i < n; ++i) {
            float ms = 0.f; hipEventElapsedTime(&ms, ev[i], ev[i + 1]);
            char line[256];
            snprintf(line, sizeof(line), "%zu\t%d\t%s\t%.1f\t%.3f\n", i, v->plan_kind[i], v->plan_desc[i].c_str(), ms * 1e3,
                     2.0 * v->plan_macs[i] * B * 1e-9);
            txt += line;
        }
        const long ncopy = std::min<long>((long)txt.size(), detail_cap - 1);
        std::memcpy(detail, txt.data(), ncopy); detail[ncopy] = 0;
    }
    for (auto& e : ev) hipEventDestroy(e);
    return rc;
}

double cfgpp_vae_flops(cfgpp_vae* v, int B) { return v && v->finalized ? 2.0 * v->dec_macs * B : 0.0; }
double cfgpp_vae_encode_flops(cfgpp_vae* v, int B) { return v && v->finalized ? 2.0 * v->enc_macs * B : 0.0; }
double cfgpp_vae_device_bytes(cfgpp_vae* v) { return v ? v->dev_bytes : 0.0; }

}  // extern "C"
