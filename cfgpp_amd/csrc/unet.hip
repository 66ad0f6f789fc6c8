// UNet engine: the MI355X-native replacement of the reference's
//     self.unet(z_in, t_in, encoder_hidden_states=c_embed[, added_cond_kwargs])['sample']
// (latent_diffusion.py:146-156, latent_sdxl.py:170-183).
//
// Architecture follows diffusers-0.27.1 UNet2DConditionModel semantics for the
// SD1.5 / SDXL configs: conv_in, time (+ text_time) embedding,
// {ResnetBlock2D, Transformer2DModel} down / mid / up blocks with skip concat,
// GroupNorm-SiLU-conv_out.  Weights are loaded by their diffusers state-dict keys
// and repacked once into the layouts the kernels want; the forward is a static
// launch plan (vector of closures) over preallocated, shape-keyed activation
// buffers in halo-padded NHWC fp16 - no allocation, no host sync, one stream.
#include "engine_base.h"

struct cfgpp_unet : EngineBase {
    cfgpp_unet_config cfg;
    int device = 0;
    bool finalized = false;

    std::vector<Op> ctx_plan;       // set_context (cross-attention K/V, added-condition embedding)
    // forward-time inputs (pointers patched per call)
    const void* in_z = nullptr; int in_z_half = 0; int in_z_rows = 0; float in_t = 0.f; void* out_eps = nullptr;
    const float* in_t_dev = nullptr;        // non-null: the timestep sinusoid reads *in_t_dev (graph replay) instead of in_t
    // whole-step graph replay (cfgpp_sample_graph_ddim): per-step scalar table, current-step block, step counter, the one cached graph
    float* d_step_tab = nullptr; int step_tab_cap = 0; float* d_step_cur = nullptr; int* d_step_idx = nullptr;
    hipStream_t cap_stream = nullptr;
    struct GraphKey { const void* z; void* z0t; void* eps; const void* euc; const void* ec; int z_half, z_rows, rows, tw, rn; float lam; long n; int tuned_serial; };
    struct Graph { GraphKey key; hipGraph_t graph; hipGraphExec_t exec; };
    std::vector<Graph> graphs;              // most recently used last; at most 4 (an invert + edit job alternates between two)
    int tuned_serial = 0;                   // bumps whenever the pins of the current batch change (a graph bakes the tiles it captured)
    static void destroy(Graph& g) { if (g.exec) hipGraphExecDestroy(g.exec); if (g.graph) hipGraphDestroy(g.graph); g.exec = nullptr; g.graph = nullptr; }
    void drop_graphs() { for (auto& g : graphs) destroy(g); graphs.clear(); }
    ~cfgpp_unet() { drop_graphs(); if (cap_stream) hipStreamDestroy(cap_stream); }
    // context inputs
    const half_t* ctx_ehs = nullptr; int ctx_rows = 0; int ctx_tokens = 77;
    const half_t* ctx_text = nullptr; const float* ctx_tids = nullptr; int ctx_cond_rows = 0;
    bool ctx_set = false;

    // time-embedding scratch
    float* d_sin_t = nullptr; float* d_emb_h = nullptr; float* d_emb_t = nullptr; float* d_emb = nullptr;
    float* d_temb_all = nullptr; int temb_total = 0;
    float* d_add_in = nullptr; float* d_add_h = nullptr; float* d_aug = nullptr;

    // transformer scratch (token-major)
    half_t *tok_x = nullptr, *tok_ln = nullptr, *tok_attn = nullptr, *tok_ff = nullptr;
    // head-major Q / K / V^T scratch, ONE SET PER LEVEL: a level has fixed (heads, tokens, d), so the
    // zero padding of the head dim (d..dp) is never overwritten by a differently shaped user.
    half_t *hq[4] = {nullptr, nullptr, nullptr, nullptr}, *hk[4] = {nullptr, nullptr, nullptr, nullptr},
           *hvt[4] = {nullptr, nullptr, nullptr, nullptr};

};

namespace {

void build_param_table(cfgpp_unet* u) {
    const cfgpp_unet_config& c = u->cfg;
    const int L = c.num_levels;
    const long c0 = c.block_out_channels[0], temb = 4 * c0;
    expect_conv(u, "conv_in", c0, c.in_channels, 3);
    expect_linear(u, "time_embedding.linear_1", temb, c0);
    expect_linear(u, "time_embedding.linear_2", temb, temb);
    if (c.addition_embed) {
        const long in = (long)c.addition_time_embed_dim * 6 + c.addition_pooled_dim;
        expect_linear(u, "add_embedding.linear_1", temb, in);
        expect_linear(u, "add_embedding.linear_2", temb, temb);
    }
    long ch = c0;
    for (int i = 0; i < L; ++i) {
        const long co = c.block_out_channels[i];
        const std::string p = "down_blocks." + std::to_string(i);
        for (int j = 0; j < c.layers_per_block; ++j) {
            expect_resnet(u, p + ".resnets." + std::to_string(j), ch, co, temb);
            if (c.level_has_attn[i])
                expect_transformer(u, p + ".attentions." + std::to_string(j), co, c.transformer_depth[i], c.cross_attention_dim);
            ch = co;
        }
        if (i != L - 1) expect_conv(u, p + ".downsamplers.0.conv", co, co, 3);
    }
    const long cm = c.block_out_channels[L - 1];
    expect_resnet(u, "mid_block.resnets.0", cm, cm, temb);
    expect_transformer(u, "mid_block.attentions.0", cm, c.transformer_depth[L - 1], c.cross_attention_dim);
    expect_resnet(u, "mid_block.resnets.1", cm, cm, temb);
    // up blocks (diffusers: reversed channels; layers_per_block+1 resnets each)
    long prev = cm;
    for (int i = 0; i < L; ++i) {
        const int lvl = L - 1 - i;
        const long co = c.block_out_channels[lvl];
        const long cin_lvl = c.block_out_channels[std::max(lvl - 1, 0)];
        const std::string p = "up_blocks." + std::to_string(i);
        for (int j = 0; j < c.layers_per_block + 1; ++j) {
            const long skip = (j == c.layers_per_block) ? cin_lvl : co;
            const long rin = (j == 0 ? prev : co) + skip;
            expect_resnet(u, p + ".resnets." + std::to_string(j), rin, co, temb);
            if (c.level_has_attn[lvl])
                expect_transformer(u, p + ".attentions." + std::to_string(j), co, c.transformer_depth[lvl], c.cross_attention_dim);
        }
        if (i != L - 1) expect_conv(u, p + ".upsamplers.0.conv", co, co, 3);
        prev = co;
    }
    expect_norm(u, "conv_norm_out", c0);
    expect_conv(u, "conv_out", c.out_channels, c0, 3);
}

// ---------------------------------------------------------------------------
// weight repacking (host) + upload
// ---------------------------------------------------------------------------
struct ResW {
    float *n1g, *n1b, *n2g, *n2b, *b1, *b2, *bsc;
    half_t *w1, *w2, *wsc;
    int temb_off;   // column offset in temb_all
    int cin, cout;
};

}  // namespace

// ---------------------------------------------------------------------------
extern "C" {

cfgpp_unet* cfgpp_unet_create(const cfgpp_unet_config* cfg, int device_id) {
    if (!cfg) { cfgpp_set_error("unet_create: null config"); return nullptr; }
    if (cfg->num_levels < 1 || cfg->num_levels > 4 || cfg->layers_per_block < 1 || cfg->max_rows < 1) {
        cfgpp_set_error("unet_create: bad config"); return nullptr;
    }
    for (int i = 0; i < cfg->num_levels; ++i) {
        const int c = cfg->block_out_channels[i];
        if (c % 64 != 0 || c % cfg->norm_groups != 0) { cfgpp_set_error("unet_create: channels %d must be a multiple of 64 and of norm_groups", c); return nullptr; }
        if (cfg->level_has_attn[i]) {
            const int d = c / cfg->num_heads[i];
            if (c % cfg->num_heads[i] != 0 || d % 4 != 0 || d > 160) { cfgpp_set_error("unet_create: head dim %d unsupported", d); return nullptr; }
        }
    }
    if (cfg->cross_attention_dim % 64 != 0) { cfgpp_set_error("unet_create: cross_attention_dim must be a multiple of 64"); return nullptr; }
    if ((cfg->sample_h % (1 << (cfg->num_levels - 1))) || (cfg->sample_w % (1 << (cfg->num_levels - 1)))) {
        cfgpp_set_error("unet_create: sample size must be divisible by 2^(levels-1)"); return nullptr;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= device_id) {
        cfgpp_set_error("unet_create: no HIP device %d (found %d) - the HIP path has no CPU fallback", device_id, ndev);
        return nullptr;
    }
    if (cfgpp_claim_device(device_id)) return nullptr;
    if (hipSetDevice(device_id) != hipSuccess) { cfgpp_set_error("unet_create: hipSetDevice failed"); return nullptr; }
    cfgpp_unet* u = new cfgpp_unet();
    u->cfg = *cfg; u->device = device_id; u->max_rows = cfg->max_rows; u->norm_groups = cfg->norm_groups;
    build_param_table(u);
    return u;
}

void cfgpp_unet_destroy(cfgpp_unet* u) {
    if (!u) return;
    delete u;
}

int cfgpp_unet_load_tensor(cfgpp_unet* u, const char* key, const void* host, int dtype, const long* shape, int ndim) {
    CFGPP_REQUIRE(u && key && host && shape, "load_tensor: null argument");
    CFGPP_REQUIRE(!u->finalized, "load_tensor: context already finalized");
    auto it = u->params.find(key);
    if (it == u->params.end()) { cfgpp_set_error("load_tensor: unknown key %s", key); return -3; }
    HostParam& p = it->second;
    long n = 1; for (int i = 0; i < ndim; ++i) n *= shape[i];
    CFGPP_REQUIRE(n == p.numel(), "load_tensor: %s has %ld elements, expected %ld", key, n, p.numel());
    if (ndim == 4) { p.shape.assign(shape, shape + 4); }   // keep OIHW (conv1x1 vs linear both fine)
    if (p.is_matrix && !(std::string(key) == "conv_in.weight")) {
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

int cfgpp_unet_missing(cfgpp_unet* u) {
    if (!u) return -1;
    int n = 0; std::string names;
    for (auto& kv : u->params) if (!kv.second.loaded) { if (n < 8) names += kv.first + " "; ++n; }
    if (n) cfgpp_set_error("missing %d parameters: %s...", n, names.c_str());
    return n;
}

int cfgpp_unet_finalize(cfgpp_unet* u) {
    CFGPP_REQUIRE(u, "finalize: null");
    CFGPP_REQUIRE(!u->finalized, "finalize: already finalized");
    if (cfgpp_unet_missing(u) != 0) { const std::string m = cfgpp_last_error(); cfgpp_set_error("finalize: %s", m.c_str()); return -2; }
    CFGPP_HIP_CHECK(hipSetDevice(u->device));
    const cfgpp_unet_config& c = u->cfg;
    const int L = c.num_levels, R = c.max_rows;
    const int c0 = c.block_out_channels[0], temb_dim = 4 * c0;
    Builder B{u};
    Plan P{u, &B, &u->plan};
    Plan PC{u, &B, &u->ctx_plan};

    // ---- scratch sizing ----
    long max_tok_c = 0, max_ff = 0;
    {
        int H = c.sample_h, W = c.sample_w;
        for (int i = 0; i < L; ++i) {
            const long C = c.block_out_channels[i];
            if (c.level_has_attn[i] || i == L - 1) {
                const long tok = (long)H * W;
                const int d = (int)(C / c.num_heads[i]), dp = round_up(d, 32);
                max_tok_c = std::max(max_tok_c, tok * C);
                max_ff = std::max(max_ff, tok * 4 * C);
                u->hq[i] = (half_t*)u->dmalloc((size_t)R * c.num_heads[i] * round_up((int)tok, 128) * dp * 2);
                u->hk[i] = (half_t*)u->dmalloc((size_t)R * c.num_heads[i] * round_up((int)tok, 128) * dp * 2);
                u->hvt[i] = (half_t*)u->dmalloc((size_t)R * c.num_heads[i] * dp * round_up((int)tok, 64) * 2);
                CFGPP_REQUIRE(u->hq[i] && u->hk[i] && u->hvt[i], "finalize: hipMalloc failed");
                if (cfgpp_op_attention_prepare_vt(u->hvt[i], R * c.num_heads[i], d, round_up((int)tok, 64), nullptr)) return -1;
            }
            if (i != L - 1) { H /= 2; W /= 2; }
        }
    }
    u->tok_x = (half_t*)u->dmalloc((size_t)R * max_tok_c * 2);
    u->tok_ln = (half_t*)u->dmalloc((size_t)R * max_tok_c * 2);
    u->tok_attn = (half_t*)u->dmalloc((size_t)R * max_tok_c * 2);
    u->tok_ff = (half_t*)u->dmalloc((size_t)R * max_ff * 2);
    u->d_gn_stats = (float*)u->dmalloc((size_t)R * (1024 * 64 * 2 + 64 * 2) * sizeof(float));
    u->d_sin_t = (float*)u->dmalloc((size_t)c0 * sizeof(float));
    u->d_emb_h = (float*)u->dmalloc((size_t)temb_dim * sizeof(float));
    u->d_emb_t = (float*)u->dmalloc((size_t)temb_dim * sizeof(float));
    u->d_emb = (float*)u->dmalloc((size_t)R * temb_dim * sizeof(float));
    CFGPP_REQUIRE(u->tok_x && u->tok_ln && u->tok_attn && u->tok_ff, "finalize: hipMalloc failed");

    // ---- collect every resnet's time_emb_proj into one [sumC][temb] matrix ----
    std::vector<std::string> res_names;
    {
        for (int i = 0; i < L; ++i)
            for (int j = 0; j < c.layers_per_block; ++j) res_names.push_back("down_blocks." + std::to_string(i) + ".resnets." + std::to_string(j));
        res_names.push_back("mid_block.resnets.0"); res_names.push_back("mid_block.resnets.1");
        for (int i = 0; i < L; ++i)
            for (int j = 0; j < c.layers_per_block + 1; ++j) res_names.push_back("up_blocks." + std::to_string(i) + ".resnets." + std::to_string(j));
    }
    std::map<std::string, int> temb_off;
    half_t* w_temb_all = nullptr; float* b_temb_all = nullptr;
    {
        std::vector<half_t> wa; std::vector<float> ba; int off = 0;
        for (auto& rn : res_names) {
            HostParam* pw = B.get(rn + ".time_emb_proj.weight"); HostParam* pb = B.get(rn + ".time_emb_proj.bias");
            CFGPP_REQUIRE(pw && pb, "finalize: %s", B.err.c_str());
            temb_off[rn] = off; off += (int)pw->shape[0];
            wa.insert(wa.end(), pw->h.begin(), pw->h.end()); ba.insert(ba.end(), pb->f.begin(), pb->f.end());
            B.drop(rn + ".time_emb_proj.weight"); B.drop(rn + ".time_emb_proj.bias");
        }
        u->temb_total = off;
        w_temb_all = B.upload(wa); b_temb_all = B.upload(ba);
        u->d_temb_all = (float*)u->dmalloc((size_t)R * off * sizeof(float));
    }
    const bool per_row_temb = c.addition_embed != 0;
    const int temb_ld = per_row_temb ? u->temb_total : 0;

    // ---- time embedding ops (head of the forward plan) ----
    {
        half_t* w1 = B.linear("time_embedding.linear_1.weight"); float* b1 = B.f32("time_embedding.linear_1.bias");
        half_t* w2 = B.linear("time_embedding.linear_2.weight"); float* b2 = B.f32("time_embedding.linear_2.bias");
        cfgpp_unet* uu = u;
        u->plan.push_back([=](hipStream_t s, int) { return cfgpp_op_sinusoid(uu->in_t_dev, uu->in_t, uu->d_sin_t, 1, c0, c0, 0, s); });
        u->plan.push_back([=](hipStream_t s, int) { return cfgpp_op_skinny_gemm(uu->d_sin_t, c0, w1, b1, nullptr, 0, uu->d_emb_h, temb_dim, 1, temb_dim, c0, 0, 1, s); });
        if (per_row_temb) {
            // emb[r] = linear_2(h) + aug[r or 0]
            u->plan.push_back([=](hipStream_t s, int rows) {
                const int add_ld = uu->ctx_cond_rows == 1 ? 0 : temb_dim;
                return cfgpp_op_skinny_gemm(uu->d_emb_h, 0, w2, b2, uu->d_aug, add_ld, uu->d_emb, temb_dim, rows, temb_dim, temb_dim, 0, 0, s);
            });
            u->plan.push_back([=](hipStream_t s, int rows) {
                return cfgpp_op_skinny_gemm(uu->d_emb, temb_dim, w_temb_all, b_temb_all, nullptr, 0, uu->d_temb_all, uu->temb_total, rows, uu->temb_total, temb_dim, 1, 0, s);
            });
        } else {
            u->plan.push_back([=](hipStream_t s, int) { return cfgpp_op_skinny_gemm(uu->d_emb_h, temb_dim, w2, b2, nullptr, 0, uu->d_emb_t, temb_dim, 1, temb_dim, temb_dim, 0, 0, s); });
            u->plan.push_back([=](hipStream_t s, int) {
                return cfgpp_op_skinny_gemm(uu->d_emb_t, temb_dim, w_temb_all, b_temb_all, nullptr, 0, uu->d_temb_all, uu->temb_total, 1, uu->temb_total, temb_dim, 1, 0, s);
            });
        }
        if (c.addition_embed) {
            const int tdim = c.addition_time_embed_dim, pooled = c.addition_pooled_dim, in = 6 * tdim + pooled;
            half_t* aw1 = B.linear("add_embedding.linear_1.weight"); float* ab1 = B.f32("add_embedding.linear_1.bias");
            half_t* aw2 = B.linear("add_embedding.linear_2.weight"); float* ab2 = B.f32("add_embedding.linear_2.bias");
            u->d_add_in = (float*)u->dmalloc((size_t)R * in * sizeof(float));
            u->d_add_h = (float*)u->dmalloc((size_t)R * temb_dim * sizeof(float));
            u->d_aug = (float*)u->dmalloc((size_t)R * temb_dim * sizeof(float));
            // add_in[r] = [text_embeds[r] | sinusoid(time_ids[r][0..5])]
            u->ctx_plan.push_back([=](hipStream_t s, int) { return cfgpp_op_f16_to_f32_rows(uu->ctx_text, uu->d_add_in, uu->ctx_cond_rows, pooled, in, 0, s); });
            u->ctx_plan.push_back([=](hipStream_t s, int) {
                // 6*cond_rows values, each -> tdim wide, laid out contiguously after the pooled part of its row
                for (int r = 0; r < uu->ctx_cond_rows; ++r) {
                    int e = cfgpp_op_sinusoid(uu->ctx_tids + (long)r * 6, 0.f, uu->d_add_in + (long)r * in + pooled, 6, tdim, tdim, 0, s);
                    if (e) return e;
                }
                return 0;
            });
            u->ctx_plan.push_back([=](hipStream_t s, int) { return cfgpp_op_skinny_gemm(uu->d_add_in, in, aw1, ab1, nullptr, 0, uu->d_add_h, temb_dim, uu->ctx_cond_rows, temb_dim, in, 0, 1, s); });
            u->ctx_plan.push_back([=](hipStream_t s, int) { return cfgpp_op_skinny_gemm(uu->d_add_h, temb_dim, aw2, ab2, nullptr, 0, uu->d_aug, temb_dim, uu->ctx_cond_rows, temb_dim, temb_dim, 0, 0, s); });
        }
    }

    // ---- layer builders ----
    auto load_res = [&](const std::string& p, int cin, int cout) {
        ResW w{}; w.cin = cin; w.cout = cout;
        w.n1g = B.f32(p + ".norm1.weight"); w.n1b = B.f32(p + ".norm1.bias");
        w.w1 = B.conv3(p + ".conv1.weight"); w.b1 = B.f32(p + ".conv1.bias");
        w.n2g = B.f32(p + ".norm2.weight"); w.n2b = B.f32(p + ".norm2.bias");
        w.w2 = B.conv3(p + ".conv2.weight"); w.b2 = B.f32(p + ".conv2.bias");
        if (cin != cout) { w.wsc = B.linear(p + ".conv_shortcut.weight"); w.bsc = B.f32(p + ".conv_shortcut.bias"); }
        w.temb_off = temb_off[p];
        return w;
    };
    // x0 (|| x1) -> new tensor
    auto resblock = [&](const std::string& p, const Tensor& x0, const Tensor* x1, int cout) {
        const int cin = x0.C + (x1 ? x1->C : 0);
        ResW w = load_res(p, cin, cout);
        Tensor g1 = u->acq(x0.H, x0.W, cin);
        P.groupnorm(x0, x1, g1.p, true, w.n1g, w.n1b, 1e-5f, true);
        Tensor h1 = u->acq(x0.H, x0.W, cout);
        P.conv3x3(g1, h1, w.w1, w.b1, 1, u->d_temb_all + w.temb_off, temb_ld, nullptr);
        u->rel(g1);
        Tensor g2 = u->acq(x0.H, x0.W, cout);
        P.groupnorm(h1, nullptr, g2.p, true, w.n2g, w.n2b, 1e-5f, true);
        u->rel(h1);
        Tensor out = u->acq(x0.H, x0.W, cout);
        if (cin != cout) {
            Tensor sc = u->acq(x0.H, x0.W, cout);
            P.conv1x1(x0, x1, sc, w.wsc, w.bsc);
            P.conv3x3(g2, out, w.w2, w.b2, 1, nullptr, 0, &sc);
            u->rel(sc);
        } else {
            P.conv3x3(g2, out, w.w2, w.b2, 1, nullptr, 0, &x0);
        }
        u->rel(g2);
        return out;
    };
    int cross_block_counter = 0;
    auto transformer = [&](const std::string& p, const Tensor& x, int depth, int nheads, int lvl) {
        half_t* const HQ = u->hq[lvl]; half_t* const HK = u->hk[lvl]; half_t* const HVT = u->hvt[lvl];
        const int C = x.C, tok = x.H * x.W, d = C / nheads, dp = round_up(d, 32);
        const int q_pad = round_up(tok, 128), k_pad = round_up(tok, 64);
        const int ck_pad = round_up(u->ctx_tokens, 64);
        float* ng = B.f32(p + ".norm.weight"); float* nb = B.f32(p + ".norm.bias");
        half_t* wpi = B.linear(p + ".proj_in.weight"); float* bpi = B.f32(p + ".proj_in.bias");
        half_t* wpo = B.linear(p + ".proj_out.weight"); float* bpo = B.f32(p + ".proj_out.bias");
        P.groupnorm(x, nullptr, u->tok_ln, false, ng, nb, 1e-6f, false);
        P.linear(u->tok_ln, C, u->tok_x, C, wpi, bpi, nullptr, tok);
        for (int k = 0; k < depth; ++k) {
            const std::string b = p + ".transformer_blocks." + std::to_string(k);
            float* l1g = B.f32(b + ".norm1.weight"); float* l1b = B.f32(b + ".norm1.bias");
            float* l2g = B.f32(b + ".norm2.weight"); float* l2b = B.f32(b + ".norm2.bias");
            float* l3g = B.f32(b + ".norm3.weight"); float* l3b = B.f32(b + ".norm3.bias");
            half_t* wqkv = B.concat({b + ".attn1.to_q.weight", b + ".attn1.to_k.weight", b + ".attn1.to_v.weight"});
            half_t* wq2 = B.linear(b + ".attn2.to_q.weight");
            half_t* wff1 = nullptr; float* bff1 = nullptr;
            B.geglu(b + ".ff.net.0.proj", C, &wff1, &bff1);
            half_t* wo1 = B.linear(b + ".attn1.to_out.0.weight"); float* bo1 = B.f32(b + ".attn1.to_out.0.bias");
            half_t* wkv2 = B.concat({b + ".attn2.to_k.weight", b + ".attn2.to_v.weight"});
            half_t* wo2 = B.linear(b + ".attn2.to_out.0.weight"); float* bo2 = B.f32(b + ".attn2.to_out.0.bias");
            half_t* wff2 = B.linear(b + ".ff.net.2.weight"); float* bff2 = B.f32(b + ".ff.net.2.bias");
            // persistent cross-attention K / V^T of this block (filled by set_context)
            half_t* ck = (half_t*)u->dmalloc((size_t)R * nheads * ck_pad * dp * 2);
            half_t* cvt = (half_t*)u->dmalloc((size_t)R * nheads * dp * ck_pad * 2);
            if (!ck || !cvt || cfgpp_op_attention_prepare_vt(cvt, R * nheads, d, ck_pad, nullptr)) { B.ok = false; B.err = "cross-attention K/V^T allocation failed"; }
            {
                cfgpp_unet* uu = u; const int Dc = c.cross_attention_dim; const int ckp = ck_pad;
                IGemmArgs a = base_args(u);
                a.C0 = Dc; a.amode = 0; a.w = wkv2; a.N = 2 * C; a.K = Dc; a.epi = EPI_HEADS;
                a.hq = nullptr; a.hk = ck; a.hvt = cvt; a.part0 = 1; a.part_width = C; a.head_dim = d; a.head_dim_pad = dp;
                a.heads = nheads; a.tok_pad = ckp; a.q_tok_pad = ckp;
                u->ctx_plan.push_back([a, uu](hipStream_t s, int) mutable {
                    IGemmArgs b2 = a; b2.a0 = uu->ctx_ehs; b2.rows_per_batch = uu->ctx_tokens; b2.M = uu->ctx_rows * uu->ctx_tokens;
                    return igemm_launch(b2, s);
                });
            }
            ++cross_block_counter;
            // self-attention
            P.layernorm(u->tok_x, u->tok_ln, l1g, l1b, tok, C);
            P.heads(u->tok_ln, C, wqkv, 3 * C, tok, 0, C, nheads, HQ, HK, HVT, q_pad, k_pad);
            P.attention(HQ, HK, HVT, u->tok_attn, nheads, d, tok, tok, q_pad, k_pad);
            P.linear(u->tok_attn, C, u->tok_x, C, wo1, bo1, u->tok_x, tok);
            // cross-attention
            P.layernorm(u->tok_x, u->tok_ln, l2g, l2b, tok, C);
            P.heads(u->tok_ln, C, wq2, C, tok, 0, C, nheads, HQ, nullptr, nullptr, q_pad, k_pad);
            {
                cfgpp_unet* uu = u;
                u->attn_macs_per_row += 2.0 * (double)nheads * tok * 77 * d;
                half_t* hq = HQ; half_t* o = u->tok_attn;
                u->plan.push_back([=](hipStream_t s, int rows) {
                    return cfgpp_op_attention(hq, ck, cvt, o, rows, nheads, d, tok, uu->ctx_tokens, q_pad, ck_pad, s);
                });
                u->tag(1, 2.0 * (double)nheads * tok * 77 * d, "cross_attn heads=" + std::to_string(nheads) + " N=" + std::to_string(tok) + " d=" + std::to_string(d));
            }
            P.linear(u->tok_attn, C, u->tok_x, C, wo2, bo2, u->tok_x, tok);
            // feed-forward (GEGLU)
            P.layernorm(u->tok_x, u->tok_ln, l3g, l3b, tok, C);
            P.linear(u->tok_ln, C, u->tok_ff, 8 * C, wff1, bff1, nullptr, tok, EPI_GEGLU);
            P.linear(u->tok_ff, 4 * C, u->tok_x, C, wff2, bff2, u->tok_x, tok);
        }
        Tensor out = u->acq(x.H, x.W, C);
        P.linear_to_padded(u->tok_x, C, out, wpo, bpo, x);
        return out;
    };

    // ---- conv_in ----
    int H = c.sample_h, W = c.sample_w;
    Tensor x = u->acq(H, W, c0);
    {
        HostParam* pw = B.get("conv_in.weight"); HostParam* pb = B.get("conv_in.bias");
        CFGPP_REQUIRE(pw && pb, "finalize: %s", B.err.c_str());
        const int Ci = c.in_channels;
        std::vector<float> r((size_t)9 * Ci * c0);
        for (int o = 0; o < c0; ++o) for (int i = 0; i < Ci; ++i) for (int t = 0; t < 9; ++t)
            r[(size_t)(t * Ci + i) * c0 + o] = pw->f[((size_t)o * Ci + i) * 9 + t];
        float* dw = B.upload(r); float* db = B.upload(pb->f);
        cfgpp_unet* uu = u; half_t* xp = x.p; const int HH = H, WW = W;
        u->plan.push_back([=](hipStream_t s, int rows) {
            return cfgpp_op_conv_in(uu->in_z, uu->in_z_half, xp, dw, db, rows, uu->in_z_rows, Ci, HH, WW, c0, s);
        });
    }
    std::vector<Tensor> skips; skips.push_back(x);
    // ---- down ----
    for (int i = 0; i < L; ++i) {
        const int co = c.block_out_channels[i];
        const std::string p = "down_blocks." + std::to_string(i);
        for (int j = 0; j < c.layers_per_block; ++j) {
            Tensor y = resblock(p + ".resnets." + std::to_string(j), x, nullptr, co);
            if (c.level_has_attn[i]) {
                Tensor z = transformer(p + ".attentions." + std::to_string(j), y, c.transformer_depth[i], c.num_heads[i], i);
                u->rel(y); y = z;
            }
            x = y; skips.push_back(x);
        }
        if (i != L - 1) {
            half_t* wd = B.conv3(p + ".downsamplers.0.conv.weight"); float* bd = B.f32(p + ".downsamplers.0.conv.bias");
            H /= 2; W /= 2;
            Tensor y = u->acq(H, W, co);
            P.conv3x3(x, y, wd, bd, 2, nullptr, 0, nullptr);
            x = y; skips.push_back(x);
        }
    }
    // ---- mid ----
    {
        const int cm = c.block_out_channels[L - 1];
        Tensor y = resblock("mid_block.resnets.0", x, nullptr, cm);          // x is also the last skip: keep it
        Tensor z = transformer("mid_block.attentions.0", y, c.transformer_depth[L - 1], c.num_heads[L - 1], L - 1);
        u->rel(y);
        Tensor w = resblock("mid_block.resnets.1", z, nullptr, cm);
        u->rel(z);
        x = w;
    }
    // ---- up ----
    for (int i = 0; i < L; ++i) {
        const int lvl = L - 1 - i;
        const int co = c.block_out_channels[lvl];
        const std::string p = "up_blocks." + std::to_string(i);
        for (int j = 0; j < c.layers_per_block + 1; ++j) {
            Tensor skip = skips.back(); skips.pop_back();
            Tensor y = resblock(p + ".resnets." + std::to_string(j), x, &skip, co);
            u->rel(x); u->rel(skip);
            if (c.level_has_attn[lvl]) {
                Tensor z = transformer(p + ".attentions." + std::to_string(j), y, c.transformer_depth[lvl], c.num_heads[lvl], lvl);
                u->rel(y); y = z;
            }
            x = y;
        }
        if (i != L - 1) {
            half_t* wu = B.conv3(p + ".upsamplers.0.conv.weight"); float* bu = B.f32(p + ".upsamplers.0.conv.bias");
            H *= 2; W *= 2;
            Tensor y = u->acq(H, W, co);
            P.conv3x3(x, y, wu, bu, 3, nullptr, 0, nullptr);
            u->rel(x); x = y;
        }
    }
    // ---- out ----
    {
        float* g = B.f32("conv_norm_out.weight"); float* b = B.f32("conv_norm_out.bias");
        Tensor gn = u->acq(H, W, c0);
        P.groupnorm(x, nullptr, gn.p, true, g, b, 1e-5f, true);
        HostParam* pw = B.get("conv_out.weight"); float* bo = B.f32("conv_out.bias");
        CFGPP_REQUIRE(pw, "finalize: %s", B.err.c_str());
        const int Co = c.out_channels;
        CFGPP_REQUIRE(Co <= 4, "finalize: out_channels %d > 4 unsupported", Co);
        std::vector<half_t> r((size_t)Co * 9 * c0);
        for (int o = 0; o < Co; ++o) for (int i = 0; i < c0; ++i) for (int t = 0; t < 9; ++t)
            r[((size_t)o * 9 + t) * c0 + i] = pw->h[((size_t)o * c0 + i) * 9 + t];
        half_t* dw = B.upload(r);
        cfgpp_unet* uu = u; half_t* gp = gn.p; const int HH = H, WW = W;
        u->macs_per_row += (double)H * W * Co * 9.0 * c0 + (double)c.sample_h * c.sample_w * c0 * 9.0 * c.in_channels;
        u->plan.push_back([=](hipStream_t s, int rows) { return cfgpp_op_conv_out(gp, uu->out_eps, 1, dw, bo, rows, HH, WW, c0, Co, s); });
    }
    CFGPP_REQUIRE(B.ok, "finalize: %s", B.err.c_str());
    CFGPP_REQUIRE(skips.empty(), "finalize: internal error, %d skips left", (int)skips.size());
    CFGPP_HIP_CHECK(hipDeviceSynchronize());
    u->plan_kind.resize(u->plan.size(), 3); u->plan_macs.resize(u->plan.size(), 0.0); u->plan_desc.resize(u->plan.size());
    u->finalized = true;
    return 0;
}

int cfgpp_unet_set_context(cfgpp_unet* u, const void* ehs, int rows, int tokens, const void* text_embeds,
                           const void* time_ids, int cond_rows, void* stream) {
    CFGPP_REQUIRE(u && u->finalized, "set_context: context not finalized");
    CFGPP_REQUIRE(ehs && rows > 0 && rows <= u->cfg.max_rows, "set_context: rows=%d (max %d)", rows, u->cfg.max_rows);
    CFGPP_REQUIRE(tokens == 77, "set_context: tokens=%d (the engine is built for 77 text tokens)", tokens);
    if (u->cfg.addition_embed) {
        CFGPP_REQUIRE(text_embeds && time_ids, "set_context: SDXL needs text_embeds and time_ids");
        CFGPP_REQUIRE(cond_rows == rows || cond_rows == 1, "set_context: cond_rows=%d must be rows (%d) or 1", cond_rows, rows);
    }
    u->ctx_ehs = (const half_t*)ehs; u->ctx_rows = rows; u->ctx_tokens = tokens;
    u->ctx_text = (const half_t*)text_embeds; u->ctx_tids = (const float*)time_ids; u->ctx_cond_rows = cond_rows;
    for (auto& op : u->ctx_plan) { int e = op((hipStream_t)stream, rows); if (e) return e; }
    u->ctx_set = true;
    return 0;
}

int cfgpp_unet_forward(cfgpp_unet* u, const void* z, int z_is_half, int z_rows, float t, void* eps_out, int rows,
                       void* stream) {
    CFGPP_REQUIRE(u && u->finalized, "forward: context not finalized");
    CFGPP_REQUIRE(u->ctx_set, "forward: set_context has not been called");
    CFGPP_REQUIRE(z && eps_out && z_rows > 0 && rows > 0 && rows <= u->cfg.max_rows, "forward: bad args (rows=%d max=%d)", rows, u->cfg.max_rows);
    CFGPP_REQUIRE(rows == u->ctx_rows, "forward: rows=%d but context was set for %d rows", rows, u->ctx_rows);
    u->in_z = z; u->in_z_half = z_is_half; u->in_z_rows = z_rows; u->in_t = t; u->in_t_dev = nullptr; u->out_eps = eps_out;
    if (u->tuned_rows != rows && igemm_autotune_enabled()) {      // first forward at this batch: in-situ tile tuning
        int e = u->tune_plan((hipStream_t)stream, rows); if (e) return e;
        ++u->tuned_serial;
    }
    for (auto& op : u->plan) { int e = op((hipStream_t)stream, rows); if (e) return e; }
    return 0;
}

// Whole-loop graph replay: the reference's DDIM loops with callback_fn None
// (latent_diffusion.py:653-674, 272-294, 160-182; latent_sdxl.py:730-752, 838-858) as ONE captured step - UNet forward at `rows` +
// the fused generalised DDIM update - replayed n_steps times.  The per-step scalars {t, c1, c2, c3, c4} (host_steps[n_steps][5],
// the same fp32 values cfgpp_unet_forward / cfgpp_step_ddim take as arguments) go into a device table; the graph's first node
// copies the current row and advances a device counter, so one graph serves every step and every later call with the same
// buffers.  z / z0t [z_rows,4,H,W] fp32 or fp16 (updated in place / written per step), eps [rows,4,H,W] fp16 scratch the UNet
// writes, eps_uc / eps_c point into it.  Capture happens on an engine-owned stream (the caller's may be the legacy default
// stream, which cannot capture) after one eager forward (tile tuning, lazy kernel attributes); replays are enqueued on `stream`.
// Results are bit-identical to the eager loop.  Returns 0, < 0 on error (a failed capture leaves no graph behind).
int cfgpp_sample_graph_ddim(cfgpp_unet* u, void* z, void* z0t, int z_is_half, int z_rows, void* eps, const void* eps_uc,
                            const void* eps_c, int rows, const float* host_steps, int n_steps, float lam, int tweedie_uc,
                            int renoise_uc, void* stream) {
    CFGPP_REQUIRE(u && u->finalized && u->ctx_set, "sample_graph: context not ready");
    CFGPP_REQUIRE(z && z0t && eps && eps_uc && eps_c && host_steps && n_steps > 0 && z_rows > 0, "sample_graph: bad args");
    CFGPP_REQUIRE(rows == u->ctx_rows && rows <= u->cfg.max_rows, "sample_graph: rows=%d but context was set for %d rows", rows, u->ctx_rows);
    hipStream_t s = (hipStream_t)stream;
    const long n = (long)z_rows * u->cfg.in_channels * u->cfg.sample_h * u->cfg.sample_w;
    if (!u->d_step_cur) {
        u->d_step_cur = (float*)u->dmalloc(8 * sizeof(float));
        u->d_step_idx = (int*)u->dmalloc(sizeof(int));
        CFGPP_REQUIRE(u->d_step_cur && u->d_step_idx, "sample_graph: out of device memory");
        CFGPP_HIP_CHECK(hipStreamCreateWithFlags(&u->cap_stream, hipStreamNonBlocking));
    }
    if (n_steps > u->step_tab_cap) {
        u->drop_graphs();                                         // the graphs hold the old table's address
        const int cap = n_steps < 64 ? 64 : n_steps;
        u->d_step_tab = (float*)u->dmalloc((size_t)cap * 8 * sizeof(float));   // (the old table is freed with the engine)
        CFGPP_REQUIRE(u->d_step_tab, "sample_graph: out of device memory");
        u->step_tab_cap = cap;
    }
    const cfgpp_unet::GraphKey want{z, z0t, eps, eps_uc, eps_c, z_is_half, z_rows, rows, tweedie_uc, renoise_uc, lam, n, 0};
    auto same = [&](const cfgpp_unet::GraphKey& k) {
        return k.z == want.z && k.z0t == want.z0t && k.eps == want.eps && k.euc == want.euc && k.ec == want.ec && k.z_half == want.z_half &&
               k.z_rows == want.z_rows && k.rows == want.rows && k.tw == want.tw && k.rn == want.rn && k.lam == want.lam && k.n == want.n &&
               k.tuned_serial == u->tuned_serial;
    };
    int hit = -1;
    for (size_t i = 0; i < u->graphs.size(); ++i) if (same(u->graphs[i].key)) hit = (int)i;
    if (hit < 0) {
        // eager first: tile tuning (bumps tuned_serial), kernel attributes and every other lazy host-side initialisation happen
        // outside the capture.  The forward only reads z, so the loop below starts from the same state.
        int e = cfgpp_unet_forward(u, z, z_is_half, z_rows, host_steps[0], eps, rows, stream);
        if (e) return e;
        CFGPP_HIP_CHECK(hipStreamSynchronize(s));
        for (size_t i = 0; i < u->graphs.size();)                 // graphs of other pins can never hit again
            if (u->graphs[i].key.tuned_serial != u->tuned_serial) { cfgpp_unet::destroy(u->graphs[i]); u->graphs.erase(u->graphs.begin() + i); } else ++i;
        cfgpp_unet::Graph g{want, nullptr, nullptr};
        g.key.tuned_serial = u->tuned_serial;
        u->in_t_dev = u->d_step_cur;
        hipError_t he = hipStreamBeginCapture(u->cap_stream, hipStreamCaptureModeRelaxed);
        int rc = 0;
        if (he == hipSuccess) {
            rc = step_advance_launch(u->d_step_tab, u->d_step_idx, u->d_step_cur, u->cap_stream);
            for (size_t i = 0; i < u->plan.size() && rc == 0; ++i) rc = u->plan[i](u->cap_stream, rows);
            if (rc == 0) rc = step_ddim_dev_launch(z, z0t, eps_uc, eps_c, 1, z_is_half, lam, u->d_step_cur + 1, tweedie_uc, renoise_uc, n, u->cap_stream);
            he = hipStreamEndCapture(u->cap_stream, &g.graph);
        }
        u->in_t_dev = nullptr;
        if (he != hipSuccess || rc != 0 || !g.graph) {
            cfgpp_unet::destroy(g);
            if (rc == 0) cfgpp_set_error("sample_graph: stream capture failed: %s", hipGetErrorString(he));
            return rc ? rc : -1;
        }
        he = hipGraphInstantiate(&g.exec, g.graph, nullptr, nullptr, 0);
        if (he != hipSuccess) { cfgpp_unet::destroy(g); cfgpp_set_error("sample_graph: hipGraphInstantiate: %s", hipGetErrorString(he)); return -1; }
        if (u->graphs.size() >= 4) { cfgpp_unet::destroy(u->graphs.front()); u->graphs.erase(u->graphs.begin()); }
        u->graphs.push_back(g);
        hit = (int)u->graphs.size() - 1;
    }
    if (hit != (int)u->graphs.size() - 1) std::swap(u->graphs[hit], u->graphs.back());
    hipGraphExec_t exec = u->graphs.back().exec;
    // this call's table (pageable host memory: the copy is staged, the buffer is free when the call returns) and counter
    std::vector<float> tab((size_t)n_steps * 8, 0.f);
    for (int i = 0; i < n_steps; ++i) for (int k = 0; k < 5; ++k) tab[(size_t)i * 8 + k] = host_steps[(size_t)i * 5 + k];
    CFGPP_HIP_CHECK(hipMemcpyAsync(u->d_step_tab, tab.data(), tab.size() * sizeof(float), hipMemcpyHostToDevice, s));
    CFGPP_HIP_CHECK(hipStreamSynchronize(s));        // `tab` leaves scope; once per sampling loop
    CFGPP_HIP_CHECK(hipMemsetAsync(u->d_step_idx, 0, sizeof(int), s));
    for (int i = 0; i < n_steps; ++i) CFGPP_HIP_CHECK(hipGraphLaunch(exec, s));
    return 0;
}

// One forward with a HIP event between every launch of the plan, on `stream` (the stream the
// kernels run on).  out_ms[k] / out_flops[k] / out_launches[k], k = 0 igemm (conv/linear),
// 1 attention, 2 norm (GroupNorm/LayerNorm), 3 small ops; flops are ALGORITHMIC (2*MAC).
int cfgpp_unet_profile(cfgpp_unet* u, const void* z, int z_is_half, int z_rows, float t, void* eps_out, int rows,
                       void* stream, double* out_ms, double* out_flops, int* out_launches, char* detail, long detail_cap) {
    CFGPP_REQUIRE(u && u->finalized && u->ctx_set, "profile: context not ready");
    CFGPP_REQUIRE(z && eps_out && out_ms && out_flops && out_launches && rows == u->ctx_rows, "profile: bad args");
    u->in_z = z; u->in_z_half = z_is_half; u->in_z_rows = z_rows; u->in_t = t; u->out_eps = eps_out;
    hipStream_t s = (hipStream_t)stream;
    const size_t n = u->plan.size();
    std::vector<hipEvent_t> ev(n + 1);
    for (auto& e : ev) CFGPP_HIP_CHECK(hipEventCreate(&e));
    CFGPP_HIP_CHECK(hipEventRecord(ev[0], s));
    int rc = 0;
    for (size_t i = 0; i < n && rc == 0; ++i) { rc = u->plan[i](s, rows); if (rc == 0 && hipEventRecord(ev[i + 1], s) != hipSuccess) rc = -1; }
    if (rc == 0 && hipStreamSynchronize(s) != hipSuccess) rc = -1;
    for (int k = 0; k < 4; ++k) { out_ms[k] = 0; out_flops[k] = 0; out_launches[k] = 0; }
    if (rc == 0) {
        for (size_t i = 0; i < n; ++i) {
            float ms = 0.f; hipEventElapsedTime(&ms, ev[i], ev[i + 1]);
            const int k = u->plan_kind[i];
            out_ms[k] += ms; out_flops[k] += 2.0 * u->plan_macs[i] * rows; out_launches[k] += 1;
        }
        if (detail && detail_cap > 0) {      // one line per launch: index, family, description, us, GFLOP
            std::string txt;
            for (size_t i = 0; i < n; ++i) {
                float ms = 0.f; hipEventElapsedTime(&ms, ev[i], ev[i + 1]);
                char line[256];
                snprintf(line, sizeof(line), "%zu\t%d\t%s\t%.1f\t%.3f\n", i, u->plan_kind[i], u->plan_desc[i].c_str(), ms * 1e3,
                         2.0 * u->plan_macs[i] * rows * 1e-9);
                txt += line;
            }
            const long ncopy = std::min<long>((long)txt.size(), detail_cap - 1);
            std::memcpy(detail, txt.data(), ncopy); detail[ncopy] = 0;
        }
    }
    for (auto& e : ev) hipEventDestroy(e);
    return rc;
}

// Export (set = 0) / import (set = 1) the tile configs the in-situ tuning pinned for batch `rows`: one int per
// igemm launch of the plan, in plan order.  Lets a profiled run (rocprofv3 --pmc) replay exactly the tiles of the
// un-profiled run without timing passes of its own.  Returns the number of hint slots, or < 0 on error.
int cfgpp_unet_tuning(cfgpp_unet* u, int rows, int* hints, int cap, int set) {
    CFGPP_REQUIRE(u && u->finalized && hints && rows > 0, "unet_tuning: bad args");
    const int n = (int)u->cfg_hints.size();
    CFGPP_REQUIRE(cap >= n, "unet_tuning: buffer of %d for %d launches", cap, n);
    if (set) {
        u->tuned_by_rows[rows] = std::vector<int>(hints, hints + n);
        if (u->tuned_rows == rows) u->tuned_rows = 0;      // re-install on the next forward
        ++u->tuned_serial;
        return n;
    }
    auto it = u->tuned_by_rows.find(rows);
    if (it == u->tuned_by_rows.end()) { cfgpp_set_error("unet_tuning: batch %d has not been tuned", rows); return -3; }
    std::copy(it->second.begin(), it->second.end(), hints);
    return n;
}

double cfgpp_unet_flops(cfgpp_unet* u, int rows) {
    if (!u || !u->finalized) return 0.0;
    return 2.0 * (u->macs_per_row + u->attn_macs_per_row) * rows;
}
double cfgpp_unet_device_bytes(cfgpp_unet* u) { return u ? u->dev_bytes : 0.0; }

// ---- single-op wrappers for tests ----------------------------------------------
// test hook: the next cfgpp_op_igemm launches write GroupNorm statistics of their output (IGemmArgs::gstat) into `buf`
// ([M / 32][N][2] floats; null = off); cfgpp_op_igemm_gstat_written() = what the last launch reported through stat_flag
static float* g_op_gstat = nullptr;
static int g_op_gstat_flag = 0;
void cfgpp_op_igemm_set_gstat(void* buf) { g_op_gstat = (float*)buf; }
int cfgpp_op_igemm_gstat_written() { return g_op_gstat_flag; }

int cfgpp_op_igemm(const void* a0, const void* a1, int C0, int C1, int taps, int amode, int H, int W,
                   const void* w, int M, int N, const float* bias, const float* temb, int temb_ld,
                   const void* resid, int rmode, int rld, void* out, int omode, int old_, int epi, void* stream) {
    IGemmArgs a = base_args();
    a.a0 = (const half_t*)a0; a.a1 = (const half_t*)a1; a.C0 = C0; a.C1 = C1; a.taps = taps; a.amode = amode; a.H = H; a.W = W;
    a.w = (const half_t*)w; a.M = M; a.N = N; a.K = taps * (C0 + C1); a.bias = bias; a.temb = temb; a.temb_ld = temb_ld;
    a.rows_per_batch = H * W; a.resid = (const half_t*)resid; a.rmode = rmode; a.rld = rld;
    a.out = (half_t*)out; a.omode = omode; a.old = old_; a.epi = epi;
    a.gstat = g_op_gstat; a.stat_flag = &g_op_gstat_flag; g_op_gstat_flag = 0;
    return igemm_launch(a, (hipStream_t)stream);
}

int cfgpp_op_igemm_heads(const void* a_, int K, const void* w, int M, int N, const float* bias, int rows_per_batch,
                         void* hq, void* hk, void* hvt, int part0, int part_width, int head_dim, int heads,
                         int q_tok_pad, int tok_pad, void* stream) {
    IGemmArgs a = base_args();
    a.a0 = (const half_t*)a_; a.C0 = K; a.amode = 0; a.w = (const half_t*)w; a.M = M; a.N = N; a.K = K; a.bias = bias;
    a.epi = EPI_HEADS; a.rows_per_batch = rows_per_batch; a.hq = (half_t*)hq; a.hk = (half_t*)hk; a.hvt = (half_t*)hvt;
    a.part0 = part0; a.part_width = part_width; a.head_dim = head_dim; a.head_dim_pad = round_up(head_dim, 32);
    a.heads = heads; a.q_tok_pad = q_tok_pad; a.tok_pad = tok_pad;
    return igemm_launch(a, (hipStream_t)stream);
}

}  // extern "C"
