// CLIP text transformer on the engine's kernels: the `text_encoder(ids)` call of the reference
// (latent_diffusion.py:105-113: last_hidden_state; latent_sdxl.py:76-93: hidden_states[-2] / [-(clip_skip+2)] and the
// projected pooled output of the second tower).  Runs once per prompt, off the per-step path - it exists so that a job needs
// no torch ops at all, not for speed.  Same building blocks as the UNet: igemm linear layers (bias / residual epilogues),
// cfgpp_op_layernorm, skinny_gemm for the pooled projection; new here: token + position embedding gather, causal
// self-attention over the 77 tokens (head dim 64 in both CLIP-L and OpenCLIP-bigG), quick_gelu / gelu.
//
// hidden_states[k], k = 0 .. L: the residual stream before layer k (k = L: after the last layer);
// last_hidden_state = final_layer_norm(hidden_states[L]); pooled = last_hidden_state[eos position] @ text_projection^T.
// Parameters use the `transformers` CLIPTextModel(WithProjection) state-dict keys unchanged.
//
// STATUS: validated on the MI355X in round 3 (tests/test_gpu_text.py against `transformers`, 4 / 4 configurations).  The
// solvers use it when it is passed as text_encoder= (cfgpp_amd/text.py); the default text path is unchanged.
#include "engine_base.h"

struct cfgpp_text : EngineBase {
    int vocab = 0, hidden = 0, layers = 0, heads = 0, inter = 0, act = 0, proj = 0, device = 0;
    bool finalized = false;
    // per-call state read by the plan's closures
    int out_layer = -1; half_t* out_hidden = nullptr; float* out_pooled = nullptr;
    // device buffers
    half_t *tok_emb = nullptr, *pos_emb = nullptr, *x = nullptr, *h = nullptr, *qkv = nullptr, *att = nullptr, *ff = nullptr;
    half_t* wproj = nullptr;
    float* pool_in = nullptr;
    int *d_ids = nullptr, *d_eos = nullptr;
};

namespace {

constexpr int T = 77;       // CLIP context length

__global__ void __launch_bounds__(128)
text_embed_kernel(const int* __restrict__ ids, const half_t* __restrict__ tok, const half_t* __restrict__ pos,
                  half_t* __restrict__ x, int H, int vocab) {
    const int row = blockIdx.x;                       // b * 77 + t
    const int t = row % T;
    int id = ids[row];
    id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
    const half_t* te = tok + (long)id * H;
    const half_t* pe = pos + (long)t * H;
    for (int c = threadIdx.x * 8; c < H; c += 128 * 8) {
        const half8_t a = *reinterpret_cast<const half8_t*>(te + c);
        const half8_t b = *reinterpret_cast<const half8_t*>(pe + c);
        half8_t o;
#pragma unroll
        for (int k = 0; k < 8; ++k) o[k] = (half_t)((float)a[k] + (float)b[k]);
        *reinterpret_cast<half8_t*>(x + (long)row * H + c) = o;
    }
}

// causal self-attention, 77 tokens, head dim 64: one workgroup per (head, sample), thread i owns query i
__global__ void __launch_bounds__(128)
text_attn_kernel(const half_t* __restrict__ qkv, half_t* __restrict__ out, int H, float scale) {
    __shared__ __attribute__((aligned(16))) half_t Ks[T][72];
    __shared__ __attribute__((aligned(16))) half_t Vs[T][72];
    const int head = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const half_t* base = qkv + (long)b * T * 3 * H + head * 64;
    for (int c = tid; c < T * 8; c += 128) {
        const int r = c >> 3, ch = c & 7;
        *reinterpret_cast<half8_t*>(&Ks[r][ch * 8]) = *reinterpret_cast<const half8_t*>(base + (long)r * 3 * H + H + ch * 8);
        *reinterpret_cast<half8_t*>(&Vs[r][ch * 8]) = *reinterpret_cast<const half8_t*>(base + (long)r * 3 * H + 2 * H + ch * 8);
    }
    __syncthreads();
    if (tid >= T) return;
    float q[64], acc[64];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const half8_t v = *reinterpret_cast<const half8_t*>(base + (long)tid * 3 * H + c * 8);
#pragma unroll
        for (int k = 0; k < 8; ++k) { q[c * 8 + k] = (float)v[k] * scale; acc[c * 8 + k] = 0.f; }
    }
    float m = -INFINITY, l = 0.f;
    for (int j = 0; j <= tid; ++j) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 64; ++k) s += q[k] * (float)Ks[j][k];
        const float mn = fmaxf(m, s);
        const float corr = __expf(m - mn), p = __expf(s - mn);
        l = l * corr + p;
#pragma unroll
        for (int k = 0; k < 64; ++k) acc[k] = acc[k] * corr + p * (float)Vs[j][k];
        m = mn;
    }
    const float inv = 1.0f / l;
    half_t* o = out + ((long)b * T + tid) * H + head * 64;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        half8_t v;
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = (half_t)(acc[c * 8 + k] * inv);
        *reinterpret_cast<half8_t*>(o + c * 8) = v;
    }
}

// in place: act 0 = quick_gelu x * sigmoid(1.702 x) (CLIP-L), 1 = gelu (erf; OpenCLIP-bigG)
__global__ void __launch_bounds__(256)
text_act_kernel(half_t* __restrict__ x, long n8, int act) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n8) return;
    half8_t v = *reinterpret_cast<half8_t*>(x + i * 8);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const float f = (float)v[k];
        const float r = act == 0 ? f / (1.0f + __expf(-1.702f * f)) : 0.5f * f * (1.0f + erff(f * 0.70710678118654752f));
        v[k] = (half_t)r;
    }
    *reinterpret_cast<half8_t*>(x + i * 8) = v;
}

// pooled input: row (b, eos[b]) of the final-LayerNorm output, as fp32
__global__ void __launch_bounds__(128)
text_pool_kernel(const half_t* __restrict__ h, const int* __restrict__ eos, float* __restrict__ out, int H) {
    const int b = blockIdx.x;
    int e = eos[b];
    e = e < 0 ? 0 : (e >= T ? T - 1 : e);
    const half_t* src = h + ((long)b * T + e) * H;
    for (int c = threadIdx.x; c < H; c += 128) out[(long)b * H + c] = (float)src[c];
}

void text_param_table(cfgpp_text* t) {
    const long H = t->hidden, I = t->inter;
    expect(t, "text_model.embeddings.token_embedding.weight", {t->vocab, H}, true);
    expect(t, "text_model.embeddings.position_embedding.weight", {T, H}, true);
    for (int l = 0; l < t->layers; ++l) {
        const std::string p = "text_model.encoder.layers." + std::to_string(l);
        expect_norm(t, p + ".layer_norm1", H); expect_norm(t, p + ".layer_norm2", H);
        for (const char* n : {"q_proj", "k_proj", "v_proj", "out_proj"}) expect_linear(t, p + ".self_attn." + n, H, H, true);
        expect_linear(t, p + ".mlp.fc1", I, H, true);
        expect_linear(t, p + ".mlp.fc2", H, I, true);
    }
    expect_norm(t, "text_model.final_layer_norm", H);
    if (t->proj) expect(t, "text_projection.weight", {t->proj, H}, true);
}

}  // namespace

extern "C" {

cfgpp_text* cfgpp_text_create(int vocab, int hidden, int layers, int heads, int intermediate, int act, int proj_dim,
                              int max_batch, int device_id) {
    if (vocab <= 0 || hidden <= 0 || layers <= 0 || heads <= 0 || hidden != heads * 64 || hidden % 64 != 0 || intermediate % 64 != 0 ||
        intermediate <= 0 || (act != 0 && act != 1) || proj_dim < 0 || proj_dim % 4 != 0 || max_batch <= 0) {
        cfgpp_set_error("text_create: unsupported geometry (hidden=%d must be heads=%d x 64, intermediate=%d a multiple of 64, act 0|1)",
                        hidden, heads, intermediate);
        return nullptr;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= device_id || hipSetDevice(device_id) != hipSuccess) {
        cfgpp_set_error("text_create: no HIP device %d - the HIP path has no CPU fallback", device_id);
        return nullptr;
    }
    if (cfgpp_claim_device(device_id)) return nullptr;
    cfgpp_text* t = new cfgpp_text();
    t->vocab = vocab; t->hidden = hidden; t->layers = layers; t->heads = heads; t->inter = intermediate; t->act = act; t->proj = proj_dim;
    t->max_rows = max_batch; t->device = device_id;
    text_param_table(t);
    return t;
}

void cfgpp_text_destroy(cfgpp_text* t) { delete t; }

int cfgpp_text_load_tensor(cfgpp_text* t, const char* key, const void* host, int dtype, const long* shape, int ndim) {
    CFGPP_REQUIRE(t && key && host && shape && !t->finalized, "text_load_tensor: bad state/args");
    auto it = t->params.find(key);
    if (it == t->params.end()) { cfgpp_set_error("text_load_tensor: unknown key %s", key); return -3; }
    HostParam& p = it->second;
    long n = 1; for (int i = 0; i < ndim; ++i) n *= shape[i];
    CFGPP_REQUIRE(n == p.numel(), "text_load_tensor: %s has %ld elements, expected %ld", key, n, p.numel());
    if (p.is_matrix) {
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

int cfgpp_text_finalize(cfgpp_text* t) {
    CFGPP_REQUIRE(t && !t->finalized, "text_finalize: bad state");
    {
        int n = 0; std::string names;
        for (auto& kv : t->params) if (!kv.second.loaded) { if (n++ < 4) names += kv.first + " "; }
        CFGPP_REQUIRE(n == 0, "text_finalize: %d parameters not loaded (%s...)", n, names.c_str());
    }
    const int H = t->hidden, I = t->inter, R = t->max_rows;
    Builder B{t};
    Plan P{t, &B, &t->plan};
    t->tok_emb = B.linear("text_model.embeddings.token_embedding.weight");
    t->pos_emb = B.linear("text_model.embeddings.position_embedding.weight");
    const size_t rows = (size_t)R * T;
    t->x = (half_t*)t->dmalloc(rows * H * sizeof(half_t));
    t->h = (half_t*)t->dmalloc(rows * H * sizeof(half_t));
    t->qkv = (half_t*)t->dmalloc(rows * 3 * H * sizeof(half_t));
    t->att = (half_t*)t->dmalloc(rows * H * sizeof(half_t));
    t->ff = (half_t*)t->dmalloc(rows * I * sizeof(half_t));
    t->pool_in = (float*)t->dmalloc((size_t)R * H * sizeof(float));
    t->d_ids = (int*)t->dmalloc(rows * sizeof(int));
    t->d_eos = (int*)t->dmalloc((size_t)R * sizeof(int));
    CFGPP_REQUIRE(t->x && t->h && t->qkv && t->att && t->ff && t->pool_in && t->d_ids && t->d_eos, "text_finalize: hipMalloc failed");
    cfgpp_text* tt = t;
    const float scale = 0.125f;                       // 64^-1/2
    // hidden_states[k] snapshot: the residual stream as it stands when layer k is about to run (k = L: after the last one)
    auto snapshot = [&](int k) {
        t->plan.push_back([tt, k, H](hipStream_t s, int r) {
            if (tt->out_layer != k || !tt->out_hidden) return 0;
            CFGPP_HIP_CHECK(hipMemcpyAsync(tt->out_hidden, tt->x, (size_t)r * T * H * sizeof(half_t), hipMemcpyDeviceToDevice, s));
            return 0;
        });
    };
    t->plan.push_back([tt, H](hipStream_t s, int r) {
        hipLaunchKernelGGL(text_embed_kernel, dim3(r * T), dim3(128), 0, s, tt->d_ids, tt->tok_emb, tt->pos_emb, tt->x, H, tt->vocab);
        CFGPP_HIP_CHECK(hipGetLastError());
        return 0;
    });
    for (int l = 0; l < t->layers && B.ok; ++l) {
        const std::string p = "text_model.encoder.layers." + std::to_string(l);
        snapshot(l);
        const float* g1 = B.f32(p + ".layer_norm1.weight"); const float* b1 = B.f32(p + ".layer_norm1.bias");
        const float* g2 = B.f32(p + ".layer_norm2.weight"); const float* b2 = B.f32(p + ".layer_norm2.bias");
        std::vector<float> bq;
        for (const char* n : {"q_proj", "k_proj", "v_proj"}) {
            HostParam* hp = B.get(p + ".self_attn." + n + ".bias");
            if (hp) bq.insert(bq.end(), hp->f.begin(), hp->f.end());
        }
        const float* bqkv = B.upload(bq);
        const half_t* wqkv = B.concat({p + ".self_attn.q_proj.weight", p + ".self_attn.k_proj.weight", p + ".self_attn.v_proj.weight"});
        const half_t* wo = B.linear(p + ".self_attn.out_proj.weight"); const float* bo = B.f32(p + ".self_attn.out_proj.bias");
        const half_t* w1 = B.linear(p + ".mlp.fc1.weight"); const float* bf1 = B.f32(p + ".mlp.fc1.bias");
        const half_t* w2 = B.linear(p + ".mlp.fc2.weight"); const float* bf2 = B.f32(p + ".mlp.fc2.bias");
        if (!B.ok) break;
        P.layernorm(t->x, t->h, g1, b1, T, H);
        P.linear(t->h, H, t->qkv, 3 * H, wqkv, bqkv, nullptr, T);
        t->plan.push_back([tt, H, scale](hipStream_t s, int r) {
            hipLaunchKernelGGL(text_attn_kernel, dim3(tt->heads, r), dim3(128), 0, s, tt->qkv, tt->att, H, scale);
            CFGPP_HIP_CHECK(hipGetLastError());
            return 0;
        });
        P.linear(t->att, H, t->x, H, wo, bo, t->x, T);                  // x += out_proj(att)   (residual in place)
        P.layernorm(t->x, t->h, g2, b2, T, H);
        P.linear(t->h, H, t->ff, I, w1, bf1, nullptr, T);
        t->plan.push_back([tt, I](hipStream_t s, int r) {
            const long n8 = (long)r * T * I / 8;
            hipLaunchKernelGGL(text_act_kernel, dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, s, tt->ff, n8, tt->act);
            CFGPP_HIP_CHECK(hipGetLastError());
            return 0;
        });
        P.linear(t->ff, I, t->x, H, w2, bf2, t->x, T);                   // x += fc2(act(fc1(ln2(x))))
    }
    CFGPP_REQUIRE(B.ok, "text_finalize: %s", B.err.c_str());
    snapshot(t->layers);
    {
        const float* gf = B.f32("text_model.final_layer_norm.weight"); const float* bf = B.f32("text_model.final_layer_norm.bias");
        P.layernorm(t->x, t->h, gf, bf, T, H);
        t->plan.push_back([tt, H](hipStream_t s, int r) {
            if (tt->out_layer >= 0 || !tt->out_hidden) return 0;
            CFGPP_HIP_CHECK(hipMemcpyAsync(tt->out_hidden, tt->h, (size_t)r * T * H * sizeof(half_t), hipMemcpyDeviceToDevice, s));
            return 0;
        });
        if (t->proj) {
            t->wproj = B.linear("text_projection.weight");
            t->plan.push_back([tt, H](hipStream_t s, int r) {
                if (!tt->out_pooled) return 0;
                hipLaunchKernelGGL(text_pool_kernel, dim3(r), dim3(128), 0, s, tt->h, tt->d_eos, tt->pool_in, H);
                CFGPP_HIP_CHECK(hipGetLastError());
                return cfgpp_op_skinny_gemm(tt->pool_in, H, tt->wproj, nullptr, nullptr, 0, tt->out_pooled, tt->proj, r, tt->proj, H, 0, 0, s);
            });
        }
    }
    CFGPP_REQUIRE(B.ok, "text_finalize: %s", B.err.c_str());
    CFGPP_HIP_CHECK(hipDeviceSynchronize());
    t->plan_kind.resize(t->plan.size(), 3); t->plan_macs.resize(t->plan.size(), 0.0); t->plan_desc.resize(t->plan.size());
    t->finalized = true;
    return 0;
}

// ids: HOST int32 [B][77]; eos_pos: HOST int32 [B] (position of the EOS token of every prompt: the pooled row);
// layer: -1 = last_hidden_state (final LayerNorm applied), k in [0, L] = hidden_states[k];
// hidden_out: DEVICE fp16 [B][77][H]; pooled_out: DEVICE fp32 [B][proj_dim] or NULL (needs proj_dim > 0).
int cfgpp_text_encode(cfgpp_text* t, const int* ids, const int* eos_pos, int B, int layer, void* hidden_out, float* pooled_out,
                      void* stream) {
    CFGPP_REQUIRE(t && t->finalized && ids && hidden_out && B > 0 && B <= t->max_rows, "text_encode: bad args (B=%d, max %d)", B, t ? t->max_rows : 0);
    CFGPP_REQUIRE(layer >= -1 && layer <= t->layers, "text_encode: layer %d outside [-1, %d]", layer, t->layers);
    CFGPP_REQUIRE(!pooled_out || (t->proj > 0 && eos_pos), "text_encode: pooled output needs a projection tower and eos positions");
    hipStream_t s = (hipStream_t)stream;
    CFGPP_HIP_CHECK(hipStreamSynchronize(s));          // the id buffers of a previous call may still be read
    CFGPP_HIP_CHECK(hipMemcpy(t->d_ids, ids, (size_t)B * T * sizeof(int), hipMemcpyHostToDevice));
    if (eos_pos) CFGPP_HIP_CHECK(hipMemcpy(t->d_eos, eos_pos, (size_t)B * sizeof(int), hipMemcpyHostToDevice));
    t->out_layer = layer; t->out_hidden = (half_t*)hidden_out; t->out_pooled = pooled_out;
    for (auto& op : t->plan) { int e = op(s, B); if (e) return e; }
    return 0;
}

double cfgpp_text_device_bytes(cfgpp_text* t) { return t ? t->dev_bytes : 0.0; }

}  // extern "C"
