// Shared executor infrastructure of the UNet and VAE engines: host-side parameter table, weight
// repacking + upload, shape-keyed activation pool (halo-padded NHWC), and the launch-plan builder
// (closures over preallocated buffers).  Header-only; each engine TU gets its own copy.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstring>
#include <deque>
#include <functional>
#include <map>
#include <string>
#include <vector>

#include "../../include/cfgpp.h"
#include "cfgpp_debug.h"
#include "igemm.h"

namespace {


struct HostParam {
    std::vector<long> shape;
    std::vector<half_t> h;     // matrices / conv kernels
    std::vector<float> f;      // 1-D params (bias, norm) and conv_in
    bool is_matrix = false;
    bool loaded = false;
    long numel() const { long n = 1; for (long s : shape) n *= s; return n; }
};

struct Tensor {   // halo-padded NHWC fp16 activation
    half_t* p = nullptr;
    int H = 0, W = 0, C = 0;
    // GroupNorm statistics of this tensor as its PRODUCER left them (IGemmArgs::gstat: [rows * H * W / 32][C][2] fp32), pooled with
    // the activation buffer; *gst_ok (host, one per acquired tensor = per producer in the plan) is set by the producer's igemm_launch
    // on every forward: 1 = the buffer holds this forward's statistics, 0 = the producer could not write them (K-split launch,
    // non-igemm producer: never set) and the consumer runs its own statistics pass.  Null on maps under 32 x 32 pixels.
    float* gst = nullptr;
    int* gst_ok = nullptr;
};

using Op = std::function<int(hipStream_t, int /*rows*/)>;


struct EngineBase {
    int max_rows = 1;
    int norm_groups = 32;
    std::map<std::string, HostParam> params;
    std::vector<void*> allocs;
    double dev_bytes = 0;
    double macs_per_row = 0;        // conv/linear MACs per batch row per forward
    double attn_macs_per_row = 0;

    std::vector<Op> plan;           // forward
    // per-op tags of `plan` for the profiler: kernel family + algorithmic MACs per batch row
    std::vector<int> plan_kind;     // 0 igemm (conv/linear), 1 attention, 2 norm (GN/LN), 3 small
    std::vector<double> plan_macs;
    std::vector<std::string> plan_desc;
    void tag(int kind, double macs, const std::string& desc = "") {
        plan_kind.resize(plan.size(), 3); plan_macs.resize(plan.size(), 0.0); plan_desc.resize(plan.size());
        if (!plan.empty()) { plan_kind.back() = kind; plan_macs.back() = macs; plan_desc.back() = desc; }
    }
    // in-situ tile tuning (igemm_kernel.hip): one hint per igemm launch of `plan`, filled by tune_plan()
    std::deque<int> cfg_hints;
    std::vector<int*> plan_hint;    // per plan op: its hint slot or null
    int tuned_rows = 0;
    std::map<int, std::vector<int>> tuned_by_rows;      // batch rows -> pinned config per hint slot (re-used when the batch alternates)
    int* new_hint(bool in_main_plan) {
        cfg_hints.push_back(0);
        int* h = &cfg_hints.back();
        if (in_main_plan) { plan_hint.resize(plan.size() + 1, nullptr); plan_hint[plan.size()] = h; }   // the op is pushed next
        return h;
    }
    // Times every hinted launch of `plan` in place - HIP events between the launches of a real forward on
    // `s`, two passes per candidate tile config - and pins the fastest (>= 3 % better than the heuristic).
    // The forward is idempotent, so the passes leave the same activations behind as one plain forward.
    int tune_plan(hipStream_t s, int rows) {
        auto cached = tuned_by_rows.find(rows);
        if (cached != tuned_by_rows.end() && cached->second.size() == cfg_hints.size()) {
            size_t k = 0;
            for (int& h : cfg_hints) h = cached->second[k++];
            tuned_rows = rows;
            return 0;
        }
        static const int cands[] = {0, 1, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 20, 24, 25, 26, 27, 28};
        const size_t n = plan.size();
        plan_hint.resize(n, nullptr);
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(s, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) return 0;
        std::vector<hipEvent_t> ev(n + 1, nullptr);
        for (auto& e : ev) if (hipEventCreate(&e) != hipSuccess) return -1;
        std::vector<float> base(n, 1e30f), best(n, 1e30f), t(n);
        std::vector<char> applied(n, 1);        // did launch i run the candidate tile (or did igemm_launch reject the hint)?
        std::vector<int> bestc(n, 0);
        int rc = 0;
        // two timed forwards with the hints as they stand; t[i] = the faster of the two for every hinted launch
        auto timed_passes = [&]() {
            std::fill(t.begin(), t.end(), 1e30f);
            for (int rep = 0; rep < 2 && rc == 0; ++rep) {
                if (hipEventRecord(ev[0], s) != hipSuccess) rc = -1;
                for (size_t i = 0; i < n && rc == 0; ++i) {
                    rc = plan[i](s, rows);
                    if (plan_hint[i]) applied[i] = (char)igemm_last_hint_applied();
                    if (rc == 0 && hipEventRecord(ev[i + 1], s) != hipSuccess) rc = -1;
                }
                if (rc == 0 && hipEventSynchronize(ev[n]) != hipSuccess) rc = -1;
                for (size_t i = 0; i < n && rc == 0; ++i) {
                    if (!plan_hint[i]) continue;
                    float ms = 0.f;
                    if (hipEventElapsedTime(&ms, ev[i], ev[i + 1]) == hipSuccess && ms < t[i]) t[i] = ms;
                }
            }
        };
        const unsigned mask = igemm_tune_mask();
        for (int c : cands) {
            if (c != 0 && !((mask >> c) & 1u)) continue;
            for (int& h : cfg_hints) h = c;
            timed_passes();
            for (size_t i = 0; i < n; ++i) {
                if (c == 0) base[i] = t[i];
                if (t[i] < best[i] && (c == 0 || applied[i])) { best[i] = t[i]; bestc[i] = c; }      // never pin a config that did not run
            }
        }
        for (size_t i = 0; i < n; ++i)
            if (plan_hint[i] && !(bestc[i] != 0 && best[i] < 0.97f * base[i])) { bestc[i] = 0; best[i] = base[i]; }
        // second stage: with the tile pinned, the tile walk (M-major / N-major; the default is a rule on operand bytes).
        // Every launch runs with its own pinned tile, so the cache state each one sees is the final plan's.
        for (int walk = 1; walk <= 2 && rc == 0 && (mask >> 31); ++walk) {
            for (size_t i = 0; i < n; ++i) if (plan_hint[i]) *plan_hint[i] = (bestc[i] & 63) | (walk << 6);
            timed_passes();
            for (size_t i = 0; i < n; ++i)
                if (plan_hint[i] && t[i] < 0.97f * best[i]) { best[i] = t[i]; bestc[i] = (bestc[i] & 63) | (walk << 6); }
        }
        for (int& h : cfg_hints) h = 0;
        for (size_t i = 0; i < n; ++i) if (plan_hint[i]) *plan_hint[i] = bestc[i];
        for (auto& e : ev) hipEventDestroy(e);
        if (rc == 0) tuned_by_rows[rows] = std::vector<int>(cfg_hints.begin(), cfg_hints.end());
        tuned_rows = rows;
        return rc;
    }
    // activation pool, keyed by shape (halo stays zero for ever); every buffer travels with its statistics buffer
    std::map<std::tuple<int, int, int>, std::vector<std::pair<half_t*, float*>>> pool;
    std::deque<int> gst_flags;      // one "producer wrote statistics" flag per acquired tensor
    float* d_gn_stats = nullptr;
    // fp32 partials of the rule-based K-split igemm launches: owned by the engine (counted in dev_bytes, freed with it) and
    // allocated while the plan is built - never inside a forward
    static constexpr long KSPLIT_WS_BYTES = 128L << 20;
    float* d_ksplit_ws = nullptr;
    float* ksplit_ws() {
        if (!d_ksplit_ws) d_ksplit_ws = (float*)dmalloc((size_t)KSPLIT_WS_BYTES, false);
        return d_ksplit_ws;
    }

    void* dmalloc(size_t bytes, bool zero = true) {
        void* p = nullptr;
        if (hipMalloc(&p, bytes) != hipSuccess) return nullptr;
        if (zero) hipMemset(p, 0, bytes);
        allocs.push_back(p);
        dev_bytes += (double)bytes;
        return p;
    }
    Tensor acq(int H, int W, int C) {
        auto key = std::make_tuple(H, W, C);
        auto& fl = pool[key];
        Tensor t; t.H = H; t.W = W; t.C = C;
        gst_flags.push_back(0);
        t.gst_ok = &gst_flags.back();
        if (!fl.empty()) { t.p = fl.back().first; t.gst = fl.back().second; fl.pop_back(); return t; }
        t.p = (half_t*)dmalloc((size_t)max_rows * (H + 2) * (W + 2) * C * sizeof(half_t));
        // (32 x 32 maps and larger: below that the one-launch slab kernel beats finalize + apply - 16 x 16: 14.6 vs 18 us per GroupNorm,
        //  profiles/r05/ab/forward_ab_gn_prestats_*)
        if (H * W >= 1024 && (H * W) % 32 == 0 && C % 8 == 0)
            t.gst = (float*)dmalloc((size_t)max_rows * (H * W / 32) * C * 2 * sizeof(float), false);
        return t;
    }
    void rel(const Tensor& t) { if (t.p) pool[std::make_tuple(t.H, t.W, t.C)].push_back(std::make_pair(t.p, t.gst)); }
    ~EngineBase() { for (void* p : allocs) hipFree(p); }
};

void expect(EngineBase* u, const std::string& key, std::vector<long> shape, bool matrix) {
    HostParam hp; hp.shape = std::move(shape); hp.is_matrix = matrix;
    u->params[key] = std::move(hp);
}
void expect_linear(EngineBase* u, const std::string& p, long out, long in, bool bias = true) {
    expect(u, p + ".weight", {out, in}, true);
    if (bias) expect(u, p + ".bias", {out}, false);
}
void expect_conv(EngineBase* u, const std::string& p, long out, long in, int k) {
    expect(u, p + ".weight", {out, in, k, k}, true);
    expect(u, p + ".bias", {out}, false);
}
void expect_norm(EngineBase* u, const std::string& p, long c) {
    expect(u, p + ".weight", {c}, false);
    expect(u, p + ".bias", {c}, false);
}
void expect_resnet(EngineBase* u, const std::string& p, long cin, long cout, long temb) {
    expect_norm(u, p + ".norm1", cin);
    expect_conv(u, p + ".conv1", cout, cin, 3);
    expect_linear(u, p + ".time_emb_proj", cout, temb);
    expect_norm(u, p + ".norm2", cout);
    expect_conv(u, p + ".conv2", cout, cout, 3);
    if (cin != cout) expect_conv(u, p + ".conv_shortcut", cout, cin, 1);
}
void expect_transformer(EngineBase* u, const std::string& p, long c, int depth, long cross) {
    expect_norm(u, p + ".norm", c);
    // proj_in/out: conv1x1 [c,c,1,1] (SD1.5) or linear [c,c] (SDXL): accept either (numel equal)
    expect(u, p + ".proj_in.weight", {c, c}, true);  expect(u, p + ".proj_in.bias", {c}, false);
    expect(u, p + ".proj_out.weight", {c, c}, true); expect(u, p + ".proj_out.bias", {c}, false);
    for (int k = 0; k < depth; ++k) {
        const std::string b = p + ".transformer_blocks." + std::to_string(k);
        expect_norm(u, b + ".norm1", c); expect_norm(u, b + ".norm2", c); expect_norm(u, b + ".norm3", c);
        expect_linear(u, b + ".attn1.to_q", c, c, false); expect_linear(u, b + ".attn1.to_k", c, c, false);
        expect_linear(u, b + ".attn1.to_v", c, c, false); expect_linear(u, b + ".attn1.to_out.0", c, c, true);
        expect_linear(u, b + ".attn2.to_q", c, c, false); expect_linear(u, b + ".attn2.to_k", c, cross, false);
        expect_linear(u, b + ".attn2.to_v", c, cross, false); expect_linear(u, b + ".attn2.to_out.0", c, c, true);
        expect_linear(u, b + ".ff.net.0.proj", 8 * c, c, true);
        expect_linear(u, b + ".ff.net.2", c, 4 * c, true);
    }
}

struct Builder {
    EngineBase* u;
    bool ok = true;
    std::string err;

    HostParam* get(const std::string& key) {
        auto it = u->params.find(key);
        if (it == u->params.end() || !it->second.loaded) { ok = false; err = "missing parameter " + key; return nullptr; }
        return &it->second;
    }
    template <typename T>
    T* upload(const std::vector<T>& v) {
        T* d = (T*)u->dmalloc(v.size() * sizeof(T), false);
        if (!d) { ok = false; err = "hipMalloc failed"; return nullptr; }
        if (hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) { ok = false; err = "hipMemcpy failed"; }
        return d;
    }
    void drop(const std::string& key) { auto& p = u->params[key]; std::vector<half_t>().swap(p.h); std::vector<float>().swap(p.f); }

    float* f32(const std::string& key) {
        HostParam* p = get(key); if (!p) return nullptr;
        float* d = upload(p->f); drop(key); return d;
    }
    // [N][K] as is (linear, or conv1x1 OIHW)
    half_t* linear(const std::string& key) {
        HostParam* p = get(key); if (!p) return nullptr;
        half_t* d = upload(p->h); drop(key); return d;
    }
    // OIHW -> [O][I/64][kh*kw][64]  (K order of the implicit GEMM: channel-block major, tap minor)
    half_t* conv3(const std::string& key) {
        HostParam* p = get(key); if (!p) return nullptr;
        const long O = p->shape[0], I = p->shape[1];
        std::vector<half_t> r((size_t)O * 9 * I);
        for (long o = 0; o < O; ++o)
            for (long i = 0; i < I; ++i)
                for (int t = 0; t < 9; ++t) r[(size_t)o * 9 * I + ((i >> 6) * 9 + t) * 64 + (i & 63)] = p->h[((size_t)o * I + i) * 9 + t];
        half_t* d = upload(r); drop(key); return d;
    }
    // concat rows of several [n_i][K] matrices
    std::vector<half_t> concat_host(const std::vector<std::string>& keys) {
        std::vector<half_t> r;
        for (auto& k : keys) { HostParam* p = get(k); if (!p) return {}; r.insert(r.end(), p->h.begin(), p->h.end()); }
        for (auto& k : keys) drop(k);
        return r;
    }
    half_t* concat(const std::vector<std::string>& keys) {
        std::vector<half_t> r = concat_host(keys);
        return ok ? upload(r) : nullptr;
    }
    // GEGLU packing: within every 64 packed rows, [0,32) value rows f, [32,64) gate rows 4C+f
    bool geglu_host(const std::string& pfx, long C, std::vector<half_t>& rw, std::vector<float>& rb) {
        HostParam* pw = get(pfx + ".weight"); HostParam* pb = get(pfx + ".bias");
        if (!pw || !pb) return false;
        const long F = 4 * C;
        rw.resize((size_t)2 * F * C); rb.resize((size_t)2 * F);
        for (long f = 0; f < F; ++f) {
            const long pv = (f / 32) * 64 + (f % 32), pg = pv + 32;
            std::memcpy(&rw[(size_t)pv * C], &pw->h[(size_t)f * C], C * sizeof(half_t));
            std::memcpy(&rw[(size_t)pg * C], &pw->h[(size_t)(F + f) * C], C * sizeof(half_t));
            rb[pv] = pb->f[f]; rb[pg] = pb->f[F + f];
        }
        drop(pfx + ".weight"); drop(pfx + ".bias");
        return true;
    }
    void geglu(const std::string& pfx, long C, half_t** w, float** b) {
        std::vector<half_t> rw; std::vector<float> rb;
        if (!geglu_host(pfx, C, rw, rb)) return;
        *w = upload(rw); *b = upload(rb);
    }
};

inline int round_up(int a, int b) { return (a + b - 1) / b * b; }

// generic igemm op helpers ----------------------------------------------------
IGemmArgs base_args(EngineBase* u = nullptr) {
    IGemmArgs a; std::memset(&a, 0, sizeof(a)); a.out_scale = 1.0f; a.taps = 1;
    if (u) { a.ws_buf = u->ksplit_ws(); a.ws_bytes = a.ws_buf ? EngineBase::KSPLIT_WS_BYTES : 0; }
    return a;
}

struct Plan {
    EngineBase* u;
    Builder* B;
    std::vector<Op>* ops;

    // conv3x3 on padded NHWC.  amode 1 normal, 2 stride-2 (src is 2H x 2W), 3 upsample (src is H/2 x W/2)
    void conv3x3(const Tensor& src, const Tensor& dst, const half_t* w, const float* bias, int amode,
                 const float* temb, int temb_ld, const Tensor* resid, int ashift = 0) {
        IGemmArgs a = base_args(u);
        a.a0 = src.p; a.C0 = src.C; a.taps = 9; a.amode = amode; a.ashift = ashift; a.H = dst.H; a.W = dst.W;
        a.w = w; a.N = dst.C; a.K = 9 * src.C; a.bias = bias; a.temb = temb; a.temb_ld = temb_ld;
        a.rows_per_batch = dst.H * dst.W;
        if (resid) { a.resid = resid->p; a.rmode = 1; a.rld = resid->C; }
        a.out = dst.p; a.omode = 1; a.old = dst.C; a.epi = EPI_STORE;
        a.gstat = dst.gst; a.stat_flag = dst.gst ? dst.gst_ok : nullptr;      // the consumer GroupNorm takes its statistics from this epilogue
        const int HW = dst.H * dst.W;
        u->macs_per_row += (double)HW * dst.C * 9.0 * src.C;
        int* hint = u->new_hint(ops == &u->plan);
        ops->push_back([a, HW, hint](hipStream_t s, int rows) mutable { IGemmArgs b = a; b.M = rows * HW; b.cfg_hint = *hint; return igemm_launch(b, s); });
        if (ops == &u->plan) u->tag(0, (double)HW * dst.C * 9.0 * src.C, "conv3x3 amode=" + std::to_string(amode) + " HW=" + std::to_string(HW) + " N=" + std::to_string(dst.C) + " K=" + std::to_string(9 * src.C) + (resid ? " +res" : "") + (temb ? " +temb" : ""));
    }
    // 1x1 conv over (src0 || src1) padded -> padded
    void conv1x1(const Tensor& s0, const Tensor* s1, const Tensor& dst, const half_t* w, const float* bias) {
        IGemmArgs a = base_args(u);
        a.a0 = s0.p; a.C0 = s0.C; if (s1) { a.a1 = s1->p; a.C1 = s1->C; }
        a.amode = 1; a.H = dst.H; a.W = dst.W; a.w = w; a.N = dst.C; a.K = a.C0 + a.C1; a.bias = bias;
        a.rows_per_batch = dst.H * dst.W; a.out = dst.p; a.omode = 1; a.old = dst.C; a.epi = EPI_STORE;
        const int HW = dst.H * dst.W;
        u->macs_per_row += (double)HW * dst.C * a.K;
        int* hint = u->new_hint(ops == &u->plan);
        ops->push_back([a, HW, hint](hipStream_t s, int rows) mutable { IGemmArgs b = a; b.M = rows * HW; b.cfg_hint = *hint; return igemm_launch(b, s); });
        if (ops == &u->plan) u->tag(0, (double)HW * dst.C * a.K, "conv1x1 HW=" + std::to_string(HW) + " N=" + std::to_string(dst.C) + " K=" + std::to_string(a.K));
    }
    // token GEMM: out[M][N] = A[M][K] W^T (+bias)(+resid, may alias out)
    void linear(const half_t* A, int K, half_t* out, int N, const half_t* w, const float* bias, const half_t* resid,
                int tokens, int epi = EPI_STORE) {
        IGemmArgs a = base_args(u);
        a.a0 = A; a.C0 = K; a.amode = 0; a.w = w; a.N = N; a.K = K; a.bias = bias;
        a.resid = resid; a.rmode = 0; a.rld = N; a.out = out; a.omode = 0;
        a.old = (epi == EPI_GEGLU) ? N / 2 : N; a.epi = epi; a.rows_per_batch = tokens;
        u->macs_per_row += (double)tokens * N * K;
        int* hint = u->new_hint(ops == &u->plan);
        ops->push_back([a, tokens, hint](hipStream_t s, int rows) mutable { IGemmArgs b = a; b.M = rows * tokens; b.cfg_hint = *hint; return igemm_launch(b, s); });
        if (ops == &u->plan) u->tag(0, (double)tokens * N * K, std::string(epi == EPI_GEGLU ? "geglu" : "linear") + " HW=" + std::to_string(tokens) + " N=" + std::to_string(N) + " K=" + std::to_string(K) + (resid ? " +res" : ""));
    }
    // tokens -> padded NHWC with residual from a padded tensor (Transformer2D proj_out)
    void linear_to_padded(const half_t* A, int K, const Tensor& dst, const half_t* w, const float* bias, const Tensor& resid) {
        IGemmArgs a = base_args(u);
        a.a0 = A; a.C0 = K; a.amode = 0; a.H = dst.H; a.W = dst.W; a.w = w; a.N = dst.C; a.K = K; a.bias = bias;
        a.resid = resid.p; a.rmode = 1; a.rld = resid.C; a.out = dst.p; a.omode = 1; a.old = dst.C;
        a.epi = EPI_STORE; a.rows_per_batch = dst.H * dst.W;
        a.gstat = dst.gst; a.stat_flag = dst.gst ? dst.gst_ok : nullptr;
        const int HW = dst.H * dst.W;
        u->macs_per_row += (double)HW * dst.C * K;
        int* hint = u->new_hint(ops == &u->plan);
        ops->push_back([a, HW, hint](hipStream_t s, int rows) mutable { IGemmArgs b = a; b.M = rows * HW; b.cfg_hint = *hint; return igemm_launch(b, s); });
        if (ops == &u->plan) u->tag(0, (double)HW * dst.C * K, "proj_out HW=" + std::to_string(HW) + " N=" + std::to_string(dst.C) + " K=" + std::to_string(K));
    }
    // projection into head-major buffers
    void heads(const half_t* A, int K, const half_t* w, int N, int tokens, int part0, int C, int nheads,
               half_t* q, half_t* k, half_t* vt, int q_tok_pad, int tok_pad, bool count = true,
               const float* bias = nullptr) {
        IGemmArgs a = base_args(u);
        const int d = C / nheads;
        a.a0 = A; a.C0 = K; a.amode = 0; a.w = w; a.N = N; a.K = K; a.epi = EPI_HEADS;
        a.bias = bias;
        a.rows_per_batch = tokens; a.hq = q; a.hk = k; a.hvt = vt; a.part0 = part0; a.part_width = C;
        a.head_dim = d; a.head_dim_pad = round_up(d, 32); a.heads = nheads; a.tok_pad = tok_pad; a.q_tok_pad = q_tok_pad;
        if (count) u->macs_per_row += (double)tokens * N * K;
        int* hint = u->new_hint(ops == &u->plan);
        ops->push_back([a, tokens, hint](hipStream_t s, int rows) mutable { IGemmArgs b = a; b.M = rows * tokens; b.cfg_hint = *hint; return igemm_launch(b, s); });
        if (ops == &u->plan) u->tag(0, count ? (double)tokens * N * K : 0.0, "heads HW=" + std::to_string(tokens) + " N=" + std::to_string(N) + " K=" + std::to_string(K));
    }
    void groupnorm(const Tensor& s0, const Tensor* s1, half_t* dst, bool dst_padded, const float* g, const float* b,
                   float eps, bool silu) {
        EngineBase* uu = u;
        const half_t* p0 = s0.p; const half_t* p1 = s1 ? s1->p : nullptr;
        const int H = s0.H, W = s0.W, C0 = s0.C, C1 = s1 ? s1->C : 0, G = u->norm_groups;
        const float* gst0 = s0.gst; const float* gst1 = s1 ? s1->gst : nullptr;
        const int* ok0 = s0.gst_ok; const int* ok1 = s1 ? s1->gst_ok : nullptr;
        ops->push_back([=](hipStream_t s, int rows) {
            // statistics from the producers' epilogues when every source's producer wrote them on THIS forward (a host flag each
            // igemm_launch sets); otherwise this op's own statistics pass
            if (cfgpp_groupnorm_prestats_enabled() && gst0 && ok0 && *ok0 && (!p1 || (gst1 && ok1 && *ok1)))
                return cfgpp_op_groupnorm_pre(p0, p1, dst, g, b, gst0, gst1, uu->d_gn_stats, rows, H, W, C0, C1, G, eps, silu ? 1 : 0,
                                              dst_padded ? 1 : 0, s);
            return cfgpp_op_groupnorm(p0, p1, dst, g, b, uu->d_gn_stats, rows, H, W, C0, C1, G, eps, silu ? 1 : 0,
                                      dst_padded ? 1 : 0, s);
        });
        if (ops == &u->plan) u->tag(2, 0.0, "groupnorm HW=" + std::to_string(H * W) + " C=" + std::to_string(C0 + C1));
    }
    void layernorm(const half_t* x, half_t* y, const float* g, const float* b, int tokens, int C) {
        ops->push_back([=](hipStream_t s, int rows) { return cfgpp_op_layernorm(x, y, g, b, (long)rows * tokens, C, 1e-5f, s); });
        if (ops == &u->plan) u->tag(2, 0.0, "layernorm HW=" + std::to_string(tokens) + " C=" + std::to_string(C));
    }
    void attention(const half_t* q, const half_t* k, const half_t* vt, half_t* o, int nheads, int d, int nq, int nk,
                   int q_tok_pad, int k_tok_pad) {
        u->attn_macs_per_row += 2.0 * (double)nheads * nq * nk * d;
        ops->push_back([=](hipStream_t s, int rows) {
            return cfgpp_op_attention(q, k, vt, o, rows, nheads, d, nq, nk, q_tok_pad, k_tok_pad, s);
        });
        if (ops == &u->plan) u->tag(1, 2.0 * (double)nheads * nq * nk * d, "self_attn heads=" + std::to_string(nheads) + " N=" + std::to_string(nq) + " d=" + std::to_string(d));
    }
};

}  // namespace
