// C-ABI of libl2s_hip.so (include/l2s.h): weight packing, workspace planning and the launch sequences of
// the visual encoder, decoder prologue, decode loop and post-net.  Host code only orchestrates; all
// arithmetic is in the kernels of gemm_nt.hip / encoder_kernels.hip / skinny.hip / decoder_kernels.hip.
#include "../../include/l2s.h"
#ifdef L2S_DIAG
#include "../../include/l2s_diag.h"
#endif
#include "l2s_common.h"
#include "l2s_model.h"
#include "pdecode.h"

#include <cmath>
#include <functional>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

namespace l2s {

// ------------------------------------------------------------------------------------------------ errors
static thread_local std::string g_err;
void set_error(const std::string& msg) { g_err = msg; }

// ------------------------------------------------------------------------------------------------ options
static Options g_default_opt;          // process defaults: what l2s_model_create copies into a new model (l2s_set_option)
int set_option_field(Options& o, const char* name, int value) {
    struct Row { const char* name; int Options::*field; };
    // what the product library (include/l2s.h) offers: precision legs, semantics, documented modes
    static const Row product[] = {
        {"persist_decode", &Options::persist}, {"use_graph", &Options::graph}, {"fold_step_weights", &Options::fold}, {"refresh_map", &Options::refresh_map},
        {"infer_bf16", &Options::infer_bf16}, {"train_bf16", &Options::train_bf16}, {"gemm_x3", &Options::gemm_x3}, {"frontend_x3", &Options::frontend_x3},
        {"trunk_x3", &Options::trunk_x3}, {"lstm_x3", &Options::lstm_x3}};
    for (auto& t : product)
        if (!std::strcmp(name, t.name)) { o.*(t.field) = value; return 0; }
#ifdef L2S_DIAG
    // block-form A/B switches of the same arithmetic (include/l2s_diag.h): libl2s_diag.so only
    static const Row diag[] = {
        {"overlap_postnet", &Options::overlap_postnet}, {"fuse_trunk", &Options::fuse_trunk}, {"fuse_s2", &Options::fuse_s2},
        {"skinny_static", &Options::skinny_static}, {"skinny_sized", &Options::skinny_sized}, {"skinny_split", &Options::skinny_split},
        {"skinny_split8", &Options::skinny_split8}, {"skinny_rc", &Options::rc_shape}, {"skinny_rc_jb", &Options::rc_jb}, {"skinny_rc_multi", &Options::rc_shape_multi},
        {"skinny_flat", &Options::skinny_flat}, {"hoist_vproj", &Options::hoist_vproj}, {"attn_lds", &Options::attn_lds}, {"flat_half", &Options::flat_half},
        {"half_min_mts", &Options::half_min_mts}, {"gemm_x3_dma", &Options::gemm_x3_dma}, {"trunk_chain", &Options::trunk_chain}, {"frontend_solo", &Options::frontend_solo}, {"flat_xcd", &Options::flat_xcd}, {"attn_skip0", &Options::attn_skip0}};
    for (auto& t : diag)
        if (!std::strcmp(name, t.name)) { o.*(t.field) = value; return 0; }
#endif
    return 1;
}

// ------------------------------------------------------------------------------------------------ profiling
struct ProfEntry { std::string name; int64_t launches = 0; double total_ms = 0; std::vector<std::pair<hipEvent_t, hipEvent_t>> pending; };
bool g_prof_on = false;
static std::vector<ProfEntry> g_prof;
static std::map<std::string, int> g_prof_idx;
static thread_local int g_prof_cur = -1;          // a begin/end pair runs on one host thread; several threads may drive launches (InflightPool)
static thread_local hipEvent_t g_prof_start;
static std::mutex g_prof_mu;                      // guards g_prof / g_prof_idx

void prof_begin(const char* name, hipStream_t s) {
    if (!g_prof_on) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    auto it = g_prof_idx.find(name);
    int idx;
    if (it == g_prof_idx.end()) {
        idx = (int)g_prof.size();
        g_prof.push_back(ProfEntry{name});
        g_prof_idx[name] = idx;
    } else {
        idx = it->second;
    }
    g_prof_cur = idx;
    (void)hipEventCreate(&g_prof_start);
    (void)hipEventRecord(g_prof_start, s);
}
void prof_end(hipStream_t s) {
    if (!g_prof_on || g_prof_cur < 0) return;
    hipEvent_t stop;
    (void)hipEventCreate(&stop);
    (void)hipEventRecord(stop, s);
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof[g_prof_cur].pending.emplace_back(g_prof_start, stop);
    g_prof[g_prof_cur].launches++;
    g_prof_cur = -1;
}
static void prof_drain() {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (auto& e : g_prof) {
        for (auto& pr : e.pending) {
            (void)hipEventSynchronize(pr.second);
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess) e.total_ms += ms;
            (void)hipEventDestroy(pr.first);
            (void)hipEventDestroy(pr.second);
        }
        e.pending.clear();
    }
}

}  // namespace l2s

namespace l2s {

// host-side blob builder: every sub-array 64-float (256 B) aligned
struct Blob {
    std::vector<float> data;
    int64_t alloc(int64_t n) {
        int64_t off = align_up((int64_t)data.size(), 64);
        data.resize(off + n, 0.f);
        return off;
    }
};

struct Packer {
    l2s_model* m;
    Blob blob;
    std::vector<std::pair<const float**, int64_t>> fixups;   // pointer slots to patch once the device address is known
    std::string missing;

    const std::vector<float>* get(const std::string& key, int64_t numel) {
        auto it = m->host.find(key);
        if (it == m->host.end()) { if (missing.empty()) missing = "missing tensor " + key; return nullptr; }
        if ((int64_t)it->second.size() != numel) {
            if (missing.empty()) missing = "tensor " + key + " has " + std::to_string(it->second.size()) + " elements, expected " + std::to_string(numel);
            return nullptr;
        }
        return &it->second;
    }
    void bind(const float** slot, int64_t off) { fixups.emplace_back(slot, off); }

    // BatchNorm (eval) as scale/shift, optionally absorbing a conv bias: v = (acc + b - mu) * s + beta
    std::vector<l2s_model::RefreshBn> bn_rec;
    std::vector<l2s_model::RefreshSum> sum_rec;
    void bn(const std::string& p, int c, const std::vector<float>* bias, const float** scale, const float** shift, const std::string& bias_key = std::string()) {
        auto g = get(p + ".weight", c), b = get(p + ".bias", c), mu = get(p + ".running_mean", c), var = get(p + ".running_var", c);
        if (!g || !b || !mu || !var) return;
        int64_t so = blob.alloc(c), ho = blob.alloc(c);
        for (int i = 0; i < c; ++i) {
            float s = (*g)[i] / std::sqrt((*var)[i] + BN_EPS);
            float cb = bias ? (*bias)[i] : 0.f;
            blob.data[so + i] = s;
            blob.data[ho + i] = (cb - (*mu)[i]) * s + (*b)[i];
        }
        bind(scale, so);
        bind(shift, ho);
        bn_rec.push_back({p, bias_key, c, so, ho});
    }
    void copy(const std::string& key, int64_t n, const float** slot) {
        auto v = get(key, n);
        if (!v) return;
        int64_t o = blob.alloc(n);
        std::memcpy(&blob.data[o], v->data(), n * sizeof(float));
        bind(slot, o);
    }
    // Conv1d weight (co, ci, k) -> [co][k*ci] (tap-major K)
    void conv1d_w(const std::string& key, int co, int ci, int k, const float** slot) {
        auto v = get(key, (int64_t)co * ci * k);
        if (!v) return;
        int64_t o = blob.alloc((int64_t)co * ci * k);
        for (int n = 0; n < co; ++n)
            for (int c = 0; c < ci; ++c)
                for (int t = 0; t < k; ++t) blob.data[o + ((int64_t)n * k + t) * ci + c] = (*v)[((int64_t)n * ci + c) * k + t];
        bind(slot, o);
    }
    // depthwise (c,1,3,3) -> [9][c]
    void dw_w(const std::string& key, int c, const float** slot) {
        auto v = get(key, (int64_t)c * 9);
        if (!v) return;
        int64_t o = blob.alloc((int64_t)c * 9);
        for (int ch = 0; ch < c; ++ch)
            for (int t = 0; t < 9; ++t) blob.data[o + (int64_t)t * c + ch] = (*v)[(int64_t)ch * 9 + t];
        bind(slot, o);
    }
    // frag16 packing of rows[n] (each K long) of a virtual [Npad][K] matrix; row_of(n) returns nullptr for zero rows
    template <typename RowFn>
    void frag16(int Npad, int K, RowFn row_of, const float** slot) {
        int64_t o = blob.alloc((int64_t)Npad * K);
        const int NC = K / 16;
        std::vector<float> row(K);
        for (int n = 0; n < Npad; ++n) {
            bool nz = row_of(n, row.data());
            if (!nz) continue;
            for (int k = 0; k < K; ++k) {
                int tile = n >> 4, i = n & 15, c = k >> 4, g = (k >> 2) & 3, e = k & 3;
                blob.data[o + ((int64_t)(tile * NC + c) * 64 + g * 16 + i) * 4 + e] = row[k];
            }
        }
        bind(slot, o);
    }
};

static int lstm_perm_row(int np, int H) {      // packed row (unit-major: 4*unit + gate) -> PyTorch row gate*H + unit
    int unit = np >> 2, gate = np & 3;
    return gate * H + unit;
}


static int pack_host(l2s_model* m, Packer& P, bool& want_enc, bool& want_dec, bool& want_spk) {
    m->w = Weights{};
    Weights& w = m->w;
    const std::string E = "encoder.", Dk = "decoder.";
    // The two halves of the path are packed independently so that a VideoExtractor or a Decoder used on its
    // own (as the reference allows: net.encoder / net.decoder) needs only its own keys.
    auto has_prefix = [&](const std::string& pre) {
        for (auto& kv : m->host) if (kv.first.compare(0, pre.size(), pre) == 0) return true;
        return false;
    };
    const std::string Sk = "speaker_encoder.";
    want_enc = has_prefix(E); want_dec = has_prefix(Dk); want_spk = has_prefix(Sk);
    if (!want_enc && !want_dec && !want_spk) { set_error("l2s_model_finalize: no encoder.* / decoder.* / speaker_encoder.* tensors were set"); return 1; }
    if (want_enc) {

    // ---- frontend: Conv3d (24,3,5,7,7) -> [slab = ci*5+kt][50][32]
    {
        auto v = P.get(E + "frontend3D.0.weight", 24 * 3 * 5 * 49);
        if (v) {
            int64_t o = P.blob.alloc(15 * 50 * 32);
            for (int co = 0; co < 24; ++co)
                for (int ci = 0; ci < 3; ++ci)
                    for (int kt = 0; kt < 5; ++kt)
                        for (int tap = 0; tap < 49; ++tap)
                            P.blob.data[o + ((int64_t)(ci * 5 + kt) * 50 + tap) * 32 + co] = (*v)[(((int64_t)co * 3 + ci) * 5 + kt) * 49 + tap];
            P.bind(&w.fe.w, o);
            // the same weights as split-bf16 operand planes (frontend3d_x3_kernel): per slab, step s = kernel rows 2s, 2s+1 (row 7: zeros),
            // 8 taps per row = one zero tap + the 7 real ones; every value split exactly into hi + mid + lo by truncation
            const int64_t o3 = P.blob.alloc(15 * 18432 / 4);
            unsigned char* base3 = reinterpret_cast<unsigned char*>(&P.blob.data[o3]);
            for (int slab = 0; slab < 15; ++slab)
                for (int st = 0; st < 4; ++st)
                    for (int n = 0; n < 32; ++n)
                        for (int k = 0; k < 16; ++k) {
                            const int kh = 2 * st + (k >> 3), kw = (k & 7) - 1, ci = slab / 5, kt = slab % 5;
                            float x = 0.f;
                            if (n < 24 && kh < 7 && kw >= 0) x = (*v)[(((int64_t)n * 3 + ci) * 5 + kt) * 49 + kh * 7 + kw];
                            uint32_t xb, hb, mb, lb; float r1, r2, tmp;
                            std::memcpy(&xb, &x, 4); hb = xb & 0xFFFF0000u; std::memcpy(&tmp, &hb, 4); r1 = x - tmp;
                            std::memcpy(&mb, &r1, 4); mb &= 0xFFFF0000u; std::memcpy(&tmp, &mb, 4); r2 = r1 - tmp;
                            std::memcpy(&lb, &r2, 4);
                            const uint16_t planes[3] = {(uint16_t)(hb >> 16), (uint16_t)(mb >> 16), (uint16_t)(lb >> 16)};
                            for (int pl = 0; pl < 3; ++pl)
                                std::memcpy(base3 + (int64_t)slab * 18432 + ((st * 3 + pl) * 32 + n) * 48 + k * 2, &planes[pl], 2);
                        }
            P.bind(&w.fe.w3, o3);
            // and as ONE bf16 plane rounded to nearest even, for the bf16 leg (option "infer_bf16")
            const int64_t o1 = P.blob.alloc(15 * 6144 / 4);
            unsigned char* base1 = reinterpret_cast<unsigned char*>(&P.blob.data[o1]);
            for (int slab = 0; slab < 15; ++slab)
                for (int st = 0; st < 4; ++st)
                    for (int n = 0; n < 32; ++n)
                        for (int k = 0; k < 16; ++k) {
                            const int kh = 2 * st + (k >> 3), kw = (k & 7) - 1, ci = slab / 5, kt = slab % 5;
                            float x = 0.f;
                            if (n < 24 && kh < 7 && kw >= 0) x = (*v)[(((int64_t)n * 3 + ci) * 5 + kt) * 49 + kh * 7 + kw];
                            uint32_t xb; std::memcpy(&xb, &x, 4);
                            const uint16_t r = (uint16_t)((xb + 0x7FFFu + ((xb >> 16) & 1u)) >> 16);
                            std::memcpy(base1 + (int64_t)slab * 6144 + (st * 32 + n) * 48 + k * 2, &r, 2);
                        }
            P.bind(&w.fe.w1, o1);
        }
        P.bn(E + "frontend3D.1", 24, nullptr, &w.fe.scale, &w.fe.shift);
        P.copy(E + "frontend3D.2.weight", 24, &w.fe.slope);
    }
    // ---- ShuffleNet units
    {
        int u = 0, cin = STAGE_CH[0];
        for (int st = 0; st < 3; ++st) {
            int cout = STAGE_CH[st + 1], half = cout / 2;
            for (int r = 0; r < STAGE_REP[st]; ++r, ++u) {
                UnitW& U = w.unit[u];
                std::string p = E + "trunk.0." + std::to_string(u) + ".";
                U.stride2 = (r == 0);
                U.cin = cin;
                U.half = half;
                int pw1_in = U.stride2 ? cin : half;
                if (U.stride2) {
                    P.dw_w(p + "banch1.0.weight", cin, &U.b1_dw.w9);
                    P.bn(p + "banch1.1", cin, nullptr, &U.b1_dw.scale, &U.b1_dw.shift);
                    P.copy(p + "banch1.2.weight", (int64_t)half * cin, &U.b1_pw.W);
                    P.bn(p + "banch1.3", half, nullptr, &U.b1_pw.scale, &U.b1_pw.shift);
                }
                P.copy(p + "banch2.0.weight", (int64_t)half * pw1_in, &U.pw1.W);
                P.bn(p + "banch2.1", half, nullptr, &U.pw1.scale, &U.pw1.shift);
                P.dw_w(p + "banch2.3.weight", half, &U.dw.w9);
                P.bn(p + "banch2.4", half, nullptr, &U.dw.scale, &U.dw.shift);
                P.copy(p + "banch2.5.weight", (int64_t)half * half, &U.pw2.W);
                P.bn(p + "banch2.6", half, nullptr, &U.pw2.scale, &U.pw2.shift);
                {   // fused-unit operands: the pointwise weights in frag16 layout, K zero-padded to a multiple of 16
                    U.kpad = pad16(half);
                    U.kin = pad16(pw1_in);
                    struct { const char* key; int K, Kp; const float** slot; } fr[3] = {
                        {"banch2.0.weight", pw1_in, U.stride2 ? U.kin : U.kpad, &U.pw1_frag},
                        {"banch2.5.weight", half, U.kpad, &U.pw2_frag},
                        {"banch1.2.weight", cin, U.kin, &U.b1_frag}};
                    for (int which = 0; which < (U.stride2 ? 3 : 2); ++which) {
                        const int K = fr[which].K, Kp = fr[which].Kp;
                        auto wv = P.get(p + fr[which].key, (int64_t)half * K);
                        if (!wv) continue;
                        P.frag16(pad16(half), Kp, [&](int n, float* row) {
                            if (n >= half) return false;
                            std::memset(row, 0, sizeof(float) * Kp);
                            std::memcpy(row, wv->data() + (int64_t)n * K, sizeof(float) * K);
                            return true;
                        }, fr[which].slot);
                    }
                }
                cin = cout;
            }
        }
        P.copy(E + "trunk.1.0.weight", (int64_t)LAST_CH * STAGE_CH[3], &w.conv_last.W);
        P.bn(E + "trunk.1.1", LAST_CH, nullptr, &w.conv_last.scale, &w.conv_last.shift);
    }
    }   // want_enc
    if (want_dec) {
    // ---- decoder prologue
    auto linear = [&](const std::string& p, int co, int ci, ConvW& c) {
        P.copy(p + ".weight", (int64_t)co * ci, &c.W);
        P.copy(p + ".bias", co, &c.shift);
    };
    P.conv1d_w(Dk + "residual_bottleneck.weight", D, 1024, 1, &w.resid.W);
    P.copy(Dk + "residual_bottleneck.bias", D, &w.resid.shift);
    linear(Dk + "encoder_site.0.linear_layer", D, 256, w.enc_site);
    P.copy(Dk + "encoder_site.1.w", D, &w.enc_site.actw);
    linear(Dk + "attention_site.0.linear_layer", D, 256, w.attn_site);
    P.copy(Dk + "attention_site.1.w", D, &w.attn_site.actw);
    linear(Dk + "E_C.linear_layer", D, 1024, w.e_c);
    linear(Dk + "encoder_proj.linear_layer", D, 1024, w.enc_proj);
    {   // BiLSTM: input weights of both directions stacked [4096][1024]; b_ih + b_hh folded into the GEMM shift
        const char* suf[2] = {"l0", "l0_reverse"};
        int64_t wo = P.blob.alloc((int64_t)4096 * 1024), bo = P.blob.alloc(4096);
        for (int d = 0; d < 2; ++d) {
            auto wi = P.get(Dk + "encoder_rnn.weight_ih_" + suf[d], (int64_t)2048 * 1024);
            auto bi = P.get(Dk + "encoder_rnn.bias_ih_" + suf[d], 2048), bh = P.get(Dk + "encoder_rnn.bias_hh_" + suf[d], 2048);
            auto wh = P.get(Dk + "encoder_rnn.weight_hh_" + suf[d], (int64_t)2048 * 512);
            if (!wi || !bi || !bh || !wh) continue;
            std::memcpy(&P.blob.data[wo + (int64_t)d * 2048 * 1024], wi->data(), sizeof(float) * 2048 * 1024);
            for (int i = 0; i < 2048; ++i) P.blob.data[bo + d * 2048 + i] = (*bi)[i] + (*bh)[i];
            P.sum_rec.push_back({Dk + "encoder_rnn.bias_ih_" + suf[d], Dk + "encoder_rnn.bias_hh_" + suf[d], 2048, 0, bo + d * 2048});
            P.frag16(2048, 512, [&](int np, float* row) {
                std::memcpy(row, wh->data() + (int64_t)lstm_perm_row(np, 512) * 512, sizeof(float) * 512);
                return true;
            }, &w.whh[d].W);
            w.whh[d].N = 2048; w.whh[d].K = 512; w.whh[d].tiles = 128;
        }
        P.bind(&w.wih_cat, wo);
        P.bind(&w.bih_cat, bo);
    }
    for (int kv = 0; kv < 2; ++kv) {
        std::string p = Dk + (kv == 0 ? "K" : "V");
        for (int j = 0; j < 4; ++j) {
            std::string c = p + ".0.conv." + std::to_string(j);
            P.conv1d_w(c + ".0.weight", D, D, MH_KS[j], &w.mh_branch[kv][j].W);
            P.bn(c + ".1", D, P.get(c + ".0.bias", D), &w.mh_branch[kv][j].scale, &w.mh_branch[kv][j].shift, c + ".0.bias");
        }
        P.conv1d_w(p + ".0.bottleneck.weight", D, 5 * D, 1, &w.mh_bott[kv].W);
        P.copy(p + ".0.bottleneck.bias", D, &w.mh_bott[kv].shift);
        P.copy(p + ".1.w", D, &w.mh_bott[kv].actw);
    }
    P.copy(Dk + "positional_encodings.pos_table", (int64_t)L2S_MAX_STEPS * D, &w.pos);
    for (int j = 0; j < 4; ++j) {
        std::string c = Dk + "content.agg." + std::to_string(j);
        P.conv1d_w(c + ".0.weight", D, D, CT_KS[j], &w.ct_branch[j].W);
        P.bn(c + ".1", D, P.get(c + ".0.bias", D), &w.ct_branch[j].scale, &w.ct_branch[j].shift, c + ".0.bias");
    }
    P.conv1d_w(Dk + "content.bottleneck.weight", 256, 5 * D, 1, &w.ct_bott.W);
    P.copy(Dk + "content.bottleneck.bias", 256, &w.ct_bott.shift);
    linear(Dk + "content.K.0", 256, 256, w.ct_k0);
    linear(Dk + "content.K.2", 256, 256, w.ct_k2);
    linear(Dk + "content.location_fc.0", 256, 256, w.ct_fc0);
    linear(Dk + "content.location_fc.2", 256, 256, w.ct_fc2);
    linear(Dk + "content.location_fc.4", VOC, 256, w.ct_fc4);
    {   // word_embeddings (501,256) -> transposed, K padded: [256][504]
        auto v = P.get(Dk + "content.word_embeddings", (int64_t)VOC * 256);
        if (v) {
            int64_t o = P.blob.alloc((int64_t)256 * VOCP);
            for (int n = 0; n < 256; ++n)
                for (int k = 0; k < VOC; ++k) P.blob.data[o + (int64_t)n * VOCP + k] = (*v)[(int64_t)k * 256 + n];
            P.bind(&w.ct_emb.W, o);
        }
    }
    // ---- decode-step weights in frag16 layout
    auto sk_linear = [&](const std::string& wkey, const std::string& bkey, int N, int K, SkW& s) {
        auto wv = P.get(wkey, (int64_t)N * K);
        int Np = pad16(N);
        if (wv)
            P.frag16(Np, K, [&](int n, float* row) {
                if (n >= N) return false;
                std::memcpy(row, wv->data() + (int64_t)n * K, sizeof(float) * K);
                return true;
            }, &s.W);
        auto bv = P.get(bkey, N);
        if (bv) {
            int64_t o = P.blob.alloc(Np);
            std::memcpy(&P.blob.data[o], bv->data(), sizeof(float) * N);
            P.bind(&s.bias, o);
        }
        s.N = N; s.K = K; s.tiles = Np / 16;
    };
    sk_linear(Dk + "prenet.0.linear_layer.weight", Dk + "prenet.0.linear_layer.bias", 256, NM, w.pre1);
    P.copy(Dk + "prenet.1.w", 256, &w.pre1.actw);
    sk_linear(Dk + "prenet.3.linear_layer.weight", Dk + "prenet.3.linear_layer.bias", 256, 256, w.pre2);
    P.copy(Dk + "prenet.4.w", 256, &w.pre2.actw);
    sk_linear(Dk + "Q.0.linear_layer.weight", Dk + "Q.0.linear_layer.bias", D, 1024, w.q);
    P.copy(Dk + "Q.1.w", D, &w.q.actw);
    sk_linear(Dk + "content.Q.0.weight", Dk + "content.Q.0.bias", 256, 1024, w.cq);
    sk_linear(Dk + "attention_proj.linear_layer.weight", Dk + "attention_proj.linear_layer.bias", 256, D, w.aproj);
    for (int l = 0; l < 2; ++l) {
        SkW& s = l == 0 ? w.lstm0 : w.lstm1;
        std::string sl = "l" + std::to_string(l);
        auto wi = P.get(Dk + "decoder_rnn.weight_ih_" + sl, (int64_t)2048 * 512), wh = P.get(Dk + "decoder_rnn.weight_hh_" + sl, (int64_t)2048 * 512);
        auto bi = P.get(Dk + "decoder_rnn.bias_ih_" + sl, 2048), bh = P.get(Dk + "decoder_rnn.bias_hh_" + sl, 2048);
        if (!wi || !wh || !bi || !bh) continue;
        P.frag16(2048, 1024, [&](int np, float* row) {
            int r = lstm_perm_row(np, 512);
            std::memcpy(row, wi->data() + (int64_t)r * 512, sizeof(float) * 512);
            std::memcpy(row + 512, wh->data() + (int64_t)r * 512, sizeof(float) * 512);
            return true;
        }, &s.W);
        int64_t o = P.blob.alloc(2048);
        for (int np = 0; np < 2048; ++np) { int r = lstm_perm_row(np, 512); P.blob.data[o + np] = (*bi)[r] + (*bh)[r]; }
        P.bind(&s.bias, o);
        P.sum_rec.push_back({Dk + "decoder_rnn.bias_ih_" + sl, Dk + "decoder_rnn.bias_hh_" + sl, 2048, 512, o});
        s.N = 2048; s.K = 1024; s.tiles = 128;
    }
    {   // fc_out (80 rows) + stop-token row over h1 (row 80) in one weight: [96][512]
        auto wf = P.get(Dk + "fc_out.linear_layer.weight", (int64_t)NM * D), bf = P.get(Dk + "fc_out.linear_layer.bias", NM);
        auto ws = P.get(Dk + "stop_token_layer.linear_layer.weight", 1024);
        if (wf && bf && ws) {
            P.frag16(96, D, [&](int n, float* row) {
                if (n < NM) std::memcpy(row, wf->data() + (int64_t)n * D, sizeof(float) * D);
                else if (n == NM) std::memcpy(row, ws->data(), sizeof(float) * D);
                else return false;
                return true;
            }, &w.fc.W);
            int64_t o = P.blob.alloc(96);
            std::memcpy(&P.blob.data[o], bf->data(), sizeof(float) * NM);
            P.bind(&w.fc.bias, o);
            int64_t t = P.blob.alloc(D);
            std::memcpy(&P.blob.data[t], ws->data() + D, sizeof(float) * D);
            P.bind(&w.stop_tail, t);
        }
        w.fc.N = NM + 1; w.fc.K = D; w.fc.tiles = 6;
        P.copy(Dk + "stop_token_layer.linear_layer.bias", 1, &w.stop_bias);
    }
    {   // Phase merging: two linear maps that are applied back to back with nothing in between are
        // pre-multiplied once, in fp64, and rounded to fp32:
        //   prenet1(fc_out(h1)) = PSine(W_p1 (W_out h1 + b_out) + b_p1) = PSine((W_p1 W_out) h1 + (W_p1 b_out + b_p1))
        //   LSTM0 gates on u = p2 + W_ap av + b_ap:  W_ih[:,256:] u = W_ih[:,256:] p2 + (W_ih[:,256:] W_ap) av + W_ih[:,256:] b_ap
        auto wp1 = P.get(Dk + "prenet.0.linear_layer.weight", (int64_t)256 * NM), bp1 = P.get(Dk + "prenet.0.linear_layer.bias", 256);
        auto wo = P.get(Dk + "fc_out.linear_layer.weight", (int64_t)NM * D), bo = P.get(Dk + "fc_out.linear_layer.bias", NM);
        if (wp1 && bp1 && wo && bo) {
            std::vector<float> wf((size_t)256 * D);
            int64_t bo_off = P.blob.alloc(256);
            std::vector<double> row(D);
            for (int n = 0; n < 256; ++n) {
                std::fill(row.begin(), row.end(), 0.0);
                double bacc = (*bp1)[n];
                for (int k = 0; k < NM; ++k) {
                    const double a = (*wp1)[(int64_t)n * NM + k];
                    const float* wr = wo->data() + (int64_t)k * D;
                    for (int j = 0; j < D; ++j) row[j] += a * wr[j];
                    bacc += a * (*bo)[k];
                }
                for (int j = 0; j < D; ++j) wf[(size_t)n * D + j] = (float)row[j];
                P.blob.data[bo_off + n] = (float)bacc;
            }
            P.frag16(256, D, [&](int n, float* r) { std::memcpy(r, wf.data() + (size_t)n * D, sizeof(float) * D); return true; }, &w.pre1f.W);
            P.bind(&w.pre1f.bias, bo_off);
            P.copy(Dk + "prenet.1.w", 256, &w.pre1f.actw);
            w.pre1f.N = 256; w.pre1f.K = D; w.pre1f.tiles = 16;
        }
        auto wi = P.get(Dk + "decoder_rnn.weight_ih_l0", (int64_t)2048 * 512), wh = P.get(Dk + "decoder_rnn.weight_hh_l0", (int64_t)2048 * 512);
        auto bi = P.get(Dk + "decoder_rnn.bias_ih_l0", 2048), bh = P.get(Dk + "decoder_rnn.bias_hh_l0", 2048);
        auto wap = P.get(Dk + "attention_proj.linear_layer.weight", (int64_t)256 * D), bap = P.get(Dk + "attention_proj.linear_layer.bias", 256);
        if (wi && wh && bi && bh && wap && bap) {
            std::vector<float> prod((size_t)2048 * D);      // (W_ih[:,256:512] @ W_ap) in PyTorch row order
            std::vector<float> badd(2048);
            std::vector<double> row(D);
            for (int r = 0; r < 2048; ++r) {
                std::fill(row.begin(), row.end(), 0.0);
                double bacc = 0.0;
                for (int k = 0; k < 256; ++k) {
                    const double a = (*wi)[(int64_t)r * 512 + 256 + k];
                    const float* wr = wap->data() + (int64_t)k * D;
                    for (int j = 0; j < D; ++j) row[j] += a * wr[j];
                    bacc += a * (*bap)[k];
                }
                for (int j = 0; j < D; ++j) prod[(size_t)r * D + j] = (float)row[j];
                badd[r] = (float)((double)(*bi)[r] + (double)(*bh)[r] + bacc);
            }
            P.frag16(2048, 1536, [&](int np, float* rowp) {
                int r = lstm_perm_row(np, 512);
                std::memcpy(rowp, wi->data() + (int64_t)r * 512, sizeof(float) * 512);          // [cc | p2] columns of W_ih
                std::memcpy(rowp + 512, prod.data() + (size_t)r * D, sizeof(float) * D);         // av columns
                std::memcpy(rowp + 1024, wh->data() + (int64_t)r * 512, sizeof(float) * 512);   // h0 columns
                return true;
            }, &w.lstm0f.W);
            int64_t o = P.blob.alloc(2048);
            for (int np = 0; np < 2048; ++np) P.blob.data[o + np] = badd[lstm_perm_row(np, 512)];
            P.bind(&w.lstm0f.bias, o);
            w.lstm0f.N = 2048; w.lstm0f.K = 1536; w.lstm0f.tiles = 128;
            // the same step with attention_proj applied to the VALUES once, in the prologue (V' = V W_ap^T + b_ap; the attention weights sum
            // to one, so a @ V' = W_ap (a @ v) + b_ap): LSTM0 then reads o = a @ V' (256 wide) through a second copy of W_ih's u columns -
            // K = 1280 instead of 1536, verbatim copies of the parameters only (the device-side refresh keeps them current by itself)
            P.frag16(2048, 1280, [&](int np, float* rowp) {
                int r = lstm_perm_row(np, 512);
                std::memcpy(rowp, wi->data() + (int64_t)r * 512, sizeof(float) * 512);                 // [cc | p2] columns of W_ih
                std::memcpy(rowp + 512, wi->data() + (int64_t)r * 512 + 256, sizeof(float) * 256);     // o: the u columns again
                std::memcpy(rowp + 768, wh->data() + (int64_t)r * 512, sizeof(float) * 512);           // h0 columns
                return true;
            }, &w.lstm0v.W);
            for (size_t fi = 0; fi < P.fixups.size(); ++fi)      // the plain LSTM0 bias (b_ih + b_hh, kept current by the refresh's bias-sum records)
                if (P.fixups[fi].first == &w.lstm0.bias) { P.bind(&w.lstm0v.bias, P.fixups[fi].second); break; }
            w.lstm0v.N = 2048; w.lstm0v.K = 1280; w.lstm0v.tiles = 128;
            P.copy(Dk + "attention_proj.linear_layer.weight", (int64_t)256 * D, &w.vproj.W);
            P.copy(Dk + "attention_proj.linear_layer.bias", 256, &w.vproj.shift);
        }
    }
    P.copy(Dk + "BOS", NM, &w.bos);
    P.copy(Dk + "temperature", 1, &w.tau);
    P.copy(Dk + "content.temperature", 1, &w.tau_c);
    // ---- postnet
    for (int i = 0; i < 5; ++i) {
        int ci = i == 0 ? NM : D, co = i == 4 ? NM : D;
        std::string c = Dk + "postnet.convolutions." + std::to_string(i);
        P.conv1d_w(c + ".0.conv.weight", co, ci, 5, &w.post[i].W);
        P.bn(c + ".1", co, P.get(c + ".0.conv.bias", co), &w.post[i].scale, &w.post[i].shift, c + ".0.conv.bias");
        if (i < 4) P.copy(Dk + "postnet.sin_activation." + std::to_string(i) + ".w", D, &w.post[i].actw);
    }
    }   // want_dec
    if (want_spk) {
        constexpr int NFFT = 400, NF = 201, NMEL = 40, NFP = 204;
        {   // hann window (periodic), real-DFT matrix [cos | sin] and HTK mel filterbank, all computed in fp64
            const double PI = 3.14159265358979323846;
            int64_t wo = P.blob.alloc(NFFT), dof = P.blob.alloc((int64_t)2 * NF * NFFT), fo = P.blob.alloc((int64_t)NMEL * NFP);
            for (int j = 0; j < NFFT; ++j) P.blob.data[wo + j] = (float)(0.5 - 0.5 * std::cos(2.0 * PI * j / NFFT));
            for (int k = 0; k < NF; ++k)
                for (int j = 0; j < NFFT; ++j) {
                    const double ang = 2.0 * PI * (double)((int64_t)k * j % NFFT) / NFFT;
                    P.blob.data[dof + (int64_t)k * NFFT + j] = (float)std::cos(ang);
                    P.blob.data[dof + (int64_t)(NF + k) * NFFT + j] = (float)std::sin(ang);
                }
            // torchaudio.functional.create_fb_matrix(n_freqs=201, f_min=0, f_max=8000, n_mels=40, sample_rate=16000, norm=None), HTK scale
            std::vector<double> fpts(NMEL + 2);
            const double m_min = 0.0, m_max = 2595.0 * std::log10(1.0 + 8000.0 / 700.0);
            for (int i = 0; i < NMEL + 2; ++i) {
                const double mpt = m_min + (m_max - m_min) * i / (NMEL + 1);
                fpts[i] = 700.0 * (std::pow(10.0, mpt / 2595.0) - 1.0);
            }
            for (int k = 0; k < NF; ++k) {
                const double f = 8000.0 * k / (NF - 1);
                for (int mm = 0; mm < NMEL; ++mm) {
                    const double down = (f - fpts[mm]) / (fpts[mm + 1] - fpts[mm]);
                    const double up = (fpts[mm + 2] - f) / (fpts[mm + 2] - fpts[mm + 1]);
                    P.blob.data[fo + (int64_t)mm * NFP + k] = (float)std::max(0.0, std::min(down, up));
                }
            }
            P.bind(&w.spk_window, wo); P.bind(&w.spk_dft, dof); P.bind(&w.spk_fbT, fo);
        }
        for (int l = 0; l < 3; ++l) {
            const int in = l == 0 ? 40 : 256;
            const std::string sl = "l" + std::to_string(l);
            auto wi = P.get(Sk + "lstm.weight_ih_" + sl, (int64_t)1024 * in), wh = P.get(Sk + "lstm.weight_hh_" + sl, (int64_t)1024 * 256);
            auto bi = P.get(Sk + "lstm.bias_ih_" + sl, 1024), bh = P.get(Sk + "lstm.bias_hh_" + sl, 1024);
            if (!wi || !wh || !bi || !bh) continue;
            P.copy(Sk + "lstm.weight_ih_" + sl, (int64_t)1024 * in, &w.spk_ih[l].W);
            int64_t bo = P.blob.alloc(1024);
            for (int i = 0; i < 1024; ++i) P.blob.data[bo + i] = (*bi)[i] + (*bh)[i];
            P.bind(&w.spk_ih[l].shift, bo);
            P.frag16(1024, 256, [&](int np, float* row) {
                std::memcpy(row, wh->data() + (int64_t)lstm_perm_row(np, 256) * 256, sizeof(float) * 256);
                return true;
            }, &w.spk_hh[l].W);
            w.spk_hh[l].N = 1024; w.spk_hh[l].K = 256; w.spk_hh[l].tiles = 64;
        }
        P.copy(Sk + "linear.weight", (int64_t)256 * 256, &w.spk_linear.W);
        P.copy(Sk + "linear.bias", 256, &w.spk_linear.shift);
    }   // want_spk
    if (!P.missing.empty()) { set_error("l2s_model_finalize: " + P.missing); return 1; }
    return 0;
}

static int build_refresh_map(l2s_model* m, const Packer& P, hipStream_t stream);

// bf16 planes of the two decoder LSTM weights (layer 0 in its unmerged [content | u | h0] form, layer 1) for the split-bf16 LSTM blocks: derived on the
// device from the packed fp32 fragments, after every pack and every device-side refresh
static int derive_lstm_planes(l2s_model* m, hipStream_t s) {
    Weights& w = m->w;
    SkW* const sk[4] = {&w.lstm0, &w.lstm1, &w.whh[0], &w.whh[1]};          // decoder layers 0 (unmerged [content | u | h0]) and 1, the BiLSTM's two directions
    for (SkW* k : sk) k->W3 = nullptr;
    if (!m->has_dec) return 0;
    int64_t bytes[4], total = 0;
    for (int i = 0; i < 4; ++i) {
        if (!sk[i]->W || sk[i]->K % 256) return 0;
        bytes[i] = (int64_t)sk[i]->tiles * sk[i]->K * 96;                    // 16 columns x K x 6 bytes per tile
        total += bytes[i];
    }
    if (!m->lstm_planes) L2S_CHECK_HIP(hipMalloc(&m->lstm_planes, total));
    char* base = reinterpret_cast<char*>(m->lstm_planes);
    for (int i = 0; i < 4; ++i) {
        if (launch_skx_planes(sk[i]->W, sk[i]->tiles, sk[i]->K, base, s)) return 1;
        sk[i]->W3 = base;
        base += bytes[i];
    }
    return 0;
}

// bf16 planes of the constant weights that meet the split-bf16 GEMM's wide tile (post-net layers 0-3, the BiLSTM input matrix, conv_last) for its LDS-DMA
// weight operand: derived on the device
// from the packed fp32 [N][K] matrices, after every pack and every device-side refresh
static int derive_gemm_planes(l2s_model* m, hipStream_t s) {
    Weights& w = m->w;
    for (int i = 0; i < 5; ++i) w.post[i].W3 = nullptr;
    w.wih_cat3 = nullptr; w.conv_last.W3 = nullptr;
    for (int kv = 0; kv < 2; ++kv) for (int j = 0; j < 4; ++j) w.mh_branch[kv][j].W3 = nullptr;
    struct Item { const float* W; int N, K; const void** slot; };
    std::vector<Item> items;
    if (m->has_dec) {
        const int Ks[4] = {5 * NM, 5 * 512, 5 * 512, 5 * 512};
        for (int i = 0; i < 4; ++i) items.push_back({w.post[i].W, 512, Ks[i], &w.post[i].W3});
        items.push_back({w.wih_cat, 4096, 1024, &w.wih_cat3});
        for (int kv = 0; kv < 2; ++kv)
            for (int j = 0; j < 4; ++j) items.push_back({w.mh_branch[kv][j].W, 512, 512 * MH_KS[j], &w.mh_branch[kv][j].W3});      // the eight MultiHop convs (k = 1, 3, 7, 11; K and V)
    }
    if (m->has_enc) items.push_back({w.conv_last.W, LAST_CH, STAGE_CH[3], &w.conv_last.W3});
    int64_t total = 0;
    for (const Item& it : items) { if (!it.W) return 0; total += (int64_t)it.N * it.K * 6; }
    if (!total) return 0;
    if (!m->gemm_planes) L2S_CHECK_HIP(hipMalloc(&m->gemm_planes, total));
    char* base = reinterpret_cast<char*>(m->gemm_planes);
    for (const Item& it : items) {
        if (launch_gemm_planes(it.W, it.N, it.K, base, s)) return 1;
        *it.slot = base;
        base += (int64_t)it.N * it.K * 6;
    }
    return 0;
}

// bf16 operand planes of the fused ShuffleNet units' pointwise convs (option "trunk_x3"): derived on the device from the packed [N][K] matrices,
// after every pack and every device-side refresh
static int derive_unit_planes(l2s_model* m, hipStream_t s) {
    Weights& w = m->w;
    for (UnitW& U : w.unit) { U.pw1_p3 = nullptr; U.pw2_p3 = nullptr; U.b1_p3 = nullptr; }
    if (!m->has_enc) return 0;
    struct Item { const float* W; int N, K; const void** slot; };
    std::vector<Item> items;
    for (UnitW& U : w.unit) {
        const int pw1_in = U.stride2 ? U.cin : U.half;
        items.push_back({U.pw1.W, U.half, pw1_in, &U.pw1_p3});
        items.push_back({U.pw2.W, U.half, U.half, &U.pw2_p3});
        if (U.stride2) items.push_back({U.b1_pw.W, U.half, U.cin, &U.b1_p3});
    }
    int64_t total = 0;
    for (const Item& it : items) { if (!it.W) return 0; total += su_planes_bytes(it.N, it.K); }
    if (!m->unit_planes) L2S_CHECK_HIP(hipMalloc(&m->unit_planes, total));
    char* base = reinterpret_cast<char*>(m->unit_planes);
    for (const Item& it : items) {
        if (launch_su_planes(it.W, it.N, it.K, base, s)) return 1;
        *it.slot = base;
        base += su_planes_bytes(it.N, it.K);
    }
    return 0;
}

static int pack_model(l2s_model* m, hipStream_t stream) {
    Packer P{m};
    bool want_enc = false, want_dec = false, want_spk = false;
    if (pack_host(m, P, want_enc, want_dec, want_spk)) return 1;
    if (m->opt.refresh_map && build_refresh_map(m, P, stream)) return 1;
    m->folded_valid = true; m->planes_valid = true;

    // upload and patch pointers
    for (auto& g : m->graphs) { (void)hipGraphExecDestroy(g.exec); (void)hipGraphDestroy(g.graph); }
    m->graphs.clear();
    if (m->blob) { (void)hipFree(m->blob); m->blob = nullptr; }
    m->blob_floats = (int64_t)P.blob.data.size();
    L2S_CHECK_HIP(hipMalloc(&m->blob, m->blob_floats * sizeof(float)));
    L2S_CHECK_HIP(hipMemcpyAsync(m->blob, P.blob.data.data(), m->blob_floats * sizeof(float), hipMemcpyHostToDevice, stream));
    L2S_CHECK_HIP(hipStreamSynchronize(stream));     // the host staging vector dies with this scope
    for (auto& f : P.fixups) *f.first = m->blob + f.second;
    m->finalized = true;
    m->has_enc = want_enc;
    m->has_dec = want_dec;
    m->has_spk = want_spk;
    if (m->lstm_planes) { (void)hipFree(m->lstm_planes); m->lstm_planes = nullptr; }
    if (derive_lstm_planes(m, stream)) return 1;
    if (m->gemm_planes) { (void)hipFree(m->gemm_planes); m->gemm_planes = nullptr; }
    if (derive_gemm_planes(m, stream)) return 1;
    if (m->unit_planes) { (void)hipFree(m->unit_planes); m->unit_planes = nullptr; }
    if (derive_unit_planes(m, stream)) return 1;
    return 0;
}

// ------------------------------------------------------------------------------------------------ device-side refresh (training)
// Which checkpoint element does each blob float copy?  Pack a shadow checkpoint whose elements carry their own global id as raw bits
// (ids < 2^31 - 2^23 are finite positive floats, so plain copies preserve them; arithmetic on them produces other patterns), then keep an
// entry only if the real blob holds exactly the value of the element the id names.  Computed entries (BatchNorm folds, bias sums) are
// refreshed from the records the packer left; the phase-merged step weights are fp64 products and are invalidated instead.
static int build_refresh_map(l2s_model* m, const Packer& P, hipStream_t stream) {
    std::vector<std::string> keys;
    for (auto& kv : m->host) keys.push_back(kv.first);
    std::sort(keys.begin(), keys.end());
    std::vector<int64_t> base(keys.size() + 1, 0);
    for (size_t i = 0; i < keys.size(); ++i) base[i + 1] = base[i] + (int64_t)m->host[keys[i]].size();
    L2S_REQUIRE(base.back() < 0x7F000000LL, "too many checkpoint elements for the refresh map");
    std::unordered_map<std::string, std::vector<float>> shadow;
    for (size_t i = 0; i < keys.size(); ++i) {
        std::vector<float> v(m->host[keys[i]].size());
        for (size_t j = 0; j < v.size(); ++j) { const uint32_t id = (uint32_t)(base[i] + (int64_t)j + 1); std::memcpy(&v[j], &id, 4); }
        shadow.emplace(keys[i], std::move(v));
    }
    Weights saved = m->w;
    m->host.swap(shadow);
    Packer P2{m};
    bool e = false, d = false, k = false;
    const int rc = pack_host(m, P2, e, d, k);
    m->host.swap(shadow);
    m->w = saved;
    if (rc) return 1;
    L2S_REQUIRE(P2.blob.data.size() == P.blob.data.size(), "refresh map: shadow pack differs in size");
    const int64_t n = (int64_t)P.blob.data.size();
    std::vector<int32_t> rk(n, -1), ri(n, 0);
    size_t cur = 0;
    int64_t copies = 0;
    for (int64_t i = 0; i < n; ++i) {
        uint32_t id; std::memcpy(&id, &P2.blob.data[i], 4);
        if (id == 0 || (int64_t)id > base.back()) continue;
        const int64_t g = (int64_t)id - 1;
        if (!(g >= base[cur] && g < base[cur + 1])) cur = (size_t)(std::upper_bound(base.begin(), base.end(), g) - base.begin()) - 1;
        const std::vector<float>& src = m->host[keys[cur]];
        const int64_t j = g - base[cur];
        uint32_t a, b; std::memcpy(&a, &P.blob.data[i], 4); std::memcpy(&b, &src[j], 4);
        if (a != b) continue;
        rk[i] = (int32_t)cur; ri[i] = (int32_t)j; ++copies;
    }
    if (m->r_key) { (void)hipFree(m->r_key); m->r_key = nullptr; }
    if (m->r_idx) { (void)hipFree(m->r_idx); m->r_idx = nullptr; }
    L2S_CHECK_HIP(hipMalloc(&m->r_key, n * sizeof(int32_t)));
    L2S_CHECK_HIP(hipMalloc(&m->r_idx, n * sizeof(int32_t)));
    L2S_CHECK_HIP(hipMemcpyAsync(m->r_key, rk.data(), n * sizeof(int32_t), hipMemcpyHostToDevice, stream));
    L2S_CHECK_HIP(hipMemcpyAsync(m->r_idx, ri.data(), n * sizeof(int32_t), hipMemcpyHostToDevice, stream));
    L2S_CHECK_HIP(hipStreamSynchronize(stream));
    m->r_keys = keys;
    m->r_bn = P.bn_rec;
    m->r_sum = P.sum_rec;
    (void)copies;
    return 0;
}

struct RBn { const float *g, *b, *mu, *var, *bias; float *scale, *shift; int c; };
struct RSum { const float *a, *b; float* dst; int n, perm_H; };

__global__ __launch_bounds__(256) void refresh_gather_kernel(float* __restrict__ blob, const int32_t* __restrict__ rk, const int32_t* __restrict__ ri,
                                                             const float* const* __restrict__ ptrs, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int k = rk[i];
        if (k < 0) continue;
        const float* src = ptrs[k];
        if (src) blob[i] = src[ri[i]];
    }
}
__global__ __launch_bounds__(256) void refresh_bn_kernel(const RBn* __restrict__ recs) {
    const RBn r = recs[blockIdx.y];
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= r.c) return;
    const float s = __fdiv_rn(r.g[i], __fsqrt_rn(r.var[i] + BN_EPS));   // correctly rounded, like the host packer's expression
    const float cb = r.bias ? r.bias[i] : 0.f;
    r.scale[i] = s;
    r.shift[i] = __fadd_rn(__fmul_rn(__fsub_rn(cb, r.mu[i]), s), r.b[i]);      // no fused multiply-add: the host packer's x86 code has none either
}
__global__ __launch_bounds__(256) void refresh_sum_kernel(const RSum* __restrict__ recs) {
    const RSum r = recs[blockIdx.y];
    const int np = blockIdx.x * 256 + threadIdx.x;
    if (np >= r.n) return;
    const int src = r.perm_H ? (np & 3) * r.perm_H + (np >> 2) : np;
    r.dst[np] = r.a[src] + r.b[src];
}

// ---- the front-end conv's bf16 operand planes (FrontendW::w3 / w1), re-derived from the bound Conv3d weight exactly as pack_host derives
// them: per slab (ci, kt), step st = kernel rows 2st, 2st+1 (row 7: zeros), 8 taps per row = one zero tap + the 7 real ones; w3 = the
// truncation split hi + mid + lo (exact), w1 = one plane rounded to nearest even
__global__ __launch_bounds__(256) void refresh_frontend_planes_kernel(const float* __restrict__ w, uint16_t* __restrict__ w3, uint16_t* __restrict__ w1) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= 15 * 4 * 32 * 16) return;
    const int k = idx & 15, n = (idx >> 4) & 31, st = (idx >> 9) & 3, slab = idx >> 11;
    const int kh = 2 * st + (k >> 3), kw = (k & 7) - 1, ci = slab / 5, kt = slab % 5;
    float x = 0.f;
    if (n < 24 && kh < 7 && kw >= 0) x = w[(((int64_t)n * 3 + ci) * 5 + kt) * 49 + kh * 7 + kw];
    const uint32_t xb = __float_as_uint(x), hb = xb & 0xFFFF0000u;
    const float r1 = __fsub_rn(x, __uint_as_float(hb));
    const uint32_t mb = __float_as_uint(r1) & 0xFFFF0000u;
    const float r2 = __fsub_rn(r1, __uint_as_float(mb));
    const uint32_t lb = __float_as_uint(r2);
    const uint16_t planes[3] = {(uint16_t)(hb >> 16), (uint16_t)(mb >> 16), (uint16_t)(lb >> 16)};
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) w3[(int64_t)slab * 9216 + ((st * 3 + pl) * 32 + n) * 24 + k] = planes[pl];      // byte offsets / 2 (18432-byte slabs, 48-byte rows)
    w1[(int64_t)slab * 3072 + (st * 32 + n) * 24 + k] = (uint16_t)((xb + 0x7FFFu + ((xb >> 16) & 1u)) >> 16);
}

// ---- device-side re-merge of the two pre-multiplied step matrices (the host packer's fp64 products, here as fp32 MFMA products of the bound
// parameters): prenet1 o fc_out -> w.pre1f, LSTM0 with attention_proj folded in -> w.lstm0f, both in the frag16 weight layout of the blob
struct MergeSeg { const float* src; int ld, col0, k_lo, k_hi; };
__global__ __launch_bounds__(256) void merge_pack_kernel(float* __restrict__ dst, int N, int K, MergeSeg s0, MergeSeg s1, MergeSeg s2, int perm_H) {
    const int64_t total = (int64_t)N * K;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int k = idx % K, np = idx / K;
        const int r = perm_H ? (np & 3) * perm_H + (np >> 2) : np;
        const MergeSeg* segs[3] = {&s0, &s1, &s2};
        float v = 0.f;
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const MergeSeg& g = *segs[q];
            if (g.src && k >= g.k_lo && k < g.k_hi) v = g.src[(int64_t)r * g.ld + g.col0 + (k - g.k_lo)];
        }
        dst[frag16_index(np, k, K)] = v;
    }
}
// out[np] = a[r] (+ b[r]) + sum_k W[r*ld + col0 + k] * x[k],  r = perm(np)
__global__ __launch_bounds__(256) void merge_bias_kernel(float* __restrict__ out, int N, const float* __restrict__ a, const float* __restrict__ b,
                                                         const float* __restrict__ W, int ld, int col0, int K, const float* __restrict__ x, int perm_H) {
    const int np = blockIdx.x * 256 + threadIdx.x;
    if (np >= N) return;
    const int r = perm_H ? (np & 3) * perm_H + (np >> 2) : np;
    double acc = (double)a[r] + (b ? (double)b[r] : 0.0);
    for (int k = 0; k < K; ++k) acc += (double)W[(int64_t)r * ld + col0 + k] * (double)x[k];
    out[np] = (float)acc;
}
static int remerge_step_weights(l2s_model* m, hipStream_t s) {
    const std::string D = "decoder.";
    auto P = [&](const char* k) { return m->canon(D + k); };
    const float *wp1 = P("prenet.0.linear_layer.weight"), *bp1 = P("prenet.0.linear_layer.bias"), *wfc = P("fc_out.linear_layer.weight"), *bfc = P("fc_out.linear_layer.bias");
    const float *wih = P("decoder_rnn.weight_ih_l0"), *whh = P("decoder_rnn.weight_hh_l0"), *bih = P("decoder_rnn.bias_ih_l0"), *bhh = P("decoder_rnn.bias_hh_l0");
    const float *wap = P("attention_proj.linear_layer.weight"), *bap = P("attention_proj.linear_layer.bias");
    if (!(wp1 && bp1 && wfc && bfc && wih && whh && bih && bhh && wap && bap) || !m->w.pre1f.W || !m->w.lstm0f.W) return 0;      // decoder not bound: stays invalid
    if (!m->merge_scratch) L2S_CHECK_HIP(hipMalloc(&m->merge_scratch, sizeof(float) * ((int64_t)2048 * 512 + (int64_t)256 * 512)));
    float* prod_ap = m->merge_scratch; float* prod_p1 = prod_ap + (int64_t)2048 * 512;
    // (W_ih[:,256:512] @ W_ap) (2048 x 512) and (W_p1 @ W_out) (256 x 512): C = A . B with B row-major is the input-gradient form of the backward GEMM
    if (launch_gemm_bwd(bwd_dx(wih + 256, 512, wap, prod_ap, 512, 1, 2048, 2048, 256, 512, 1, 0, false), s, "train_merge_step_weights")) return 1;
    if (launch_gemm_bwd(bwd_dx(wp1, NM, wfc, prod_p1, 512, 1, 256, 256, NM, 512, 1, 0, false), s, "train_merge_step_weights")) return 1;
    ProfScope ps("train_merge_step_weights", s);
    const MergeSeg none{nullptr, 0, 0, 0, 0};
    hipLaunchKernelGGL(merge_pack_kernel, dim3(4096), dim3(256), 0, s, const_cast<float*>(m->w.lstm0f.W), 2048, 1536, MergeSeg{wih, 512, 0, 0, 512},
                       MergeSeg{prod_ap, 512, 0, 512, 1024}, MergeSeg{whh, 512, 0, 1024, 1536}, 512);
    hipLaunchKernelGGL(merge_pack_kernel, dim3(512), dim3(256), 0, s, const_cast<float*>(m->w.pre1f.W), 256, 512, MergeSeg{prod_p1, 512, 0, 0, 512}, none, none, 0);
    hipLaunchKernelGGL(merge_bias_kernel, dim3(8), dim3(256), 0, s, const_cast<float*>(m->w.lstm0f.bias), 2048, bih, bhh, wih, 512, 256, 256, bap, 512);
    hipLaunchKernelGGL(merge_bias_kernel, dim3(1), dim3(256), 0, s, const_cast<float*>(m->w.pre1f.bias), 256, bp1, (const float*)nullptr, wp1, NM, 0, NM, bfc, 0);
    L2S_CHECK_HIP(hipGetLastError());
    m->folded_valid = true;
    return 0;
}

static int refresh_weights(l2s_model* m, hipStream_t s) {
    L2S_REQUIRE(m->finalized && m->r_key && m->r_idx, "no refresh map: set option refresh_map=1 before l2s_model_finalize");
    const size_t nk = m->r_keys.size(), nb = m->r_bn.size(), ns = m->r_sum.size();
    const size_t off_bn = align_up((int64_t)(nk * sizeof(float*)), 64), off_sum = off_bn + align_up((int64_t)(nb * sizeof(RBn)), 64);
    const size_t total = off_sum + ns * sizeof(RSum) + 64;
    m->r_tables_host.assign(total, 0);
    const float** ptrs = reinterpret_cast<const float**>(m->r_tables_host.data());
    for (size_t i = 0; i < nk; ++i) ptrs[i] = m->canon(m->r_keys[i]);
    RBn* bn = reinterpret_cast<RBn*>(m->r_tables_host.data() + off_bn);
    size_t nb_live = 0;
    int maxc = 1;
    for (const auto& r : m->r_bn) {
        RBn d{m->canon(r.p + ".weight"), m->canon(r.p + ".bias"), m->canon(r.p + ".running_mean"), m->canon(r.p + ".running_var"),
              r.bias_key.empty() ? nullptr : m->canon(r.bias_key), m->blob + r.so, m->blob + r.ho, r.c};
        if (!d.g && !d.b && !d.mu && !d.var) continue;                   // a module that is not bound at all (e.g. frozen) keeps its packed values
        L2S_REQUIRE(d.g && d.b && d.mu && d.var && (r.bias_key.empty() || d.bias), "refresh: BatchNorm tensors of a layer are only partly bound");
        bn[nb_live++] = d; maxc = std::max(maxc, r.c);
    }
    RSum* sm = reinterpret_cast<RSum*>(m->r_tables_host.data() + off_sum);
    size_t ns_live = 0;
    int maxn = 1;
    for (const auto& r : m->r_sum) {
        RSum d{m->canon(r.a), m->canon(r.b), m->blob + r.dst, r.n, r.perm_H};
        if (!d.a && !d.b) continue;
        L2S_REQUIRE(d.a && d.b, "refresh: bias pair only partly bound");
        sm[ns_live++] = d; maxn = std::max(maxn, r.n);
    }
    if ((int64_t)total > m->r_tables_bytes) {
        if (m->r_tables) (void)hipFree(m->r_tables);
        L2S_CHECK_HIP(hipMalloc(&m->r_tables, total));
        m->r_tables_bytes = (int64_t)total;
        m->r_tables_uploaded.clear();
    }
    // the tables only change when tensors are (re)bound: upload them then, not on every optimizer step (the upload comes from a pageable vector,
    // so it ends in a stream synchronise - once per step that drained the pipeline between steps)
    if (m->r_tables_host != m->r_tables_uploaded) {
        L2S_CHECK_HIP(hipMemcpyAsync(m->r_tables, m->r_tables_host.data(), total, hipMemcpyHostToDevice, s));
        L2S_CHECK_HIP(hipStreamSynchronize(s));                         // pageable staging buffer: the copy must have left the host vector
        m->r_tables_uploaded = m->r_tables_host;
    }
    char* T = (char*)m->r_tables;
    {
        ProfScope ps("train_refresh_gather", s);
        hipLaunchKernelGGL(refresh_gather_kernel, dim3(4096), dim3(256), 0, s, m->blob, m->r_key, m->r_idx, reinterpret_cast<const float* const*>(T), m->blob_floats);
    }
    if (nb_live) hipLaunchKernelGGL(refresh_bn_kernel, dim3((maxc + 255) / 256, (unsigned)nb_live), dim3(256), 0, s, reinterpret_cast<const RBn*>(T + off_bn));
    if (ns_live) hipLaunchKernelGGL(refresh_sum_kernel, dim3((maxn + 255) / 256, (unsigned)ns_live), dim3(256), 0, s, reinterpret_cast<const RSum*>(T + off_sum));
    L2S_CHECK_HIP(hipGetLastError());
    // the front-end's bf16 operand planes are splits of the old weights: re-split them from the bound Conv3d weight (an encoder that is not
    // bound keeps its packed planes, like every other unbound module)
    if (const float* w3d = m->canon("encoder.frontend3D.0.weight"); w3d && m->w.fe.w3 && m->w.fe.w1) {
        ProfScope ps("train_refresh_frontend_planes", s);
        hipLaunchKernelGGL(refresh_frontend_planes_kernel, dim3(15 * 4 * 32 * 16 / 256), dim3(256), 0, s, w3d,
                           reinterpret_cast<uint16_t*>(const_cast<float*>(m->w.fe.w3)), reinterpret_cast<uint16_t*>(const_cast<float*>(m->w.fe.w1)));
        L2S_CHECK_HIP(hipGetLastError());
    }
    if (m->lstm_planes && derive_lstm_planes(m, s)) return 1;      // the LSTM weights' bf16 planes are splits of the old weights too
    if (m->gemm_planes && derive_gemm_planes(m, s)) return 1;
    if (m->unit_planes && derive_unit_planes(m, s)) return 1;
    m->folded_valid = false;        // W_p1 W_out and W_ih W_ap are products of the old parameters ...
    if (remerge_step_weights(m, s)) return 1;      // ... rebuilt here when the decoder's tensors are bound (then the 4-launch step stays valid)
    for (auto& g : m->graphs) { (void)hipGraphExecDestroy(g.exec); (void)hipGraphDestroy(g.graph); }
    m->graphs.clear();
    return 0;
}

// ------------------------------------------------------------------------------------------------ workspace
struct EncPlan {
    int NF, Hp;
    int64_t act_a, act_b, t1, t2, last;
};
static EncPlan enc_plan(int B, int T, int H) {
    EncPlan p{};
    p.NF = B * T;
    p.Hp = H / 4;
    int64_t hw = (int64_t)p.Hp * p.Hp;
    int64_t amax = (int64_t)p.NF * hw * STAGE_CH[0], t1 = 0, t2 = 0;
    int cin = STAGE_CH[0];
    int h = p.Hp;
    for (int st = 0; st < 3; ++st) {
        int cout = STAGE_CH[st + 1], half = cout / 2;
        int ho = (h + 1) / 2;
        int64_t in_px = (int64_t)p.NF * h * h, out_px = (int64_t)p.NF * ho * ho;
        t1 = std::max(t1, std::max(in_px * half, out_px * half));     // pw1 output (stride-2: input resolution)
        t2 = std::max(t2, std::max(out_px * cin, out_px * half));     // dw outputs
        amax = std::max(amax, out_px * cout);
        cin = cout;
        h = ho;
    }
    p.act_a = amax; p.act_b = amax; p.t1 = t1; p.t2 = t2;
    p.last = (int64_t)p.NF * h * h * LAST_CH;
    return p;
}
static int64_t enc_ws_floats(int B, int T, int H) {
    EncPlan p = enc_plan(B, T, H);
    return p.act_a + p.act_b + p.t1 + p.t2 + p.last + 64 * 8;
}

static int64_t prologue_ws_floats(int B, int T) {
    int L[4];
    int m = content_lens(T, L);
    int64_t BT = (int64_t)B * T, n = 0;
    n += BT * 4096;            // BiLSTM input gates
    n += BT * 1024;            // rnn_out
    n += BT * 512;             // residual
    n += (int64_t)B * 512 * 2; // s_e, s_a
    n += (int64_t)pad16(B) * 512 * 6;   // h,c frags for both directions (ping-pong h)
    n += (int64_t)B * 1024;    // cell cat
    n += BT * 4608;            // cat buffer [x | K branches | V branches]
    for (int j = 0; j < 4; ++j) n += (int64_t)B * L[j] * 512;
    n += (int64_t)B * m * 2560;
    n += (int64_t)B * m * 256 * 4;
    n += (int64_t)B * m * (VOC + VOCP);
    n += 8 * std::max((int64_t)B * m * 256, (int64_t)B * 512);   // split-K partial products
    for (int j = 1; j < 4; ++j) n += (int64_t)CT_KS[j] * B * L[j] * 512;   // per-tap partial products of Content.agg
    n += 8 * BT * 512;                                                      // split-K partial products of the two MultiHop bottlenecks
    if (B <= 2) n += pbilstm_ws_bytes() / 4 + 64;                           // exchange granules of the persistent BiLSTM
    return n + 64 * 44;
}
static int64_t decode_ws_floats(int B) {
    int64_t Bp = pad16(B);
    return Bp * (512 * 4 + 512 * 2 + 512 + 256 * 4 + 96) + (int64_t)B * (512 + 256 + 256) + 64 * 24 + (B <= 8 ? pdecode_ws_bytes(B) / 4 + 64 : 0);
}
constexpr int POST_TAPSPLIT_ROWS = 640;      // a batch with at most this many post-net rows (one or two clips of 300 frames) runs its Conv1d layers one K slice per tap
static int64_t postnet_ws_floats(int B, int S) { return (int64_t)B * S * 512 * ((int64_t)B * S <= POST_TAPSPLIT_ROWS * MAX_GROUP ? 9 : 4) + 64 * 7; }

// ------------------------------------------------------------------------------------------------ encoder

static GemmP pw_gemm(const float* A, int lda, int a_off, const ConvW& c, float* C, int ldc, int c_off, int cstride,
                     int64_t M, int N, int K, int act) {
    GemmP p = gemm_plain(A + a_off, lda, c.W, C + c_off, ldc, (int)M, N, K);
    p.scale = c.scale; p.shift = c.shift; p.actw = c.actw; p.act = act; p.c_cstride = cstride;
    p.W3 = c.W3;
    return p;
}

static FrameSrc frame_src(const float* video, int B) { FrameSrc f{}; f.p[0] = video; f.per = B; return f; }

static int encoder_run(l2s_model* m, const FrameSrc& video, int B, int T, int H, int W, const float* emb, float* vis,
                       float* feat, void* ws, int64_t ws_bytes, hipStream_t s) {
    X3Scope x3scope(m->opt.infer_bf16 ? 0 : m->opt.gemm_x3);
    Bf16Scope bf16scope(m->opt.infer_bf16);      // the bf16 leg: bf16-operand GEMM / Conv1d kernels instead of the f32 / split-bf16 ones
    const Weights& w = m->w;
    EncPlan pl = enc_plan(B, T, H);
    Bump bp(ws, ws_bytes);
    float* a = bp.f(pl.act_a); float* b = bp.f(pl.act_b); float* t1 = bp.f(pl.t1); float* t2 = bp.f(pl.t2); float* last = bp.f(pl.last);
    L2S_REQUIRE(!bp.overflow, "encoder workspace too small");
    FrontendW fe = w.fe;
    if (!m->opt.frontend_x3 || !m->planes_valid) fe.w3 = nullptr;        // after a device-side refresh the split planes are stale
    fe.pair = m->opt.frontend_x3 >= 2; fe.pipe = m->opt.frontend_x3 == 3;
    fe.solo = m->opt.frontend_solo && (m->opt.frontend_solo >= 2 || chains_hint() >= 2);
    if (!m->opt.infer_bf16 || !m->planes_valid) fe.w1 = nullptr;
    if (launch_frontend(fe, video, B, T, H, W, a, s)) return 1;
    float* x = a; float* y = b;
    int h = pl.Hp;
    const int NF = pl.NF;
    for (int u = 0; u < N_UNITS; ++u) {
        const UnitW& U = w.unit[u];
        const int half = U.half, cout = 2 * half;
        if (U.stride2 && m->opt.fuse_trunk && m->opt.fuse_s2) {
            ShuffleS2P sp{};
            sp.x = x; sp.out = y;
            sp.wd1 = U.b1_dw.w9; sp.sd1 = U.b1_dw.scale; sp.bd1 = U.b1_dw.shift;
            sp.wb1f = U.b1_frag; sp.sb1 = U.b1_pw.scale; sp.bb1 = U.b1_pw.shift;
            sp.w1f = U.pw1_frag; sp.s1 = U.pw1.scale; sp.b1 = U.pw1.shift;
            sp.wd = U.dw.w9; sp.sd = U.dw.scale; sp.bd = U.dw.shift;
            sp.w2f = U.pw2_frag; sp.s2 = U.pw2.scale; sp.b2 = U.pw2.shift;
            sp.NF = NF; sp.h = h; sp.ho = (h + 1) / 2; sp.cin = U.cin; sp.half = half; sp.Kin = U.kin; sp.Kh = U.kpad;
            sp.Ro = U.cin == 232 ? 3 : 2;                      // informational: fixed by the kernel instance
            if (m->opt.trunk_x3) { sp.wb1p = U.b1_p3; sp.w1p = U.pw1_p3; sp.w2p = U.pw2_p3; }
            if (launch_shuffle_s2(sp, s)) return 1;
            h = sp.ho;
        } else if (U.stride2) {
            const int cin = U.cin, ho = (h + 1) / 2;
            const int64_t in_px = (int64_t)NF * h * h, out_px = (int64_t)NF * ho * ho;
            // banch1: dw s2 (+BN) -> pw (+BN+ReLU) -> even output channels
            if (launch_dwconv(x, NF, h, h, cin, 0, cin, 2, U.b1_dw.w9, U.b1_dw.scale, U.b1_dw.shift, t2, cin, 0, s)) return 1;
            if (launch_gemm1(pw_gemm(t2, cin, 0, U.b1_pw, y, cout, 0, 2, out_px, half, cin, ACT_RELU), s, "shuffle_pw_gemm")) return 1;
            // banch2: pw -> dw s2 -> pw -> odd output channels
            if (launch_gemm1(pw_gemm(x, cin, 0, U.pw1, t1, half, 0, 1, in_px, half, cin, ACT_RELU), s, "shuffle_pw_gemm")) return 1;
            if (launch_dwconv(t1, NF, h, h, half, 0, half, 2, U.dw.w9, U.dw.scale, U.dw.shift, t2, half, 0, s)) return 1;
            if (launch_gemm1(pw_gemm(t2, half, 0, U.pw2, y, cout, 1, 2, out_px, half, half, ACT_RELU), s, "shuffle_pw_gemm")) return 1;
            h = ho;
        } else if (m->opt.fuse_trunk) {
            auto s1_params = [&](const UnitW& V, const float* in, float* outp) {
                ShuffleS1P sp{};
                sp.x = in; sp.out = outp;
                sp.w1f = V.pw1_frag; sp.s1 = V.pw1.scale; sp.b1 = V.pw1.shift;
                sp.wd = V.dw.w9; sp.sd = V.dw.scale; sp.bd = V.dw.shift;
                sp.w2f = V.pw2_frag; sp.s2 = V.pw2.scale; sp.b2 = V.pw2.shift;
                sp.NF = NF; sp.h = h; sp.half = V.half; sp.Kpad = V.kpad;
                sp.F = h >= 11 ? 1 : 2;                        // informational: fixed by the kernel instance
                if (m->opt.trunk_x3) { sp.w1p = V.pw1_p3; sp.w2p = V.pw2_p3; }      // pointwise convs on the bf16 matrix cores (exact split)
                return sp;
            };
            // the run of stride-1 units that starts here (the rest of the stage) as ONE launch: the map stays on chip between the units
            int run = 1;
            while (u + run < N_UNITS && !w.unit[u + run].stride2 && w.unit[u + run].half == half) ++run;
            if (m->opt.trunk_chain && (h >= 6 || m->opt.trunk_chain >= 2) && m->opt.trunk_x3 && U.pw1_p3 && U.pw2_p3 && run >= 2 && run <= S1_CHAIN_MAX) {
                ShuffleS1P chain[S1_CHAIN_MAX];
                for (int i = 0; i < run; ++i) chain[i] = s1_params(w.unit[u + i], x, y);
                if (launch_shuffle_s1_chain(chain, run, s)) return 1;
                u += run - 1;
            } else {
                const ShuffleS1P sp = s1_params(U, x, y);
                if (launch_shuffle_s1(sp, s)) return 1;
            }
        } else {
            const int64_t px = (int64_t)NF * h * h;
            if (launch_copy_cols(x, cout, 0, y, cout, 0, 2, px, half, s)) return 1;
            if (launch_gemm1(pw_gemm(x, cout, half, U.pw1, t1, half, 0, 1, px, half, half, ACT_RELU), s, "shuffle_pw_gemm")) return 1;
            if (launch_dwconv(t1, NF, h, h, half, 0, half, 1, U.dw.w9, U.dw.scale, U.dw.shift, t2, half, 0, s)) return 1;
            if (launch_gemm1(pw_gemm(t2, half, 0, U.pw2, y, cout, 1, 2, px, half, half, ACT_RELU), s, "shuffle_pw_gemm")) return 1;
        }
        std::swap(x, y);
    }
    const int64_t px = (int64_t)NF * h * h;
    {
        GemmP pc = pw_gemm(x, STAGE_CH[3], 0, w.conv_last, last, LAST_CH, 0, 1, px, LAST_CH, STAGE_CH[3], ACT_RELU);
        if (!m->opt.gemm_x3_dma) pc.W3 = nullptr;
        if (launch_gemm1(pc, s, "conv_last_gemm")) return 1;
    }
    if (launch_pool_norm_cat(last, NF, h * h, LAST_CH, emb, L2S_D_EMB, T, vis, L2S_D_VIS, feat, s)) return 1;
    return 0;
}

// ------------------------------------------------------------------------------------------------ decoder prologue
static GemmP conv_gemm(const float* X, int lda, int B, int Tin, int Cin, const ConvW& c, int Cout, int taps, int stride, int pad,
                       float* out, int ldc, int act) {
    const int Tout = (Tin + 2 * pad - taps) / stride + 1;
    GemmP p = gemm_plain(X, lda, c.W, out, ldc, B * Tout, Cout, taps * Cin);
    p.Tout = Tout; p.Tin = Tin; p.taps = taps; p.stride = stride; p.pad = pad; p.Cin = Cin;
    p.scale = c.scale; p.shift = c.shift; p.actw = c.actw; p.act = act;
    p.W3 = c.W3;
    return p;
}

static SkinnyP sk_base(const SkW& sw, int B) {
    SkinnyP p{};
    p.W = sw.W; p.W3 = sw.W3; p.bias = sw.bias; p.actw = sw.actw;
    p.B = B; p.N = sw.N; p.K = sw.K;
    p.act = ACT_NONE; p.epi = SK_PLAIN;
    return p;
}

static int prologue_run(l2s_model* m, const float* vis, const float* emb, const float* gumbel, int B, int T,
                        float* state, float* content_dis, void* ws, int64_t ws_bytes, hipStream_t s) {
    X3Scope x3scope(m->opt.infer_bf16 ? 0 : m->opt.gemm_x3);
    Bf16Scope bf16scope(m->opt.infer_bf16);      // the bf16 leg: bf16-operand GEMM / Conv1d kernels instead of the f32 / split-bf16 ones
    const Weights& w = m->w;
    StateLayout sl = state_layout(B, T);
    int L[4];
    const int mT = content_lens(T, L);
    L2S_REQUIRE(T >= 7 && T <= L2S_MAX_STEPS, "T must be in [7, 300] (Content.agg stride-7 branch; positional table)");
    const int BT = B * T, Bp = pad16(B);
    Bump bp(ws, ws_bytes);
    float* gin = bp.f((int64_t)BT * 4096);
    float* rnn = bp.f((int64_t)BT * 1024);
    float* resid = bp.f((int64_t)BT * 512);
    float* s_e = bp.f((int64_t)B * 512);
    float* s_a = bp.f((int64_t)B * 512);
    float* hf[2][2]; float* cf[2];
    for (int d = 0; d < 2; ++d) { hf[d][0] = bp.f((int64_t)Bp * 512); hf[d][1] = bp.f((int64_t)Bp * 512); cf[d] = bp.f((int64_t)Bp * 512); }
    float* cellcat = bp.f((int64_t)B * 1024);
    float* cat = bp.f((int64_t)BT * 4608);
    float* cmap[4];
    for (int j = 0; j < 4; ++j) cmap[j] = bp.f((int64_t)B * L[j] * 512);
    float* pooled = bp.f((int64_t)B * mT * 2560);
    float* wv = bp.f((int64_t)B * mT * 256);
    float* tA = bp.f((int64_t)B * mT * 256);
    float* tB = bp.f((int64_t)B * mT * 256);
    float* tC = bp.f((int64_t)B * mT * 256);
    float* logits = bp.f((int64_t)B * mT * VOC);
    float* z = bp.f((int64_t)B * mT * VOCP);
    float* part = bp.f(8 * std::max((int64_t)B * mT * 256, (int64_t)B * 512));
    int64_t tap_floats = 0;
    for (int j = 1; j < 4; ++j) tap_floats += (int64_t)CT_KS[j] * B * L[j] * 512;
    float* tap_part = bp.f(tap_floats);
    float* bott_part = bp.f((int64_t)8 * BT * 512);
    // one or two clips of a single-batch call: the BiLSTM recurrence as ONE persistent launch (pdecode.hip pbilstm_kernel; option "persist_decode")
    // (the envelope of the latency path; a persistent launch that timed out since the last call fails THIS call once: pdecode_gate)
    const int pgate = (m->opt.persist > 0 && B <= m->opt.persist && !grouped_entry() && pbilstm_supported(B, T) && pdecode_supported(B, T, mT)) ? pdecode_gate() : 0;
    if (pgate < 0) return 1;
    const bool pbi = pgate > 0;
    float* pbx = pbi ? bp.f(pbilstm_ws_bytes() / 4 + 64) : nullptr;
    L2S_REQUIRE(!bp.overflow, "prologue workspace too small");

    // residual_bottleneck, site embeddings
    {
        GemmP p = gemm_plain(vis, 1024, w.resid.W, resid, 512, BT, 512, 1024);
        p.shift = w.resid.shift;
        if (launch_gemm_splitk(p, 4, bott_part, s, "prologue_gemm")) return 1;          // 120 tiles: four K slices
        GemmBatch gb{};
        gb.p[0] = gemm_plain(emb, 256, w.enc_site.W, s_e, 512, B, 512, 256);
        gb.p[0].shift = w.enc_site.shift; gb.p[0].act = ACT_PSINE; gb.p[0].actw = w.enc_site.actw;
        gb.p[1] = gemm_plain(emb, 256, w.attn_site.W, s_a, 512, B, 512, 256);
        gb.p[1].shift = w.attn_site.shift; gb.p[1].act = ACT_PSINE; gb.p[1].actw = w.attn_site.actw;
        gb.count = 2;
        if (launch_gemm(gb, s, "prologue_gemm")) return 1;
    }
    // BiLSTM input gates for both directions: (B*T,1024) x (1024,4096)
    {
        GemmP p = gemm_plain(vis, 1024, w.wih_cat, gin, 4096, BT, 4096, 1024);
        p.shift = w.bih_cat;
        if (m->opt.gemm_x3_dma) p.W3 = w.wih_cat3;
        if (launch_gemm1(p, s, "bilstm_input_gemm")) return 1;
    }
    if (pbi) {
        PBiP q{};
        q.Whh0 = w.whh[0].W; q.Whh1 = w.whh[1].W; q.gin = gin; q.s_e = s_e; q.rnn = rnn; q.h_state = state + sl.h; q.cellcat = cellcat; q.B = B; q.T = T;
        if (launch_pbilstm(q, pbx, pbilstm_ws_bytes(), s)) return 1;
    } else {
    // recurrence: h0 = c0 = s_e for both directions (decoder.py:386-389)
    for (int d = 0; d < 2; ++d) {
        if (launch_to_frag(s_e, 512, B, 512, hf[d][0], 512, 0, 0, s)) return 1;
        if (launch_to_frag(s_e, 512, B, 512, cf[d], 512, 0, 0, s)) return 1;
        if (launch_fill(hf[d][1], (int64_t)Bp * 512, 0.f, s)) return 1;
    }
    for (int step = 0; step < T; ++step) {
        SkinnyBatch sb{};
        const int cur = step & 1, nxt = cur ^ 1;
        for (int d = 0; d < 2; ++d) {
            const int t = d == 0 ? step : T - 1 - step;
            SkinnyP p = sk_base(w.whh[d], B);
            p.seg[0] = {hf[d][cur], 32}; p.nseg = 1;
            p.epi = SK_LSTM; p.H = 512;
            p.pre = gin + (int64_t)t * 4096 + d * 2048; p.ld_pre = (int64_t)T * 4096;
            p.c_in = cf[d]; p.c_out = cf[d];
            p.h_out = hf[d][nxt]; p.h_out_K = 512; p.h_out_off = 0;
            p.h_seq = rnn + (int64_t)t * 1024 + d * 512; p.ld_hseq = (int64_t)T * 1024;
            sb.p[d] = p; sb.ntiles[d] = w.whh[d].tiles;
        }
        sb.count = 2;
        if (launch_skinny(sb, s, "bilstm_step", m->opt)) return 1;
    }
    const int fin = T & 1;     // buffer holding the final hidden states
    // decoder initial hidden = BiLSTM finals (fwd -> layer 0, bwd -> layer 1); kept in the state buffer as frag16
    L2S_CHECK_HIP(hipMemcpyAsync(state + sl.h, hf[0][fin], sizeof(float) * Bp * 512, hipMemcpyDeviceToDevice, s));
    L2S_CHECK_HIP(hipMemcpyAsync(state + sl.h + (int64_t)Bp * 512, hf[1][fin], sizeof(float) * Bp * 512, hipMemcpyDeviceToDevice, s));
    // encoder_cell = E_C(cat(c_fwd, c_bwd))
    if (launch_from_frag(cf[0], 512, B, 512, cellcat, 1024, 0, s)) return 1;
    if (launch_from_frag(cf[1], 512, B, 512, cellcat, 1024, 512, s)) return 1;
    }
    {
        GemmP p = gemm_plain(cellcat, 1024, w.e_c.W, state + sl.ecell, 512, B, 512, 1024);
        p.shift = w.e_c.shift;
        if (launch_gemm_splitk(p, 8, part, s, "prologue_gemm")) return 1;      // B rows x 512 columns = 8 tiles: split K = 1024 eight ways
        if (launch_stop_const(state + sl.ecell, w.stop_tail, w.stop_bias, B, state + sl.stopc, s)) return 1;
    }
    // enc = encoder_proj(rnn_out) + s_a (broadcast over T) + residual  -> cat[:, 0:512] and the state
    {
        GemmP p = gemm_plain(rnn, 1024, w.enc_proj.W, cat, 4608, BT, 512, 1024);
        p.shift = w.enc_proj.shift;
        p.R1 = resid; p.ldr1 = 512; p.r1_mod = 0;
        p.R2 = s_a; p.ldr2 = 512; p.r2_div = T;     // attention_site embedding, broadcast over the T frames of a clip
        if (launch_gemm_splitk(p, 4, bott_part, s, "prologue_gemm")) return 1;
        if (launch_copy_cols(cat, 4608, 0, state + sl.enc, 512, 0, 1, BT, 512, s)) return 1;
    }
    // MultiHopConv branches of K and V (8 convs, one grouped launch), then the two bottlenecks (+PSine +pos)
    {
        // longest K first: the groups are dispatched in order, and the 11-tap convs (K = 5632) would otherwise start last and run alone
        GemmBatch gb{};
        int g = 0;
        for (int j = 3; j >= 0; --j)
            for (int kv = 0; kv < 2; ++kv)
                gb.p[g++] = conv_gemm(cat, 4608, B, T, 512, w.mh_branch[kv][j], 512, MH_KS[j], 1, MH_KS[j] / 2,
                                      cat + 512 + (kv * 4 + j) * 512, 4608, ACT_SILU);
        gb.count = 8;
        if (!m->opt.gemm_x3_dma) for (int i = 0; i < 8; ++i) gb.p[i].W3 = nullptr;
        if (launch_gemm(gb, s, "multihop_conv_gemm")) return 1;
        GemmBatch bb{};
        for (int kv = 0; kv < 2; ++kv) {
            GemmP p = gemm_plain(cat, 4608, w.mh_bott[kv].W, state + (kv == 0 ? sl.k : sl.v), 512, BT, 512, 2560);
            if (kv == 1) { p.a_split = 512; p.a_gap = 2048; }     // V reads [x | V branches]
            p.shift = w.mh_bott[kv].shift; p.act = ACT_PSINE; p.actw = w.mh_bott[kv].actw;
            p.R1 = w.pos; p.ldr1 = 512; p.r1_mod = T;             // + pos_table[t]
            bb.p[kv] = p;
        }
        bb.count = 2;
        // 2 x 120 tiles with K = 2560 is one block per CU and 80 dependent K iterations: four K slices each
        if (launch_gemm_splitk_group(bb, 4, bott_part, s, "multihop_bottleneck_gemm")) return 1;
        // V' = V W_ap^T + b_ap: attention_proj (decoder.py:420) applied to the values once per clip instead of to a @ v at every step
        if (w.vproj.W) {
            GemmP p = gemm_plain(state + sl.v, 512, w.vproj.W, state + sl.vp, 256, BT, 256, 512);
            p.shift = w.vproj.shift;
            if (launch_gemm1(p, s, "prologue_gemm")) return 1;
        }
    }
    // Content.encode (decoder.py:239-260)
    {
        GemmBatch gb{};
        for (int j = 0; j < 4; ++j)      // K = 512 * ks is the same for every branch here; rows shrink with ks - largest map first
            gb.p[j] = conv_gemm(cat, 4608, B, T, 512, w.ct_branch[j], 512, CT_KS[j], CT_KS[j], 0, cmap[j], 512, ACT_SILU);
        gb.count = 4;
        // (4..29) x B rows by 512 columns: 16-120 tiles per branch with K up to 3584 - one slice per tap instead (16 slices, 472 tiles of K = 512)
        if (launch_gemm_tapsplit(gb, tap_part, s, "content_agg_gemm")) return 1;
        PoolCatP pc{};
        pc.x[0] = cat; pc.L[0] = T; pc.ld[0] = 4608;
        for (int j = 0; j < 4; ++j) { pc.x[j + 1] = cmap[j]; pc.L[j + 1] = L[j]; pc.ld[j + 1] = 512; }
        pc.nmaps = 5; pc.B = B; pc.m = mT; pc.C = 512; pc.out = pooled;
        if (launch_pool_cat(pc, s)) return 1;
        const int R = B * mT;
        GemmP p = gemm_plain(pooled, 2560, w.ct_bott.W, wv, 256, R, 256, 2560);
        p.shift = w.ct_bott.shift;
        if (launch_gemm_splitk(p, 8, part, s, "content_gemm")) return 1;       // 4B rows x 256 columns = 8 tiles with K = 2560
        GemmBatch g1{};
        g1.p[0] = gemm_plain(wv, 256, w.ct_k0.W, tA, 256, R, 256, 256); g1.p[0].shift = w.ct_k0.shift; g1.p[0].act = ACT_SILU;
        g1.p[1] = gemm_plain(wv, 256, w.ct_fc0.W, tB, 256, R, 256, 256); g1.p[1].shift = w.ct_fc0.shift; g1.p[1].act = ACT_SILU;
        g1.count = 2;
        if (launch_gemm(g1, s, "content_gemm")) return 1;
        GemmBatch g2{};
        g2.p[0] = gemm_plain(tA, 256, w.ct_k2.W, state + sl.ckey, 256, R, 256, 256); g2.p[0].shift = w.ct_k2.shift; g2.p[0].act = ACT_SILU;
        g2.p[1] = gemm_plain(tB, 256, w.ct_fc2.W, tC, 256, R, 256, 256); g2.p[1].shift = w.ct_fc2.shift; g2.p[1].act = ACT_SILU;
        g2.count = 2;
        if (launch_gemm(g2, s, "content_gemm")) return 1;
        GemmP p3 = gemm_plain(tC, 256, w.ct_fc4.W, logits, VOC, R, VOC, 256);
        p3.shift = w.ct_fc4.shift; p3.act = ACT_SILU;
        if (launch_gemm1(p3, s, "content_gemm")) return 1;
        if (launch_gumbel_softmax(logits, gumbel, R, VOC, 0.1f, z, VOCP, content_dis, s)) return 1;
        GemmP p4 = gemm_plain(z, VOCP, w.ct_emb.W, state + sl.cval, 256, R, 256, VOCP);
        if (launch_gemm1(p4, s, "content_gemm")) return 1;
    }
    // decoder cell state starts at zero (decoder.py:406)
    if (launch_fill(state + sl.c, (int64_t)Bp * 512 * 2, 0.f, s)) return 1;
    return 0;
}

// ------------------------------------------------------------------------------------------------ decode loop
// options (l2s_common.h Options, per model): "fold_step_weights" - phase-merged step (4 launches) vs the literal 6-phase step; "overlap_postnet" -
// l2s_inference runs the post-net in time windows on a second stream under the decode loop (bit-identical; measured SLOWER on MI355X, 13.1 vs
// 12.0 ms: the GEMM blocks delay the latency-critical step launches); "use_graph" - replay the loop from a captured hipGraph (measured slower
// than stream launches on MI355X: 15.0 vs 13.7 ms)

struct DecodeBufs {
    float *h0[2], *h1[2], *c0, *c1, *av, *p1, *cc, *uu, *yf, *p2f, *q, *qc, *p2;
};

// on_frames(n): called (if set) right after the launch that completes mel frames [0, n) has been enqueued on `s`
static int decode_launches(l2s_model* m, float* state, int B, int T, int S, const float* teacher, const uint8_t* teacher_mask,
                           float* mel, float* stop, float* attn, int attn_logits, void* ws, int64_t ws_bytes, hipStream_t s, bool fold,
                           const std::function<int(int)>* on_frames = nullptr) {
    const Weights& w = m->w;
    StateLayout sl = state_layout(B, T);
    const int Bp = pad16(B);
    const bool vhoist = fold && m->opt.hoist_vproj && w.lstm0v.W && w.vproj.W;      // attention_proj applied to the values in the prologue (option "hoist_vproj")
    const bool vsum = vhoist && m->opt.hoist_vproj >= 2 && skinny_sum_supported(m->opt);     // ... and u = prenet + o formed by LSTM0's operand loader (K = 1024)
    Bump bp(ws, ws_bytes);
    DecodeBufs d;
    for (int i = 0; i < 2; ++i) d.h0[i] = bp.f((int64_t)Bp * 512);
    for (int i = 0; i < 2; ++i) d.h1[i] = bp.f((int64_t)Bp * 512);
    d.c0 = bp.f((int64_t)Bp * 512); d.c1 = bp.f((int64_t)Bp * 512); d.av = bp.f((int64_t)Bp * 512);
    d.p1 = bp.f((int64_t)Bp * 256); d.cc = bp.f((int64_t)Bp * 256); d.uu = bp.f((int64_t)Bp * 256); d.p2f = bp.f((int64_t)Bp * 256);
    d.yf = bp.f((int64_t)Bp * 96);
    d.q = bp.f((int64_t)B * 512); d.qc = bp.f((int64_t)B * 256); d.p2 = bp.f((int64_t)B * 256);
    L2S_REQUIRE(!bp.overflow, "decode workspace too small");

    // initial state: h from the prologue, c = 0, y = BOS; padded rows of every frag buffer zero
    L2S_CHECK_HIP(hipMemcpyAsync(d.h0[0], state + sl.h, sizeof(float) * Bp * 512, hipMemcpyDeviceToDevice, s));
    L2S_CHECK_HIP(hipMemcpyAsync(d.h1[0], state + sl.h + (int64_t)Bp * 512, sizeof(float) * Bp * 512, hipMemcpyDeviceToDevice, s));
    // one fill over the contiguous run h0[1] .. p2f is not possible (h0[0]/h1[0] sit in between); fill individually
    float* zero512[] = {d.h0[1], d.h1[1], d.c0, d.c1, d.av};
    for (float* z : zero512) if (launch_fill(z, (int64_t)Bp * 512, 0.f, s)) return 1;
    float* zero256[] = {d.p1, d.cc, d.uu, d.p2f};
    for (float* z : zero256) if (launch_fill(z, (int64_t)Bp * 256, 0.f, s)) return 1;
    if (launch_to_frag(w.bos, 0, B, 80, d.yf, 80, 0, 1, s)) return 1;

    auto fc_group = [&](int step, const float* h1buf, bool write_y) {
        SkinnyP a = sk_base(w.fc, B);
        a.seg[0] = {h1buf, 32}; a.nseg = 1; a.epi = SK_MEL;
        a.mel = mel + (int64_t)step * NM; a.ld_mel_b = (int64_t)S * NM;
        a.stop = stop + step; a.ld_stop_b = S; a.stop_const = state + sl.stopc; a.yfrag = write_y ? d.yf : nullptr;
        return a;
    };

    for (int i = 0; i < S; ++i) {
        const int cur = i & 1, nxt = cur ^ 1;
        const bool forced = teacher && teacher_mask && teacher_mask[i];
        if (forced)
            if (launch_to_frag(teacher + (int64_t)i * NM, S * NM, B, 80, d.yf, 80, 0, 0, s)) return 1;
        {   // phase A: prenet layer 1, Q (+PSine +pos[i]), content Q (+SiLU) [, mel frame + stop logit of step i-1]
            SkinnyBatch sb{};
            const bool from_frame = !fold || i == 0 || forced;      // prenet input is an explicit frame (BOS / teacher / unfolded y)
            SkinnyP a = sk_base(from_frame ? w.pre1 : w.pre1f, B);
            if (from_frame) a.seg[0] = {d.yf, 5}; else a.seg[0] = {d.h1[cur], 32};
            a.nseg = 1; a.act = ACT_PSINE; a.epi = SK_FRAG; a.out = d.p1; a.ldo = 256;
            SkinnyP b = sk_base(w.q, B);
            b.seg[0] = {d.h0[cur], 32}; b.seg[1] = {d.h1[cur], 32}; b.nseg = 2; b.act = ACT_PSINE; b.epi = SK_PLAIN; b.out = d.q; b.ldo = 512;
            b.addrow = w.pos + (int64_t)i * 512;
            SkinnyP c = sk_base(w.cq, B);
            c.seg[0] = {d.c0, 32}; c.seg[1] = {d.c1, 32}; c.nseg = 2; c.act = ACT_SILU; c.epi = SK_PLAIN; c.out = d.qc; c.ldo = 256;
            sb.p[0] = a; sb.ntiles[0] = 16;
            sb.p[1] = b; sb.ntiles[1] = w.q.tiles;
            sb.p[2] = c; sb.ntiles[2] = w.cq.tiles;
            sb.count = 3;
            if (fold && i > 0) { sb.p[3] = fc_group(i - 1, d.h1[cur], false); sb.ntiles[3] = w.fc.tiles; sb.count = 4; }
            if (launch_skinny(sb, s, fold ? "step_prenet1_q_cq_fc" : "step_prenet1_q_cq", m->opt)) return 1;
            if (fold && i > 0 && on_frames && (*on_frames)(i)) return 1;
        }
        {   // phase B: attention + content attention per batch row; prenet layer 2
            AttnP at{};
            at.q = d.q; at.ldq = 512; at.k = state + sl.k; at.v = state + sl.v; at.tau = w.tau; at.av_frag = d.av;
            if (fold && vhoist) at.vp = state + sl.vp;      // d.av then holds o = a @ V' (frag16, K = 256)
            at.attn_out = attn ? attn + (int64_t)i * T : nullptr; at.ld_attn_b = (int64_t)S * T; at.attn_logits = attn_logits;
            at.qc = d.qc; at.ldqc = 256; at.ckey = state + sl.ckey; at.cval = state + sl.cval; at.tau_c = w.tau_c; at.cc_frag = d.cc;
            at.B = B; at.T = T; at.m = sl.m;
            SkinnyP pr = sk_base(w.pre2, B);
            pr.seg[0] = {d.p1, 16}; pr.nseg = 1; pr.act = ACT_PSINE;
            if (fold) { pr.epi = SK_FRAG; pr.out = d.p2f; pr.ldo = 256; }
            else { pr.epi = SK_PLAIN; pr.out = d.p2; pr.ldo = 256; }
            if (launch_step_attn(at, pr, w.pre2.tiles, s, m->opt.attn_lds, m->opt.attn_skip0)) return 1;
        }
        if (!fold) {   // phase C: u = prenet + attention_proj(a @ v)
            SkinnyBatch sb{};
            SkinnyP a = sk_base(w.aproj, B);
            a.seg[0] = {d.av, 32}; a.nseg = 1; a.epi = SK_FRAG; a.out = d.uu; a.ldo = 256; a.add = d.p2; a.ld_add = 256;
            sb.p[0] = a; sb.ntiles[0] = w.aproj.tiles; sb.count = 1;
            if (launch_skinny(sb, s, "step_attention_proj", m->opt)) return 1;
        }
        {   // phase D: LSTM layer 0 on cat(content, u), h0  (folded: cat(content, prenet, a@v) against [W_ih | W_ih_u W_ap | W_hh])
            SkinnyBatch sb{};
            SkinnyP a = sk_base(fold ? (vsum ? w.lstm0 : vhoist ? w.lstm0v : w.lstm0f) : w.lstm0, B);
            if (vsum) { a.seg[0] = {d.cc, 16}; a.seg[1] = {d.p2f, 16}; a.a_sum = d.av; a.seg[2] = {d.h0[cur], 32}; a.nseg = 3; }      // u = prenet + o, summed by the loader
            else if (fold) { a.seg[0] = {d.cc, 16}; a.seg[1] = {d.p2f, 16}; a.seg[2] = {d.av, vhoist ? 16 : 32}; a.seg[3] = {d.h0[cur], 32}; a.nseg = 4; }
            else { a.seg[0] = {d.cc, 16}; a.seg[1] = {d.uu, 16}; a.seg[2] = {d.h0[cur], 32}; a.nseg = 3; }
            a.epi = SK_LSTM; a.H = 512; a.c_in = d.c0; a.c_out = d.c0; a.h_out = d.h0[nxt]; a.h_out_K = 512; a.h_out_off = 0;
            sb.p[0] = a; sb.ntiles[0] = 128; sb.count = 1;
            if (launch_skinny(sb, s, "step_lstm_cell", m->opt)) return 1;
        }
        {   // phase E: LSTM layer 1 on the new h0
            SkinnyBatch sb{};
            SkinnyP a = sk_base(w.lstm1, B);
            a.seg[0] = {d.h0[nxt], 32}; a.seg[1] = {d.h1[cur], 32}; a.nseg = 2;
            a.epi = SK_LSTM; a.H = 512; a.c_in = d.c1; a.c_out = d.c1; a.h_out = d.h1[nxt]; a.h_out_K = 512; a.h_out_off = 0;
            sb.p[0] = a; sb.ntiles[0] = w.lstm1.tiles; sb.count = 1;
            if (launch_skinny(sb, s, "step_lstm_cell", m->opt)) return 1;
        }
        if (!fold || i == S - 1) {   // phase F: mel frame + stop logit (folded mode: only the last step needs its own launch)
            SkinnyBatch sb{};
            sb.p[0] = fc_group(i, d.h1[nxt], !fold); sb.ntiles[0] = w.fc.tiles; sb.count = 1;
            if (launch_skinny(sb, s, "step_fc_out_stop", m->opt)) return 1;
            if (on_frames && (*on_frames)(i + 1)) return 1;
        }
    }
    return 0;
}

static int decode_run(l2s_model* m, float* state, int B, int T, int S, const float* teacher, const uint8_t* teacher_mask,
                      float* mel, float* stop, float* attn, int attn_logits, void* ws, int64_t ws_bytes, hipStream_t s) {
    L2S_REQUIRE(S >= 1 && S <= L2S_MAX_STEPS, "S must be in [1, 300] (positional table)");
    const bool fold = m->opt.fold != 0 && m->folded_valid;
    if (!teacher && fold && m->opt.persist > 0 && B <= m->opt.persist && !grouped_entry()) {      // the latency form: one launch for the whole loop
        const Weights& w = m->w;
        StateLayout sl = state_layout(B, T);
        const int pgate = (pdecode_supported(B, T, sl.m) && w.vproj.W && w.pre1f.W && w.lstm0.W && w.lstm1.W) ? pdecode_gate() : 0;
        if (pgate < 0) return 1;      // an earlier persistent launch on this device gave up (its outputs are NaN): reported here, once
        if (pgate > 0) {
            PDecP p{};
            p.Wq = w.q.W; p.bq = w.q.bias; p.aq = w.q.actw;
            p.Wcq = w.cq.W; p.bcq = w.cq.bias;
            p.Wp1f = w.pre1f.W; p.bp1f = w.pre1f.bias; p.ap1 = w.pre1f.actw;
            p.Wp1 = w.pre1.W; p.bp1 = w.pre1.bias;
            p.Wp2 = w.pre2.W; p.bp2 = w.pre2.bias; p.ap2 = w.pre2.actw;
            p.Wl0 = w.lstm0.W; p.bl0 = w.lstm0.bias; p.Wl1 = w.lstm1.W; p.bl1 = w.lstm1.bias;
            p.Wfc = w.fc.W; p.bfc = w.fc.bias;
            p.pos = w.pos; p.tau = w.tau; p.tau_c = w.tau_c; p.bos = w.bos;
            p.k = state + sl.k; p.vp = state + sl.vp; p.ckey = state + sl.ckey; p.cval = state + sl.cval;
            p.h_init = state + sl.h; p.stop_const = state + sl.stopc;
            p.mel = mel; p.stop = stop; p.attn = attn; p.attn_logits = attn_logits;
            p.B = B; p.T = T; p.m = sl.m; p.S = S;
            return launch_pdecode(p, ws, ws_bytes, s);
        }
    }
    const bool use_graph = m->opt.graph && !teacher && !g_prof_on;
    if (!use_graph) return decode_launches(m, state, B, T, S, teacher, teacher_mask, mel, stop, attn, attn_logits, ws, ws_bytes, s, fold);

    std::lock_guard<std::mutex> side_lock(m->side_mu);      // graph cache, side stream and events are per model; chains of other threads wait here
    if (!m->side) {
        L2S_CHECK_HIP(hipStreamCreateWithFlags(&m->side, hipStreamNonBlocking));
        L2S_CHECK_HIP(hipEventCreateWithFlags(&m->ev_in, hipEventDisableTiming));
        L2S_CHECK_HIP(hipEventCreateWithFlags(&m->ev_out, hipEventDisableTiming));
    }
    l2s_model::GraphEntry* hit = nullptr;
    for (auto& g : m->graphs)
        if (g.B == B && g.T == T && g.S == S && g.attn_logits == attn_logits && g.fold == (int)fold && g.state == state && g.mel == mel &&
            g.stop == stop && g.attn == attn && g.ws == ws) { hit = &g; break; }
    if (!hit) {
        hipGraph_t graph = nullptr;
        L2S_CHECK_HIP(hipStreamBeginCapture(m->side, hipStreamCaptureModeThreadLocal));
        int rc = decode_launches(m, state, B, T, S, nullptr, nullptr, mel, stop, attn, attn_logits, ws, ws_bytes, m->side, fold);
        hipError_t ce = hipStreamEndCapture(m->side, &graph);
        if (rc) { if (graph) (void)hipGraphDestroy(graph); return 1; }
        L2S_CHECK_HIP(ce);
        hipGraphExec_t exec = nullptr;
        L2S_CHECK_HIP(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
        if (m->graphs.size() >= 8) {           // small FIFO: shapes/buffers rarely change in a serving loop
            (void)hipGraphExecDestroy(m->graphs.front().exec);
            (void)hipGraphDestroy(m->graphs.front().graph);
            m->graphs.erase(m->graphs.begin());
        }
        m->graphs.push_back({B, T, S, attn_logits, (int)fold, state, mel, stop, attn, ws, graph, exec});
        hit = &m->graphs.back();
    }
    L2S_CHECK_HIP(hipEventRecord(m->ev_in, s));
    L2S_CHECK_HIP(hipStreamWaitEvent(m->side, m->ev_in, 0));
    L2S_CHECK_HIP(hipGraphLaunch(hit->exec, m->side));
    L2S_CHECK_HIP(hipEventRecord(m->ev_out, m->side));
    L2S_CHECK_HIP(hipStreamWaitEvent(s, m->ev_out, 0));
    return 0;
}

// ------------------------------------------------------------------------------------------------ postnet
struct PostBufs { float* x[4]; float* part = nullptr; };

static int postnet_alloc(Bump& bp, int B, int S, PostBufs& pb) {
    for (int i = 0; i < 4; ++i) pb.x[i] = bp.f((int64_t)B * S * 512);
    // few rows (one or two clips alone: 5-10 row tiles x 8 column tiles on 256 CUs, K = 2560 deep: 67 us per layer): one K slice per tap in one grouped
    // launch + a finish kernel that adds the taps in order (launch_gemm_tapsplit; 40 -> 200 tiles).  Decided on the rows of ONE batch, so that a batch
    // meets the same arithmetic alone and in a group.
    pb.part = ((int64_t)B * S / gemm_x3_group() <= POST_TAPSPLIT_ROWS) ? bp.f((int64_t)B * S * 512 * 5) : nullptr;
    return bp.overflow ? 1 : 0;
}

// One post-net layer (decoder.py:143-156) over the frames [t0, t1) of every sequence; layer 0..4.
// Layer i reads buffer i (mel for i = 0) and writes buffer i+1 (mel_post, channel-first, for i = 4).
static int postnet_layer(const Weights& w, int layer, const float* mel, const PostBufs& pb, float* mel_post, int B, int S, int t0, int t1, hipStream_t s, bool dma_weights) {
    if (t1 <= t0) return 0;
    float* const* bufs = pb.x;
    const float* in = layer == 0 ? mel : bufs[layer - 1];
    const int cin = layer == 0 ? NM : 512;
    GemmP p = conv_gemm(in, cin, B, S, cin, w.post[layer], layer == 4 ? NM : 512, 5, 1, 2, layer == 4 ? mel_post : bufs[layer], layer == 4 ? NM : 512,
                        layer == 4 ? ACT_NONE : ACT_PSINE);
    p.M = B * (t1 - t0); p.Tout = t1 - t0; p.win_T = S; p.win_off = t0;
    if (layer >= 1 && layer <= 3) { p.R1 = in; p.ldr1 = 512; p.r1_mod = 0; }
    if (layer == 4) { p.R1 = mel; p.ldr1 = NM; p.r1_mod = 0; p.c_tr_T = S; }
    if (!dma_weights) p.W3 = nullptr;
    if (pb.part && t0 == 0 && t1 == S) {
        p.win_T = 0; p.win_off = 0; p.W3 = nullptr;
        GemmBatch gb{};
        gb.p[0] = p; gb.count = 1;
        return launch_gemm_tapsplit(gb, pb.part, s, "postnet_conv_gemm");
    }
    return launch_gemm1(p, s, "postnet_conv_gemm");
}

static int postnet_run(l2s_model* m, const float* mel, int B, int S, float* mel_post, float* mel_cf, void* ws, int64_t ws_bytes, hipStream_t s) {
    X3Scope x3scope(m->opt.infer_bf16 ? 0 : m->opt.gemm_x3);
    Bf16Scope bf16scope(m->opt.infer_bf16);      // the bf16 leg: bf16-operand GEMM / Conv1d kernels instead of the f32 / split-bf16 ones
    const Weights& w = m->w;
    Bump bp(ws, ws_bytes);
    PostBufs pb;
    L2S_REQUIRE(postnet_alloc(bp, B, S, pb) == 0, "postnet workspace too small");
    for (int layer = 0; layer < 5; ++layer)
        if (postnet_layer(w, layer, mel, pb, mel_post, B, S, 0, S, s, m->opt.gemm_x3_dma != 0)) return 1;
    if (mel_cf && launch_transpose_bsc(mel, B, S, NM, mel_cf, s)) return 1;
    return 0;
}

// ------------------------------------------------------------------------------------------------ speaker encoder
static int64_t spk_ws_floats(int B, int N) {
    const int64_t L = N / 160 + 1, R = (int64_t)B * L;
    return R * (400 + 402 + 204 + 40 + 1024 + 256 * 2) + (int64_t)pad16(B) * 256 * 3 + (int64_t)B * 256 + 64 * 16;
}

// SpeakerEncoder.inference (audio.py:131-150): mel40 -> 3 x LSTM(256), zero initial state -> Linear(h_last) -> ReLU -> L2 norm
static int speaker_run(l2s_model* m, const float* audio, int B, int N, float* emb, void* ws, int64_t ws_bytes, hipStream_t s) {
    X3Scope x3scope(m->opt.infer_bf16 ? 0 : m->opt.gemm_x3);
    Bf16Scope bf16scope(m->opt.infer_bf16);      // the bf16 leg: bf16-operand GEMM / Conv1d kernels instead of the f32 / split-bf16 ones
    const Weights& w = m->w;
    L2S_REQUIRE(N > 200, "audio shorter than the reflect padding (200 samples)");
    const int L = N / 160 + 1, Bp = pad16(B);
    const int64_t R = (int64_t)B * L;
    Bump bp(ws, ws_bytes);
    float* frames = bp.f(R * 400); float* spec = bp.f(R * 402); float* power = bp.f(R * 204); float* mel = bp.f(R * 40);
    float* pre = bp.f(R * 1024); float* hseq[2] = {bp.f(R * 256), bp.f(R * 256)};
    float* hf[2] = {bp.f((int64_t)Bp * 256), bp.f((int64_t)Bp * 256)}; float* cf = bp.f((int64_t)Bp * 256);
    float* lin = bp.f((int64_t)B * 256);
    L2S_REQUIRE(!bp.overflow, "speaker-encoder workspace too small");
    if (launch_frame_window(audio, B, N, L, 400, 160, w.spk_window, frames, s)) return 1;
    if (launch_gemm1(gemm_plain(frames, 400, w.spk_dft, spec, 402, (int)R, 402, 400), s, "spk_dft_gemm")) return 1;
    if (launch_power(spec, 402, R, 201, power, 204, s)) return 1;
    if (launch_gemm1(gemm_plain(power, 204, w.spk_fbT, mel, 40, (int)R, 40, 204), s, "spk_mel_gemm")) return 1;
    const float* x = mel;
    int xin = 40;
    for (int l = 0; l < 3; ++l) {
        GemmP g = gemm_plain(x, xin, w.spk_ih[l].W, pre, 1024, (int)R, 1024, xin);
        g.shift = w.spk_ih[l].shift;
        if (launch_gemm1(g, s, "spk_lstm_input_gemm")) return 1;
        if (launch_fill(hf[0], (int64_t)Bp * 256, 0.f, s)) return 1;
        if (launch_fill(hf[1], (int64_t)Bp * 256, 0.f, s)) return 1;
        if (launch_fill(cf, (int64_t)Bp * 256, 0.f, s)) return 1;
        float* out = hseq[l & 1];
        for (int t = 0; t < L; ++t) {
            SkinnyBatch sb{};
            SkinnyP p = sk_base(w.spk_hh[l], B);
            p.seg[0] = {hf[t & 1], 16}; p.nseg = 1;
            p.epi = SK_LSTM; p.H = 256;
            p.pre = pre + (int64_t)t * 1024; p.ld_pre = (int64_t)L * 1024;
            p.c_in = cf; p.c_out = cf;
            p.h_out = hf[(t & 1) ^ 1]; p.h_out_K = 256; p.h_out_off = 0;
            p.h_seq = out + (int64_t)t * 256; p.ld_hseq = (int64_t)L * 256;
            sb.p[0] = p; sb.ntiles[0] = 64; sb.count = 1;
            if (launch_skinny(sb, s, "spk_lstm_step", m->opt)) return 1;
        }
        x = out;
        xin = 256;
    }
    // embeds = normalize(relu(linear(h_last))), h_last = top layer's output at the last frame
    GemmP g = gemm_plain(x + (int64_t)(L - 1) * 256, L * 256, w.spk_linear.W, lin, 256, B, 256, 256);
    g.shift = w.spk_linear.shift; g.act = ACT_RELU;
    if (launch_gemm1(g, s, "spk_linear_gemm")) return 1;
    return launch_pool_norm_cat(lin, B, 1, 256, nullptr, 0, 1, nullptr, 0, emb, s);
}

}  // namespace l2s

// ================================================================================================ C ABI
using namespace l2s;

extern "C" {

int l2s_abi_version(void) { return 2; }
const char* l2s_last_error(void) { return g_err.c_str(); }

int l2s_model_create(l2s_model** out) {
    L2S_REQUIRE(out != nullptr, "null out pointer");
    *out = new l2s_model();
    (*out)->opt = g_default_opt;
    return 0;
}
int l2s_model_set_tensor(l2s_model* m, const char* key, const float* host_data, int64_t numel) {
    L2S_REQUIRE(m && key && host_data && numel >= 0, "bad arguments");
    m->host[key].assign(host_data, host_data + numel);
    m->finalized = false;
    return 0;
}
int l2s_model_finalize(l2s_model* m, void* stream) {
    L2S_REQUIRE(m != nullptr, "null model");
    int rc = pack_model(m, (hipStream_t)stream);
    if (rc == 0) m->host.clear();
    return rc;
}
int l2s_model_destroy(l2s_model* m) {
    if (!m) return 0;
    for (auto& g : m->graphs) { (void)hipGraphExecDestroy(g.exec); (void)hipGraphDestroy(g.graph); }
    if (m->side) { (void)hipStreamDestroy(m->side); (void)hipEventDestroy(m->ev_in); (void)hipEventDestroy(m->ev_out); }
    for (auto e : m->ev_pool) (void)hipEventDestroy(e);
    if (m->blob) (void)hipFree(m->blob);
    if (m->r_key) (void)hipFree(m->r_key);
    if (m->r_idx) (void)hipFree(m->r_idx);
    if (m->r_tables) (void)hipFree(m->r_tables);
    if (m->merge_scratch) (void)hipFree(m->merge_scratch);
    if (m->lstm_planes) (void)hipFree(m->lstm_planes);
    if (m->gemm_planes) (void)hipFree(m->gemm_planes);
    if (m->unit_planes) (void)hipFree(m->unit_planes);
    delete m;
    return 0;
}

int l2s_min_T(int T) { int L[4]; return content_lens(T, L); }

int64_t l2s_workspace_bytes(int B, int T, int H, int W, int S) {
    (void)W;
    int64_t enc = enc_ws_floats(B, T, H), pro = prologue_ws_floats(B, T), dec = decode_ws_floats(B), post = postnet_ws_floats(B, S);
    int64_t io = (int64_t)B * T * 1024 + l2s_state_floats(B, T) + (int64_t)B * S * (NM + 1) + 64 * 8;   // l2s_inference intermediates
    int64_t mx = std::max(std::max(enc, pro), dec + post + 64);      // decode and post-net buffers are live together (overlap)
    return (mx + io) * (int64_t)sizeof(float) + (1 << 16);
}
int64_t l2s_state_floats(int B, int T) { return state_layout(B, T).total; }
int64_t l2s_state_offset(int B, int T, int field) {
    StateLayout s = state_layout(B, T);
    switch (field) {
        case L2S_ST_K: return s.k;
        case L2S_ST_V: return s.v;
        case L2S_ST_CKEY: return s.ckey;
        case L2S_ST_CVAL: return s.cval;
        case L2S_ST_ECELL: return s.ecell;
        case L2S_ST_H: return s.h;
        case L2S_ST_C: return s.c;
        case L2S_ST_ENC: return s.enc;
        case L2S_ST_STOPC: return s.stopc;
        case L2S_ST_VP: return s.vp;
    }
    return -1;
}

#define L2S_MODEL_READY(m) L2S_REQUIRE((m) && (m)->finalized, "model not finalized (call l2s_model_finalize)")
#define L2S_ENC_READY(m) L2S_MODEL_READY(m); L2S_REQUIRE((m)->has_enc, "model holds no encoder.* weights")
#define L2S_DEC_READY(m) L2S_MODEL_READY(m); L2S_REQUIRE((m)->has_dec, "model holds no decoder.* weights")

int l2s_encoder_fwd(l2s_model* m, const float* video, int B, int T, int H, int W, float* feat, void* ws, int64_t ws_bytes, void* stream) {
    L2S_ENC_READY(m);
    L2S_REQUIRE(video && feat && ws && B > 0 && T > 0, "bad arguments");
    return encoder_run(m, frame_src(video, B), B, T, H, W, nullptr, nullptr, feat, ws, ws_bytes, (hipStream_t)stream);
}

int l2s_normalise_pad_frames(const uint8_t* packed_u8, const int64_t* offsets, const int32_t* frames, int B, int T, int H, int W, float* video,
                             void* stream) {
    L2S_REQUIRE(packed_u8 && offsets && frames && video && B > 0 && T > 0, "bad arguments");
    return launch_normalise_pad(packed_u8, offsets, frames, B, T, H, W, video, (hipStream_t)stream);
}

int l2s_build_visual(const float* feat, const float* emb, int B, int T, float* vis, void* stream) {
    L2S_REQUIRE(feat && emb && vis && B > 0 && T > 0, "bad arguments");
    hipStream_t s = (hipStream_t)stream;
    if (launch_copy_cols(feat, L2S_D_FEAT, 0, vis, L2S_D_VIS, 0, 1, (int64_t)B * T, L2S_D_FEAT, s)) return 1;
    return launch_tile_rows(emb, L2S_D_EMB, vis + L2S_D_FEAT, L2S_D_VIS, B, T, L2S_D_EMB, s);
}

int l2s_decoder_prologue(l2s_model* m, const float* vis, const float* emb, const float* gumbel, int B, int T, float* state,
                         float* content_dis, void* ws, int64_t ws_bytes, void* stream) {
    L2S_DEC_READY(m);
    L2S_REQUIRE(vis && emb && gumbel && state && ws && B > 0, "bad arguments");
    return prologue_run(m, vis, emb, gumbel, B, T, state, content_dis, ws, ws_bytes, (hipStream_t)stream);
}

int l2s_decode_steps(l2s_model* m, float* state, int B, int T, int S, const float* teacher, const uint8_t* teacher_mask, float* mel,
                     float* stop, float* attn, int attn_logits, void* ws, int64_t ws_bytes, void* stream) {
    L2S_DEC_READY(m);
    L2S_REQUIRE(state && mel && stop && ws && B > 0, "bad arguments");
    return decode_run(m, state, B, T, S, teacher, teacher_mask, mel, stop, attn, attn_logits, ws, ws_bytes, (hipStream_t)stream);
}

int l2s_postnet(l2s_model* m, const float* mel, int B, int S, float* mel_post, float* mel_cf, void* ws, int64_t ws_bytes, void* stream) {
    L2S_DEC_READY(m);
    L2S_REQUIRE(mel && mel_post && ws && B > 0 && S > 0, "bad arguments");
    return postnet_run(m, mel, B, S, mel_post, mel_cf, ws, ws_bytes, (hipStream_t)stream);
}

int64_t l2s_speaker_workspace_bytes(int B, int n_samples) { return spk_ws_floats(B, n_samples) * (int64_t)sizeof(float) + (1 << 12); }

int l2s_speaker_encoder_fwd(l2s_model* m, const float* audio, int B, int n_samples, float* emb, void* ws, int64_t ws_bytes, void* stream) {
    L2S_MODEL_READY(m);
    L2S_REQUIRE(m->has_spk, "model holds no speaker_encoder.* weights");
    L2S_REQUIRE(audio && emb && ws && B > 0, "bad arguments");
    return speaker_run(m, audio, B, n_samples, emb, ws, ws_bytes, (hipStream_t)stream);
}

int l2s_output_lengths(const float* stop, int B, int S, int64_t* lengths, void* stream) {
    L2S_REQUIRE(stop && lengths && B > 0 && S > 0, "bad arguments");
    return launch_output_lengths(stop, B, S, lengths, (hipStream_t)stream);
}

// What one pass over a batch hands back.  inference(): mel_post + lengths (+ post-softmax attention); forward(tf_ratio) in eval mode
// (decoder.py:320-379): mel_cf, mel_post, stop, attention LOGITS, content_dis.
struct PathOut {
    float* mel_post = nullptr; float* mel_cf = nullptr; float* stop = nullptr; int64_t* lengths = nullptr;
    float* attn = nullptr; int attn_logits = 0; float* content_dis = nullptr;
};

static int path_run(l2s_model* m, const FrameSrc& video, const float* emb, const float* gumbel, int B, int T, int H, int W, int S,
                    const float* teacher, const uint8_t* teacher_mask, const PathOut& o, void* ws, int64_t ws_bytes, hipStream_t s) {
    X3Scope x3scope(m->opt.infer_bf16 ? 0 : m->opt.gemm_x3);
    Bf16Scope bf16scope(m->opt.infer_bf16);      // the bf16 leg: bf16-operand GEMM / Conv1d kernels instead of the f32 / split-bf16 ones
    Bump bp(ws, ws_bytes);
    float* vis = bp.f((int64_t)B * T * 1024);
    float* state = bp.f(l2s_state_floats(B, T));
    float* mel = bp.f((int64_t)B * S * NM);
    float* stop = o.stop ? o.stop : bp.f((int64_t)B * S);
    L2S_REQUIRE(!bp.overflow, "workspace too small (l2s_workspace_bytes)");
    void* rest = (char*)ws + bp.off;
    const int64_t rest_bytes = ws_bytes - bp.off;
    if (encoder_run(m, video, B, T, H, W, emb, vis, nullptr, rest, rest_bytes, s)) return 1;
    if (prologue_run(m, vis, emb, gumbel, B, T, state, o.content_dis, rest, rest_bytes, s)) return 1;
    const bool plain = !m->opt.overlap_postnet || g_prof_on || m->opt.graph || teacher || o.mel_cf;
    if (plain) {
        if (decode_run(m, state, B, T, S, teacher, teacher_mask, mel, stop, o.attn, o.attn_logits, rest, rest_bytes, s)) return 1;
        if (postnet_run(m, mel, B, S, o.mel_post, o.mel_cf, rest, rest_bytes, s)) return 1;
    } else {
        // The decode loop is a chain of small latency-bound launches that leaves most CUs idle, and the post-net of frame t
        // only needs mel frames t-10..t+10: run the post-net in time windows on a second stream while later steps decode.
        std::lock_guard<std::mutex> side_lock(m->side_mu);      // the side stream and its events are per model: one chain at a time enqueues on them
        if (!m->side) {
            L2S_CHECK_HIP(hipStreamCreateWithFlags(&m->side, hipStreamNonBlocking));
            L2S_CHECK_HIP(hipEventCreateWithFlags(&m->ev_in, hipEventDisableTiming));
            L2S_CHECK_HIP(hipEventCreateWithFlags(&m->ev_out, hipEventDisableTiming));
        }
        Bump pbump((char*)rest + align_up(decode_ws_floats(B) * (int64_t)sizeof(float), 256), rest_bytes - align_up(decode_ws_floats(B) * (int64_t)sizeof(float), 256));
        PostBufs pb;
        L2S_REQUIRE(postnet_alloc(pbump, B, S, pb) == 0, "workspace too small (l2s_workspace_bytes)");
        int done[5] = {0, 0, 0, 0, 0};      // frames finished per post-net layer
        size_t ev_used = 0;
        const int chunk = 64, tail = 12;
        std::function<int(int)> on_frames = [&](int n) -> int {
            const bool boundary = (n == S) || (n % chunk == 0 && n < S - tail) || (n == S - tail && S > tail);
            if (!boundary) return 0;
            if (ev_used >= m->ev_pool.size()) {
                hipEvent_t e;
                L2S_CHECK_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
                m->ev_pool.push_back(e);
            }
            hipEvent_t ev = m->ev_pool[ev_used++];
            L2S_CHECK_HIP(hipEventRecord(ev, s));
            L2S_CHECK_HIP(hipStreamWaitEvent(m->side, ev, 0));
            for (int layer = 0; layer < 5; ++layer) {
                const int end = n == S ? S : std::max(done[layer], n - 2 * (layer + 1));
                if (postnet_layer(m->w, layer, mel, pb, o.mel_post, B, S, done[layer], end, m->side, m->opt.gemm_x3_dma != 0)) return 1;
                done[layer] = end;
            }
            return 0;
        };
        // the side stream must not start before earlier work on `s` (previous users of these buffers) is done
        L2S_CHECK_HIP(hipEventRecord(m->ev_in, s));
        L2S_CHECK_HIP(hipStreamWaitEvent(m->side, m->ev_in, 0));
        if (decode_launches(m, state, B, T, S, nullptr, nullptr, mel, stop, o.attn, o.attn_logits, rest, rest_bytes, s, m->opt.fold != 0 && m->folded_valid, &on_frames)) return 1;
        L2S_CHECK_HIP(hipEventRecord(m->ev_out, m->side));
        L2S_CHECK_HIP(hipStreamWaitEvent(s, m->ev_out, 0));
    }
    return o.lengths ? launch_output_lengths(stop, B, S, o.lengths, s) : 0;
}

static int inference_run(l2s_model* m, const FrameSrc& video, const float* emb, const float* gumbel, int B, int T, int H, int W, int S,
                         float* mel_post, int64_t* lengths, float* attn, void* ws, int64_t ws_bytes, hipStream_t s) {
    PathOut o;
    o.mel_post = mel_post; o.lengths = lengths; o.attn = attn;
    return path_run(m, video, emb, gumbel, B, T, H, W, S, nullptr, nullptr, o, ws, ws_bytes, s);
}

int l2s_inference(l2s_model* m, const float* video, const float* emb, const float* gumbel, int B, int T, int H, int W, int S,
                  float* mel_post, int64_t* lengths, float* attn, void* ws, int64_t ws_bytes, void* stream) {
    L2S_ENC_READY(m);
    L2S_DEC_READY(m);
    L2S_REQUIRE(video && emb && gumbel && mel_post && lengths && ws && B > 0, "bad arguments");
    return inference_run(m, frame_src(video, B), emb, gumbel, B, T, H, W, S, mel_post, lengths, attn, ws, ws_bytes, (hipStream_t)stream);
}

// Grouped inference: the G batches are rows g*B .. g*B+B-1 of ONE launch chain on ONE weight blob.  Every kernel of the path is row-independent
// (a row's arithmetic does not depend on how many rows share the launch), so each batch's results are bit-identical to l2s_inference on it.
int64_t l2s_workspace_bytes_multi(int G, int B, int T, int H, int W, int S) {
    int L[4];
    const int64_t rows = (int64_t)G * B;
    return l2s_workspace_bytes((int)rows, T, H, W, S) + align_up(rows * L2S_D_EMB * 4, 256) + align_up(rows * content_lens(T, L) * VOC * 4, 256) +
           align_up(rows * S * NM * 4, 256);      // + the gathered teacher frames of l2s_forward_eval_multi
}
int l2s_inference_multi(l2s_model* m, int G, const float* const* video, const float* const* emb, const float* const* gumbel, int B, int T, int H,
                        int W, int S, float* mel_post, int64_t* lengths, float* attn, void* ws, int64_t ws_bytes, void* stream) {
    L2S_ENC_READY(m);
    L2S_DEC_READY(m);
    L2S_REQUIRE(G >= 1 && G <= L2S_MAX_GROUP && video && emb && gumbel && mel_post && lengths && ws && B > 0, "bad arguments");
    hipStream_t s = (hipStream_t)stream;
    int L[4];
    const int mT = content_lens(T, L);
    Bump bp(ws, ws_bytes);
    float* emb_all = bp.f((int64_t)G * B * L2S_D_EMB);
    float* gum_all = bp.f((int64_t)G * B * mT * VOC);
    L2S_REQUIRE(!bp.overflow, "workspace too small (l2s_workspace_bytes_multi)");
    FrameSrc src{};
    src.per = B;
    for (int g = 0; g < G; ++g) {
        L2S_REQUIRE(video[g] && emb[g] && gumbel[g], "null batch pointer");
        src.p[g] = video[g];
        // the small per-batch operands are gathered (G x 32 KB + G x 256 KB at B = 32); the frames (102.6 MB per batch) are read in place
        L2S_CHECK_HIP(hipMemcpyAsync(emb_all + (int64_t)g * B * L2S_D_EMB, emb[g], sizeof(float) * B * L2S_D_EMB, hipMemcpyDeviceToDevice, s));
        L2S_CHECK_HIP(hipMemcpyAsync(gum_all + (int64_t)g * B * mT * VOC, gumbel[g], sizeof(float) * B * mT * VOC, hipMemcpyDeviceToDevice, s));
    }
    X3Group x3group(G);
    return inference_run(m, src, emb_all, gum_all, G * B, T, H, W, S, mel_post, lengths, attn, (char*)ws + bp.off, ws_bytes - bp.off, s);
}

// Lip2Speech.forward(..., tf_ratio) in eval() mode / under no_grad (model.py:23-40 + decoder.py:320-379; what evaluate.py:38 runs at
// tf_ratio = 1): S = mels.shape[2] steps, attention LOGITS out, optional teacher frames for the steps the caller's scheduled-sampling
// draws selected.  One launch chain, like l2s_inference.
int l2s_forward_eval(l2s_model* m, const float* video, const float* emb, const float* gumbel, int B, int T, int H, int W, int S,
                     const float* teacher, const uint8_t* teacher_mask, float* mel_cf, float* mel_post, float* stop, float* attn_logits,
                     float* content_dis, void* ws, int64_t ws_bytes, void* stream) {
    L2S_ENC_READY(m);
    L2S_DEC_READY(m);
    L2S_REQUIRE(video && emb && gumbel && mel_post && stop && ws && B > 0, "bad arguments");
    L2S_REQUIRE((teacher != nullptr) == (teacher_mask != nullptr), "teacher frames and teacher_mask come together");
    PathOut o;
    o.mel_post = mel_post; o.mel_cf = mel_cf; o.stop = stop; o.attn = attn_logits; o.attn_logits = 1; o.content_dis = content_dis;
    return path_run(m, frame_src(video, B), emb, gumbel, B, T, H, W, S, teacher, teacher_mask, o, ws, ws_bytes, (hipStream_t)stream);
}

// The grouped form: G batches of the evaluate loop as rows of ONE launch chain (see l2s_inference_multi).  The batches of a group share S
// and the scheduled-sampling mask (at tf_ratio = 1 - evaluate.py - no step is ever teacher-forced, so any G batches group).
int l2s_forward_eval_multi(l2s_model* m, int G, const float* const* video, const float* const* emb, const float* const* gumbel,
                           const float* const* teacher, const uint8_t* teacher_mask, int B, int T, int H, int W, int S, float* mel_cf,
                           float* mel_post, float* stop, float* attn_logits, float* content_dis, void* ws, int64_t ws_bytes, void* stream) {
    L2S_ENC_READY(m);
    L2S_DEC_READY(m);
    L2S_REQUIRE(G >= 1 && G <= L2S_MAX_GROUP && video && emb && gumbel && mel_post && stop && ws && B > 0, "bad arguments");
    L2S_REQUIRE((teacher != nullptr) == (teacher_mask != nullptr), "teacher frames and teacher_mask come together");
    L2S_REQUIRE(S >= 1 && S <= L2S_MAX_STEPS, "S must be in [1, 300] (positional table)");
    hipStream_t s = (hipStream_t)stream;
    int L[4];
    const int mT = content_lens(T, L);
    Bump bp(ws, ws_bytes);
    float* emb_all = bp.f((int64_t)G * B * L2S_D_EMB);
    float* gum_all = bp.f((int64_t)G * B * mT * VOC);
    float* teach_all = teacher ? bp.f((int64_t)G * B * S * NM) : nullptr;
    L2S_REQUIRE(!bp.overflow, "workspace too small (l2s_workspace_bytes_multi)");
    FrameSrc src{};
    src.per = B;
    for (int g = 0; g < G; ++g) {
        L2S_REQUIRE(video[g] && emb[g] && gumbel[g] && (!teacher || teacher[g]), "null batch pointer");
        src.p[g] = video[g];
        L2S_CHECK_HIP(hipMemcpyAsync(emb_all + (int64_t)g * B * L2S_D_EMB, emb[g], sizeof(float) * B * L2S_D_EMB, hipMemcpyDeviceToDevice, s));
        L2S_CHECK_HIP(hipMemcpyAsync(gum_all + (int64_t)g * B * mT * VOC, gumbel[g], sizeof(float) * B * mT * VOC, hipMemcpyDeviceToDevice, s));
        if (teacher)
            L2S_CHECK_HIP(hipMemcpyAsync(teach_all + (int64_t)g * B * S * NM, teacher[g], sizeof(float) * B * S * NM, hipMemcpyDeviceToDevice, s));
    }
    X3Group x3group(G);
    PathOut o;
    o.mel_post = mel_post; o.mel_cf = mel_cf; o.stop = stop; o.attn = attn_logits; o.attn_logits = 1; o.content_dis = content_dis;
    return path_run(m, src, emb_all, gum_all, G * B, T, H, W, S, teach_all, teacher_mask, o, (char*)ws + bp.off, ws_bytes - bp.off, s);
}

// ---- operator-level entry points
#ifdef L2S_DIAG      // operator-level test hooks: libl2s_diag.so (include/l2s_diag.h)
int l2s_op_gemm(const float* A, const float* Wt, const float* scale, const float* shift, const float* actw, float* C, int M, int N,
                int K, int act, void* stream) {
    GemmP p = gemm_plain(A, K, Wt, C, N, M, N, K);
    p.scale = scale; p.shift = shift; p.actw = actw; p.act = act;
    return launch_gemm1(p, (hipStream_t)stream, "op_gemm");
}
int l2s_op_conv1d(const float* X, const float* Wp, const float* scale, const float* shift, const float* actw, float* out, int B,
                  int Tin, int Cin, int Cout, int taps, int stride, int pad, int act, void* stream) {
    ConvW c; c.W = Wp; c.scale = scale; c.shift = shift; c.actw = actw;
    GemmP p = conv_gemm(X, Cin, B, Tin, Cin, c, Cout, taps, stride, pad, out, Cout, act);
    return launch_gemm1(p, (hipStream_t)stream, "op_conv1d");
}
// flags bit 8 of the two operators below: the weight operand as pre-split bf16 planes fetched by LDS-DMA (GemmP::W3; needs N % 256 == 0 and
// K % 16 == 0, ignored otherwise) - derived per call into a scratch buffer the library owns (operator tests and tools; the model paths keep their own)
static void* g_op_planes = nullptr;
static int64_t g_op_planes_bytes = 0;
static std::mutex g_op_planes_mu;
static const void* op_planes(const float* W, int N, int K, hipStream_t s) {
    if (N % 256 || K % 16) return nullptr;
    std::lock_guard<std::mutex> lk(g_op_planes_mu);
    const int64_t need = (int64_t)N * K * 6;
    if (need > g_op_planes_bytes) {
        if (g_op_planes) { (void)hipDeviceSynchronize(); (void)hipFree(g_op_planes); g_op_planes = nullptr; g_op_planes_bytes = 0; }
        if (hipMalloc(&g_op_planes, need) != hipSuccess) { g_op_planes = nullptr; return nullptr; }
        g_op_planes_bytes = need;
    }
    return launch_gemm_planes(W, N, K, g_op_planes, s) ? nullptr : g_op_planes;
}
int l2s_op_gemm_ex(const float* A, const float* Wt, const float* scale, const float* shift, const float* actw, float* C, int M, int N,
                   int K, int act, int flags, void* stream) {
    X3Scope x3scope((flags & 1) ? (3 | (flags & 4)) : 0);      // forced; flags bit 4: the narrow tile
    Bf16Scope bf16scope((flags & 2) ? 1 : 0);
    GemmP p = gemm_plain(A, K, Wt, C, N, M, N, K);
    p.scale = scale; p.shift = shift; p.actw = actw; p.act = act;
    if (flags & 8) p.W3 = op_planes(Wt, N, K, (hipStream_t)stream);
    return launch_gemm1(p, (hipStream_t)stream, "op_gemm");
}
int l2s_op_conv1d_ex(const float* X, const float* Wp, const float* scale, const float* shift, const float* actw, float* out, int B,
                     int Tin, int Cin, int Cout, int taps, int stride, int pad, int act, int flags, void* stream) {
    X3Scope x3scope((flags & 1) ? (3 | (flags & 4)) : 0);      // forced; flags bit 4: the narrow tile
    Bf16Scope bf16scope((flags & 2) ? 1 : 0);
    ConvW c; c.W = Wp; c.scale = scale; c.shift = shift; c.actw = actw;
    if (flags & 8) c.W3 = op_planes(Wp, Cout, taps * Cin, (hipStream_t)stream);
    GemmP p = conv_gemm(X, Cin, B, Tin, Cin, c, Cout, taps, stride, pad, out, Cout, act);
    return launch_gemm1(p, (hipStream_t)stream, "op_conv1d");
}
int l2s_op_conv1d_bwd(const float* dZ, const float* X, const float* Wp, float* dX, float* dWp, int B, int Tin, int Cin, int Cout, int taps,
                      int stride, int pad, void* stream) {
    L2S_REQUIRE(dZ && X && Wp, "bad arguments");
    const int Tout = (Tin + 2 * pad - taps) / stride + 1;
    if (dX) {
        L2S_REQUIRE(stride == 1, "dX of a strided Conv1d is computed on the (B*Tout, taps*Cin) view by the caller");
        if (launch_gemm_bwd(bwd_dx(dZ, Cout, Wp, dX, Cin, B, Tout, Tin, Cout, Cin, taps, pad, false), (hipStream_t)stream, "op_conv1d_dx")) return 1;
    }
    if (dWp && launch_gemm_bwd(bwd_dw(dZ, Cout, X, Cin, dWp, B, Tout, Tin, Cout, Cin, taps, stride, pad, false), (hipStream_t)stream, "op_conv1d_dw")) return 1;
    return 0;
}
int l2s_op_frontend(l2s_model* m, const float* video, int B, int T, int H, int W, float* out, void* stream) {
    L2S_ENC_READY(m);
    FrontendW fe = m->w.fe;
    if (!m->opt.frontend_x3 || !m->planes_valid) fe.w3 = nullptr;
    fe.pair = m->opt.frontend_x3 >= 2; fe.pipe = m->opt.frontend_x3 == 3;
    if (!m->opt.infer_bf16 || !m->planes_valid) fe.w1 = nullptr;
    return launch_frontend(fe, frame_src(video, B), B, T, H, W, out, (hipStream_t)stream);
}

#endif

int l2s_train_set_bn(l2s_model* m, int batch_stats, float momentum) {
    L2S_REQUIRE(m != nullptr && momentum >= 0.f && momentum <= 1.f, "bad arguments");
    m->bn_batch = batch_stats != 0;
    m->bn_momentum = momentum;
    return 0;
}

int l2s_train_refresh_weights(l2s_model* m, void* stream) {
    L2S_REQUIRE(m != nullptr, "null model");
    return refresh_weights(m, (hipStream_t)stream);
}

int l2s_set_option(const char* name, int value) {
    L2S_REQUIRE(name != nullptr, "null option name");
    if (set_option_field(g_default_opt, name, value)) { set_error(std::string("unknown option ") + name); return 1; }
    if (!std::strcmp(name, "persist_decode") && value > 0) pdecode_rearm();
    return 0;
}
int l2s_model_set_option(l2s_model* m, const char* name, int value) {
    L2S_REQUIRE(m != nullptr && name != nullptr, "bad arguments");
    if (set_option_field(m->opt, name, value)) { set_error(std::string("unknown option ") + name); return 1; }
    if (!std::strcmp(name, "persist_decode") && value > 0) pdecode_rearm();      // asking for the persistent forms (again) forgives the device's earlier time-outs
    return 0;
}

#ifdef L2S_DIAG      // chain microbenches, timelines, probes: libl2s_diag.so (include/l2s_diag.h)
// Average duration of the decoder LSTM-cell kernel (the kernel with the largest share of GPU time) measured with ONE pair of HIP
// events around a chain of n_pairs x {layer 0 (K=1536), layer 1 (K=1024)} launches on `stream` - the same launches the decode loop
// issues, on zeroed state.  Per-launch event brackets (l2s_profile_*) add ~1.8 us to a 6 us kernel; this does not.  Synchronises.
int l2s_op_lstm_cell_chain(l2s_model* m, int B, int n_pairs, void* ws, int64_t ws_bytes, void* stream, double* avg_us) {
    L2S_DEC_READY(m);
    L2S_REQUIRE(ws && avg_us && B > 0 && n_pairs > 0, "bad arguments");
    hipStream_t s = (hipStream_t)stream;
    const Weights& w = m->w;
    const int Bp = pad16(B);
    Bump bp(ws, ws_bytes);
    float* h0[2] = {bp.f((int64_t)Bp * 512), bp.f((int64_t)Bp * 512)};
    float* h1[2] = {bp.f((int64_t)Bp * 512), bp.f((int64_t)Bp * 512)};
    float* c0 = bp.f((int64_t)Bp * 512); float* c1 = bp.f((int64_t)Bp * 512); float* av = bp.f((int64_t)Bp * 512);
    float* cc = bp.f((int64_t)Bp * 256); float* p2f = bp.f((int64_t)Bp * 256);
    L2S_REQUIRE(!bp.overflow, "workspace too small");
    for (float* z : {h0[0], h0[1], h1[0], h1[1], c0, c1, av}) if (launch_fill(z, (int64_t)Bp * 512, 0.f, s)) return 1;
    for (float* z : {cc, p2f}) if (launch_fill(z, (int64_t)Bp * 256, 0.f, s)) return 1;
    hipEvent_t e0, e1;
    L2S_CHECK_HIP(hipEventCreate(&e0));
    L2S_CHECK_HIP(hipEventCreate(&e1));
    auto pair = [&](int cur) -> int {
        const int nxt = cur ^ 1;
        SkinnyBatch sb{};
        const bool vhoist = m->opt.hoist_vproj && w.lstm0v.W && w.vproj.W;      // the layer-0 launch of the production step (decode_launches)
        const bool vsum = vhoist && m->opt.hoist_vproj >= 2 && skinny_sum_supported(m->opt);
        SkinnyP a = sk_base(vsum ? w.lstm0 : vhoist ? w.lstm0v : w.lstm0f, B);
        if (vsum) { a.seg[0] = {cc, 16}; a.seg[1] = {p2f, 16}; a.a_sum = av; a.seg[2] = {h0[cur], 32}; a.nseg = 3; }
        else { a.seg[0] = {cc, 16}; a.seg[1] = {p2f, 16}; a.seg[2] = {av, vhoist ? 16 : 32}; a.seg[3] = {h0[cur], 32}; a.nseg = 4; }
        a.epi = SK_LSTM; a.H = 512; a.c_in = c0; a.c_out = c0; a.h_out = h0[nxt]; a.h_out_K = 512; a.h_out_off = 0;
        sb.p[0] = a; sb.ntiles[0] = 128; sb.count = 1;
        if (launch_skinny(sb, s, "step_lstm_cell", m->opt)) return 1;
        SkinnyBatch sc{};
        SkinnyP b = sk_base(w.lstm1, B);
        b.seg[0] = {h0[nxt], 32}; b.seg[1] = {h1[cur], 32}; b.nseg = 2;
        b.epi = SK_LSTM; b.H = 512; b.c_in = c1; b.c_out = c1; b.h_out = h1[nxt]; b.h_out_K = 512; b.h_out_off = 0;
        sc.p[0] = b; sc.ntiles[0] = 128; sc.count = 1;
        return launch_skinny(sc, s, "step_lstm_cell", m->opt);
    };
    for (int i = 0; i < 8; ++i) if (pair(i & 1)) return 1;                   // warm-up
    L2S_CHECK_HIP(hipEventRecord(e0, s));
    for (int i = 0; i < n_pairs; ++i) if (pair(i & 1)) return 1;
    L2S_CHECK_HIP(hipEventRecord(e1, s));
    L2S_CHECK_HIP(hipEventSynchronize(e1));
    float ms = 0.f;
    L2S_CHECK_HIP(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    *avg_us = (double)ms * 1e3 / (2.0 * n_pairs);
    return 0;
}

/* measurement: ts_dev != NULL routes every skinny launch to the stamped build (8 wall-clock stamps per block into ts_dev); NULL restores */
// measurement: n back-to-back launches of the step's attention kernel alone (same K / V / content state every launch, zero queries) - does a
// clip's 119 KB of K / V stay in its XCD's L2 from one launch to the next when no weight stream runs in between?  (tools/attn_l2_probe.py)
int l2s_op_step_attn_chain(l2s_model* m, float* state, int B, int T, int n_launches, void* ws, int64_t ws_bytes, void* stream) {
    L2S_DEC_READY(m);
    L2S_REQUIRE(state && ws && B >= 1 && n_launches >= 1, "bad arguments");
    hipStream_t s = (hipStream_t)stream;
    const Weights& w = m->w;
    StateLayout sl = state_layout(B, T);
    const int Bp = pad16(B);
    Bump bp(ws, ws_bytes);
    float* q = bp.f((int64_t)B * 512); float* qc = bp.f((int64_t)B * 256);
    float* av = bp.f((int64_t)Bp * 512); float* cc = bp.f((int64_t)Bp * 256); float* p1 = bp.f((int64_t)Bp * 256); float* p2f = bp.f((int64_t)Bp * 256);
    L2S_REQUIRE(!bp.overflow, "workspace too small");
    if (launch_fill(q, (int64_t)B * 512, 0.f, s) || launch_fill(qc, (int64_t)B * 256, 0.f, s) || launch_fill(p1, (int64_t)Bp * 256, 0.f, s)) return 1;
    for (int i = 0; i < n_launches; ++i) {
        AttnP at{};
        at.q = q; at.ldq = 512; at.k = state + sl.k; at.v = state + sl.v; at.tau = w.tau; at.av_frag = av;
        if (m->opt.hoist_vproj && w.vproj.W) at.vp = state + sl.vp;      // the production form: 256 value columns
        at.attn_out = nullptr; at.ld_attn_b = 0; at.attn_logits = 0;
        at.qc = qc; at.ldqc = 256; at.ckey = state + sl.ckey; at.cval = state + sl.cval; at.tau_c = w.tau_c; at.cc_frag = cc;
        at.B = B; at.T = T; at.m = sl.m;
        SkinnyP pr = sk_base(w.pre2, B);
        pr.seg[0] = {p1, 16}; pr.nseg = 1; pr.act = ACT_PSINE; pr.epi = SK_FRAG; pr.out = p2f; pr.ldo = 256;
        if (launch_step_attn(at, pr, w.pre2.tiles, s, m->opt.attn_lds, m->opt.attn_skip0)) return 1;
    }
    return 0;
}
int l2s_op_skinny_timeline(void* ts_dev) { skinny_set_timeline((unsigned long long*)ts_dev); return 0; }
int l2s_op_attn_timeline(void* ts_dev) { attn_set_timeline((unsigned long long*)ts_dev); return 0; }
int l2s_op_flat_timeline(void* ts_dev) { skinny_set_flat_timeline((unsigned long long*)ts_dev); return 0; }
int l2s_op_pdecode_timeline(void* ts_dev, int step) { pdecode_set_timeline((unsigned long long*)ts_dev, step); return 0; }
int l2s_op_gemm_x3_timeline(void* ts_dev, int block) { gemm_x3_set_timeline((unsigned long long*)ts_dev, block); return 0; }
int l2s_op_fused_unit_timeline(void* ts_dev, int h) { shuffle_set_timeline((unsigned long long*)ts_dev, h); return 0; }
int l2s_op_stamp_log(void* log_dev, int64_t capacity) {
    L2S_REQUIRE(set_stamp_log((unsigned long long*)log_dev, (long long)capacity) == 0, "stamp log: hipMemcpyToSymbol failed");
    return 0;
}

int l2s_op_launch_chain(int kind, int n_launches, int blocks, int n_per_block, const float* in, float* out, void* stream) {
    for (int i = 0; i < n_launches; ++i)
        if (launch_probe(kind, blocks, n_per_block, in, out, (hipStream_t)stream)) return 1;
    return 0;
}

int l2s_op_launch_chain2(int kind, int n_launches, int blocks, int n_per_block, const float* in, float* out, void* stream_a, void* stream_b) {
    for (int i = 0; i < n_launches; ++i) {
        if (launch_probe(kind, blocks, n_per_block, in, out, (hipStream_t)stream_a)) return 1;
        if (launch_probe(kind, blocks, n_per_block, in + (int64_t)(1 << 24), out + 2048, (hipStream_t)stream_b)) return 1;
    }
    return 0;
}
#endif      // L2S_DIAG

int l2s_set_thread_chains(int n) { chains_hint() = n < 1 ? 1 : n; return 0; }
int l2s_persist_available(void) { return pdecode_device_ok() ? 1 : 0; }
int l2s_persist_timeouts(void) { return pdecode_timeouts(); }

int l2s_profile_enable(int on) { g_prof_on = on != 0; return 0; }
int l2s_profile_reset(void) { prof_drain(); g_prof.clear(); g_prof_idx.clear(); return 0; }
int l2s_profile_count(void) { prof_drain(); return (int)g_prof.size(); }
int l2s_profile_get(int idx, const char** name, int64_t* launches, double* total_ms) {
    prof_drain();
    L2S_REQUIRE(idx >= 0 && idx < (int)g_prof.size(), "profile index");
    if (name) *name = g_prof[idx].name.c_str();
    if (launches) *launches = g_prof[idx].launches;
    if (total_ms) *total_ms = g_prof[idx].total_ms;
    return 0;
}

}  // extern "C"
