// Model-side declarations shared by the HIP translation units that orchestrate launches (l2s_api.hip, train_decoder.hip).
#pragma once
#include "../../include/l2s.h"
#include "l2s_common.h"

#include <algorithm>
#include <map>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

namespace l2s {

// ------------------------------------------------------------------------------------------------ geometry
constexpr int STAGE_CH[4] = {24, 116, 232, 464};
constexpr int STAGE_REP[3] = {4, 8, 4};
constexpr int N_UNITS = 16;
constexpr int LAST_CH = 768;
constexpr int MH_KS[4] = {1, 3, 7, 11};
constexpr int CT_KS[4] = {1, 3, 5, 7};
constexpr int D = 512;          // decoder width
constexpr int NM = L2S_N_MELS;
constexpr int VOC = L2S_VOCAB;
constexpr int VOCP = 504;       // vocabulary padded to a multiple of 4 for float4 operand loads
constexpr float BN_EPS = 1e-5f;

static inline int64_t align_up(int64_t x, int64_t a) { return (x + a - 1) / a * a; }
static inline int pad16(int b) { return (b + 15) & ~15; }

// ------------------------------------------------------------------------------------------------ model
struct ConvW { const float* W = nullptr; const float* scale = nullptr; const float* shift = nullptr; const float* actw = nullptr;
               const void* W3 = nullptr; };      // W3: W as pre-split bf16 planes (GemmP::W3), derived on the device (derive_gemm_planes)
struct DwW { const float* w9 = nullptr; const float* scale = nullptr; const float* shift = nullptr; };
struct UnitW {
    bool stride2 = false;
    int cin = 0, half = 0;
    DwW b1_dw; ConvW b1_pw;            // stride-2 units only
    ConvW pw1; DwW dw; ConvW pw2;      // banch2
    const float* pw1_frag = nullptr; const float* pw2_frag = nullptr; int kpad = 0;   // fused units: frag16 [pad16(half)][K]; kpad = pad16(half)
    const float* b1_frag = nullptr; int kin = 0;                                      // stride-2 units: banch1 pw, and pw1, have K = kin = pad16(cin)
    const void* pw1_p3 = nullptr; const void* pw2_p3 = nullptr; const void* b1_p3 = nullptr;   // the same as bf16 operand planes (launch_su_planes; option "trunk_x3")
};
struct SkW { const float* W = nullptr; const float* bias = nullptr; const float* actw = nullptr; int N = 0, K = 0, tiles = 0; const void* W3 = nullptr; };      // W3: bf16 planes (split-bf16 LSTM blocks)

struct Weights {
    FrontendW fe;
    UnitW unit[N_UNITS];
    ConvW conv_last;
    // decoder prologue
    ConvW resid, enc_site, attn_site, e_c, enc_proj;
    const float* wih_cat = nullptr; const float* bih_cat = nullptr;     // [4096][1024], [4096]
    const void* wih_cat3 = nullptr;                                      // wih_cat as pre-split bf16 planes (GemmP::W3)
    SkW whh[2];                                                         // BiLSTM recurrent weights, permuted rows
    ConvW mh_branch[2][4], mh_bott[2];                                  // K, V
    const float* pos = nullptr;                                         // [300][512]
    ConvW ct_branch[4], ct_bott, ct_k0, ct_k2, ct_fc0, ct_fc2, ct_fc4, ct_emb;
    // decode step
    SkW pre1, pre2, q, cq, aproj, lstm0, lstm1, fc;
    SkW pre1f, lstm0f;                 // phase-merged forms: prenet1 o fc_out over h1; LSTM0 with attention_proj folded in
    SkW lstm0v;                        // LSTM0 over [content | prenet | o | h0] (K = 1280), o = a @ V' with V' = V W_ap^T + b_ap hoisted into the prologue:
                                       //   [W_ih[:, :256] | W_ih[:, 256:] | W_ih[:, 256:] | W_hh], the plain LSTM0 bias
    ConvW vproj;                       // attention_proj as a row-major [256][512] GEMM weight + bias (the prologue's V' = V W_ap^T + b_ap)
    const float* stop_tail = nullptr; const float* stop_bias = nullptr;
    const float* bos = nullptr; const float* tau = nullptr; const float* tau_c = nullptr;
    // postnet
    ConvW post[5];
    // speaker encoder (audio.py:110-150): mel40 front-end tables + 3-layer LSTM(256) + Linear
    const float* spk_window = nullptr; const float* spk_dft = nullptr; const float* spk_fbT = nullptr;   // [400], [402][400], [40][204]
    ConvW spk_ih[3];                 // input weights [1024][in] with shift = b_ih + b_hh
    SkW spk_hh[3];                   // recurrent weights, frag16, rows permuted to (unit, gate)
    ConvW spk_linear;
};

}  // namespace l2s

struct l2s_model {
    std::unordered_map<std::string, std::vector<float>> host;
    float* blob = nullptr;
    int64_t blob_floats = 0;
    bool finalized = false;
    bool has_enc = false, has_dec = false, has_spk = false;
    l2s::Weights w;
    l2s::Options opt;            // this model's run-time options (l2s_model_set_option; defaults from l2s_set_option at creation)
    // captured decode loops (hipGraph), replayed on a private non-blocking stream fenced against the caller's stream
    struct GraphEntry { int B, T, S, attn_logits, fold; const void *state, *mel, *stop, *attn, *ws; hipGraph_t graph; hipGraphExec_t exec; };
    std::vector<GraphEntry> graphs;
    hipStream_t side = nullptr;
    hipEvent_t ev_in = nullptr, ev_out = nullptr;
    std::vector<hipEvent_t> ev_pool;
    std::mutex side_mu;          // guards side / ev_in / ev_out / ev_pool / graphs: several host threads may drive ONE model (parallel.InflightPool)
    // training: device pointers of the canonical (PyTorch-layout) parameters and of their gradient slots, bound by key
    std::unordered_map<std::string, std::pair<float*, float*>> bound;
    // training: device-side refresh of the packed blob from the bound tensors (l2s_train_refresh_weights).  Built by l2s_model_finalize
    // when the "refresh_map" option is on: for every blob float that is a verbatim copy of a checkpoint element, its source (key index,
    // element index); plus the records of the entries that are computed (eval-BatchNorm folds, bias sums).
    struct RefreshBn { std::string p, bias_key; int c; int64_t so, ho; };
    struct RefreshSum { std::string a, b; int n, perm_H; int64_t dst; };
    std::vector<RefreshBn> r_bn;
    std::vector<RefreshSum> r_sum;
    std::vector<std::string> r_keys;
    int32_t* r_key = nullptr; int32_t* r_idx = nullptr;      // device [blob_floats]
    void* r_tables = nullptr; int64_t r_tables_bytes = 0;     // device scratch for the pointer / descriptor tables
    std::vector<char> r_tables_host, r_tables_uploaded;       // what the next refresh needs / what the device table holds
    bool folded_valid = true;                                 // the phase-merged step weights match the current parameters
    bool planes_valid = true;                                 // the front-end's bf16 operand planes (w3 / w1) match the current parameters
    float* merge_scratch = nullptr;                           // device: the two products of the device-side re-merge (l2s_train_refresh_weights)
    void* gemm_planes = nullptr;                              // device: bf16 planes of the post-net's Conv1d weights, the BiLSTM input matrix and conv_last (option "gemm_x3_dma"); rebuilt like lstm_planes
    void* unit_planes = nullptr;                              // device: bf16 operand planes of the fused ShuffleNet units' pointwise convs (option "trunk_x3"); rebuilt like lstm_planes
    void* lstm_planes = nullptr;                              // device: bf16 planes of the decoder LSTM weights (split-bf16 LSTM blocks, option "lstm_x3"); rebuilt
                                                              //   from the packed fp32 fragments after every pack / device-side refresh
    // training: BatchNorm layers normalise with batch statistics and update their running statistics (nn.Module.train()); off = running statistics
    bool bn_batch = false; float bn_momentum = 0.1f;
    const float* canon(const std::string& key) const { auto it = bound.find(key); return it == bound.end() ? nullptr : it->second.first; }
    float* grad(const std::string& key) const { auto it = bound.find(key); return it == bound.end() ? nullptr : it->second.second; }
};


namespace l2s {
constexpr int NM_ = L2S_N_MELS;
struct Bump {
    char* base; int64_t cap; int64_t off = 0; bool overflow = false;
    Bump(void* p, int64_t c) : base((char*)p), cap(c) {}
    float* f(int64_t n) {
        int64_t bytes = align_up(n * (int64_t)sizeof(float), 256);
        if (off + bytes > cap) { overflow = true; off += bytes; return nullptr; }
        float* r = (float*)(base + off);
        off += bytes;
        return r;
    }
};
extern bool g_prof_on;

inline int content_lens(int T, int L[4]) {
    int m = T;
    for (int j = 0; j < 4; ++j) {
        L[j] = T >= CT_KS[j] ? (T - CT_KS[j]) / CT_KS[j] + 1 : 0;
        m = std::min(m, L[j]);
    }
    return m;
}

struct StateLayout { int64_t k, v, ckey, cval, ecell, h, c, enc, stopc, vp, total; int m; };
inline StateLayout state_layout(int B, int T) {
    StateLayout s{};
    int L[4];
    s.m = content_lens(T, L);
    int64_t o = 0;
    auto take = [&](int64_t n) { int64_t r = o; o += align_up(n, 64); return r; };
    s.k = take((int64_t)B * T * 512);
    s.v = take((int64_t)B * T * 512);
    s.ckey = take((int64_t)B * s.m * 256);
    s.cval = take((int64_t)B * s.m * 256);
    s.ecell = take((int64_t)B * 512);
    s.h = take((int64_t)pad16(B) * 512 * 2);
    s.c = take((int64_t)pad16(B) * 512 * 2);
    s.enc = take((int64_t)B * T * 512);
    s.stopc = take(B);
    s.vp = take((int64_t)B * T * 256);      // V' = V W_ap^T + b_ap (decoder.py:420 applied to the values once instead of to a @ v every step)
    s.total = o;
    return s;
}

}  // namespace l2s
