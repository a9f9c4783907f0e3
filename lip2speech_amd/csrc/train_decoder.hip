// Training path of the decoder (reference: Decoder.forward /root/reference/model/modules/decoder.py:320-379 under
// loss.backward(), train.py:184): forward with a tape of the intermediates the backward needs, and the backward itself.
// Stage 1 (this file, so far): the post-net (decoder.py:107-156).  Eval-mode normalisation statistics and no dropout - the
// configuration SURVEY.md §8(a) a16(iii) pins with gradient goldens; batch-statistics BatchNorm and dropout masks are the
// next increment.
#include "../../include/l2s.h"
#include "l2s_common.h"
#include "l2s_model.h"

#include <algorithm>

namespace l2s {

// ---------------------------------------------------------------------------------------------------------------------
// Backward of a fused GEMM epilogue  y = act(z) [+ residual],  z = conv * s + shift  (s, shift = eval-mode BatchNorm and/or bias):
//   dpre = dy * act'(z);  dconv = dpre * s;  per-column sums  r0 = sum dpre,  r1 = sum dpre * (z - beta)/gamma,  r2 = sum dy * sin(z)
// Two-stage column reduction (row splits -> partials -> final), deterministic.
constexpr int AB_RS = 32;     // row splits

struct ActBwdP {
    const float* dy; const float* z; float* dconv;     // [rows][C]
    int64_t rows; int C;
    int act;                                            // ACT_NONE / ACT_SILU / ACT_PSINE / ACT_RELU
    const float* actw;                                  // psine w
    const float* scale;                                 // BN scale s (null = 1)
    const float* gamma; const float* beta;              // BN affine (null = no BN)
    float* partials;                                    // [AB_RS][3][C]
};

__global__ __launch_bounds__(256) void act_bwd_kernel(const ActBwdP p) {
    __shared__ float sh[3][4][64];
    const int col = blockIdx.x * 64 + (threadIdx.x & 63), rl = threadIdx.x >> 6, rs = blockIdx.y;
    const int64_t chunk = (p.rows + AB_RS - 1) / AB_RS;
    const int64_t r_begin = rs * chunk, r_end = r_begin + chunk < p.rows ? r_begin + chunk : p.rows;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
    if (col < p.C) {
        const float s = p.scale ? p.scale[col] : 1.f;
        const float w = p.act == ACT_PSINE ? p.actw[col] : 0.f;
        const float be = p.beta ? p.beta[col] : 0.f, ig = p.gamma ? 1.f / p.gamma[col] : 0.f;
        for (int64_t r = r_begin + rl; r < r_end; r += 4) {
            const float z = p.z[r * p.C + col], dy = p.dy[r * p.C + col];
            float dpre = dy;
            if (p.act == ACT_PSINE) { dpre = dy * cosf(z) * w; a2 += dy * sinf(z); }
            else if (p.act == ACT_SILU) { const float sg = 1.f / (1.f + expf(-z)); dpre = dy * sg * (1.f + z * (1.f - sg)); }
            else if (p.act == ACT_RELU) { dpre = z > 0.f ? dy : 0.f; }
            a0 += dpre;
            a1 += dpre * (z - be) * ig;
            p.dconv[r * p.C + col] = dpre * s;
        }
    }
    sh[0][rl][threadIdx.x & 63] = a0; sh[1][rl][threadIdx.x & 63] = a1; sh[2][rl][threadIdx.x & 63] = a2;
    __syncthreads();
    if (rl == 0 && col < p.C) {
        const int c = threadIdx.x & 63;
#pragma unroll
        for (int k = 0; k < 3; ++k)
            p.partials[((int64_t)rs * 3 + k) * p.C + col] = (sh[k][0][c] + sh[k][1][c]) + (sh[k][2][c] + sh[k][3][c]);
    }
}

// final stage: out_k[col] (+)= mul_k(col) * sum_rs partial[rs][k][col];   k=0 -> shift-like grad (BN beta or plain bias), conv bias = s * r0
__global__ __launch_bounds__(256) void act_bwd_final_kernel(const float* __restrict__ partials, int C, const float* __restrict__ scale,
                                                            float* __restrict__ d_shift, float* __restrict__ d_gamma, float* __restrict__ d_actw,
                                                            float* __restrict__ d_convbias, int accumulate) {
    const int col = blockIdx.x * 256 + threadIdx.x;
    if (col >= C) return;
    float r[3] = {0.f, 0.f, 0.f};
    for (int rs = 0; rs < AB_RS; ++rs)
#pragma unroll
        for (int k = 0; k < 3; ++k) r[k] += partials[((int64_t)rs * 3 + k) * C + col];
    auto put = [&](float* dst, float v) { if (dst) dst[col] = accumulate ? dst[col] + v : v; };
    put(d_shift, r[0]);
    put(d_gamma, r[1]);
    put(d_actw, r[2]);
    put(d_convbias, r[0] * (scale ? scale[col] : 1.f));
}

static int act_bwd(const ActBwdP& p, float* d_shift, float* d_gamma, float* d_actw, float* d_convbias, bool accumulate, hipStream_t s) {
    ProfScope ps("train_act_bn_bwd", s);
    hipLaunchKernelGGL(act_bwd_kernel, dim3((p.C + 63) / 64, AB_RS), dim3(256), 0, s, p);
    hipLaunchKernelGGL(act_bwd_final_kernel, dim3((p.C + 255) / 256), dim3(256), 0, s, p.partials, p.C, p.scale, d_shift, d_gamma, d_actw, d_convbias,
                       accumulate ? 1 : 0);
    L2S_CHECK_HIP(hipGetLastError());
    return 0;
}

// dWp [co][tap][ci] -> canonical Conv1d gradient (co, ci, tap)
__global__ __launch_bounds__(256) void conv1d_grad_to_canonical_kernel(const float* __restrict__ dwp, int co, int ci, int taps, float* __restrict__ out, int accumulate) {
    const int64_t total = (int64_t)co * ci * taps;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int t = idx % taps;
        const int64_t r = idx / taps;
        const int c = r % ci, n = r / ci;
        const float v = dwp[((int64_t)n * taps + t) * ci + c];
        out[idx] = accumulate ? out[idx] + v : v;
    }
}
static int conv1d_grad_to_canonical(const float* dwp, int co, int ci, int taps, float* out, bool accumulate, hipStream_t s) {
    const int64_t total = (int64_t)co * ci * taps;
    int blocks = (int)std::min<int64_t>((total + 255) / 256, 8192);
    ProfScope ps("train_conv1d_grad_layout", s);
    hipLaunchKernelGGL(conv1d_grad_to_canonical_kernel, dim3(blocks), dim3(256), 0, s, dwp, co, ci, taps, out, accumulate ? 1 : 0);
    L2S_CHECK_HIP(hipGetLastError());
    return 0;
}

// out[i] (+)= a[i] (+ b[i]);  (B,C,S) -> (B,S,C) transposing variant for the channel-first loss gradients
__global__ __launch_bounds__(256) void add_transposed_bcs_kernel(const float* __restrict__ a_bcs, int B, int C, int S, float* __restrict__ out_bsc, int accumulate) {
    const int64_t total = (int64_t)B * S * C;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int c = idx % C;
        const int64_t r = idx / C;
        const int sidx = r % S, b = r / S;
        const float v = a_bcs[((int64_t)b * C + c) * S + sidx];
        out_bsc[idx] = accumulate ? out_bsc[idx] + v : v;
    }
}
static int add_transposed_bcs(const float* a_bcs, int B, int C, int S, float* out_bsc, bool accumulate, hipStream_t s) {
    const int64_t total = (int64_t)B * S * C;
    int blocks = (int)std::min<int64_t>((total + 255) / 256, 8192);
    ProfScope ps("train_transpose_add", s);
    hipLaunchKernelGGL(add_transposed_bcs_kernel, dim3(blocks), dim3(256), 0, s, a_bcs, B, C, S, out_bsc, accumulate ? 1 : 0);
    L2S_CHECK_HIP(hipGetLastError());
    return 0;
}
__global__ __launch_bounds__(256) void axpy_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) y[i] += x[i];
}
static int add_into(const float* x, float* y, int64_t n, hipStream_t s) {
    int blocks = (int)std::min<int64_t>((n + 255) / 256, 8192);
    ProfScope ps("train_add", s);
    hipLaunchKernelGGL(axpy_kernel, dim3(blocks), dim3(256), 0, s, x, y, n);
    L2S_CHECK_HIP(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// Post-net.  Tape: z_l (B*S, C_l) for l = 0..3 (pre-PSine), x_l (layer outputs) for l = 0..3.
struct PostTape { float* z[5]; float* x[4]; };
static int64_t post_tape_floats(int B, int S) { return (int64_t)B * S * 512 * 9 + 64 * 9; }
static PostTape post_tape(float* base, int B, int S) {
    PostTape t;
    const int64_t n = align_up((int64_t)B * S * 512, 64);
    for (int l = 0; l < 4; ++l) { t.z[l] = base + (2 * l) * n; t.x[l] = base + (2 * l + 1) * n; }
    t.z[4] = base + 8 * n;                  // (B*S, 80): output of the last BatchNorm, before the residual
    return t;
}

static int postnet_train_fwd(l2s_model* m, const float* mel, int B, int S, float* tape, float* mel_post, hipStream_t s) {
    const Weights& w = m->w;
    PostTape t = post_tape(tape, B, S);
    for (int l = 0; l < 5; ++l) {
        const float* in = l == 0 ? mel : t.x[l - 1];
        const int cin = l == 0 ? NM_ : 512, cout = l == 4 ? NM_ : 512;
        GemmP p = gemm_plain(in, cin, w.post[l].W, l == 4 ? mel_post : t.x[l], cout, B * S, cout, 5 * cin);
        p.Tout = S; p.Tin = S; p.taps = 5; p.stride = 1; p.pad = 2; p.Cin = cin;
        p.scale = w.post[l].scale; p.shift = w.post[l].shift;
        p.Zout = t.z[l];
        if (l < 4) { p.act = ACT_PSINE; p.actw = w.post[l].actw; }
        if (l >= 1 && l <= 3) { p.R1 = in; p.ldr1 = 512; }
        if (l == 4) { p.R1 = mel; p.ldr1 = NM_; p.c_tr_T = S; }
        if (launch_gemm1(p, s, "train_postnet_conv_gemm")) return 1;
    }
    return 0;
}

// dmel_post (B,80,S) channel-first -> accumulates into dmel (B,S,80); parameter gradients into the bound slots
static int postnet_train_bwd(l2s_model* m, const float* mel, const float* dmel_post_cf, int B, int S, float* tape, float* dmel, void* ws,
                             int64_t ws_bytes, hipStream_t s) {
    const Weights& w = m->w;
    PostTape t = post_tape(tape, B, S);
    const int64_t R = (int64_t)B * S;
    Bump bp(ws, ws_bytes);
    float* g = bp.f(R * 512);          // gradient wrt the current layer's output
    float* gconv = bp.f(R * 512);      // gradient wrt the conv output
    float* gprev = bp.f(R * 512);
    float* dwp = bp.f((int64_t)512 * 5 * 512);
    float* partials = bp.f((int64_t)AB_RS * 3 * 512);
    L2S_REQUIRE(!bp.overflow, "post-net backward workspace too small");
    const std::string P = "decoder.postnet.";
    // mel_post = z4 + mel, channel-first: g4 (B*S,80) = transpose(dmel_post); dmel += g4
    if (add_transposed_bcs(dmel_post_cf, B, NM_, S, g, false, s)) return 1;
    if (add_into(g, dmel, R * NM_, s)) return 1;
    for (int l = 4; l >= 0; --l) {
        const int cin = l == 0 ? NM_ : 512, cout = l == 4 ? NM_ : 512;
        const float* xin = l == 0 ? mel : t.x[l - 1];
        const std::string c = P + "convolutions." + std::to_string(l);
        ActBwdP a{};
        a.dy = g; a.z = t.z[l]; a.dconv = gconv; a.rows = R; a.C = cout;
        a.act = l < 4 ? ACT_PSINE : ACT_NONE; a.actw = w.post[l].actw;
        a.scale = w.post[l].scale; a.gamma = m->canon(c + ".1.weight"); a.beta = m->canon(c + ".1.bias");
        a.partials = partials;
        if (act_bwd(a, m->grad(c + ".1.bias"), m->grad(c + ".1.weight"), l < 4 ? m->grad(P + "sin_activation." + std::to_string(l) + ".w") : nullptr,
                    m->grad(c + ".0.conv.bias"), false, s)) return 1;
        // weight gradient (tap-major) -> canonical layout
        if (launch_gemm_bwd(bwd_dw(gconv, cout, xin, cin, dwp, B, S, S, cout, cin, 5, 1, 2, false), s, "train_postnet_dw")) return 1;
        if (float* gw = m->grad(c + ".0.conv.weight")) { if (conv1d_grad_to_canonical(dwp, cout, cin, 5, gw, false, s)) return 1; }
        // input gradient (+ residual path for layers 1..3)
        float* gin = gprev;
        if (launch_gemm_bwd(bwd_dx(gconv, cout, w.post[l].W, gin, cin, B, S, S, cout, cin, 5, 2, false), s, "train_postnet_dx")) return 1;
        if (l >= 1 && l <= 3) { if (add_into(g, gin, R * 512, s)) return 1; }     // x_l = PSine(..) + x_{l-1}
        if (l == 0) { if (add_into(gin, dmel, R * NM_, s)) return 1; }
        std::swap(g, gprev);
    }
    return 0;
}

}  // namespace l2s

// ================================================================================================ C ABI
using namespace l2s;

extern "C" {

int l2s_train_bind(l2s_model* m, const char* key, float* param_dev, float* grad_dev) {
    L2S_REQUIRE(m && key && param_dev, "bad arguments");
    m->bound[key] = {param_dev, grad_dev};
    return 0;
}

int64_t l2s_train_postnet_tape_floats(int B, int S) { return post_tape_floats(B, S); }
int64_t l2s_train_postnet_ws_bytes(int B, int S) {
    return ((int64_t)B * S * 512 * 3 + (int64_t)512 * 5 * 512 + (int64_t)AB_RS * 3 * 512 + 64 * 8) * (int64_t)sizeof(float);
}

int l2s_train_postnet_fwd(l2s_model* m, const float* mel, int B, int S, float* tape, float* mel_post, void* stream) {
    L2S_REQUIRE(m && m->finalized && m->has_dec && mel && tape && mel_post, "bad arguments");
    return postnet_train_fwd(m, mel, B, S, tape, mel_post, (hipStream_t)stream);
}

int l2s_train_postnet_bwd(l2s_model* m, const float* mel, const float* dmel_post, int B, int S, float* tape, float* dmel, void* ws,
                          int64_t ws_bytes, void* stream) {
    L2S_REQUIRE(m && m->finalized && m->has_dec && mel && dmel_post && tape && dmel && ws, "bad arguments");
    L2S_REQUIRE(m->canon("decoder.postnet.convolutions.0.1.weight") != nullptr, "parameters not bound (l2s_train_bind)");
    return postnet_train_bwd(m, mel, dmel_post, B, S, tape, dmel, ws, ws_bytes, (hipStream_t)stream);
}

}  // extern "C"
