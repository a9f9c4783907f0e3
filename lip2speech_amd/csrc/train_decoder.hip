// Training path of the decoder (reference: Decoder.forward /root/reference/model/modules/decoder.py:320-379 under
// loss.backward(), train.py:184): forward with a tape of the intermediates the backward needs, and the backward itself.
// Three stages, each with its own tape and C-ABI pair: the post-net (decoder.py:107-156), the autoregressive loop with back-propagation
// through time (:353-375) and the prologue (:321-351).  BatchNorm runs on running statistics (eval) or batch statistics (train, see
// l2s_train_set_bn); the dropout sites take their multipliers as inputs.  Gradient parity: tests/test_grad_goldens.py (reference
// goldens, both modes), tests/test_training_pieces.py (per stage against autograd through the oracle).
#include "../../include/l2s.h"
#include "l2s_common.h"
#include "l2s_model.h"

#include <algorithm>

namespace l2s {

// ---------------------------------------------------------------------------------------------------------------------
// Backward of a fused GEMM epilogue  y = act(z) [+ residual],  z = conv * s + shift  (s, shift = eval-mode BatchNorm and/or bias):
//   dpre = dy * act'(z);  dconv = dpre * s;  per-column sums  r0 = sum dpre,  r1 = sum dpre * (z - beta)/gamma,  r2 = sum dy * sin(z)
// Two-stage column reduction (row splits -> partials -> final), deterministic.
__global__ __launch_bounds__(256) void act_bwd_kernel(const ActBwdP p, int nsplit) {
    __shared__ float sh[3][4][64];
    const int col = blockIdx.x * 64 + (threadIdx.x & 63), rl = threadIdx.x >> 6, rs = blockIdx.y;
    const int64_t chunk = (p.rows + nsplit - 1) / nsplit;
    const int64_t r_begin = rs * chunk, r_end = r_begin + chunk < p.rows ? r_begin + chunk : p.rows;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
    const int64_t ldy = p.ld_dy ? p.ld_dy : p.C, ldz = p.ld_z ? p.ld_z : p.C, ldc = p.ld_dconv ? p.ld_dconv : p.C;
    const int csy = p.cs_dy ? p.cs_dy : 1, csz = p.cs_z ? p.cs_z : 1;
    if (col < p.C) {
        const float s = p.scale ? p.scale[col] : 1.f;
        const float w = (p.act == ACT_PSINE || p.act == ACT_PRELU) ? p.actw[col] : 0.f;
        const float be = p.beta ? p.beta[col] : 0.f, ig = p.gamma ? 1.f / p.gamma[col] : 0.f;
        // U rows per trip, their 2U loads issued before the first use: one row per trip was a chain of dependent round trips
        // (28 us for an 8 MB map). Same accumulation order: rows ascending within the thread.
        constexpr int U = 4;
        const float* zp = p.z + p.co_z + (int64_t)col * csz;
        const float* yp = p.dy + p.co_dy + (int64_t)col * csy;
        for (int64_t r0 = r_begin + rl; r0 < r_end; r0 += 4 * U) {
            float zz[U], dd[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t r = r0 + 4 * u < r_end ? r0 + 4 * u : r_begin + rl;
                zz[u] = zp[r * ldz]; dd[u] = yp[r * ldy];
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t r = r0 + 4 * u;
                if (r < r_end) {
                    const float z = zz[u], dy = dd[u];
                    float dpre = dy;
                    if (p.act == ACT_PSINE) { dpre = dy * cosf(z) * w; a2 += dy * sinf(z); }
                    else if (p.act == ACT_SILU) { const float sg = 1.f / (1.f + expf(-z)); dpre = dy * sg * (1.f + z * (1.f - sg)); }
                    else if (p.act == ACT_RELU) { dpre = z > 0.f ? dy : 0.f; }
                    else if (p.act == ACT_PRELU) { dpre = z >= 0.f ? dy : dy * w; a2 += z >= 0.f ? 0.f : dy * z; }
                    a0 += dpre;
                    a1 += dpre * (z - be) * ig;
                    p.dconv[r * ldc + col] = dpre * s;
                }
            }
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
// block = 64 columns x 16 lanes striding over the row splits
__global__ __launch_bounds__(1024) void act_bwd_final_kernel(const float* __restrict__ partials, int nsplit, int C, const float* __restrict__ scale,
                                                            float* __restrict__ d_shift, float* __restrict__ d_gamma, float* __restrict__ d_actw,
                                                            float* __restrict__ d_convbias, int accumulate, float* __restrict__ totals) {
    __shared__ float sh[3][16][64];
    const int cl = threadIdx.x & 63, lane = threadIdx.x >> 6;
    const int col = blockIdx.x * 64 + cl;
    float r[3] = {0.f, 0.f, 0.f};
    if (col < C)
#pragma unroll 4
        for (int rs = lane; rs < nsplit; rs += 16)
#pragma unroll
            for (int k = 0; k < 3; ++k) r[k] += partials[((int64_t)rs * 3 + k) * C + col];
#pragma unroll
    for (int k = 0; k < 3; ++k) sh[k][lane][cl] = r[k];
    __syncthreads();
    if (lane != 0 || col >= C) return;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        float t = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) t += sh[k][q][cl];
        r[k] = t;
    }
    auto put = [&](float* dst, float v) { if (dst) dst[col] = accumulate ? dst[col] + v : v; };
    put(d_shift, r[0]);
    put(d_gamma, r[1]);
    put(d_actw, r[2]);
    put(d_convbias, r[0] * (scale ? scale[col] : 1.f));
    if (totals) { totals[col] = r[0]; totals[C + col] = r[1]; }
}

int act_bwd(const ActBwdP& p, float* d_shift, float* d_gamma, float* d_actw, float* d_convbias, bool accumulate, hipStream_t s, float* totals) {
    ProfScope ps("train_act_bn_bwd", s);
    // row splits: the long, narrow maps of the encoder (24-116 channels = one or two column blocks) need them to fill the chip
    const int nsplit = (int)std::min<int64_t>(AB_RS, std::max<int64_t>(32, (p.rows + 63) / 64));
    hipLaunchKernelGGL(act_bwd_kernel, dim3((p.C + 63) / 64, nsplit), dim3(256), 0, s, p, nsplit);
    hipLaunchKernelGGL(act_bwd_final_kernel, dim3((p.C + 63) / 64), dim3(1024), 0, s, p.partials, nsplit, p.C, p.scale, d_shift, d_gamma, d_actw, d_convbias,
                       accumulate ? 1 : 0, totals);
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
struct PostTape { float* z[5]; float* x[4]; float* bn; float* stats; };    // bn: this batch's (scale, shift) per layer, 2*512 floats each
static int64_t post_tape_floats(int B, int S) { return (int64_t)B * S * 512 * 9 + 5 * 1024 + gemm_stats_floats(B * S, 512) + 64 * 12; }
static PostTape post_tape(float* base, int B, int S) {
    PostTape t;
    const int64_t n = align_up((int64_t)B * S * 512, 64);
    for (int l = 0; l < 4; ++l) { t.z[l] = base + (2 * l) * n; t.x[l] = base + (2 * l + 1) * n; }
    t.z[4] = base + 8 * n;                  // (B*S, 80): output of the last BatchNorm, before the residual
    t.bn = base + 9 * n;
    t.stats = t.bn + 5 * 1024 + 64;
    return t;
}
static BnLayer dec_bn_layer(const l2s_model* m, const std::string& bn_key, const std::string& bias_key, float* slot, int C) {
    BnLayer L{};
    L.gamma = m->canon(bn_key + ".weight"); L.beta = m->canon(bn_key + ".bias");
    L.rmean = const_cast<float*>(m->canon(bn_key + ".running_mean")); L.rvar = const_cast<float*>(m->canon(bn_key + ".running_var"));
    L.conv_bias = bias_key.empty() ? nullptr : m->canon(bias_key);
    L.scale = slot; L.shift = slot + 512; L.C = C;
    return L;
}

// dropout masks of the post-net (decoder.py:152,154), channel-last like the activations: layers 0..3 (B*S,512) each, layer 4 (B*S,80)
static const float* post_mask(const float* drop, int B, int S, int l) { return drop ? drop + (int64_t)l * B * S * 512 : nullptr; }

static int postnet_train_fwd(l2s_model* m, const float* mel, int B, int S, float* tape, float* mel_post, const float* drop, hipStream_t s) {
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
        p.mask = post_mask(drop, B, S, l); p.ldmask = cout; p.mask_pre = l == 4;     // the mel residual is added outside the Postnet module
        if (m->bn_batch) {                     // batch statistics: stats pass of the same conv, then this batch's scale/shift in the fused epilogue
            const std::string c = "decoder.postnet.convolutions." + std::to_string(l);
            GemmP q = p; q.stats = t.stats; q.stats_raw = 1; q.scale = nullptr; q.shift = nullptr;
            if (launch_gemm1(q, s, "train_postnet_conv_stats")) return 1;
            BnLayer L = dec_bn_layer(m, c + ".1", c + ".0.conv.bias", t.bn + l * 1024, cout);
            if (bn_stats_finalize(t.stats, (B * S + 63) / 64, 2 * cout, (int64_t)B * S, L, m->bn_momentum, s)) return 1;
            p.scale = L.scale; p.shift = L.shift;
            GemmBatch fb{}; fb.p[0] = p; fb.count = 1;          // the product is parked at z_l: the epilogue runs over it, not a second conv
            if (launch_gemm_finish(fb, s, "train_postnet_conv_epilogue")) return 1;
            continue;
        }
        if (launch_gemm1(p, s, "train_postnet_conv_gemm")) return 1;
    }
    return 0;
}

// dmel_post (B,80,S) channel-first -> accumulates into dmel (B,S,80); parameter gradients into the bound slots
__global__ __launch_bounds__(256) void mul_inplace_kernel(float* __restrict__ x, const float* __restrict__ m, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) x[i] *= m[i];
}
__global__ __launch_bounds__(256) void mul_out_kernel(float* __restrict__ out, const float* __restrict__ a, const float* __restrict__ m, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) out[i] = a[i] * m[i];
}
static int mul_inplace(float* x, const float* m, int64_t n, hipStream_t s) {
    ProfScope ps("train_dropout_mask", s);
    hipLaunchKernelGGL(mul_inplace_kernel, dim3((unsigned)std::min<int64_t>((n + 255) / 256, 8192)), dim3(256), 0, s, x, m, n);
    L2S_CHECK_HIP(hipGetLastError());
    return 0;
}

static int postnet_train_bwd(l2s_model* m, const float* mel, const float* dmel_post_cf, int B, int S, float* tape, float* dmel, const float* drop, void* ws,
                             int64_t ws_bytes, hipStream_t s) {
    const Weights& w = m->w;
    PostTape t = post_tape(tape, B, S);
    const int64_t R = (int64_t)B * S;
    Bump bp(ws, ws_bytes);
    float* g = bp.f(R * 512);          // gradient wrt the current layer's output
    float* gconv = bp.f(R * 512);      // gradient wrt the conv output
    float* gprev = bp.f(R * 512);
    float* skp = bp.f(5 * R * 512);    // split-K partials of the input gradients (one slice per tap)
    float* dwp = bp.f((int64_t)512 * 5 * 512);
    float* partials = bp.f((int64_t)AB_RS * 3 * 512);
    float* totals = bp.f(1024);
    L2S_REQUIRE(!bp.overflow, "post-net backward workspace too small");
    const bool bnb = m->bn_batch;
    const std::string P = "decoder.postnet.";
    // mel_post = z4 + mel, channel-first: g4 (B*S,80) = transpose(dmel_post); dmel += g4
    if (add_transposed_bcs(dmel_post_cf, B, NM_, S, g, false, s)) return 1;
    if (add_into(g, dmel, R * NM_, s)) return 1;
    for (int l = 4; l >= 0; --l) {
        const int cin = l == 0 ? NM_ : 512, cout = l == 4 ? NM_ : 512;
        const float* xin = l == 0 ? mel : t.x[l - 1];
        const std::string c = P + "convolutions." + std::to_string(l);
        if (drop) { if (mul_inplace(g, post_mask(drop, B, S, l), R * cout, s)) return 1; }     // dropout sits after the residual add (layers 0..3) / after the last conv
        ActBwdP a{};
        a.dy = g; a.z = t.z[l]; a.dconv = gconv; a.rows = R; a.C = cout;
        a.act = l < 4 ? ACT_PSINE : ACT_NONE; a.actw = w.post[l].actw;
        a.scale = bnb ? t.bn + l * 1024 : w.post[l].scale; a.gamma = m->canon(c + ".1.weight"); a.beta = m->canon(c + ".1.bias");
        a.partials = partials;
        if (act_bwd(a, m->grad(c + ".1.bias"), m->grad(c + ".1.weight"), l < 4 ? m->grad(P + "sin_activation." + std::to_string(l) + ".w") : nullptr,
                    bnb ? nullptr : m->grad(c + ".0.conv.bias"), false, s, bnb ? totals : nullptr)) return 1;
        if (bnb) {      // batch statistics: the mean subtraction removes the conv bias (zero gradient) and couples every row of a channel
            if (bn_train_fix(gconv, cout, t.z[l], cout, 1, 0, a.gamma, a.beta, a.scale, totals, R, cout, s)) return 1;
            if (float* gb = m->grad(c + ".0.conv.bias")) { if (launch_fill(gb, cout, 0.f, s)) return 1; }
        }
        // weight gradient (tap-major) -> canonical layout
        if (launch_gemm_bwd(bwd_dw(gconv, cout, xin, cin, dwp, B, S, S, cout, cin, 5, 1, 2, false), s, "train_postnet_dw")) return 1;
        if (float* gw = m->grad(c + ".0.conv.weight")) { if (conv1d_grad_to_canonical(dwp, cout, cin, 5, gw, false, s)) return 1; }
        // input gradient (+ residual path for layers 1..3)
        float* gin = gprev;
        // B*S rows by <= 512 columns is 80 tiles at B = 8 with K = 5 * 512: five K slices
        if (launch_gemm_bwd_splitk(bwd_dx(gconv, cout, w.post[l].W, gin, cin, B, S, S, cout, cin, 5, 2, false), 5, skp, s, "train_postnet_dx")) return 1;
        if (l >= 1 && l <= 3) { if (add_into(g, gin, R * 512, s)) return 1; }     // x_l = PSine(..) + x_{l-1}
        if (l == 0) { if (add_into(gin, dmel, R * NM_, s)) return 1; }
        std::swap(g, gprev);
    }
    return 0;
}

}  // namespace l2s


// =====================================================================================================================
// Stage 2: the autoregressive loop (decoder.py:353-375) - forward with a tape, then back-propagation through time.
// =====================================================================================================================
#include "skinny_dev.h"
#include <cstdlib>
namespace l2s {

constexpr int TR_MAXC = 16;      // K <= 2048 for the backward products with the LSTM gate gradients

struct TrainSkinnyBatch { SkinnyTrain t[SKINNY_MAX_GROUP]; };

__global__ __launch_bounds__(512) void train_skinny_kernel(const SkinnyBatch batch, const TrainSkinnyBatch tb) {
    __shared__ float red[SK_RED_FLOATS];
    const int g = blockIdx.z;
    skinny_block<true, TR_MAXC>(batch.p[g], blockIdx.x, blockIdx.y, red, batch.ntiles[g], &tb.t[g]);
}
// the forward LSTM cells on the inference path's four-wave straight-line blocks (one wave per SIMD, compile-time K layout, exact waits, buffer loads;
// skinny_block_rcs<.., NW = 4, TRAIN>): per output element the arithmetic of the eight-wave general block, operation for operation - the same bits -
// for the launch's two K layouts: [cc | p2 | a@v | h0] (K = 1536, phase-merged layer 0) and [h0' | h1] (K = 1024, layer 1)
template <class LAY>
__global__ __launch_bounds__(256, 1) void train_lstm_rc4_kernel(const SkinnyBatch batch, const TrainSkinnyBatch tb, int mts) {
    __shared__ float red[SkRc<1, 1>::RED_FLOATS];
    skinny_block_rcs<1, 1, LAY, 4, true, false, 4, false, true>(batch.p[0], blockIdx.x, blockIdx.y, red, batch.ntiles[0], mts, nullptr, &tb.t[0]);
}
static int launch_train_skinny(const SkinnyBatch& b, const TrainSkinnyBatch& tb, hipStream_t s, const char* name) {
    static const bool rc4 = !(getenv("L2S_TRAIN_RC4") && getenv("L2S_TRAIN_RC4")[0] == '0');      // L2S_TRAIN_RC4=0: the general eight-wave block (A/B)
    if (rc4 && b.count == 1 && b.p[0].epi == SK_LSTM && !b.p[0].a_sum && !b.p[0].pre) {
        const SkinnyP& p = b.p[0];
        const int n0 = p.seg[0].nchunks, n1 = p.nseg > 1 ? p.seg[1].nchunks : 0, n2 = p.nseg > 2 ? p.seg[2].nchunks : 0, n3 = p.nseg > 3 ? p.seg[3].nchunks : 0;
        const int mts1 = (p.B + 15) / 16;
        SkinnyBatch bl = b;
        for (int j = bl.p[0].nseg; j < 4; ++j) { bl.p[0].seg[j].nchunks = 0; bl.p[0].seg[j].a = bl.p[0].seg[0].a; }
        if (n0 == 16 && n1 == 16 && n2 == 32 && n3 == 32) {
            ProfScope ps(name, s);
            hipLaunchKernelGGL((train_lstm_rc4_kernel<SegLay<16, 16, 32, 32>>), dim3(b.ntiles[0], mts1, 1), dim3(256), 0, s, bl, tb, mts1);
            L2S_CHECK_HIP(hipGetLastError());
            return 0;
        }
        if (n0 == 32 && n1 == 32 && n2 == 0 && n3 == 0) {
            ProfScope ps(name, s);
            hipLaunchKernelGGL((train_lstm_rc4_kernel<SegLay<32, 32, 0, 0>>), dim3(b.ntiles[0], mts1, 1), dim3(256), 0, s, bl, tb, mts1);
            L2S_CHECK_HIP(hipGetLastError());
            return 0;
        }
    }
    int maxt = 0, mts = 0;
    for (int i = 0; i < b.count; ++i) {
        const SkinnyP& p = b.p[i];
        int k = 0;
        for (int j = 0; j < 4; ++j) k += 16 * p.seg[j].nchunks;
        L2S_REQUIRE(k == p.K && p.K % 16 == 0 && p.K <= 16 * SK_WAVES * TR_MAXC, "train skinny K segments");
        maxt = std::max(maxt, b.ntiles[i]);
        mts = (p.B + 15) / 16;
    }
    ProfScope ps(name, s);
    hipLaunchKernelGGL(train_skinny_kernel, dim3(maxt, mts, b.count), dim3(512), 0, s, b, tb);
    L2S_CHECK_HIP(hipGetLastError());
    return 0;
}

struct TrainStepB { AttnP at; AttnTrain att; SkinnyP pre2; SkinnyTrain pre2t; int pre2_tiles; };
__global__ __launch_bounds__(512) void train_step_attn_kernel(const TrainStepB sb) {
    __shared__ __attribute__((aligned(16))) float sm[ATT_SM_FLOATS > SK_RED_FLOATS ? ATT_SM_FLOATS : SK_RED_FLOATS];
    const int nb = sb.at.B, bid = blockIdx.x;
    if (bid < nb) attention_block<true>(sb.at, bid, sm, &sb.att);
    else if (bid < 2 * nb) content_block<true>(sb.at, bid - nb, sm, &sb.att);
    else { const int j = bid - 2 * nb; skinny_block<true, TR_MAXC>(sb.pre2, j % sb.pre2_tiles, j / sb.pre2_tiles, sm, 1 << 30, &sb.pre2t); }
}

// ---- tape of the loop: time-major plain tensors
struct StepTape {
    float *h0, *h1, *c0, *c1;       // (S+1, B, 512): state before step s (s = S: final)
    float *z1, *z2, *zc, *cc, *u;   // (S, B, 256)
    float *zq, *av;                 // (S, B, 512)
    float *alpha;                   // (S, B, 16)
    float *g0, *g1;                 // (S, B, 2048) gates i,f,g,o after their nonlinearities
    float *yprev;                   // (S, B, 96)  frame fed to the prenet at step s (cols 80.. zero)
};
static int64_t step_tape_floats(int B, int S) {
    return (int64_t)(S + 1) * B * 512 * 4 + (int64_t)S * B * (256 * 5 + 512 * 2 + 16 + 2048 * 2 + 96) + 64 * 20;
}
static StepTape step_tape(float* base, int B, int S) {
    StepTape t;
    int64_t o = 0;
    auto take = [&](int64_t n) { float* r = base + o; o += align_up(n, 64); return r; };
    const int64_t SB = (int64_t)S * B, S1B = (int64_t)(S + 1) * B;
    t.h0 = take(S1B * 512); t.h1 = take(S1B * 512); t.c0 = take(S1B * 512); t.c1 = take(S1B * 512);
    t.z1 = take(SB * 256); t.z2 = take(SB * 256); t.zc = take(SB * 256); t.cc = take(SB * 256); t.u = take(SB * 256);
    t.zq = take(SB * 512); t.av = take(SB * 512); t.alpha = take(SB * 16); t.g0 = take(SB * 2048); t.g1 = take(SB * 2048);
    t.yprev = take(SB * 96);
    return t;
}

// yprev[s][b][n] = s == 0 ? BOS[n] : (forced[s] ? teacher[b][s][n] : mel[b][s-1][n]);  teacher[b][0] = BOS by construction
__global__ __launch_bounds__(256) void build_yprev_kernel(const float* __restrict__ mel, const float* __restrict__ teacher, const unsigned char* __restrict__ mask,
                                                          const float* __restrict__ bos, int B, int S, float* __restrict__ yprev) {
    const int64_t total = (int64_t)S * B * 96;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int n = idx % 96;
        const int64_t r = idx / 96;
        const int b = r % B, sidx = r / B;
        float v = 0.f;
        if (n < 80) {
            if (sidx == 0) v = bos[n];
            else if (teacher && mask && mask[sidx]) v = teacher[((int64_t)b * S + sidx) * 80 + n];
            else v = mel[((int64_t)b * S + sidx - 1) * 80 + n];
        }
        yprev[idx] = v;
    }
}

// ---- backward elementwise kernels (B <= 96 rows; one thread per element)
// LSTM cell backward: dh = dh_a (+ dh_b); dc = dc_carry + dh*o*(1-tanh(c_new)^2); pre-activation gate gradients in canonical order
__global__ __launch_bounds__(256) void lstm_bwd_kernel(const float* __restrict__ dh_a, int ld_a, const float* __restrict__ dh_b, int ld_b,
                                                       float* __restrict__ dc_carry, const float* __restrict__ gates, const float* __restrict__ c_prev,
                                                       const float* __restrict__ c_new, int B, int H, float* __restrict__ dg_frag, float* __restrict__ dg_stack,
                                                       float* __restrict__ dg_stack2 = nullptr, int64_t ld_stack2_b = 0,
                                                       const float* __restrict__ mask_b = nullptr) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= B * H) return;
    const int b = idx / H, u = idx - b * H;
    float dh = dh_a[(int64_t)b * ld_a + u];
    if (dh_b) dh += dh_b[(int64_t)b * ld_b + u] * (mask_b ? mask_b[idx] : 1.f);     // mask_b: inter-layer dropout on the path into the next layer
    const float* g = gates + (int64_t)b * 4 * H + u;
    const float gi = g[0], gf = g[H], gg = g[2 * H], go = g[3 * H];
    const float tc = tanhf(c_new[idx]);
    const float dc = dc_carry[idx] + dh * go * (1.f - tc * tc);
    const float dzi = dc * gg * gi * (1.f - gi);
    const float dzf = dc * c_prev[idx] * gf * (1.f - gf);
    const float dzg = dc * gi * (1.f - gg * gg);
    const float dzo = dh * tc * go * (1.f - go);
    dc_carry[idx] = dc * gf;
    const float vals[4] = {dzi, dzf, dzg, dzo};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        dg_frag[frag16_index(b, k * H + u, 4 * H)] = vals[k];
        dg_stack[(int64_t)b * 4 * H + k * H + u] = vals[k];
        if (dg_stack2) dg_stack2[(int64_t)b * ld_stack2_b + k * H + u] = vals[k];
    }
}

// loss gradient of every frame: dmel[:, s] with the stop-logit gradient in column 80, as one frag16 (K = 96) per step and in the (S,B,96) stack
__global__ __launch_bounds__(256) void build_dy_all_kernel(const float* __restrict__ dmel, const float* __restrict__ dstop, int B, int S, float* __restrict__ frags,
                                                           float* __restrict__ stack) {
    const int Bp = (B + 15) & ~15;
    const int64_t total = (int64_t)S * Bp * 96;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int n = idx % 96; const int64_t q = idx / 96;
        const int b = q % Bp, sidx = q / Bp;
        float v = 0.f;
        if (b < B) {
            if (n < 80) v = dmel[((int64_t)b * S + sidx) * 80 + n];
            else if (n == 80) v = dstop[(int64_t)b * S + sidx];
            stack[((int64_t)sidx * B + b) * 96 + n] = v;
        }
        frags[(int64_t)sidx * Bp * 96 + frag16_index(b, n, 96)] = v;
    }
}
// the carries join the stack after the loop: stack[s][b][:80] += dyc[s+1][b][:] where step s+1 took its own previous output as input
struct CarryFlags { uint8_t on[ATT_MAXT]; };
__global__ __launch_bounds__(256) void add_carry_kernel(float* __restrict__ stack, const float* __restrict__ dyc, CarryFlags f, int B, int S) {
    const int64_t total = (int64_t)(S - 1) * B * 80;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int n = idx % 80; const int64_t q = idx / 80;
        const int b = q % B, sidx = q / B;
        if (f.on[sidx]) stack[((int64_t)sidx * B + b) * 96 + n] += dyc[((int64_t)(sidx + 1) * B + b) * 80 + n];
    }
}
// ---- attention + content attention backward, one 512-thread block per batch row (decoder.py:414-419, 262-271)
struct AttnBwdP {
    const float* dav; int ld_dav;     // [B][ld_dav] (0 = 512)
    const float* dcc; int ld_dcc;     // [B][ld]
    const float* logits; int64_t ld_logit_b;   // logits of this step (after dropout): [b*ld + t]
    const float* lmask; int ld_lmask;          // dropout multiplier of the logits [b*ld + t] or null
    const float* k; const float* v;   // [B][T][512]
    const float* zq; const float* wq; const float* pos; const float* tau;
    const float* alpha;               // [B][16]
    const float* ckey; const float* cval; const float* zc; const float* tau_c;
    float* dk; float* dv; float* dckey; float* dcval;         // accumulated over steps
    float* dzq_frag; float* dq_stack;                         // frag16 K=512; [B][512]
    float* dzc_frag; float* dqc_stack;                        // frag16 K=256; [B][256]
    float* dtau_part; float* dtauc_part;                      // [B] partial sums of this step
    int B, T, m;
};

__global__ __launch_bounds__(512) void attn_bwd_kernel(const AttnBwdP p) {
    __shared__ float s_dav[512], s_a[ATT_MAXT], s_dl[ATT_MAXT], s_red[16], s_dcc[256], s_qc[256], s_dlc[16], s_al[16];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int T = p.T, m = p.m;
    const float tau = p.tau[0], tau_c = p.tau_c[0];
    const float* kb = p.k + (int64_t)b * T * 512;
    const float* vb = p.v + (int64_t)b * T * 512;
    s_dav[tid] = p.dav[(int64_t)b * (p.ld_dav ? p.ld_dav : 512) + tid];
    // softmax of the stored logits
    const bool on = tid < T;
    const float l = on ? p.logits[(int64_t)b * p.ld_logit_b + tid] : -INFINITY;
    float mx = wave_max_f(l);
    __syncthreads();
    if (lane == 0) s_red[wave] = mx;
    __syncthreads();
    mx = s_red[0];
    for (int i = 1; i < 8; ++i) mx = fmaxf(mx, s_red[i]);
    const float ex = on ? expf(l - mx) : 0.f;
    float sm = wave_sum_f(ex);
    __syncthreads();
    if (lane == 0) s_red[wave] = sm;
    __syncthreads();
    sm = 0.f;
    for (int i = 0; i < 8; ++i) sm += s_red[i];
    if (on) s_a[tid] = ex / sm;
    __syncthreads();
    // da[t] = dav . v[t]  (wave per row)
    for (int t = wave; t < T; t += 8) {
        float acc = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) acc = fmaf(s_dav[lane * 8 + e], vb[(int64_t)t * 512 + lane * 8 + e], acc);
        acc = wave_sum_f(acc);
        if (lane == 0) s_dl[t] = acc;                    // holds da for now
    }
    __syncthreads();
    float part = on ? s_a[tid] * s_dl[tid] : 0.f;        // sum_t a*da
    part = wave_sum_f(part);
    __syncthreads();
    if (lane == 0) s_red[wave] = part;
    __syncthreads();
    float sad = 0.f;
    for (int i = 0; i < 8; ++i) sad += s_red[i];
    __syncthreads();
    float dlt = 0.f;
    if (on) { dlt = s_a[tid] * (s_dl[tid] - sad); }
    __syncthreads();
    if (on) s_dl[tid] = dlt;
    // d tau: sum_t dl_t * (l_t / tau)
    float pt = on ? dlt * (l / tau) : 0.f;
    pt = wave_sum_f(pt);
    __syncthreads();
    if (lane == 0) s_red[wave] = pt;
    __syncthreads();
    if (tid == 0) { float tt = 0.f; for (int i = 0; i < 8; ++i) tt += s_red[i]; p.dtau_part[b] = tt; }
    // stored logit L = mask * tau q.k: the sum above is already d tau (mask*L_raw = L); q and k see the masked gradient
    if (on && p.lmask) s_dl[tid] = dlt * p.lmask[(int64_t)b * p.ld_lmask + tid];
    __syncthreads();
    // per column c = tid: q, dq, dk, dv
    {
        const float zq = p.zq[(int64_t)b * 512 + tid], wq = p.wq[tid];
        const float q = sinf(zq) * wq + p.pos[tid];
        const float dav = s_dav[tid];
        float dq = 0.f;
        float* dkb = p.dk + (int64_t)b * T * 512 + tid;
        float* dvb = p.dv + (int64_t)b * T * 512 + tid;
        // dk / dv accumulate over the steps in HBM: a read-modify-write per (t, column). Eight frames per trip with all 24 loads
        // issued before the first store - one frame at a time was a chain of 29 dependent round trips (23 us per launch)
        constexpr int TC = 8;
        for (int t0 = 0; t0 < T; t0 += TC) {
            float kk[TC], dkv[TC], dvv[TC];
#pragma unroll
            for (int u = 0; u < TC; ++u) {
                const int64_t o = (int64_t)min(t0 + u, T - 1) * 512;
                kk[u] = kb[o + tid]; dkv[u] = dkb[o]; dvv[u] = dvb[o];
            }
#pragma unroll
            for (int u = 0; u < TC; ++u) {
                const int t = t0 + u;
                if (t < T) {
                    const float dl_t = s_dl[t];
                    dq = fmaf(dl_t, kk[u], dq);
                    dkb[(int64_t)t * 512] = dkv[u] + tau * dl_t * q;
                    dvb[(int64_t)t * 512] = dvv[u] + s_a[t] * dav;
                }
            }
        }
        dq *= tau;
        p.dq_stack[(int64_t)b * 512 + tid] = dq;
        p.dzq_frag[frag16_index(b, tid, 512)] = dq * cosf(zq) * wq;
    }
    // ---- content attention
    __syncthreads();
    if (tid < 256) {
        s_dcc[tid] = p.dcc[(int64_t)b * p.ld_dcc + tid];
        const float zc = p.zc[(int64_t)b * 256 + tid];
        s_qc[tid] = zc / (1.f + expf(-zc));
    }
    if (tid < 16) s_al[tid] = tid < m ? p.alpha[(int64_t)b * 16 + tid] : 0.f;
    __syncthreads();
    const float* keyb = p.ckey + (int64_t)b * m * 256;
    const float* valb = p.cval + (int64_t)b * m * 256;
    __shared__ float s_dal[16], s_dot[16];
    for (int j = wave; j < m; j += 8) {                  // d alpha_j = dcc . value_j ; dot_j = qc . key_j
        float a0 = 0.f, a1 = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            a0 = fmaf(s_dcc[lane * 4 + e], valb[(int64_t)j * 256 + lane * 4 + e], a0);
            a1 = fmaf(s_qc[lane * 4 + e], keyb[(int64_t)j * 256 + lane * 4 + e], a1);
        }
        a0 = wave_sum_f(a0); a1 = wave_sum_f(a1);
        if (lane == 0) { s_dal[j] = a0; s_dot[j] = a1; }
    }
    __syncthreads();
    if (tid == 0) {
        float sa = 0.f;
        for (int j = 0; j < m; ++j) sa += s_al[j] * s_dal[j];
        float dtc = 0.f;
        for (int j = 0; j < m; ++j) { const float d = s_al[j] * (s_dal[j] - sa); s_dlc[j] = d; dtc += d * s_dot[j]; }
        p.dtauc_part[b] = dtc;
    }
    __syncthreads();
    if (tid < 256) {
        float dqc = 0.f;
        constexpr int MC = 4;
        for (int j0 = 0; j0 < m; j0 += MC) {                 // same read-modify-write pattern: loads first
            float kk[MC], dkv[MC], dvv[MC];
#pragma unroll
            for (int u = 0; u < MC; ++u) {
                const int64_t o = ((int64_t)b * m + min(j0 + u, m - 1)) * 256 + tid;
                kk[u] = keyb[(int64_t)min(j0 + u, m - 1) * 256 + tid]; dkv[u] = p.dckey[o]; dvv[u] = p.dcval[o];
            }
#pragma unroll
            for (int u = 0; u < MC; ++u) {
                const int j = j0 + u;
                if (j < m) {
                    dqc = fmaf(s_dlc[j], kk[u], dqc);
                    p.dckey[((int64_t)b * m + j) * 256 + tid] = dkv[u] + tau_c * s_dlc[j] * s_qc[tid];
                    p.dcval[((int64_t)b * m + j) * 256 + tid] = dvv[u] + s_al[j] * s_dcc[tid];
                }
            }
        }
        dqc *= tau_c;
        p.dqc_stack[(int64_t)b * 256 + tid] = dqc;
        const float zc = p.zc[(int64_t)b * 256 + tid], sg = 1.f / (1.f + expf(-zc));
        p.dzc_frag[frag16_index(b, tid, 256)] = dqc * sg * (1.f + zc * (1.f - sg));
    }
}

// transposed frag16 pack of up to two canonical source blocks: F[n'][k'] = src[(k'-k_lo)*ld + (n'-n_lo)]
struct PackSeg { const float* src; int ld, n_lo, n_hi, k_lo, k_hi; };
__global__ __launch_bounds__(256) void pack_fragT_kernel(float* __restrict__ dst, int Npad, int K, PackSeg s0, PackSeg s1, PackSeg s2) {
    const int64_t total = (int64_t)Npad * K;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int k = idx % K, n = idx / K;
        float v = 0.f;
        const PackSeg* segs[3] = {&s0, &s1, &s2};
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const PackSeg& sg = *segs[q];
            if (sg.src && n >= sg.n_lo && n < sg.n_hi && k >= sg.k_lo && k < sg.k_hi) v = sg.src[(int64_t)(k - sg.k_lo) * sg.ld + (n - sg.n_lo)];
        }
        dst[frag16_index(n, k, K)] = v;
    }
}
static int pack_fragT(float* dst, int Npad, int K, PackSeg s0, PackSeg s1, PackSeg s2, hipStream_t s) {
    const int64_t total = (int64_t)Npad * K;
    int blocks = (int)std::min<int64_t>((total + 255) / 256, 8192);
    ProfScope ps("train_pack_transposed_weights", s);
    hipLaunchKernelGGL(pack_fragT_kernel, dim3(blocks), dim3(256), 0, s, dst, Npad, K, s0, s1, s2);
    L2S_CHECK_HIP(hipGetLastError());
    return 0;
}

// column sums of a (rows x C) stack, two-stage (reuses the act_bwd machinery with the identity activation)
static int colsum(const float* x, int64_t rows, int C, float* partials, float* scratch_dconv, float* out, bool accumulate, hipStream_t s) {
    ActBwdP a{};
    a.dy = x; a.z = x; a.dconv = scratch_dconv; a.rows = rows; a.C = C; a.act = ACT_NONE; a.partials = partials;
    return act_bwd(a, out, nullptr, nullptr, nullptr, accumulate, s);
}

}  // namespace l2s

namespace l2s {

static SkinnyP tsk(const SkW& sw, int B) {
    SkinnyP p{};
    p.W = sw.W; p.bias = sw.bias; p.actw = sw.actw; p.B = B; p.N = sw.N; p.K = sw.K; p.act = ACT_NONE; p.epi = SK_PLAIN;
    return p;
}

__global__ __launch_bounds__(256) void psine_fwd_kernel(const float* __restrict__ z, const float* __restrict__ w, int64_t rows, int C, float* __restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < rows * C; i += (int64_t)gridDim.x * 256) out[i] = sinf(z[i]) * w[i % C];
}

// ---- forward of the loop with the tape (the literal 6-phase step; eval-mode statistics, no dropout)
static int64_t step_fwd_ws_floats(int B) { return (int64_t)pad16(B) * (512 * 8 + 256 * 4 + 96) + (int64_t)B * (512 + 256 + 256) + 64 * 20; }

// dropout multipliers of the loop (train mode; any may be null): prenet (S,B,256) after the first PSine (decoder.py:308), attention
// logits (S,B,T) (:363), inter-layer LSTM dropout (S,B,512) on h0 as the input of layer 1 (:312)
struct StepDrop { const float* prenet; const float* attn; const float* rnn; };

static int decode_train_fwd(l2s_model* m, float* state, int B, int T, int S, const float* teacher, const uint8_t* mask, const uint8_t* mask_dev,
                            float* tape_base, float* mel, float* stop, float* attn_logits, StepDrop drop, void* ws, int64_t ws_bytes, hipStream_t s) {
    const Weights& w = m->w;
    StateLayout sl = state_layout(B, T);
    StepTape tp = step_tape(tape_base, B, S);
    const int Bp = pad16(B);
    Bump bp(ws, ws_bytes);
    float* h0[2] = {bp.f((int64_t)Bp * 512), bp.f((int64_t)Bp * 512)};
    float* h1[2] = {bp.f((int64_t)Bp * 512), bp.f((int64_t)Bp * 512)};
    float* c0 = bp.f((int64_t)Bp * 512); float* c1 = bp.f((int64_t)Bp * 512); float* av = bp.f((int64_t)Bp * 512);
    float* p1 = bp.f((int64_t)Bp * 256); float* cc = bp.f((int64_t)Bp * 256); float* uu = bp.f((int64_t)Bp * 256);
    float* yf = bp.f((int64_t)Bp * 96);
    float* q = bp.f((int64_t)B * 512); float* qc = bp.f((int64_t)B * 256); float* p2 = bp.f((int64_t)B * 256);
    float* h0d = bp.f((int64_t)Bp * 512);
    float* p2f = bp.f((int64_t)Bp * 256);
    L2S_REQUIRE(!bp.overflow, "training decode workspace too small");
    // The phase-merged step of inference (prenet1 o fc_out over h1, attention_proj folded into LSTM0's input weights) is used
    // here too when the merged matrices are valid (packed by the host, or re-merged on the device by l2s_train_refresh_weights): 4 launches per
    // step instead of 6.  The tape is the same (z1, z2, zq, zc, alpha, a@v, gates, cells, hidden states); u = attention_proj(a@v) + prenet is
    // only needed by the parameter gradients and is rebuilt for all steps at once after the loop.
    const bool fold = m->opt.fold != 0 && m->folded_valid && w.pre1f.W && w.lstm0f.W;
    // every buffer above starts at zero (padded rows of the fragments must not hold NaN garbage): ONE fill over the whole run, then the initial state
    if (launch_fill(h0[0], bp.off / (int64_t)sizeof(float), 0.f, s)) return 1;
    L2S_CHECK_HIP(hipMemcpyAsync(h0[0], state + sl.h, sizeof(float) * Bp * 512, hipMemcpyDeviceToDevice, s));
    L2S_CHECK_HIP(hipMemcpyAsync(h1[0], state + sl.h + (int64_t)Bp * 512, sizeof(float) * Bp * 512, hipMemcpyDeviceToDevice, s));
    if (launch_to_frag(w.bos, 0, B, 80, yf, 80, 0, 1, s)) return 1;
    // tape row 0 of the state sequences
    if (launch_from_frag(h0[0], 512, B, 512, tp.h0, 512, 0, s)) return 1;
    if (launch_from_frag(h1[0], 512, B, 512, tp.h1, 512, 0, s)) return 1;
    if (launch_fill(tp.c0, (int64_t)B * 512, 0.f, s)) return 1;
    if (launch_fill(tp.c1, (int64_t)B * 512, 0.f, s)) return 1;

    for (int i = 0; i < S; ++i) {
        const int cur = i & 1, nxt = cur ^ 1;
        const int64_t r256 = (int64_t)i * B * 256, r512 = (int64_t)i * B * 512;
        if (teacher && mask && mask[i])
            if (launch_to_frag(teacher + (int64_t)i * NM_, S * NM_, B, 80, yf, 80, 0, 0, s)) return 1;
        {
            SkinnyBatch sb{}; TrainSkinnyBatch tb{};
            const bool from_frame = !fold || i == 0 || (teacher && mask && mask[i]);      // prenet input is an explicit frame (BOS / teacher / unfolded y)
            SkinnyP a = tsk(from_frame ? w.pre1 : w.pre1f, B);
            if (from_frame) a.seg[0] = {yf, 5}; else a.seg[0] = {h1[cur], 32};
            a.nseg = 1; a.act = ACT_PSINE; a.epi = SK_FRAG; a.out = p1; a.ldo = 256;
            tb.t[0].zsave = tp.z1 + r256; tb.t[0].ld_z = 256;
            if (drop.prenet) { tb.t[0].out_mask = drop.prenet + r256; tb.t[0].ld_mask = 256; }
            SkinnyP b = tsk(w.q, B);
            b.seg[0] = {h0[cur], 32}; b.seg[1] = {h1[cur], 32}; b.nseg = 2; b.act = ACT_PSINE; b.out = q; b.ldo = 512;
            b.addrow = w.pos + (int64_t)i * 512;
            tb.t[1].zsave = tp.zq + r512; tb.t[1].ld_z = 512;
            SkinnyP c = tsk(w.cq, B);
            c.seg[0] = {c0, 32}; c.seg[1] = {c1, 32}; c.nseg = 2; c.act = ACT_SILU; c.out = qc; c.ldo = 256;
            tb.t[2].zsave = tp.zc + r256; tb.t[2].ld_z = 256;
            sb.p[0] = a; sb.ntiles[0] = 16; sb.p[1] = b; sb.ntiles[1] = w.q.tiles; sb.p[2] = c; sb.ntiles[2] = w.cq.tiles; sb.count = 3;
            if (fold && i > 0) {          // mel frame + stop logit of step i-1 ride in this launch
                SkinnyP f = tsk(w.fc, B);
                f.seg[0] = {h1[cur], 32}; f.nseg = 1; f.epi = SK_MEL;
                f.mel = mel + (int64_t)(i - 1) * NM_; f.ld_mel_b = (int64_t)S * NM_; f.stop = stop + (i - 1); f.ld_stop_b = S;
                f.stop_const = state + sl.stopc; f.yfrag = nullptr;
                sb.p[3] = f; sb.ntiles[3] = w.fc.tiles; sb.count = 4;
            }
            if (launch_train_skinny(sb, tb, s, fold ? "train_step_prenet1_q_cq_fc" : "train_step_prenet1_q_cq")) return 1;
        }
        {
            TrainStepB sb{};
            AttnP& at = sb.at;
            at.q = q; at.ldq = 512; at.k = state + sl.k; at.v = state + sl.v; at.tau = w.tau; at.av_frag = av;
            at.attn_out = attn_logits + (int64_t)i * T; at.ld_attn_b = (int64_t)S * T; at.attn_logits = 1;
            at.qc = qc; at.ldqc = 256; at.ckey = state + sl.ckey; at.cval = state + sl.cval; at.tau_c = w.tau_c; at.cc_frag = cc;
            at.B = B; at.T = T; at.m = sl.m;
            if (drop.attn) { sb.att.logit_mask = drop.attn + (int64_t)i * B * T; sb.att.ld_lmask = T; }
            sb.att.alpha = tp.alpha + (int64_t)i * B * 16; sb.att.ld_alpha = 16; sb.att.av_plain = tp.av + r512; sb.att.cc_plain = tp.cc + r256;
            sb.pre2 = tsk(w.pre2, B);
            sb.pre2.seg[0] = {p1, 16}; sb.pre2.nseg = 1; sb.pre2.act = ACT_PSINE; sb.pre2.out = p2; sb.pre2.ldo = 256;
            if (fold) { sb.pre2.epi = SK_FRAG; sb.pre2.out = p2f; }
            sb.pre2t.zsave = tp.z2 + r256; sb.pre2t.ld_z = 256;
            sb.pre2_tiles = w.pre2.tiles;
            ProfScope ps("train_step_attention_prenet2", s);
            hipLaunchKernelGGL(train_step_attn_kernel, dim3(2 * B + w.pre2.tiles * (Bp / 16)), dim3(512), 0, s, sb);
            L2S_CHECK_HIP(hipGetLastError());
        }
        if (!fold) {
            SkinnyBatch sb{}; TrainSkinnyBatch tb{};
            SkinnyP a = tsk(w.aproj, B);
            a.seg[0] = {av, 32}; a.nseg = 1; a.epi = SK_FRAG; a.out = uu; a.ldo = 256; a.add = p2; a.ld_add = 256;
            tb.t[0].out_plain = tp.u + r256; tb.t[0].ld_out = 256;
            sb.p[0] = a; sb.ntiles[0] = w.aproj.tiles; sb.count = 1;
            if (launch_train_skinny(sb, tb, s, "train_step_attention_proj")) return 1;
        }
        for (int layer = 0; layer < 2; ++layer) {
            SkinnyBatch sb{}; TrainSkinnyBatch tb{};
            SkinnyP a = tsk(layer == 0 ? (fold ? w.lstm0f : w.lstm0) : w.lstm1, B);
            if (layer == 0 && fold) { a.seg[0] = {cc, 16}; a.seg[1] = {p2f, 16}; a.seg[2] = {av, 32}; a.seg[3] = {h0[cur], 32}; a.nseg = 4; }
            else if (layer == 0) { a.seg[0] = {cc, 16}; a.seg[1] = {uu, 16}; a.seg[2] = {h0[cur], 32}; a.nseg = 3; }
            else { a.seg[0] = {drop.rnn ? h0d : h0[nxt], 32}; a.seg[1] = {h1[cur], 32}; a.nseg = 2; }
            a.epi = SK_LSTM; a.H = 512;
            if (layer == 0 && drop.rnn) { tb.t[0].h_drop = h0d; tb.t[0].h_drop_K = 512; tb.t[0].h_mask = drop.rnn + r512; tb.t[0].ld_hmask = 512; }
            a.c_in = layer == 0 ? c0 : c1; a.c_out = a.c_in == c0 ? c0 : c1;
            a.h_out = layer == 0 ? h0[nxt] : h1[nxt]; a.h_out_K = 512; a.h_out_off = 0;
            a.h_seq = (layer == 0 ? tp.h0 : tp.h1) + (int64_t)(i + 1) * B * 512; a.ld_hseq = 512;
            tb.t[0].gates = (layer == 0 ? tp.g0 : tp.g1) + (int64_t)i * B * 2048; tb.t[0].ld_gates = 2048;
            tb.t[0].c_new = (layer == 0 ? tp.c0 : tp.c1) + (int64_t)(i + 1) * B * 512; tb.t[0].ld_c = 512;
            sb.p[0] = a; sb.ntiles[0] = 128; sb.count = 1;
            if (launch_train_skinny(sb, tb, s, "train_step_lstm_cell")) return 1;
        }
        if (!fold || i == S - 1) {
            SkinnyBatch sb{}; TrainSkinnyBatch tb{};
            SkinnyP a = tsk(w.fc, B);
            a.seg[0] = {h1[nxt], 32}; a.nseg = 1; a.epi = SK_MEL;
            a.mel = mel + (int64_t)i * NM_; a.ld_mel_b = (int64_t)S * NM_; a.stop = stop + i; a.ld_stop_b = S;
            a.stop_const = state + sl.stopc; a.yfrag = yf;
            sb.p[0] = a; sb.ntiles[0] = w.fc.tiles; sb.count = 1;
            if (launch_train_skinny(sb, tb, s, "train_step_fc_out_stop")) return 1;
        }
    }
    if (fold) {      // u = attention_proj(a@v) + prenet for all S*B rows: PSine(z2) first, then the product with the prenet as its addend, in place
        const float* wap = m->canon("decoder.attention_proj.linear_layer.weight"); const float* bap = m->canon("decoder.attention_proj.linear_layer.bias");
        const float* w2 = m->canon("decoder.prenet.4.w");
        L2S_REQUIRE(wap && bap && w2, "decoder parameters not bound");
        const int64_t SB = (int64_t)S * B;
        hipLaunchKernelGGL(psine_fwd_kernel, dim3((unsigned)std::min<int64_t>((SB * 256 + 255) / 256, 4096)), dim3(256), 0, s, tp.z2, w2, SB, 256, tp.u);
        GemmP g = gemm_plain(tp.av, 512, wap, tp.u, 256, (int)SB, 256, 512);
        g.shift = bap; g.R1 = tp.u; g.ldr1 = 256;
        if (launch_gemm1(g, s, "train_rebuild_u")) return 1;
    }
    {
        const int64_t total = (int64_t)S * B * 96;
        ProfScope ps("train_build_yprev", s);
        hipLaunchKernelGGL(build_yprev_kernel, dim3((int)std::min<int64_t>((total + 255) / 256, 4096)), dim3(256), 0, s, mel, teacher, mask_dev, w.bos, B, S, tp.yprev);
        L2S_CHECK_HIP(hipGetLastError());
    }
    return 0;
}

// ---- transposed step weights for the backward products, packed on the device from the canonical parameters
struct TrainW { float *fc, *l1, *l0, *ap, *q, *cq, *p2, *p1, *bhh[2], *fc4, *prod_ap, *prod_p1, *fcp; };     // prod_ap: W_ih_l0[:,256:512] @ W_ap (2048 x 512); prod_p1: W_p1 @ W_out (256 x 512); fcp: [fc_out | stop | prod_p1]^T, K = 96 + 256
static int64_t train_w_floats() { return (int64_t)512 * 96 + (int64_t)256 * 512 + (int64_t)512 * 352 + 2 * (int64_t)1024 * 2048 + 2 * (int64_t)512 * 2048 + (int64_t)512 * 256 + (int64_t)1024 * 512 + (int64_t)1024 * 256 + 256 * 256 + 80 * 256 + 2 * (int64_t)512 * 2048 + 504 * 256 + 64 * 14; }
static TrainW train_w(float* base) {
    TrainW t; int64_t o = 0;
    auto take = [&](int64_t n) { float* r = base + o; o += align_up(n, 64); return r; };
    t.fc = take((int64_t)512 * 96); t.l1 = take((int64_t)1024 * 2048); t.l0 = take((int64_t)1536 * 2048); t.ap = take((int64_t)512 * 256);
    t.prod_ap = take((int64_t)2048 * 512); t.prod_p1 = take((int64_t)256 * 512); t.fcp = take((int64_t)512 * 352);
    t.q = take((int64_t)1024 * 512); t.cq = take((int64_t)1024 * 256); t.p2 = take(256 * 256); t.p1 = take(80 * 256);      // p1: unused slot (kept for the layout)
    t.bhh[0] = take((int64_t)512 * 2048); t.bhh[1] = take((int64_t)512 * 2048); t.fc4 = take(504 * 256);
    return t;
}
static int pack_train_weights(l2s_model* m, float* wbuf, hipStream_t s) {
    TrainW t = train_w(wbuf);
    const std::string D = "decoder.";
    auto P = [&](const std::string& k) { return m->canon(D + k); };
    const PackSeg none{nullptr, 0, 0, 0, 0, 0};
    L2S_REQUIRE(P("fc_out.linear_layer.weight") && P("decoder_rnn.weight_ih_l0") && P("prenet.0.linear_layer.weight"), "decoder parameters not bound");
    if (pack_fragT(t.fc, 512, 96, PackSeg{P("fc_out.linear_layer.weight"), 512, 0, 512, 0, 80}, PackSeg{P("stop_token_layer.linear_layer.weight"), 1024, 0, 512, 80, 81}, none, s)) return 1;
    if (pack_fragT(t.l1, 1024, 2048, PackSeg{P("decoder_rnn.weight_ih_l1"), 512, 0, 512, 0, 2048}, PackSeg{P("decoder_rnn.weight_hh_l1"), 512, 512, 1024, 0, 2048}, none, s)) return 1;
    // layer 0's transposed block also yields d(a@v) directly: its last 512 output columns are (W_ih[:,256:512] W_ap)^T, so the attention_proj
    // backward product does not need a launch of its own in the loop (d(u) still leaves as columns 256..511 for the parameter gradients)
    L2S_REQUIRE(P("attention_proj.linear_layer.weight"), "decoder parameters not bound");
    if (launch_gemm_bwd(bwd_dx(P("decoder_rnn.weight_ih_l0") + 256, 512, P("attention_proj.linear_layer.weight"), t.prod_ap, 512, 1, 2048, 2048, 256, 512, 1, 0, false), s,
                        "train_merge_step_weights")) return 1;
    if (pack_fragT(t.l0, 1536, 2048, PackSeg{P("decoder_rnn.weight_ih_l0"), 512, 0, 512, 0, 2048}, PackSeg{P("decoder_rnn.weight_hh_l0"), 512, 512, 1024, 0, 2048},
                   PackSeg{t.prod_ap, 512, 1024, 1536, 0, 2048}, s)) return 1;
    if (pack_fragT(t.q, 1024, 512, PackSeg{P("Q.0.linear_layer.weight"), 1024, 0, 1024, 0, 512}, none, none, s)) return 1;
    if (pack_fragT(t.cq, 1024, 256, PackSeg{P("content.Q.0.weight"), 1024, 0, 1024, 0, 256}, none, none, s)) return 1;
    if (pack_fragT(t.p2, 256, 256, PackSeg{P("prenet.3.linear_layer.weight"), 256, 0, 256, 0, 256}, none, none, s)) return 1;
    // the carry of a frame gradient into the previous step, d(z1) -> prenet layer 1 -> fc_out -> d(h1), as ONE product with the loss gradient of
    // that step: K = [96 loss columns | 256 d(z1) columns] against [fc_out ; stop ; W_p1 W_out]^T - the prenet-1 backward product leaves the loop
    if (launch_gemm_bwd(bwd_dx(P("prenet.0.linear_layer.weight"), 80, P("fc_out.linear_layer.weight"), t.prod_p1, 512, 1, 256, 256, 80, 512, 1, 0, false), s,
                        "train_merge_step_weights")) return 1;
    if (pack_fragT(t.fcp, 512, 352, PackSeg{P("fc_out.linear_layer.weight"), 512, 0, 512, 0, 80}, PackSeg{P("stop_token_layer.linear_layer.weight"), 1024, 0, 512, 80, 81},
                   PackSeg{t.prod_p1, 512, 0, 512, 96, 352}, s)) return 1;
    if (P("encoder_rnn.weight_hh_l0") && P("content.location_fc.4.weight")) {     // prologue backward operands
        if (pack_fragT(t.bhh[0], 512, 2048, PackSeg{P("encoder_rnn.weight_hh_l0"), 512, 0, 512, 0, 2048}, none, none, s)) return 1;
        if (pack_fragT(t.bhh[1], 512, 2048, PackSeg{P("encoder_rnn.weight_hh_l0_reverse"), 512, 0, 512, 0, 2048}, none, none, s)) return 1;
        if (launch_fill(t.fc4, 504 * 256, 0.f, s)) return 1;
        L2S_CHECK_HIP(hipMemcpyAsync(t.fc4, P("content.location_fc.4.weight"), sizeof(float) * 501 * 256, hipMemcpyDeviceToDevice, s));
    }
    return 0;
}

// small helpers of the backward loop
static SkinnyP bsk(const float* Wfrag, int N, int K, int B, const float* afrag, float* out, int ldo, const float* add = nullptr, int ld_add = 0) {
    SkinnyP p{};
    p.W = Wfrag; p.B = B; p.N = N; p.K = K; p.act = ACT_NONE; p.epi = SK_PLAIN;
    p.seg[0] = {afrag, K / 16}; p.nseg = 1; p.out = out; p.ldo = ldo; p.add = add; p.ld_add = ld_add;
    return p;
}
static int run1(const SkinnyP& p, hipStream_t s, const char* name) {
    SkinnyBatch sb{}; TrainSkinnyBatch tb{};
    sb.p[0] = p; sb.ntiles[0] = (p.N + 15) / 16; sb.count = 1;
    return launch_train_skinny(sb, tb, s, name);
}
static int run1t(const SkinnyP& p, const SkinnyTrain& t, hipStream_t s, const char* name) {
    SkinnyBatch sb{}; TrainSkinnyBatch tb{};
    sb.p[0] = p; sb.ntiles[0] = (p.N + 15) / 16; sb.count = 1;
    tb.t[0] = t;
    return launch_train_skinny(sb, tb, s, name);
}
static int ew(int n) { return (n + 255) / 256; }

// de_c[b][j] = (sum_i dstop[b][i]) * ws[512 + j];  d ws[512 + j] = sum_b (sum_i dstop[b][i]) * e_c[b][j]
__global__ __launch_bounds__(512) void stop_tail_bwd_kernel(const float* __restrict__ dstop, int B, int S, const float* __restrict__ ws, const float* __restrict__ ecell,
                                                            float* __restrict__ de_c, float* __restrict__ dws_tail) {
    __shared__ float sb[128];
    const int j = threadIdx.x;
    for (int b = j; b < B; b += 512) { float a = 0.f; for (int i = 0; i < S; ++i) a += dstop[(int64_t)b * S + i]; sb[b] = a; }
    __syncthreads();
    float g = 0.f;
    for (int b = 0; b < B; ++b) { de_c[(int64_t)b * 512 + j] = sb[b] * ws[512 + j]; g += sb[b] * ecell[(int64_t)b * 512 + j]; }
    if (dws_tail) dws_tail[j] = g;
}
__global__ void sum_small_kernel(const float* __restrict__ x, int n, float* __restrict__ out) {   // one thread: n <= a few thousand
    if (blockIdx.x == 0 && threadIdx.x == 0) { double a = 0.0; for (int i = 0; i < n; ++i) a += x[i]; out[0] = (float)a; }
}
__global__ __launch_bounds__(256) void copy_rows_kernel(const float* __restrict__ src, int ld_src, float* __restrict__ dst, int ld_dst, int rows, int cols) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx < rows * cols) dst[(int64_t)(idx / cols) * ld_dst + idx % cols] = src[(int64_t)(idx / cols) * ld_src + idx % cols];
}

static int64_t step_bwd_ws_floats(int B, int S) {
    const int64_t Bp = pad16(B), SB = (int64_t)S * B;
    return SB * (96 + 2048 * 3 + 256 * 4 + 512 + 2 + 256 + 256) + (int64_t)S * Bp * 96 + Bp * (96 + 2048 * 2 + 256 * 4 + 512) + (int64_t)B * (512 * 6 + 1024 * 6 + 256 + 80) +
           (int64_t)AB_RS * 3 * 2048 + (int64_t)96 * 512 + 4096 + 64 * 60;
}

// Back-propagation through the S steps.  Inputs: dmel (B,S,80) total gradient of the pre-postnet frames, dstop (B,S).
// Outputs: parameter gradients (bound slots), dk/dv (B,T,512), dckey/dcval (B,m,256), dh_init (2,B,512), de_c (B,512).
static int decode_train_bwd(l2s_model* m, float* state, int B, int T, int S, const uint8_t* mask, float* tape_base, const float* attn_logits,
                            const float* dmel, const float* dstop, float* wbuf, float* dk, float* dv, float* dckey, float* dcval, float* dh_init,
                            float* de_c, StepDrop drop, void* ws, int64_t ws_bytes, hipStream_t s) {
    const Weights& w = m->w;
    StateLayout sl = state_layout(B, T);
    StepTape tp = step_tape(tape_base, B, S);
    TrainW tw = train_w(wbuf);
    const int Bp = pad16(B);
    const int64_t SB = (int64_t)S * B;
    Bump bp(ws, ws_bytes);
    // stacks (time-major)
    float* st_dyt = bp.f(SB * 96); float* st_dg1 = bp.f(SB * 2048); float* st_dg0 = bp.f(SB * 2048); float* st_du = bp.f(SB * 256);
    float* st_dq = bp.f(SB * 512); float* st_dqc = bp.f(SB * 256); float* st_dp1 = bp.f(SB * 256); float* st_dtau = bp.f(SB); float* st_dtauc = bp.f(SB);
    float* st_tmp = bp.f(SB * 2048); float* st_p1 = bp.f(SB * 256);
    // per-step fragments and plain buffers
    float* f_dyl = bp.f((int64_t)S * Bp * 96); float* st_dz1 = bp.f(SB * 256); float* f_dg1 = bp.f((int64_t)Bp * 2048); float* f_dg0 = bp.f((int64_t)Bp * 2048);
    float* f_dzq = bp.f((int64_t)Bp * 512); float* f_dzc = bp.f((int64_t)Bp * 256);
    float* f_dz2 = bp.f((int64_t)Bp * 256); float* f_dz1 = bp.f((int64_t)Bp * 256);
    float* dh1lin = bp.f((int64_t)B * 512); float* d01 = bp.f((int64_t)B * 1536); float* d0x = bp.f((int64_t)B * 1536);      // row pitch 1536 both: [dcc 256 | du 256 | dh0 512 | d(a@v) 512]; d01 uses 1024 of it
    float* dp1 = bp.f((int64_t)B * 256);
    // carries into the previous step, both layers side by side: (B,1024) = [layer 0 | layer 1], row pitch 1024
    float* dhc = bp.f((int64_t)B * 1024); float* dcc = bp.f((int64_t)B * 1024);
    float* dh0c = dhc; float* dh1c = dhc + 512; float* dc0c = dcc; float* dc1c = dcc + 512;
    float* partials = bp.f((int64_t)AB_RS * 3 * 2048); float* tmp96 = bp.f((int64_t)96 * 512); float* small = bp.f(4096);
    L2S_REQUIRE(!bp.overflow, "training backward workspace too small");
    for (float* z : {dhc, dcc}) if (launch_fill(z, (int64_t)B * 1024, 0.f, s)) return 1;
    if (launch_fill(dk, (int64_t)B * T * 512, 0.f, s) || launch_fill(dv, (int64_t)B * T * 512, 0.f, s)) return 1;
    if (launch_fill(dckey, (int64_t)B * sl.m * 256, 0.f, s) || launch_fill(dcval, (int64_t)B * sl.m * 256, 0.f, s)) return 1;
    const std::string D = "decoder.";
    const float* wq = m->canon(D + "Q.1.w"); const float* w1 = m->canon(D + "prenet.1.w"); const float* w2 = m->canon(D + "prenet.4.w");
    L2S_REQUIRE(wq && w1 && w2, "decoder parameters not bound");

    // Five launches per step, each a product whose epilogue also does the elementwise work that follows it: the LSTM-cell backwards ride on the
    // products that yield their dh, d(a@v), d(u) and prenet layer 2's PSine backward on the layer-0 product, the carries into step i-1 and prenet
    // layer 1's PSine backward on the Q / content-Q / prenet-2 launch; the step's first product takes the loss gradient of frame i AND the carry
    // d(z1) of step i+1 as two K segments (were 10 launches at the start of round 2).
    hipLaunchKernelGGL(build_dy_all_kernel, dim3(ew(S * Bp * 96)), dim3(256), 0, s, dmel, dstop, B, S, f_dyl, st_dyt);
    for (int i = S - 1; i >= 0; --i) {
        const int64_t r256 = (int64_t)i * B * 256, r512 = (int64_t)i * B * 512, r2048 = (int64_t)i * B * 2048;
        {
            SkinnyTrain t{};
            t.lb_gates = tp.g1 + r2048; t.lb_cprev = tp.c1 + r512; t.lb_cnew = tp.c1 + r512 + (int64_t)B * 512; t.lb_dc = dc1c; t.lb_ld_dc = 1024;
            t.lb_frag = f_dg1; t.lb_stack = st_dg1 + r2048; t.lb_H = 512;
            const float* dyl = f_dyl + (int64_t)i * Bp * 96;
            const bool carry = i < S - 1 && !(mask && mask[i + 1]);         // step i+1 consumed this step's own output
            SkinnyP p = carry ? bsk(tw.fcp, 512, 352, B, dyl, dh1lin, 512, dh1c, 1024) : bsk(tw.fc, 512, 96, B, dyl, dh1lin, 512, dh1c, 1024);
            if (carry) { p.seg[0] = {dyl, 6}; p.seg[1] = {f_dz1, 16}; p.nseg = 2; }
            if (run1t(p, t, s, "train_bwd_fc")) return 1;
        }
        {
            SkinnyTrain t{};
            t.lb_gates = tp.g0 + r2048; t.lb_cprev = tp.c0 + r512; t.lb_cnew = tp.c0 + r512 + (int64_t)B * 512; t.lb_dc = dc0c; t.lb_ld_dc = 1024;
            t.lb_dha = dh0c; t.lb_ld_a = 1024; t.lb_mask = drop.rnn ? drop.rnn + r512 : nullptr;
            t.lb_frag = f_dg0; t.lb_stack = st_dg0 + r2048; t.lb_H = 512;
            if (run1t(bsk(tw.l1, 1024, 2048, B, f_dg1, d01, 1536), t, s, "train_bwd_lstm_dx")) return 1;
        }
        {
            SkinnyTrain t{};                // columns 256..511 of d0x = d(u): fragment + stack, and through prenet layer 2's PSine
            t.sd_lo = 256; t.sd_hi = 512; t.sd_z = tp.z2 + r256; t.sd_w = w2; t.sd_stack = st_du + r256; t.sd_frag_dz = f_dz2;
            if (run1t(bsk(tw.l0, 1536, 2048, B, f_dg0, d0x, 1536), t, s, "train_bwd_lstm_dx")) return 1;
        }
        {
            AttnBwdP a{};
            a.dav = d0x + 1024; a.ld_dav = 1536; a.dcc = d0x; a.ld_dcc = 1536; a.logits = attn_logits + (int64_t)i * T; a.ld_logit_b = (int64_t)S * T;
            if (drop.attn) { a.lmask = drop.attn + (int64_t)i * B * T; a.ld_lmask = T; }
            a.k = state + sl.k; a.v = state + sl.v; a.zq = tp.zq + r512; a.wq = wq; a.pos = w.pos + (int64_t)i * 512; a.tau = w.tau;
            a.alpha = tp.alpha + (int64_t)i * B * 16; a.ckey = state + sl.ckey; a.cval = state + sl.cval; a.zc = tp.zc + r256; a.tau_c = w.tau_c;
            a.dk = dk; a.dv = dv; a.dckey = dckey; a.dcval = dcval; a.dzq_frag = f_dzq; a.dq_stack = st_dq + r512; a.dzc_frag = f_dzc;
            a.dqc_stack = st_dqc + r256; a.dtau_part = st_dtau + (int64_t)i * B; a.dtauc_part = st_dtauc + (int64_t)i * B;
            a.B = B; a.T = T; a.m = sl.m;
            hipLaunchKernelGGL(attn_bwd_kernel, dim3(B), dim3(512), 0, s, a);
        }
        {
            SkinnyBatch sb{}; TrainSkinnyBatch tb{};
            // dh carries: [d0x[:,512:] | d01[:,512:]] + d(h) through Q; dc carries += d(c) through the content query; dp1 through prenet-1's PSine
            sb.p[0] = bsk(tw.q, 1024, 512, B, f_dzq, dhc, 1024, d0x + 512, 1536); sb.ntiles[0] = 64;
            tb.t[0].add_hi = d01; tb.t[0].add_hi_from = 512;
            sb.p[1] = bsk(tw.cq, 1024, 256, B, f_dzc, dcc, 1024, dcc, 1024); sb.ntiles[1] = 64;
            sb.p[2] = bsk(tw.p2, 256, 256, B, f_dz2, dp1, 256); sb.ntiles[2] = 16;
            tb.t[2].sd_lo = 0; tb.t[2].sd_hi = 256; tb.t[2].sd_z = tp.z1 + r256; tb.t[2].sd_w = w1; tb.t[2].sd_mask = drop.prenet ? drop.prenet + r256 : nullptr;
            tb.t[2].sd_stack = st_dp1 + r256; tb.t[2].sd_frag_dz = f_dz1; tb.t[2].sd_stack_dz = st_dz1 + r256;
            sb.count = 3;
            if (launch_train_skinny(sb, tb, s, "train_bwd_q_cq_prenet2")) return 1;
        }
        L2S_CHECK_HIP(hipGetLastError());
    }
    // ---- state gradients
    hipLaunchKernelGGL(copy_rows_kernel, dim3(ew(B * 512)), dim3(256), 0, s, dh0c, 1024, dh_init, 512, B, 512);
    hipLaunchKernelGGL(copy_rows_kernel, dim3(ew(B * 512)), dim3(256), 0, s, dh1c, 1024, dh_init + (int64_t)B * 512, 512, B, 512);
    const float* ws_can = m->canon(D + "stop_token_layer.linear_layer.weight");
    float* g_ws = m->grad(D + "stop_token_layer.linear_layer.weight");
    hipLaunchKernelGGL(stop_tail_bwd_kernel, dim3(1), dim3(512), 0, s, dstop, B, S, ws_can, state + sl.ecell, de_c, g_ws ? g_ws + 512 : nullptr);
    // ---- the carries join the frame-gradient stack: dyc = d(z1) @ W_p1 for all steps at once, added where the next step was not teacher-forced
    {
        const float* wp1 = m->canon(D + "prenet.0.linear_layer.weight");
        L2S_REQUIRE(wp1, "decoder parameters not bound");
        if (launch_gemm_bwd(bwd_dx(st_dz1, 256, wp1, st_tmp, 80, 1, (int)SB, (int)SB, 256, 80, 1, 0, false), s, "train_bwd_prenet1_all")) return 1;
        CarryFlags cf{};
        L2S_REQUIRE(S <= ATT_MAXT, "too many decode steps");
        for (int i = 0; i + 1 < S; ++i) cf.on[i] = (mask && mask[i + 1]) ? 0 : 1;
        if (S > 1) hipLaunchKernelGGL(add_carry_kernel, dim3(ew((S - 1) * B * 80)), dim3(256), 0, s, st_dyt, st_tmp, cf, B, S);
        // BOS is the frame of step 0: its gradient is the first B rows of the same product, summed over the batch
        if (float* g = m->grad(D + "BOS")) { if (colsum(st_tmp, B, 80, partials, st_tmp + SB * 96, g, false, s)) return 1; }
    }
    // ---- parameter gradients from the stacks
    auto dw = [&](const float* dz, int ldz, int nout, const float* x, int ldx, int cin, float* out, int ldc) -> int {
        if (!out) return 0;
        BwdGemmP p = bwd_dw(dz, ldz, x, ldx, out, 1, (int)SB, (int)SB, nout, cin, 1, 1, 0, false);
        p.ldc = ldc;
        return launch_gemm_bwd(p, s, "train_bwd_step_dw");
    };
    const float* h0prev = tp.h0; const float* h0new = tp.h0 + (int64_t)B * 512; const float* h1prev = tp.h1; const float* h1new = tp.h1 + (int64_t)B * 512;
    // fc_out + stop (h1 half): (96 x 512) product, rows 0..79 -> fc_out.weight, row 80 -> stop weight[:512]
    if (dw(st_dyt, 96, 96, h1new, 512, 512, tmp96, 512)) return 1;
    if (float* g = m->grad(D + "fc_out.linear_layer.weight")) hipLaunchKernelGGL(copy_rows_kernel, dim3(ew(80 * 512)), dim3(256), 0, s, tmp96, 512, g, 512, 80, 512);
    if (g_ws) hipLaunchKernelGGL(copy_rows_kernel, dim3(ew(512)), dim3(256), 0, s, tmp96 + 80 * 512, 512, g_ws, 512, 1, 512);
    if (colsum(st_dyt, SB, 96, partials, st_tmp, small, false, s)) return 1;
    if (float* g = m->grad(D + "fc_out.linear_layer.bias")) hipLaunchKernelGGL(copy_rows_kernel, dim3(1), dim3(256), 0, s, small, 96, g, 80, 1, 80);
    if (float* g = m->grad(D + "stop_token_layer.linear_layer.bias")) hipLaunchKernelGGL(copy_rows_kernel, dim3(1), dim3(256), 0, s, small + 80, 96, g, 1, 1, 1);
    // LSTM layers
    const float* l1_in = h0new;
    if (drop.rnn) {                                     // layer 1 saw the dropped h0
        hipLaunchKernelGGL(mul_out_kernel, dim3(4096), dim3(256), 0, s, st_tmp, h0new, drop.rnn, SB * 512);
        l1_in = st_tmp;
    }
    if (dw(st_dg1, 2048, 2048, l1_in, 512, 512, m->grad(D + "decoder_rnn.weight_ih_l1"), 512)) return 1;
    if (dw(st_dg1, 2048, 2048, h1prev, 512, 512, m->grad(D + "decoder_rnn.weight_hh_l1"), 512)) return 1;
    if (colsum(st_dg1, SB, 2048, partials, st_tmp, small, false, s)) return 1;
    for (const char* k : {"decoder_rnn.bias_ih_l1", "decoder_rnn.bias_hh_l1"})
        if (float* g = m->grad(D + k)) hipLaunchKernelGGL(copy_rows_kernel, dim3(ew(2048)), dim3(256), 0, s, small, 2048, g, 2048, 1, 2048);
    if (float* g = m->grad(D + "decoder_rnn.weight_ih_l0")) {
        if (dw(st_dg0, 2048, 2048, tp.cc, 256, 256, g, 512)) return 1;
        if (dw(st_dg0, 2048, 2048, tp.u, 256, 256, g + 256, 512)) return 1;
    }
    if (dw(st_dg0, 2048, 2048, h0prev, 512, 512, m->grad(D + "decoder_rnn.weight_hh_l0"), 512)) return 1;
    if (colsum(st_dg0, SB, 2048, partials, st_tmp, small, false, s)) return 1;
    for (const char* k : {"decoder_rnn.bias_ih_l0", "decoder_rnn.bias_hh_l0"})
        if (float* g = m->grad(D + k)) hipLaunchKernelGGL(copy_rows_kernel, dim3(ew(2048)), dim3(256), 0, s, small, 2048, g, 2048, 1, 2048);
    // attention_proj
    if (dw(st_du, 256, 256, tp.av, 512, 512, m->grad(D + "attention_proj.linear_layer.weight"), 512)) return 1;
    if (colsum(st_du, SB, 256, partials, st_tmp, m->grad(D + "attention_proj.linear_layer.bias") ? m->grad(D + "attention_proj.linear_layer.bias") : small, false, s)) return 1;
    // Q (PSine): dz stack + bias + w gradients, then the weight
    {
        ActBwdP a{}; a.dy = st_dq; a.z = tp.zq; a.dconv = st_tmp; a.rows = SB; a.C = 512; a.act = ACT_PSINE; a.actw = wq; a.partials = partials;
        if (act_bwd(a, m->grad(D + "Q.0.linear_layer.bias"), nullptr, m->grad(D + "Q.1.w"), nullptr, false, s)) return 1;
        if (float* g = m->grad(D + "Q.0.linear_layer.weight")) {
            if (dw(st_tmp, 512, 512, h0prev, 512, 512, g, 1024)) return 1;
            if (dw(st_tmp, 512, 512, h1prev, 512, 512, g + 512, 1024)) return 1;
        }
    }
    {   // content Q (SiLU)
        ActBwdP a{}; a.dy = st_dqc; a.z = tp.zc; a.dconv = st_tmp; a.rows = SB; a.C = 256; a.act = ACT_SILU; a.partials = partials;
        if (act_bwd(a, m->grad(D + "content.Q.0.bias"), nullptr, nullptr, nullptr, false, s)) return 1;
        if (float* g = m->grad(D + "content.Q.0.weight")) {
            if (dw(st_tmp, 256, 256, tp.c0, 512, 512, g, 1024)) return 1;
            if (dw(st_tmp, 256, 256, tp.c1, 512, 512, g + 512, 1024)) return 1;
        }
    }
    {   // prenet layer 2: input p1 = PSine(z1)
        hipLaunchKernelGGL(psine_fwd_kernel, dim3(2048), dim3(256), 0, s, tp.z1, w1, SB, 256, st_p1);
        if (drop.prenet) { if (mul_inplace(st_p1, drop.prenet, SB * 256, s)) return 1; }
        ActBwdP a{}; a.dy = st_du; a.z = tp.z2; a.dconv = st_tmp; a.rows = SB; a.C = 256; a.act = ACT_PSINE; a.actw = w2; a.partials = partials;
        if (act_bwd(a, m->grad(D + "prenet.3.linear_layer.bias"), nullptr, m->grad(D + "prenet.4.w"), nullptr, false, s)) return 1;
        if (dw(st_tmp, 256, 256, st_p1, 256, 256, m->grad(D + "prenet.3.linear_layer.weight"), 256)) return 1;
    }
    {   // prenet layer 1: input = the frame fed at each step
        ActBwdP a{}; a.dy = st_dp1; a.z = tp.z1; a.dconv = st_tmp; a.rows = SB; a.C = 256; a.act = ACT_PSINE; a.actw = w1; a.partials = partials;
        if (act_bwd(a, m->grad(D + "prenet.0.linear_layer.bias"), nullptr, m->grad(D + "prenet.1.w"), nullptr, false, s)) return 1;
        if (dw(st_tmp, 256, 256, tp.yprev, 96, 80, m->grad(D + "prenet.0.linear_layer.weight"), 80)) return 1;
    }
    if (float* g = m->grad(D + "temperature")) hipLaunchKernelGGL(sum_small_kernel, dim3(1), dim3(1), 0, s, st_dtau, (int)SB, g);
    if (float* g = m->grad(D + "content.temperature")) hipLaunchKernelGGL(sum_small_kernel, dim3(1), dim3(1), 0, s, st_dtauc, (int)SB, g);
    L2S_CHECK_HIP(hipGetLastError());
    return 0;
}

}  // namespace l2s

// =====================================================================================================================
// Stage 3: the decoder prologue (decoder.py:321-351): forward with a tape, and its backward down to the visual features.
// =====================================================================================================================
namespace l2s {

struct ProTape {
    float *resid, *z_se, *s_e, *z_sa, *s_a;                 // (BT,512), (B,512) x 4
    float *gin;                                             // (BT,4096) BiLSTM input gates (both directions)
    float *rnn;                                             // (B,T,1024) BiLSTM outputs
    float *gates[2], *cproc[2], *hproc[2];                  // per direction, in processing order: (T,B,2048), (T+1,B,512), (T+1,B,512)
    float *cellcat;                                         // (B,1024)
    float *cat;                                             // (BT,4608) [x | K branches | V branches] (post-activation)
    float *zcat;                                            // (BT,4608): pre-SiLU (post-BN) of the 8 MultiHop branches, same columns as cat
    float *zkv[2];                                          // (BT,512) pre-PSine bottleneck outputs
    float *zagg[4], *cmap[4];                               // (B*L_j,512)
    float *pooled, *wv, *zk0, *tA, *zk2, *zf0, *tB, *zf2, *tC, *zf4, *logits, *zsoft, *dis;
    float* bn;                                              // batch-statistics BatchNorm: (scale, shift) of the 8 MultiHop + 4 content branches, 1024 floats each
    float* stats; int64_t stats_group;                      // statistics scratch: one region of stats_group floats per grouped conv
    int L[4], m;
};
static int64_t pro_tape_floats(int B, int T) {
    int L[4]; const int m = content_lens(T, L);
    const int64_t BT = (int64_t)B * T, R = (int64_t)B * m;
    int64_t n = BT * (512 + 4096 + 1024 + 4608 * 2 + 512 * 2) + (int64_t)B * 512 * 4 + 2 * ((int64_t)T * B * 2048 + 2 * (int64_t)(T + 1) * B * 512) + (int64_t)B * 1024;
    for (int j = 0; j < 4; ++j) n += 2 * (int64_t)B * L[j] * 512;
    n += R * (2560 + 256 * 8 + 504 * 4);
    n += 12 * 1024 + 8 * align_up(gemm_stats_floats((int)BT, 512), 64);
    return n + 64 * 64;
}
static ProTape pro_tape(float* base, int B, int T) {
    ProTape t; int64_t o = 0;
    auto take = [&](int64_t n) { float* r = base + o; o += align_up(n, 64); return r; };
    t.m = content_lens(T, t.L);
    const int64_t BT = (int64_t)B * T, R = (int64_t)B * t.m;
    t.resid = take(BT * 512); t.z_se = take((int64_t)B * 512); t.s_e = take((int64_t)B * 512); t.z_sa = take((int64_t)B * 512); t.s_a = take((int64_t)B * 512);
    t.gin = take(BT * 4096); t.rnn = take(BT * 1024);
    for (int d = 0; d < 2; ++d) { t.gates[d] = take((int64_t)T * B * 2048); t.cproc[d] = take((int64_t)(T + 1) * B * 512); t.hproc[d] = take((int64_t)(T + 1) * B * 512); }
    t.cellcat = take((int64_t)B * 1024); t.cat = take(BT * 4608);
    t.zcat = take(BT * 4608);
    for (int j = 0; j < 2; ++j) t.zkv[j] = take(BT * 512);
    for (int j = 0; j < 4; ++j) { t.zagg[j] = take((int64_t)B * t.L[j] * 512); t.cmap[j] = take((int64_t)B * t.L[j] * 512); }
    t.pooled = take(R * 2560); t.wv = take(R * 256); t.zk0 = take(R * 256); t.tA = take(R * 256); t.zk2 = take(R * 256);
    t.zf0 = take(R * 256); t.tB = take(R * 256); t.zf2 = take(R * 256); t.tC = take(R * 256); t.zf4 = take(R * 504); t.logits = take(R * 504); t.zsoft = take(R * 504); t.dis = take(R * 504);
    t.bn = take(12 * 1024);
    t.stats_group = align_up(gemm_stats_floats((int)BT, 512), 64);
    t.stats = take(8 * t.stats_group);
    return t;
}

static GemmP tconv(const float* X, int lda, int B, int Tin, int Cin, const ConvW& c, int Cout, int taps, int stride, int pad, float* out, int ldc, int act, float* zout) {
    const int Tout = (Tin + 2 * pad - taps) / stride + 1;
    GemmP p = gemm_plain(X, lda, c.W, out, ldc, B * Tout, Cout, taps * Cin);
    p.Tout = Tout; p.Tin = Tin; p.taps = taps; p.stride = stride; p.pad = pad; p.Cin = Cin;
    p.scale = c.scale; p.shift = c.shift; p.actw = c.actw; p.act = act; p.Zout = zout;
    return p;
}
static GemmP tlin(const float* A, int lda, const ConvW& c, float* out, int ldc, int M, int N, int K, int act, float* zout) {
    GemmP p = gemm_plain(A, lda, c.W, out, ldc, M, N, K);
    p.shift = c.shift; p.actw = c.actw; p.act = act; p.Zout = zout;
    return p;
}

static int64_t pro_fwd_ws_floats(int B) { return (int64_t)pad16(B) * 512 * 6 + 64 * 8; }

static int prologue_train_fwd(l2s_model* m, const float* vis, const float* emb, const float* gumbel, int B, int T, float* state, float* content_dis,
                              float* tape_base, void* ws, int64_t ws_bytes, hipStream_t s) {
    const Weights& w = m->w;
    StateLayout sl = state_layout(B, T);
    ProTape tp = pro_tape(tape_base, B, T);
    L2S_REQUIRE(T >= 7 && T <= L2S_MAX_STEPS, "T must be in [7, 300]");
    const int BT = B * T, Bp = pad16(B), R = B * tp.m;
    Bump bp(ws, ws_bytes);
    float* hf[2][2]; float* cf[2];
    for (int d = 0; d < 2; ++d) { hf[d][0] = bp.f((int64_t)Bp * 512); hf[d][1] = bp.f((int64_t)Bp * 512); cf[d] = bp.f((int64_t)Bp * 512); }
    L2S_REQUIRE(!bp.overflow, "training prologue workspace too small");
    {
        GemmP p = gemm_plain(vis, 1024, w.resid.W, tp.resid, 512, BT, 512, 1024); p.shift = w.resid.shift;
        if (launch_gemm1(p, s, "train_prologue_gemm")) return 1;
        GemmBatch gb{};
        gb.p[0] = tlin(emb, 256, w.enc_site, tp.s_e, 512, B, 512, 256, ACT_PSINE, tp.z_se);
        gb.p[1] = tlin(emb, 256, w.attn_site, tp.s_a, 512, B, 512, 256, ACT_PSINE, tp.z_sa);
        gb.count = 2;
        if (launch_gemm(gb, s, "train_prologue_gemm")) return 1;
        GemmP g = gemm_plain(vis, 1024, w.wih_cat, tp.gin, 4096, BT, 4096, 1024); g.shift = w.bih_cat;
        if (launch_gemm1(g, s, "train_bilstm_input_gemm")) return 1;
    }
    for (int d = 0; d < 2; ++d) {
        if (launch_to_frag(tp.s_e, 512, B, 512, hf[d][0], 512, 0, 0, s)) return 1;
        if (launch_to_frag(tp.s_e, 512, B, 512, cf[d], 512, 0, 0, s)) return 1;
        if (launch_fill(hf[d][1], (int64_t)Bp * 512, 0.f, s)) return 1;
        L2S_CHECK_HIP(hipMemcpyAsync(tp.hproc[d], tp.s_e, sizeof(float) * B * 512, hipMemcpyDeviceToDevice, s));
        L2S_CHECK_HIP(hipMemcpyAsync(tp.cproc[d], tp.s_e, sizeof(float) * B * 512, hipMemcpyDeviceToDevice, s));
    }
    for (int step = 0; step < T; ++step) {
        SkinnyBatch sb{}; TrainSkinnyBatch tb{};
        const int cur = step & 1, nxt = cur ^ 1;
        for (int d = 0; d < 2; ++d) {
            const int t = d == 0 ? step : T - 1 - step;
            SkinnyP p = tsk(w.whh[d], B);
            p.seg[0] = {hf[d][cur], 32}; p.nseg = 1; p.epi = SK_LSTM; p.H = 512;
            p.pre = tp.gin + (int64_t)t * 4096 + d * 2048; p.ld_pre = (int64_t)T * 4096;
            p.c_in = cf[d]; p.c_out = cf[d]; p.h_out = hf[d][nxt]; p.h_out_K = 512; p.h_out_off = 0;
            p.h_seq = tp.rnn + (int64_t)t * 1024 + d * 512; p.ld_hseq = (int64_t)T * 1024;
            p.h_plain = tp.hproc[d] + (int64_t)(step + 1) * B * 512; p.ld_hplain = 512;
            tb.t[d].gates = tp.gates[d] + (int64_t)step * B * 2048; tb.t[d].ld_gates = 2048;
            tb.t[d].c_new = tp.cproc[d] + (int64_t)(step + 1) * B * 512; tb.t[d].ld_c = 512;
            sb.p[d] = p; sb.ntiles[d] = w.whh[d].tiles;
        }
        sb.count = 2;
        if (launch_train_skinny(sb, tb, s, "train_bilstm_step")) return 1;
    }
    const int fin = T & 1;
    L2S_CHECK_HIP(hipMemcpyAsync(state + sl.h, hf[0][fin], sizeof(float) * Bp * 512, hipMemcpyDeviceToDevice, s));
    L2S_CHECK_HIP(hipMemcpyAsync(state + sl.h + (int64_t)Bp * 512, hf[1][fin], sizeof(float) * Bp * 512, hipMemcpyDeviceToDevice, s));
    if (launch_from_frag(cf[0], 512, B, 512, tp.cellcat, 1024, 0, s)) return 1;
    if (launch_from_frag(cf[1], 512, B, 512, tp.cellcat, 1024, 512, s)) return 1;
    {
        GemmP p = gemm_plain(tp.cellcat, 1024, w.e_c.W, state + sl.ecell, 512, B, 512, 1024); p.shift = w.e_c.shift;
        if (launch_gemm1(p, s, "train_prologue_gemm")) return 1;
        if (launch_stop_const(state + sl.ecell, w.stop_tail, w.stop_bias, B, state + sl.stopc, s)) return 1;
        GemmP e = gemm_plain(tp.rnn, 1024, w.enc_proj.W, tp.cat, 4608, BT, 512, 1024);
        e.shift = w.enc_proj.shift; e.R1 = tp.resid; e.ldr1 = 512; e.R2 = tp.s_a; e.ldr2 = 512; e.r2_div = T;
        if (launch_gemm1(e, s, "train_prologue_gemm")) return 1;
        if (launch_copy_cols(tp.cat, 4608, 0, state + sl.enc, 512, 0, 1, BT, 512, s)) return 1;
    }
    {
        GemmBatch gb{};
        for (int kv = 0; kv < 2; ++kv)
            for (int j = 0; j < 4; ++j)
                gb.p[kv * 4 + j] = tconv(tp.cat, 4608, B, T, 512, w.mh_branch[kv][j], 512, MH_KS[j], 1, MH_KS[j] / 2, tp.cat + 512 + (kv * 4 + j) * 512, 4608, ACT_SILU, nullptr);
        gb.count = 8;
        for (int q = 0; q < 8; ++q) gb.p[q].Zout = tp.zcat + 512 + q * 512;       // Zout shares C's addressing (ld 4608)
        if (m->bn_batch) {
            GemmBatch sbt = gb;
            for (int q = 0; q < 8; ++q) { sbt.p[q].stats = tp.stats + q * tp.stats_group; sbt.p[q].stats_raw = 1; sbt.p[q].scale = nullptr; sbt.p[q].shift = nullptr; }
            if (launch_gemm(sbt, s, "train_multihop_conv_stats")) return 1;
            for (int q = 0; q < 8; ++q) {
                const std::string c = std::string("decoder.") + (q < 4 ? "K" : "V") + ".0.conv." + std::to_string(q % 4);
                BnLayer L = dec_bn_layer(m, c + ".1", c + ".0.bias", tp.bn + q * 1024, 512);
                if (bn_stats_finalize(tp.stats + q * tp.stats_group, (BT + 63) / 64, 1024, BT, L, m->bn_momentum, s)) return 1;
                gb.p[q].scale = L.scale; gb.p[q].shift = L.shift;
            }
        }
        if (m->bn_batch ? launch_gemm_finish(gb, s, "train_multihop_conv_epilogue") : launch_gemm(gb, s, "train_multihop_conv_gemm")) return 1;
        GemmBatch bb{};
        for (int kv = 0; kv < 2; ++kv) {
            GemmP p = gemm_plain(tp.cat, 4608, w.mh_bott[kv].W, state + (kv == 0 ? sl.k : sl.v), 512, BT, 512, 2560);
            if (kv == 1) { p.a_split = 512; p.a_gap = 2048; }
            p.shift = w.mh_bott[kv].shift; p.act = ACT_PSINE; p.actw = w.mh_bott[kv].actw; p.R1 = w.pos; p.ldr1 = 512; p.r1_mod = T; p.Zout = tp.zkv[kv];
            bb.p[kv] = p;
        }
        bb.count = 2;
        if (launch_gemm(bb, s, "train_multihop_bottleneck_gemm")) return 1;
    }
    {
        GemmBatch gb{};
        for (int j = 0; j < 4; ++j)
            gb.p[j] = tconv(tp.cat, 4608, B, T, 512, w.ct_branch[j], 512, CT_KS[j], CT_KS[j], 0, tp.cmap[j], 512, ACT_SILU, tp.zagg[j]);
        gb.count = 4;
        if (m->bn_batch) {
            GemmBatch sbt = gb;
            for (int j = 0; j < 4; ++j) { sbt.p[j].stats = tp.stats + j * tp.stats_group; sbt.p[j].stats_raw = 1; sbt.p[j].scale = nullptr; sbt.p[j].shift = nullptr; }
            if (launch_gemm(sbt, s, "train_content_agg_stats")) return 1;
            for (int j = 0; j < 4; ++j) {
                const std::string c = "decoder.content.agg." + std::to_string(j);
                const int rows = B * tp.L[j];
                BnLayer L = dec_bn_layer(m, c + ".1", c + ".0.bias", tp.bn + (8 + j) * 1024, 512);
                if (bn_stats_finalize(tp.stats + j * tp.stats_group, (rows + 63) / 64, 1024, rows, L, m->bn_momentum, s)) return 1;
                gb.p[j].scale = L.scale; gb.p[j].shift = L.shift;
            }
        }
        if (m->bn_batch ? launch_gemm_finish(gb, s, "train_content_agg_epilogue") : launch_gemm(gb, s, "train_content_agg_gemm")) return 1;
        PoolCatP pc{};
        pc.x[0] = tp.cat; pc.L[0] = T; pc.ld[0] = 4608;
        for (int j = 0; j < 4; ++j) { pc.x[j + 1] = tp.cmap[j]; pc.L[j + 1] = tp.L[j]; pc.ld[j + 1] = 512; }
        pc.nmaps = 5; pc.B = B; pc.m = tp.m; pc.C = 512; pc.out = tp.pooled;
        if (launch_pool_cat(pc, s)) return 1;
        if (launch_gemm1(tlin(tp.pooled, 2560, w.ct_bott, tp.wv, 256, R, 256, 2560, ACT_NONE, nullptr), s, "train_content_gemm")) return 1;
        GemmBatch g1{};
        g1.p[0] = tlin(tp.wv, 256, w.ct_k0, tp.tA, 256, R, 256, 256, ACT_SILU, tp.zk0);
        g1.p[1] = tlin(tp.wv, 256, w.ct_fc0, tp.tB, 256, R, 256, 256, ACT_SILU, tp.zf0);
        g1.count = 2;
        if (launch_gemm(g1, s, "train_content_gemm")) return 1;
        GemmBatch g2{};
        g2.p[0] = tlin(tp.tA, 256, w.ct_k2, state + sl.ckey, 256, R, 256, 256, ACT_SILU, tp.zk2);
        g2.p[1] = tlin(tp.tB, 256, w.ct_fc2, tp.tC, 256, R, 256, 256, ACT_SILU, tp.zf2);
        g2.count = 2;
        if (launch_gemm(g2, s, "train_content_gemm")) return 1;
        GemmP p3 = tlin(tp.tC, 256, w.ct_fc4, tp.logits, VOC, R, VOC, 256, ACT_SILU, tp.zf4);
        if (launch_gemm1(p3, s, "train_content_gemm")) return 1;
        if (launch_gumbel_softmax(tp.logits, gumbel, R, VOC, 0.1f, tp.zsoft, VOCP, content_dis ? content_dis : tp.dis, s)) return 1;
        if (content_dis) L2S_CHECK_HIP(hipMemcpyAsync(tp.dis, content_dis, sizeof(float) * R * VOC, hipMemcpyDeviceToDevice, s));
        if (launch_gemm1(gemm_plain(tp.zsoft, VOCP, w.ct_emb.W, state + sl.cval, 256, R, 256, VOCP), s, "train_content_gemm")) return 1;
    }
    if (launch_fill(state + sl.c, (int64_t)Bp * 512 * 2, 0.f, s)) return 1;
    return 0;
}

}  // namespace l2s

namespace l2s {

// d logits of Content.encode: y = softmax((l+g)/tau) (gumbel) and dis = softmax(l):
//   dl = y*(dy - sum y dy)/tau + dis*(ddis - sum dis ddis);  one block per row; output ld 504 (padding columns zeroed)
__global__ __launch_bounds__(256) void content_softmax_bwd_kernel(const float* __restrict__ y, const float* __restrict__ dy, int ldy, const float* __restrict__ dis,
                                                                  const float* __restrict__ ddis, int n, float inv_tau, float* __restrict__ dl, int ldl) {
    __shared__ float sh[8];
    const int row = blockIdx.x, tid = threadIdx.x;
    float s1 = 0.f, s2 = 0.f;
    for (int j = tid; j < n; j += 256) {
        s1 += y[(int64_t)row * ldy + j] * dy[(int64_t)row * ldy + j];
        if (ddis) s2 += dis[(int64_t)row * n + j] * ddis[(int64_t)row * n + j];
    }
    s1 = wave_sum_f(s1); s2 = wave_sum_f(s2);
    if ((tid & 63) == 0) { sh[tid >> 6] = s1; sh[4 + (tid >> 6)] = s2; }
    __syncthreads();
    s1 = (sh[0] + sh[1]) + (sh[2] + sh[3]); s2 = (sh[4] + sh[5]) + (sh[6] + sh[7]);
    for (int j = tid; j < ldl; j += 256) {
        float v = 0.f;
        if (j < n) {
            v = y[(int64_t)row * ldy + j] * (dy[(int64_t)row * ldy + j] - s1) * inv_tau;
            if (ddis) v += dis[(int64_t)row * n + j] * (ddis[(int64_t)row * n + j] - s2);
        }
        dl[(int64_t)row * ldl + j] = v;
    }
}

// backward of the adaptive average pooling + concatenation: dpooled (B,m,nmaps*C) -> d map_j[b][t][c] (+)= sum over bins containing t of dpooled / binsize
struct PoolBwdP { float* dx[5]; int L[5]; int ld[5]; int acc[5]; int nmaps; int B, m, C; const float* dp; };
__global__ __launch_bounds__(256) void pool_cat_bwd_kernel(const PoolBwdP p) {
    int64_t total = 0;
    for (int j = 0; j < p.nmaps; ++j) total += (int64_t)p.B * p.L[j] * p.C;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        int64_t r = idx; int j = 0;
        while (r >= (int64_t)p.B * p.L[j] * p.C) { r -= (int64_t)p.B * p.L[j] * p.C; ++j; }
        const int c = r % p.C; r /= p.C;
        const int L = p.L[j];
        const int t = r % L, b = r / L;
        float g = 0.f;
        for (int i = 0; i < p.m; ++i) {
            const int st = (i * L) / p.m, en = ((i + 1) * L + p.m - 1) / p.m;
            if (t >= st && t < en) g += p.dp[((int64_t)b * p.m + i) * (p.nmaps * p.C) + j * p.C + c] / (float)(en - st);
        }
        float* dst = p.dx[j] + ((int64_t)b * L + t) * p.ld[j] + c;
        *dst = p.acc[j] ? *dst + g : g;
    }
}
// out[b][c] = sum_t x[b][t][c]
__global__ __launch_bounds__(256) void sum_time_kernel(const float* __restrict__ x, int B, int T, int C, int ld, float* __restrict__ out) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= B * C) return;
    const int b = idx / C, c = idx - b * C;
    float a = 0.f;
    for (int t = 0; t < T; ++t) a += x[((int64_t)b * T + t) * ld + c];
    out[idx] = a;
}
__global__ __launch_bounds__(256) void add3_kernel(const float* __restrict__ a, int lda, const float* __restrict__ b, int ldb, float* __restrict__ out, int ldo, int64_t rows, int C) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < rows * C; i += (int64_t)gridDim.x * 256) {
        const int c = i % C; const int64_t r = i / C;
        out[r * ldo + c] = a[r * lda + c] + (b ? b[r * ldb + c] : 0.f);
    }
}

static int64_t pro_bwd_ws_floats(int B, int T) {
    int L[4]; const int m = content_lens(T, L);
    const int64_t BT = (int64_t)B * T, R = (int64_t)B * m, Bp = pad16(B);
    int64_t n = BT * (4608 + 512 * 4 + 1024 + 2048 * 3) + (int64_t)512 * 11 * 512 + (int64_t)AB_RS * 3 * 4608 + R * (2560 + 504 * 3 + 256 * 6) + 504 * 256 +
                (int64_t)B * (512 * 10 + 1024 * 2) + Bp * 2048 * 4 + BT * 2048 + 4 * BT * 512 + 8192 + 16 * BT * 512;
    return n + 64 * 65;
}

// Inputs: gradients of the state the loop consumed.  Outputs: every prologue parameter gradient (bound slots) and dvis (B,T,1024).
static int prologue_train_bwd(l2s_model* m, const float* vis, const float* emb, int B, int T, float* state, float* tape_base, float* wbuf,
                              const float* dk, const float* dv, const float* dckey, const float* dcval, const float* dh_init, const float* de_c,
                              const float* ddis, float* dvis, void* ws, int64_t ws_bytes, hipStream_t s) {
    const Weights& w = m->w;
    ProTape tp = pro_tape(tape_base, B, T);
    TrainW tw = train_w(wbuf);
    const int BT = B * T, Bp = pad16(B), R = B * tp.m, mT = tp.m;
    const std::string D = "decoder.";
    Bump bp(ws, ws_bytes);
    float* dcat = bp.f((int64_t)BT * 4608); float* gk = bp.f((int64_t)BT * 512); float* gconv = bp.f((int64_t)BT * 512);
    float* dxc = bp.f((int64_t)BT * 512); float* denc = bp.f((int64_t)BT * 512);
    float* drnn = bp.f((int64_t)BT * 1024); float* dgbt[2] = {bp.f((int64_t)BT * 2048), bp.f((int64_t)BT * 2048)};
    float* dgk2[2] = {bp.f((int64_t)BT * 2048), bp.f((int64_t)BT * 2048)};      // processing-order stacks of the two directions
    float* dwp = bp.f((int64_t)512 * 11 * 512); float* partials = bp.f((int64_t)AB_RS * 3 * 4608);
    float* dpooled = bp.f((int64_t)R * 2560); float* dzs = bp.f((int64_t)R * 504); float* dl = bp.f((int64_t)R * 504); float* dz4 = bp.f((int64_t)R * 504);
    float* r256[6]; for (auto& q : r256) q = bp.f((int64_t)R * 256);
    float* tmpE = bp.f(504 * 256);
    float* ds_a = bp.f((int64_t)B * 512); float* ds_e = bp.f((int64_t)B * 512); float* dzs_site = bp.f((int64_t)B * 512);
    float* dcellcat = bp.f((int64_t)B * 1024); float* dhn = bp.f((int64_t)B * 512);
    float* dhd[2] = {bp.f((int64_t)B * 512), bp.f((int64_t)B * 512)}; float* dccd[2] = {bp.f((int64_t)B * 512), bp.f((int64_t)B * 512)};
    float* f_dgd[2][2] = {{bp.f((int64_t)Bp * 2048), bp.f((int64_t)Bp * 2048)}, {bp.f((int64_t)Bp * 2048), bp.f((int64_t)Bp * 2048)}};
    float* dmap[4]; for (int j = 0; j < 4; ++j) dmap[j] = bp.f((int64_t)B * tp.L[j] * 512);
    float* small = bp.f(8192);
    float* skp = bp.f((int64_t)16 * BT * 512);                  // split-K partials of the wide input gradients
    L2S_REQUIRE(!bp.overflow, "training prologue backward workspace too small");
    auto G = [&](const std::string& k) { return m->grad(D + k); };
    auto Cn = [&](const std::string& k) { return m->canon(D + k); };
    auto dW = [&](const float* dz, int ldz, int nout, const float* x, int ldx, int cin, int rows, float* out, int ldc) -> int {   // linear weight gradient
        if (!out) return 0;
        BwdGemmP p = bwd_dw(dz, ldz, x, ldx, out, 1, rows, rows, nout, cin, 1, 1, 0, false);
        p.ldc = ldc;
        return launch_gemm_bwd(p, s, "train_bwd_prologue_dw");
    };
    auto dX = [&](const float* dz, int ldz, int nout, const float* Wf, int ldw, float* out, int ldo, int cin, int rows, bool acc) -> int {   // linear input gradient
        BwdGemmP p = bwd_dx(dz, ldz, Wf, out, ldo, 1, rows, rows, nout, cin, 1, 0, acc);
        p.ldb = ldw;
        // B*T (or B) rows by <= 1024 columns are a few dozen 64x64 tiles with K up to 2048: split the reduction so that the grid fills the chip
        const int tiles = ((rows + 63) / 64) * ((cin + 63) / 64);
        int splits = nout >= 1024 && tiles < 128 ? std::min(8, 256 / tiles) : 1;
        while (splits > 1 && gemm_bwd_splitk_floats(p, splits) > (int64_t)16 * BT * 512) --splits;
        if (splits > 1) return launch_gemm_bwd_splitk(p, splits, skp, s, "train_bwd_prologue_dx");
        return launch_gemm_bwd(p, s, "train_bwd_prologue_dx");
    };
    auto act = [&](const float* dy, int ldy, const float* z, int ldz, float* dconv, int ldc, int64_t rows, int C, int actk, const float* aw, const float* scale,
                   const float* gamma, const float* beta, float* g_shift, float* g_gamma, float* g_aw, float* g_cb) -> int {
        ActBwdP a{}; a.dy = dy; a.ld_dy = ldy; a.z = z; a.ld_z = ldz; a.dconv = dconv; a.ld_dconv = ldc; a.rows = rows; a.C = C; a.act = actk; a.actw = aw;
        a.scale = scale; a.gamma = gamma; a.beta = beta; a.partials = partials;
        return act_bwd(a, g_shift, g_gamma, g_aw, g_cb, false, s);
    };
    // conv + BatchNorm + SiLU branch: with batch statistics the epilogue backward is followed by the coupling term, the conv bias gets a zero gradient
    const bool bnb = m->bn_batch;
    auto act_bn = [&](const float* dy, int ldy, const float* z, int ldz, float* dconv, int64_t rows, const float* eval_scale, int slot, const std::string& c) -> int {
        ActBwdP a{}; a.dy = dy; a.ld_dy = ldy; a.z = z; a.ld_z = ldz; a.dconv = dconv; a.ld_dconv = 512; a.rows = rows; a.C = 512; a.act = ACT_SILU;
        a.scale = bnb ? tp.bn + slot * 1024 : eval_scale; a.gamma = Cn(c + ".1.weight"); a.beta = Cn(c + ".1.bias"); a.partials = partials;
        if (act_bwd(a, G(c + ".1.bias"), G(c + ".1.weight"), nullptr, bnb ? nullptr : G(c + ".0.bias"), false, s, bnb ? small : nullptr)) return 1;
        if (bnb) {
            if (bn_train_fix(dconv, 512, z, ldz, 1, 0, a.gamma, a.beta, a.scale, small, rows, 512, s)) return 1;
            if (float* gb = G(c + ".0.bias")) { if (launch_fill(gb, 512, 0.f, s)) return 1; }
        }
        return 0;
    };

    // ---- A. K / V bottlenecks: k = PSine(bott([x | branches])) + pos
    for (int kv = 0; kv < 2; ++kv) {
        const std::string kn = kv == 0 ? "K" : "V";
        if (act(kv == 0 ? dk : dv, 512, tp.zkv[kv], 512, gk, 512, BT, 512, ACT_PSINE, Cn(kn + ".1.w"), nullptr, nullptr, nullptr, G(kn + ".0.bottleneck.bias"), nullptr,
                G(kn + ".1.w"), nullptr)) return 1;
        float* gw = G(kn + ".0.bottleneck.weight");
        if (kv == 0) {
            if (dW(gk, 512, 512, tp.cat, 4608, 2560, BT, gw, 2560)) return 1;
            if (dX(gk, 512, 512, w.mh_bott[0].W, 2560, dcat, 4608, 2560, BT, false)) return 1;
        } else {
            if (dW(gk, 512, 512, tp.cat, 4608, 512, BT, gw, 2560)) return 1;
            if (dW(gk, 512, 512, tp.cat + 2560, 4608, 2048, BT, gw ? gw + 512 : nullptr, 2560)) return 1;
            if (dX(gk, 512, 512, w.mh_bott[1].W, 2560, dcat, 4608, 512, BT, true)) return 1;
            if (dX(gk, 512, 512, w.mh_bott[1].W + 512, 2560, dcat + 2560, 4608, 2048, BT, false)) return 1;
        }
    }
    // ---- B. the 8 MultiHop branches: SiLU(BN(conv_k(x)))
    for (int q = 0; q < 8; ++q) {
        const int kv = q / 4, j = q % 4, k = MH_KS[j];
        const std::string c = std::string(kv == 0 ? "K" : "V") + ".0.conv." + std::to_string(j);
        if (act_bn(dcat + 512 + q * 512, 4608, tp.zcat + 512 + q * 512, 4608, gconv, BT, w.mh_branch[kv][j].scale, q, c)) return 1;
        if (float* gw = G(c + ".0.weight")) {
            if (launch_gemm_bwd(bwd_dw(gconv, 512, tp.cat, 4608, dwp, B, T, T, 512, 512, k, 1, k / 2, false), s, "train_bwd_multihop_dw")) return 1;
            if (conv1d_grad_to_canonical(dwp, 512, 512, k, gw, false, s)) return 1;
        }
        BwdGemmP px = bwd_dx(gconv, 512, w.mh_branch[kv][j].W, dcat, 4608, B, T, T, 512, 512, k, k / 2, true);
        if (launch_gemm_bwd_splitk(px, std::min(16, 4 * k), skp, s, "train_bwd_multihop_dx")) return 1;      // 32 output tiles, K = 512 k
    }
    // ---- C. Content.encode
    if (launch_fill(dxc, (int64_t)BT * 512, 0.f, s)) return 1;
    {
        float *dtC = r256[0], *dtB = r256[1], *dwv = r256[2], *dtA = r256[3], *dzt = r256[4];
        // value = zsoft @ word_embeddings
        if (launch_fill(dzs, (int64_t)R * 504, 0.f, s)) return 1;
        if (launch_gemm1(gemm_plain(dcval, 256, Cn("content.word_embeddings"), dzs, 504, R, VOC, 256), s, "train_bwd_content_gemm")) return 1;
        if (float* g = G("content.word_embeddings")) {
            if (dW(tp.zsoft, 504, 504, dcval, 256, 256, R, tmpE, 256)) return 1;
            hipLaunchKernelGGL(copy_rows_kernel, dim3(ew(VOC * 256)), dim3(256), 0, s, tmpE, 256, g, 256, VOC, 256);
        }
        hipLaunchKernelGGL(content_softmax_bwd_kernel, dim3(R), dim3(256), 0, s, tp.zsoft, dzs, 504, tp.dis, ddis, VOC, 10.0f, dl, 504);
        // location_fc.4 (+SiLU): dz4 padded to 504 columns
        if (launch_fill(dz4, (int64_t)R * 504, 0.f, s)) return 1;
        if (act(dl, 504, tp.zf4, VOC, dz4, 504, R, VOC, ACT_SILU, nullptr, nullptr, nullptr, nullptr, G("content.location_fc.4.bias"), nullptr, nullptr, nullptr)) return 1;
        if (float* g = G("content.location_fc.4.weight")) {
            if (dW(dz4, 504, 504, tp.tC, 256, 256, R, tmpE, 256)) return 1;
            hipLaunchKernelGGL(copy_rows_kernel, dim3(ew(VOC * 256)), dim3(256), 0, s, tmpE, 256, g, 256, VOC, 256);
        }
        if (dX(dz4, 504, 504, tw.fc4, 256, dtC, 256, 256, R, false)) return 1;
        // location_fc.2, .0
        if (act(dtC, 256, tp.zf2, 256, dzt, 256, R, 256, ACT_SILU, nullptr, nullptr, nullptr, nullptr, G("content.location_fc.2.bias"), nullptr, nullptr, nullptr)) return 1;
        if (dW(dzt, 256, 256, tp.tB, 256, 256, R, G("content.location_fc.2.weight"), 256)) return 1;
        if (dX(dzt, 256, 256, w.ct_fc2.W, 256, dtB, 256, 256, R, false)) return 1;
        if (act(dtB, 256, tp.zf0, 256, dzt, 256, R, 256, ACT_SILU, nullptr, nullptr, nullptr, nullptr, G("content.location_fc.0.bias"), nullptr, nullptr, nullptr)) return 1;
        if (dW(dzt, 256, 256, tp.wv, 256, 256, R, G("content.location_fc.0.weight"), 256)) return 1;
        if (dX(dzt, 256, 256, w.ct_fc0.W, 256, dwv, 256, 256, R, false)) return 1;
        // key = SiLU(K.2(SiLU(K.0(w))))
        if (act(dckey, 256, tp.zk2, 256, dzt, 256, R, 256, ACT_SILU, nullptr, nullptr, nullptr, nullptr, G("content.K.2.bias"), nullptr, nullptr, nullptr)) return 1;
        if (dW(dzt, 256, 256, tp.tA, 256, 256, R, G("content.K.2.weight"), 256)) return 1;
        if (dX(dzt, 256, 256, w.ct_k2.W, 256, dtA, 256, 256, R, false)) return 1;
        if (act(dtA, 256, tp.zk0, 256, dzt, 256, R, 256, ACT_SILU, nullptr, nullptr, nullptr, nullptr, G("content.K.0.bias"), nullptr, nullptr, nullptr)) return 1;
        if (dW(dzt, 256, 256, tp.wv, 256, 256, R, G("content.K.0.weight"), 256)) return 1;
        if (dX(dzt, 256, 256, w.ct_k0.W, 256, dwv, 256, 256, R, true)) return 1;
        // bottleneck over the pooled concatenation
        if (dW(dwv, 256, 256, tp.pooled, 2560, 2560, R, G("content.bottleneck.weight"), 2560)) return 1;
        if (float* g = G("content.bottleneck.bias")) { if (colsum(dwv, R, 256, partials, dzt, g, false, s)) return 1; }
        if (dX(dwv, 256, 256, w.ct_bott.W, 2560, dpooled, 2560, 2560, R, false)) return 1;
        PoolBwdP pb{};
        pb.dx[0] = dcat; pb.L[0] = T; pb.ld[0] = 4608; pb.acc[0] = 1;
        for (int j = 0; j < 4; ++j) { pb.dx[j + 1] = dmap[j]; pb.L[j + 1] = tp.L[j]; pb.ld[j + 1] = 512; pb.acc[j + 1] = 0; }
        pb.nmaps = 5; pb.B = B; pb.m = mT; pb.C = 512; pb.dp = dpooled;
        hipLaunchKernelGGL(pool_cat_bwd_kernel, dim3(2048), dim3(256), 0, s, pb);
        for (int j = 0; j < 4; ++j) {
            const int k = CT_KS[j], Lj = tp.L[j];
            const std::string c = "content.agg." + std::to_string(j);
            if (act_bn(dmap[j], 512, tp.zagg[j], 512, gconv, (int64_t)B * Lj, w.ct_branch[j].scale, 8 + j, c)) return 1;
            if (float* gw = G(c + ".0.weight")) {
                if (launch_gemm_bwd(bwd_dw(gconv, 512, tp.cat, 4608, dwp, B, Lj, T, 512, 512, k, k, 0, false), s, "train_bwd_content_agg_dw")) return 1;
                if (conv1d_grad_to_canonical(dwp, 512, 512, k, gw, false, s)) return 1;
            }
            // stride = kernel, no padding: each input frame belongs to one window -> a plain linear map on the (B*L, k*512) view of x
            BwdGemmP px{};
            px.mode = BWD_DX; px.A = gconv; px.lda = 512; px.B = w.ct_branch[j].W; px.ldb = k * 512; px.C = dxc; px.ldc = k * 512;
            px.M = B * Lj; px.N = k * 512; px.K = 512; px.Tx = B * Lj; px.Tz = B * Lj; px.taps = 1; px.stride = 1; px.pad = 0; px.padp = 0; px.Nout = 512; px.Cin = k * 512;
            px.c_T = Lj; px.c_seq_stride = (int64_t)T * 512; px.alpha = 1.f; px.accumulate = 1;
            if (launch_gemm_bwd(px, s, "train_bwd_content_agg_dx")) return 1;
        }
    }
    // ---- D. enc = encoder_proj(rnn_out) + s_a + residual_bottleneck(vis)
    hipLaunchKernelGGL(add3_kernel, dim3(2048), dim3(256), 0, s, dcat, 4608, dxc, 512, denc, 512, (int64_t)BT, 512);
    if (dW(denc, 512, 512, tp.rnn, 1024, 1024, BT, G("encoder_proj.linear_layer.weight"), 1024)) return 1;
    if (dW(denc, 512, 512, vis, 1024, 1024, BT, G("residual_bottleneck.weight"), 1024)) return 1;
    if (colsum(denc, BT, 512, partials, gconv, small, false, s)) return 1;
    for (const char* k : {"encoder_proj.linear_layer.bias", "residual_bottleneck.bias"})
        if (float* g = G(k)) hipLaunchKernelGGL(copy_rows_kernel, dim3(ew(512)), dim3(256), 0, s, small, 512, g, 512, 1, 512);
    if (dX(denc, 512, 512, w.enc_proj.W, 1024, drnn, 1024, 1024, BT, false)) return 1;
    if (dX(denc, 512, 512, w.resid.W, 1024, dvis, 1024, 1024, BT, false)) return 1;
    hipLaunchKernelGGL(sum_time_kernel, dim3(ew(B * 512)), dim3(256), 0, s, denc, B, T, 512, 512, ds_a);
    if (act(ds_a, 512, tp.z_sa, 512, dzs_site, 512, B, 512, ACT_PSINE, Cn("attention_site.1.w"), nullptr, nullptr, nullptr, G("attention_site.0.linear_layer.bias"), nullptr,
            G("attention_site.1.w"), nullptr)) return 1;
    if (dW(dzs_site, 512, 512, emb, 256, 256, B, G("attention_site.0.linear_layer.weight"), 256)) return 1;
    // ---- E. encoder_cell = E_C(cat(c_fwd, c_bwd))
    if (dW(de_c, 512, 512, tp.cellcat, 1024, 1024, B, G("E_C.linear_layer.weight"), 1024)) return 1;
    if (float* g = G("E_C.linear_layer.bias")) { if (colsum(de_c, B, 512, partials, gconv, g, false, s)) return 1; }
    if (dX(de_c, 512, 512, w.e_c.W, 1024, dcellcat, 1024, 1024, B, false)) return 1;
    // ---- F. BiLSTM back-propagation through time (both directions; the decoder's initial hidden states are its final hidden states)
    // Both directions advance in the same launches, and the cell backward of step-1 rides in the epilogue of the product that yields its dh
    // (d h_{step-1} = dg_step @ W_hh + d rnn_out): one launch per time step for the pair (was four: cell kernel + product, per direction).
    if (launch_fill(ds_e, (int64_t)B * 512, 0.f, s)) return 1;
    auto t_of = [&](int d, int step) { return d == 0 ? step : T - 1 - step; };
    for (int d = 0; d < 2; ++d) {       // last processed step first: its dh is the decoder's initial-state gradient plus d rnn_out
        hipLaunchKernelGGL(copy_rows_kernel, dim3(ew(B * 512)), dim3(256), 0, s, dcellcat + d * 512, 1024, dccd[d], 512, B, 512);
        const int step = T - 1, t = t_of(d, step);
        hipLaunchKernelGGL(lstm_bwd_kernel, dim3(ew(B * 512)), dim3(256), 0, s, dh_init + (int64_t)d * B * 512, 512, drnn + (int64_t)t * 1024 + d * 512, T * 1024, dccd[d],
                           tp.gates[d] + (int64_t)step * B * 2048, tp.cproc[d] + (int64_t)step * B * 512, tp.cproc[d] + (int64_t)(step + 1) * B * 512, B, 512, f_dgd[d][0],
                           dgk2[d] + (int64_t)step * B * 2048, dgbt[d] + (int64_t)t * 2048, (int64_t)T * 2048);
    }
    for (int step = T - 1; step >= 0; --step) {
        const int cur = (T - 1 - step) & 1;
        SkinnyBatch sb{}; TrainSkinnyBatch tb{};
        for (int d = 0; d < 2; ++d) {
            sb.p[d] = bsk(tw.bhh[d], 512, 2048, B, f_dgd[d][cur], dhd[d], 512); sb.ntiles[d] = 32;
            if (step > 0) {             // the cell of step-1 in this product's epilogue
                const int ps = step - 1, t = t_of(d, ps);
                SkinnyTrain& q = tb.t[d];
                q.lb_gates = tp.gates[d] + (int64_t)ps * B * 2048; q.lb_cprev = tp.cproc[d] + (int64_t)ps * B * 512; q.lb_cnew = tp.cproc[d] + (int64_t)(ps + 1) * B * 512;
                q.lb_dc = dccd[d]; q.lb_dha = drnn + (int64_t)t * 1024 + d * 512; q.lb_ld_a = T * 1024;
                q.lb_frag = f_dgd[d][cur ^ 1]; q.lb_stack = dgk2[d] + (int64_t)ps * B * 2048; q.lb_H = 512;
                q.lb_stack2 = dgbt[d] + (int64_t)t * 2048; q.lb_ld_stack2 = (int64_t)T * 2048;
            }
        }
        sb.count = 2;
        if (launch_train_skinny(sb, tb, s, "train_bwd_bilstm_dx")) return 1;
    }
    for (int d = 0; d < 2; ++d) {
        const std::string suf = d == 0 ? "l0" : "l0_reverse";
        hipLaunchKernelGGL(add3_kernel, dim3(ew(B * 512)), dim3(256), 0, s, dhd[d], 512, dccd[d], 512, dhn, 512, (int64_t)B, 512);      // h0 = c0 = s_e
        if (add_into(dhn, ds_e, (int64_t)B * 512, s)) return 1;
        if (dW(dgk2[d], 2048, 2048, tp.hproc[d], 512, 512, BT, G("encoder_rnn.weight_hh_" + suf), 512)) return 1;
        if (dW(dgbt[d], 2048, 2048, vis, 1024, 1024, BT, G("encoder_rnn.weight_ih_" + suf), 1024)) return 1;
        if (colsum(dgk2[d], BT, 2048, partials, dcat /*scratch: dcat is dead by now*/, small, false, s)) return 1;
        for (const std::string& k : {"encoder_rnn.bias_ih_" + suf, "encoder_rnn.bias_hh_" + suf})
            if (float* g = G(k)) hipLaunchKernelGGL(copy_rows_kernel, dim3(ew(2048)), dim3(256), 0, s, small, 2048, g, 2048, 1, 2048);
        if (dX(dgbt[d], 2048, 2048, w.wih_cat + (int64_t)d * 2048 * 1024, 1024, dvis, 1024, 1024, BT, true)) return 1;
    }
    // ---- G. encoder_site embedding (initial h and c of both directions)
    if (act(ds_e, 512, tp.z_se, 512, dzs_site, 512, B, 512, ACT_PSINE, Cn("encoder_site.1.w"), nullptr, nullptr, nullptr, G("encoder_site.0.linear_layer.bias"), nullptr,
            G("encoder_site.1.w"), nullptr)) return 1;
    if (dW(dzs_site, 512, 512, emb, 256, 256, B, G("encoder_site.0.linear_layer.weight"), 256)) return 1;
    L2S_CHECK_HIP(hipGetLastError());
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

int64_t l2s_train_steps_tape_floats(int B, int S) { return step_tape_floats(B, S); }
int64_t l2s_train_steps_weights_floats(void) { return train_w_floats(); }
int64_t l2s_train_steps_ws_bytes(int B, int S) { return (std::max(step_fwd_ws_floats(B), step_bwd_ws_floats(B, S)) + 1024) * (int64_t)sizeof(float); }

int l2s_train_steps_pack_weights(l2s_model* m, float* wbuf, void* stream) {
    L2S_REQUIRE(m && wbuf, "bad arguments");
    return pack_train_weights(m, wbuf, (hipStream_t)stream);
}

int l2s_train_steps_fwd(l2s_model* m, float* state, int B, int T, int S, const float* teacher, const uint8_t* teacher_mask,
                        const uint8_t* teacher_mask_dev, float* tape, float* mel, float* stop, float* attn_logits, const float* drop_prenet,
                        const float* drop_attn, const float* drop_rnn, void* ws, int64_t ws_bytes, void* stream) {
    L2S_REQUIRE(m && m->finalized && m->has_dec && state && tape && mel && stop && attn_logits && ws, "bad arguments");
    L2S_REQUIRE(S >= 1 && S <= L2S_MAX_STEPS && B <= 96, "sizes");
    L2S_REQUIRE(!teacher || (teacher_mask && teacher_mask_dev), "teacher frames need the step mask on host and device");
    return decode_train_fwd(m, state, B, T, S, teacher, teacher_mask, teacher_mask_dev, tape, mel, stop, attn_logits, StepDrop{drop_prenet, drop_attn, drop_rnn}, ws,
                            ws_bytes, (hipStream_t)stream);
}

int l2s_train_steps_bwd(l2s_model* m, float* state, int B, int T, int S, const uint8_t* teacher_mask, float* tape, const float* attn_logits,
                        const float* dmel, const float* dstop, float* wbuf, float* dk, float* dv, float* dckey, float* dcval, float* dh_init,
                        float* de_c, const float* drop_prenet, const float* drop_attn, const float* drop_rnn, void* ws, int64_t ws_bytes, void* stream) {
    L2S_REQUIRE(m && m->finalized && m->has_dec && state && tape && attn_logits && dmel && dstop && wbuf && dk && dv && dckey && dcval && dh_init && de_c && ws,
                "bad arguments");
    return decode_train_bwd(m, state, B, T, S, teacher_mask, tape, attn_logits, dmel, dstop, wbuf, dk, dv, dckey, dcval, dh_init, de_c,
                            StepDrop{drop_prenet, drop_attn, drop_rnn}, ws, ws_bytes, (hipStream_t)stream);
}

int64_t l2s_train_prologue_tape_floats(int B, int T) { return pro_tape_floats(B, T); }
int64_t l2s_train_prologue_ws_bytes(int B, int T) { return (std::max(pro_fwd_ws_floats(B), pro_bwd_ws_floats(B, T)) + 1024) * (int64_t)sizeof(float); }

int l2s_train_prologue_fwd(l2s_model* m, const float* vis, const float* emb, const float* gumbel, int B, int T, float* state, float* content_dis,
                           float* tape, void* ws, int64_t ws_bytes, void* stream) {
    L2S_REQUIRE(m && m->finalized && m->has_dec && vis && emb && gumbel && state && tape && ws, "bad arguments");
    L2S_REQUIRE(B >= 1 && B <= 96, "sizes");
    Bf16Scope bf16scope(m->opt.train_bf16);      // GEMMs / Conv1d stacks with bf16 operands when the model asks for it (fp32 accumulation)
    return prologue_train_fwd(m, vis, emb, gumbel, B, T, state, content_dis, tape, ws, ws_bytes, (hipStream_t)stream);
}

int l2s_train_prologue_bwd(l2s_model* m, const float* vis, const float* emb, int B, int T, float* state, float* tape, float* wbuf, const float* dk,
                           const float* dv, const float* dckey, const float* dcval, const float* dh_init, const float* de_c, const float* dcontent_dis,
                           float* dvis, void* ws, int64_t ws_bytes, void* stream) {
    L2S_REQUIRE(m && m->finalized && m->has_dec && vis && emb && state && tape && wbuf && dk && dv && dckey && dcval && dh_init && de_c && dvis && ws, "bad arguments");
    L2S_REQUIRE(B >= 1 && B <= 96 && T >= 7 && T <= L2S_MAX_STEPS, "sizes");
    L2S_REQUIRE(m->canon("decoder.encoder_rnn.weight_hh_l0") != nullptr, "parameters not bound (l2s_train_bind)");
    Bf16Scope bf16scope(m->opt.train_bf16);      // GEMMs / Conv1d stacks with bf16 operands when the model asks for it (fp32 accumulation)
    return prologue_train_bwd(m, vis, emb, B, T, state, tape, wbuf, dk, dv, dckey, dcval, dh_init, de_c, dcontent_dis, dvis, ws, ws_bytes, (hipStream_t)stream);
}

int64_t l2s_train_postnet_tape_floats(int B, int S) { return post_tape_floats(B, S); }
int64_t l2s_train_postnet_ws_bytes(int B, int S) {
    return ((int64_t)B * S * 512 * 3 + (int64_t)512 * 5 * 512 + (int64_t)AB_RS * 3 * 512 + 1024 + 64 * 11 + (int64_t)5 * B * S * 512) * (int64_t)sizeof(float);
}

int l2s_train_postnet_fwd(l2s_model* m, const float* mel, int B, int S, float* tape, float* mel_post, const float* drop, void* stream) {
    L2S_REQUIRE(m && m->finalized && m->has_dec && mel && tape && mel_post, "bad arguments");
    Bf16Scope bf16scope(m->opt.train_bf16);      // GEMMs / Conv1d stacks with bf16 operands when the model asks for it (fp32 accumulation)
    return postnet_train_fwd(m, mel, B, S, tape, mel_post, drop, (hipStream_t)stream);
}

int l2s_train_postnet_bwd(l2s_model* m, const float* mel, const float* dmel_post, int B, int S, float* tape, float* dmel, const float* drop, void* ws,
                          int64_t ws_bytes, void* stream) {
    L2S_REQUIRE(m && m->finalized && m->has_dec && mel && dmel_post && tape && dmel && ws, "bad arguments");
    L2S_REQUIRE(m->canon("decoder.postnet.convolutions.0.1.weight") != nullptr, "parameters not bound (l2s_train_bind)");
    Bf16Scope bf16scope(m->opt.train_bf16);      // GEMMs / Conv1d stacks with bf16 operands when the model asks for it (fp32 accumulation)
    return postnet_train_bwd(m, mel, dmel_post, B, S, tape, dmel, drop, ws, ws_bytes, (hipStream_t)stream);
}

}  // extern "C"
