// Training path of the visual encoder (reference: VideoExtractor.forward /root/reference/model/modules/video.py:76-87 and
// shufflenetv2.py:42-152 under loss.backward(), train.py:184): forward with a tape (every unit's intermediate maps are kept - 288 GB
// of HBM make recomputation pointless) and the backward down to the Conv3d weight.  Eval-mode normalisation statistics (the
// configuration the gradient goldens pin).
//
//   conv_last/pool/normalise : L2-normalise + AvgPool backward in one kernel, then the generic epilogue/BN backward + dX/dW GEMMs
//   ShuffleNet units         : 1x1 convs = dX / split-K dW GEMMs (gemm_bwd.hip) around the shared epilogue backward (act_bwd, column
//                              strides pick the shuffled channel positions); depthwise 3x3 = transposed depthwise conv for dX and a
//                              two-stage per-tap reduction for dW
//   front-end                : MaxPool argmax gather + PReLU + BN backward fused in one pass over the saved pre-activation map, then
//                              dW of the Conv3d as a split-K GEMM over an explicit im2col of the clip (chunks of frames)
#include "../../include/l2s.h"
#include "l2s_common.h"
#include "l2s_model.h"

#include <algorithm>
#include <string>

namespace l2s {

// ------------------------------------------------------------------------------------------------------------ tape
struct EncTape {
    float* z0;                 // (NF, H/2, W/2, 24) pre-PReLU front-end map
    float* x[N_UNITS + 1];     // x[0] = pooled front-end map (NF, H/4, W/4, 24); x[u+1] = output of unit u (channel-last)
    float* t1[N_UNITS];        // banch2 first 1x1 conv output (post-ReLU)
    float* t2[N_UNITS];        // banch2 depthwise output (post-BN)
    float* b1[N_UNITS];        // stride-2 units: banch1 depthwise output (post-BN)
    float* last;               // conv_last output (NF*h*h, 768), post-ReLU
    float* t1z[N_UNITS];       // batch-statistics mode only: pre-ReLU values of t1 / of the branch columns of x[u+1] / of last (the batch-stat
    float* yz[N_UNITS];        //   backward needs xhat on every row, also where the ReLU is off)
    float* lastz;
    float* bn;                 // batch-statistics BatchNorm: this batch's (scale, shift) per layer, slot id * 2*768 floats
    float* stats; int64_t stats_floats;   // scratch of the statistics passes
    int h[N_UNITS + 1];        // spatial size of x[u]
    int NF;
};
// BatchNorm layer ids: 0 front-end; 1 + 5u + {0: banch1.1, 1: banch1.3, 2: banch2.1, 3: banch2.4, 4: banch2.6}; 81 conv_last
constexpr int ENC_BN_SLOTS = 82, ENC_BN_SLOT = 2 * LAST_CH;
static float* bn_scale(const EncTape& t, int id) { return t.bn + (int64_t)id * ENC_BN_SLOT; }
static float* bn_shift(const EncTape& t, int id) { return t.bn + (int64_t)id * ENC_BN_SLOT + LAST_CH; }
static BnLayer enc_bn_layer(const l2s_model* m, const EncTape& t, int id, const std::string& key, int C) {
    BnLayer L{};
    L.gamma = m->canon("encoder." + key + ".weight"); L.beta = m->canon("encoder." + key + ".bias");
    L.rmean = const_cast<float*>(m->canon("encoder." + key + ".running_mean")); L.rvar = const_cast<float*>(m->canon("encoder." + key + ".running_var"));
    L.scale = bn_scale(t, id); L.shift = bn_shift(t, id); L.C = C;
    return L;
}
static int64_t enc_tape_layout(EncTape* t, float* base, int B, int T, int H) {
    int64_t o = 0;
    auto take = [&](int64_t n) { float* r = base ? base + o : nullptr; o += align_up(n, 64); return r; };
    const int NF = B * T;
    if (t) t->NF = NF;
    float* z0 = take((int64_t)NF * (H / 2) * (H / 2) * 24);
    int h = H / 4, cin = STAGE_CH[0], u = 0;
    float* x0 = take((int64_t)NF * h * h * cin);
    if (t) { t->z0 = z0; t->x[0] = x0; t->h[0] = h; }
    for (int st = 0; st < 3; ++st) {
        const int cout = STAGE_CH[st + 1], half = cout / 2;
        for (int r = 0; r < STAGE_REP[st]; ++r, ++u) {
            const bool s2 = r == 0;
            const int ho = s2 ? (h + 1) / 2 : h;
            const int64_t in_px = (int64_t)NF * h * h, out_px = (int64_t)NF * ho * ho;
            float* t1 = take(in_px * half); float* t2 = take(out_px * half);
            float* b1 = s2 ? take(out_px * cin) : nullptr;
            float* y = take(out_px * cout);
            float* t1z = take(in_px * half); float* yz = take(out_px * cout);
            if (t) { t->t1[u] = t1; t->t2[u] = t2; t->b1[u] = b1; t->x[u + 1] = y; t->h[u + 1] = ho; t->t1z[u] = t1z; t->yz[u] = yz; }
            h = ho; cin = cout;
        }
    }
    float* last = take((int64_t)NF * h * h * LAST_CH);
    float* lastz = take((int64_t)NF * h * h * LAST_CH);
    if (t) t->lastz = lastz;
    float* bn = take((int64_t)ENC_BN_SLOTS * ENC_BN_SLOT);
    const int64_t sf = std::max<int64_t>((int64_t)NF * (H / 4) * (H / 4) * 116 / 16 + 8192, (int64_t)NF * 8 * 48 + (int64_t)DWS_RS * 2 * 512);
    float* stats = take(sf);
    if (t) { t->last = last; t->bn = bn; t->stats = stats; t->stats_floats = sf; }
    return o + 64;
}

static GemmP pw(const float* A, int lda, int a_off, const ConvW& c, float* C, int ldc, int c_off, int cstride, int64_t M, int N, int K) {
    GemmP p = gemm_plain(A + a_off, lda, c.W, C + c_off, ldc, (int)M, N, K);
    p.scale = c.scale; p.shift = c.shift; p.act = ACT_RELU; p.c_cstride = cstride;
    return p;
}

static int encoder_train_fwd(l2s_model* m, const float* video, int B, int T, int H, int W, const float* emb, float* vis, float* feat, float* tape_base,
                             hipStream_t s) {
    const Weights& w = m->w;
    EncTape tp; enc_tape_layout(&tp, tape_base, B, T, H);
    const int NF = tp.NF;
    const bool bnb = m->bn_batch;
    // 1x1 conv + BN + ReLU; with batch statistics: stats pass -> finalize -> the same fused kernel with this batch's scale/shift
    auto run_pw = [&](GemmP p, float* zbuf, int id, const std::string& key, const char* name) -> int {
        if (bnb) {
            p.Zout = zbuf;
            L2S_REQUIRE(gemm_stats_floats(p.M, p.N) <= tp.stats_floats, "statistics scratch too small");
            GemmP q = p; q.stats = tp.stats; q.stats_raw = 1; q.scale = nullptr; q.shift = nullptr;
            if (launch_gemm1(q, s, "train_pw_gemm_stats")) return 1;
            BnLayer L = enc_bn_layer(m, tp, id, key, p.N);
            if (bn_stats_finalize(tp.stats, (p.M + 63) / 64, 2 * p.N, p.M, L, m->bn_momentum, s)) return 1;
            p.scale = L.scale; p.shift = L.shift;
            GemmBatch fb{}; fb.p[0] = p; fb.count = 1;          // the product is parked at zbuf: epilogue pass, not a second product
            return launch_gemm_finish(fb, s, name);
        }
        return launch_gemm1(p, s, name);
    };
    auto run_dw = [&](const float* in, int h, int ldi, int C, int stride, const DwW& d, float* out, int id, const std::string& key) -> int {
        const float* sc = d.scale; const float* sh = d.shift;
        if (bnb) {
            const int ho = (h + 2 - 3) / stride + 1;
            if (launch_dwconv_stats(in, NF, h, h, ldi, 0, C, stride, d.w9, tp.stats, s, out, C, 0)) return 1;      // raw conv output parked in `out`
            BnLayer L = enc_bn_layer(m, tp, id, key, C);
            if (bn_stats_finalize(tp.stats, DWS_RS, 2 * C, (int64_t)NF * ho * ho, L, m->bn_momentum, s)) return 1;
            return launch_bn_apply(out, (int64_t)NF * ho * ho, C, C, 0, L.scale, L.shift, s);
        }
        return launch_dwconv(in, NF, h, h, ldi, 0, C, stride, d.w9, sc, sh, out, C, 0, s);
    };
    {
        FrontendW fe = w.fe;
        fe.w3 = nullptr; fe.w1 = nullptr;     // training forward: f32 MFMA kernel (it also writes the pre-PReLU map)
        if (bnb) {      // the conv runs once: its statistics pass parks the raw map in the tape's z0, BatchNorm is applied in place, PReLU + pool follow
            int nblk = 0;
            if (launch_frontend_stats(w.fe, video, B, T, H, W, tp.stats, &nblk, s, tp.z0)) return 1;
            BnLayer L = enc_bn_layer(m, tp, 0, "frontend3D.1", 24);
            const int64_t px = (int64_t)NF * (H / 2) * (W / 2);
            if (bn_stats_finalize(tp.stats, nblk, 48, px, L, m->bn_momentum, s)) return 1;
            if (launch_bn_apply(tp.z0, px, 24, 24, 0, L.scale, L.shift, s, true)) return 1;
            if (launch_frontend_pool(tp.z0, w.fe.slope, NF, H / 2, W / 2, tp.x[0], s)) return 1;
        } else if (launch_frontend(fe, video, B, T, H, W, tp.x[0], s, tp.z0)) return 1;
    }
    for (int u = 0; u < N_UNITS; ++u) {
        const UnitW& U = w.unit[u];
        const int half = U.half, cout = 2 * half, h = tp.h[u], ho = tp.h[u + 1];
        const float* x = tp.x[u]; float* y = tp.x[u + 1];
        const int64_t in_px = (int64_t)NF * h * h, out_px = (int64_t)NF * ho * ho;
        const std::string p = "trunk.0." + std::to_string(u) + ".";
        const int id = 1 + 5 * u;
        if (U.stride2) {
            const int cin = U.cin;
            if (run_dw(x, h, cin, cin, 2, U.b1_dw, tp.b1[u], id + 0, p + "banch1.1")) return 1;
            if (run_pw(pw(tp.b1[u], cin, 0, U.b1_pw, y, cout, 0, 2, out_px, half, cin), tp.yz[u], id + 1, p + "banch1.3", "train_shuffle_pw_gemm")) return 1;
            if (run_pw(pw(x, cin, 0, U.pw1, tp.t1[u], half, 0, 1, in_px, half, cin), tp.t1z[u], id + 2, p + "banch2.1", "train_shuffle_pw_gemm")) return 1;
            if (run_dw(tp.t1[u], h, half, half, 2, U.dw, tp.t2[u], id + 3, p + "banch2.4")) return 1;
            if (run_pw(pw(tp.t2[u], half, 0, U.pw2, y, cout, 1, 2, out_px, half, half), tp.yz[u] + 1, id + 4, p + "banch2.6", "train_shuffle_pw_gemm")) return 1;
        } else {
            if (launch_copy_cols(x, cout, 0, y, cout, 0, 2, in_px, half, s)) return 1;
            if (run_pw(pw(x, cout, half, U.pw1, tp.t1[u], half, 0, 1, in_px, half, half), tp.t1z[u], id + 2, p + "banch2.1", "train_shuffle_pw_gemm")) return 1;
            if (run_dw(tp.t1[u], h, half, half, 1, U.dw, tp.t2[u], id + 3, p + "banch2.4")) return 1;
            if (run_pw(pw(tp.t2[u], half, 0, U.pw2, y, cout, 1, 2, in_px, half, half), tp.yz[u] + 1, id + 4, p + "banch2.6", "train_shuffle_pw_gemm")) return 1;
        }
    }
    const int hl = tp.h[N_UNITS];
    const int64_t px = (int64_t)NF * hl * hl;
    if (run_pw(pw(tp.x[N_UNITS], STAGE_CH[3], 0, w.conv_last, tp.last, LAST_CH, 0, 1, px, LAST_CH, STAGE_CH[3]), tp.lastz, 81, "trunk.1.1", "train_conv_last_gemm")) return 1;
    if (launch_pool_norm_cat(tp.last, NF, hl * hl, LAST_CH, emb, L2S_D_EMB, T, vis, L2S_D_VIS, feat, s)) return 1;
    return 0;
}

// ------------------------------------------------------------------------------------------------------------ backward kernels
// feat = v / max(|v|, eps), v = mean over the P pixels: dlast[f][p][c] = ((df - feat*(feat.df)) / max(|v|, eps))[c] / P   (video.py:81-85)
__global__ __launch_bounds__(256) void pool_norm_bwd_kernel(const float* __restrict__ last, int P, int C, const float* __restrict__ dfeat, int ld_df,
                                                            float* __restrict__ dlast) {
    const int f = blockIdx.x, tid = threadIdx.x;
    __shared__ float red[8];
    float v[4], g[4];
    float ss = 0.f, dot = 0.f;
    int cnt = 0;
    for (int c = tid; c < C; c += 256, ++cnt) {
        float acc = 0.f;
        for (int p = 0; p < P; ++p) acc += last[((int64_t)f * P + p) * C + c];
        acc /= (float)P;
        v[cnt] = acc; g[cnt] = dfeat[(int64_t)f * ld_df + c];
        ss += acc * acc; dot += acc * g[cnt];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { ss += __shfl_xor(ss, o); dot += __shfl_xor(dot, o); }
    if ((tid & 63) == 0) { red[tid >> 6] = ss; red[4 + (tid >> 6)] = dot; }
    __syncthreads();
    ss = (red[0] + red[1]) + (red[2] + red[3]); dot = (red[4] + red[5]) + (red[6] + red[7]);
    const float nrm = sqrtf(ss);
    const bool clamped = nrm < 1e-12f;
    const float inv = 1.f / (clamped ? 1e-12f : nrm);
    cnt = 0;
    for (int c = tid; c < C; c += 256, ++cnt) {
        // d(v/n): (g - v (v.g)/n^2)/n ; when the norm is clamped the denominator is the constant eps
        const float dv = clamped ? g[cnt] * inv : (g[cnt] - v[cnt] * dot * inv * inv) * inv;
        for (int p = 0; p < P; ++p) dlast[((int64_t)f * P + p) * C + c] = dv / (float)P;
    }
}

// transposed depthwise 3x3 (pad 1): dx[n][ih][iw][c] (+)= sum_{kh,kw} gd[n][oh][ow][c] * w9[kh*3+kw][c], oh*stride + kh - 1 = ih
template <int STRIDE>
__global__ __launch_bounds__(256) void dwconv_bwd_dx_kernel(const float* __restrict__ gd, int N, int Ho, int Wo, int C, const float* __restrict__ w9,
                                                            float* __restrict__ dx, int Hi, int Wi, int ldx, int xoff, int accumulate) {
    // 32-bit index arithmetic (the maps hold < 2^31 elements; checked by the launcher), compile-time stride, and all nine taps fetched with
    // clamped addresses before the first use (64-bit divisions and a branch per tap made this gather ALU- and round-trip-bound: 26 us for a 12x12 map)
    const unsigned total = (unsigned)N * Hi * Wi * C;
    for (unsigned idx = blockIdx.x * 256u + threadIdx.x; idx < total; idx += gridDim.x * 256u) {
        const unsigned c = idx % C;
        unsigned r = idx / C;
        const int iw = r % Wi; r /= Wi;
        const int ih = r % Hi;
        const unsigned n = r / Hi;
        float g[9];
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const int th = ih + 1 - kh;
            const int oh = th / STRIDE;
            const bool okh = th >= 0 && (STRIDE == 1 || (th & 1) == 0) && oh < Ho;
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int tw = iw + 1 - kw;
                const int ow = tw / STRIDE;
                const bool ok = okh && tw >= 0 && (STRIDE == 1 || (tw & 1) == 0) && ow < Wo;
                const unsigned src = ok ? ((n * Ho + oh) * Wo + ow) * C + c : c;
                const float v = gd[src];
                g[kh * 3 + kw] = ok ? v : 0.f;
            }
        }
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < 9; ++k) acc = fmaf(g[k], w9[k * C + c], acc);
        float* dst = dx + (size_t)((n * Hi + ih) * Wi + iw) * ldx + xoff + c;
        *dst = accumulate ? *dst + acc : acc;
    }
}

constexpr int DW_RS = 1024;    // row splits of the per-tap reductions (58-232 channels = 1-4 column blocks: the splits fill the chip)
// partial[rs][k][c] = sum over this split's output pixels of gd[.][c] * x[shifted by tap k][c]
__global__ __launch_bounds__(256) void dwconv_bwd_dw_kernel(const float* __restrict__ gd, const float* __restrict__ x, int N, int Hi, int Wi, int ldx, int xoff,
                                                            int Ho, int Wo, int C, int stride, float* __restrict__ partials) {
    __shared__ float sh[9][4][64];
    const int col = blockIdx.x * 64 + (threadIdx.x & 63), rl = threadIdx.x >> 6, rs = blockIdx.y;
    const int64_t rows = (int64_t)N * Ho * Wo, chunk = (rows + DW_RS - 1) / DW_RS;
    const int64_t r_begin = rs * chunk, r_end = r_begin + chunk < rows ? r_begin + chunk : rows;
    float a[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) a[k] = 0.f;
    if (col < C) {
        // rows are wave-uniform (lanes = channels): 32-bit pixel arithmetic on the scalar unit, and U rows per trip so that
        // 10 x U independent loads are in flight instead of one dependent round trip per output pixel
        constexpr int U = 4;
        const int cx = xoff + col;
        for (int r0 = (int)r_begin + rl; r0 < (int)r_end; r0 += 4 * U) {
            float g[U], xv[U][9];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int r = r0 + 4 * u;
                const bool live = r < (int)r_end;
                const int rr = live ? r : (int)r_begin;
                const int ow = rr % Wo, q = rr / Wo;
                const int oh = q % Ho, n = q / Ho;
                g[u] = live ? gd[(int64_t)rr * C + col] : 0.f;
#pragma unroll
                for (int kh = 0; kh < 3; ++kh) {
                    const int ih = oh * stride + kh - 1;
#pragma unroll
                    for (int kw = 0; kw < 3; ++kw) {
                        const int iw = ow * stride + kw - 1;
                        const bool in = live && ih >= 0 && ih < Hi && iw >= 0 && iw < Wi;
                        xv[u][kh * 3 + kw] = in ? x[(((int64_t)n * Hi + ih) * Wi + iw) * ldx + cx] : 0.f;
                    }
                }
            }
            // same accumulation order as one row per trip: rows ascending within the thread
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int k = 0; k < 9; ++k) a[k] = fmaf(g[u], xv[u][k], a[k]);
        }
    }
#pragma unroll
    for (int k = 0; k < 9; ++k) sh[k][rl][threadIdx.x & 63] = a[k];
    __syncthreads();
    if (rl == 0 && col < C) {
        const int c = threadIdx.x & 63;
#pragma unroll
        for (int k = 0; k < 9; ++k) partials[((int64_t)rs * 9 + k) * C + col] = (sh[k][0][c] + sh[k][1][c]) + (sh[k][2][c] + sh[k][3][c]);
    }
}
// out[c*so_c + k*so_k] (+)= sum_blk partials[blk*blk_stride + k*C + c]
// 64 (k,c) columns x 16 block-lanes per workgroup: every lane sums nblk/16 partials with all its loads in flight, then the 16
// lane sums are added in a fixed order (deterministic). One thread per column walking all nblk partials was a 30 us serial chain.
__global__ __launch_bounds__(1024) void reduce_partials_kernel(const float* __restrict__ partials, int nblk, int blk_stride, int K, int C, float* __restrict__ out, int so_k,
                                                               int so_c, int accumulate) {
    __shared__ float sh[16][64];
    const int lane = threadIdx.x & 63, bl = threadIdx.x >> 6;
    const int idx = blockIdx.x * 64 + lane;
    float a = 0.f;
    if (idx < K * C) {
#pragma unroll 8
        for (int b = bl; b < nblk; b += 16) a += partials[(int64_t)b * blk_stride + idx];
    }
    sh[bl][lane] = a;
    __syncthreads();
    if (bl == 0 && idx < K * C) {
        float t = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) t += sh[q][lane];
        const int c = idx % C, k = idx / C;
        float* dst = out + (int64_t)c * so_c + (int64_t)k * so_k;
        *dst = accumulate ? *dst + t : t;
    }
}

// Front-end backward: MaxPool(1x3x3, s 2, p 1) argmax gather + PReLU + BN(eval) backward over the saved pre-PReLU map z (NF,Hc,Wc,24).
//   dy(pixel) = sum over the <= 4 pool windows that contain it and whose first maximum it is;  dz = dy * (z >= 0 ? 1 : slope)
//   dconv = dz * bn_scale;  partial sums per block: [0] sum dz, [1] sum dz (z - beta)/gamma, [2] sum dy * min(z, 0)
// One block = one frame x FB_PR pooled rows: the z rows its windows touch are staged in LDS ONCE (2*FB_PR + 3 conv rows), every window's first
// maximum is found by one thread per (window, channel), and the block's 2*FB_PR conv rows gather their <= 4 windows from that table.  (The
// first version walked the nine neighbours of each of a pixel's four windows in global memory behind data-dependent branches: 470 us for 232
// frames; this one streams z once.)  Per pixel the arithmetic - window order of the dy sum, dz, the three partial-sum terms - is the same.
constexpr int FB_PR = 4;
__host__ __device__ constexpr int fb_strips(int Hc) { return (Hc / 2 + FB_PR - 1) / FB_PR; }
template <int HC>
__global__ __launch_bounds__(256) void frontend_bwd_kernel(const float* __restrict__ z, const float* __restrict__ dpool, int NF, const float* __restrict__ slope,
                                                           const float* __restrict__ scale, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           float* __restrict__ dconv, float* __restrict__ partials) {
    constexpr int Hc = HC, Wc = HC, Hp = HC / 2, Wp = HC / 2, ROWS = 2 * FB_PR + 3, WR = FB_PR + 1;
    __shared__ float zs[ROWS * Wc * 24];                  // z rows r_lo .. r_lo + ROWS - 1 (rows outside the map are never read)
    __shared__ unsigned char win[WR * Wp * 24];           // per (window row, window col, channel): 3*dr + dc of the first maximum (dr, dc in 0..2)
    __shared__ float red[3][8][24];
    const int f = blockIdx.y, p0 = blockIdx.x * FB_PR;    // pooled rows p0 .. p0 + FB_PR (the extra one: windows of the block's last odd conv row)
    const int r_lo = 2 * p0 - 1;                          // conv row of tile row 0
    const int tid = threadIdx.x;
    const float* zf = z + (int64_t)f * Hc * Wc * 24;
    for (int i = tid; i < ROWS * Wc * 6; i += 256) {      // float4 = 4 channels
        const int row = i / (Wc * 6), q = i - row * (Wc * 6), r = r_lo + row;
        if (r >= 0 && r < Hc) *reinterpret_cast<float4*>(&zs[row * Wc * 24 + q * 4]) = *reinterpret_cast<const float4*>(zf + (int64_t)r * Wc * 24 + q * 4);
    }
    __syncthreads();
    for (int i = tid; i < WR * Wp * 24; i += 256) {
        const int ch = i % 24, pc = (i / 24) % Wp, wr = i / (24 * Wp), pr = p0 + wr;
        if (pr >= Hp) continue;
        const float sl = slope[ch];
        float best = 0.f; int code = -1;
#pragma unroll
        for (int dr = 0; dr < 3; ++dr) {
            const int rr = 2 * pr - 1 + dr;
            if (rr < 0 || rr >= Hc) continue;
#pragma unroll
            for (int dc = 0; dc < 3; ++dc) {
                const int cc = 2 * pc - 1 + dc;
                if (cc < 0 || cc >= Wc) continue;
                float v = zs[((rr - r_lo) * Wc + cc) * 24 + ch];
                v = v >= 0.f ? v : sl * v;
                if (code < 0 || v > best) { best = v; code = 3 * dr + dc; }     // strict: the FIRST maximum in row-major order wins
            }
        }
        win[i] = (unsigned char)code;
    }
    __syncthreads();
    const int ch = tid % 24, pl = tid / 24;               // 10 pixel lanes x 24 channels (threads 240..255 idle)
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
    if (pl < 10) {
        const float sl = slope[ch], sc = scale[ch], be = beta[ch], ig = 1.f / gamma[ch];
        for (int px = pl; px < 2 * FB_PR * Wc; px += 10) {
            const int lr = px / Wc, c = px - lr * Wc, r = 2 * p0 + lr;
            if (r >= Hc) break;
            const float zv = zs[((r - r_lo) * Wc + c) * 24 + ch];
            float dy = 0.f;
            const int pr0 = r >> 1, pr1 = (r & 1) ? pr0 + 1 : pr0;         // pool rows whose window holds conv row r
            const int pc0 = c >> 1, pc1 = (c & 1) ? pc0 + 1 : pc0;
            for (int pr = pr0; pr <= pr1; ++pr) {
                if (pr >= Hp) continue;
                for (int pc = pc0; pc <= pc1; ++pc) {
                    if (pc >= Wp) continue;
                    const int code = 3 * (r - (2 * pr - 1)) + (c - (2 * pc - 1));
                    if (win[((pr - p0) * Wp + pc) * 24 + ch] == code) dy += dpool[(((int64_t)f * Hp + pr) * Wp + pc) * 24 + ch];
                }
            }
            const float dz = zv >= 0.f ? dy : dy * sl;
            a0 += dz; a1 += dz * (zv - be) * ig; a2 += zv >= 0.f ? 0.f : dy * zv;
            dconv[(((int64_t)f * Hc + r) * Wc + c) * 24 + ch] = dz * sc;
        }
    }
    if (pl < 8) { red[0][pl][ch] = a0; red[1][pl][ch] = a1; red[2][pl][ch] = a2; }
    __syncthreads();
    if (pl >= 8 && pl < 10) { red[0][pl - 8][ch] += a0; red[1][pl - 8][ch] += a1; red[2][pl - 8][ch] += a2; }
    __syncthreads();
    if (pl == 0) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            float t = 0.f;
            for (int j = 0; j < 8; ++j) t += red[k][j][ch];
            partials[((int64_t)(blockIdx.y * gridDim.x + blockIdx.x) * 3 + k) * 24 + ch] = t;
        }
    }
}

// Conv3d weight gradient of the front-end as an implicit GEMM, no im2col (the explicit patch matrix was 1.6 GB written and read back
// per step at B=8): dW[co][slab][tap] = sum over conv pixels of dconv[p][co] * x[input pixel of (p, tap) in slab (ci, kt)].
// Block = frame x 12 conv rows, wave = 3 of those rows. v_mfma_f32_32x32x2_f32 with M = 24 channels (lanes 24..31 feed zeros),
// N = 49 taps (two 32-column tiles), K = the wave's 144 pixels: the dconv operand sits in registers for the whole block (72 values per
// lane), the input operand is read from the LDS slab with compile-time offsets along a row. Per slab the four waves' tiles are added
// through LDS and written as this block's partial [24][15][49]; reduce_partials_kernel adds the blocks in order.
constexpr int FD_CR = 12;                        // conv rows per block
constexpr int FD_XROWS = 2 * (FD_CR - 1) + 7;    // input rows of a slab
constexpr int FD_XLD = 104;                      // slab row stride (floats); input column x lives at x + 4
constexpr int FD_PART = 24 * 735;                // floats per block partial
template <int HW>
__global__ __launch_bounds__(256, 2) void frontend_dw_kernel(const float* __restrict__ video, const float* __restrict__ dconv, int T, float* __restrict__ partials) {
    constexpr int H = HW, W = HW, Hc = H / 2, Wc = W / 2, C2 = Wc / 2, XS = FD_XROWS * FD_XLD;
    __shared__ __attribute__((aligned(16))) float Xs[XS];
    __shared__ float red[4 * 2 * 1024];
    const int f = blockIdx.y, b = f / T, t = f - b * T;
    const int r0 = blockIdx.x * FD_CR;
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, lg = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* part = partials + (int64_t)(blockIdx.y * gridDim.x + blockIdx.x) * FD_PART;
    for (int i = tid; i < XS; i += 256) Xs[i] = 0.f;                       // column pads stay zero for every slab
    // the block's dconv pixels: lane (channel li, parity lg) keeps pixel (row, 2*c2 + lg) of its wave's three rows
    float areg[3][C2];
#pragma unroll
    for (int rr = 0; rr < 3; ++rr) {
        const int row = r0 + wave * 3 + rr;
#pragma unroll
        for (int c2 = 0; c2 < C2; ++c2)
            areg[rr][c2] = (li < 24 && row < Hc) ? dconv[(((int64_t)f * Hc + row) * Wc + 2 * c2 + lg) * 24 + li] : 0.f;
    }
    int off[2];                                                             // tap offset of this lane's column in each 32-tap tile
#pragma unroll
    for (int n = 0; n < 2; ++n) { const int tap = n * 32 + li; off[n] = tap < 49 ? (tap / 7) * FD_XLD + tap % 7 : 0; }
    const int gy0 = 2 * r0 - 3;                                             // frame row of slab row 0
    for (int slab = 0; slab < 15; ++slab) {
        const int ci = slab / 5, kt = slab - ci * 5, tt = t + kt - 2;
        if (tt < 0 || tt >= T) {                                            // temporal zero padding: this slab's weights get nothing from this frame
            for (int idx = tid; idx < 24 * 49; idx += 256) { const int co = idx / 49; part[co * 735 + slab * 49 + idx - co * 49] = 0.f; }
            continue;
        }
        __syncthreads();                                                    // previous slab (and its reduction) fully consumed
        const float* src = video + ((int64_t)(b * 3 + ci) * T + tt) * (H * W);
        for (int i = tid; i < FD_XROWS * (W / 4); i += 256) {
            const int row = i / (W / 4), q = i - row * (W / 4), gy = gy0 + row;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (gy >= 0 && gy < H) v = *reinterpret_cast<const float4*>(src + gy * W + 4 * q);
            *reinterpret_cast<float4*>(&Xs[row * FD_XLD + 4 + 4 * q]) = v;
        }
        __syncthreads();
        f32x16 acc[2];
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
#pragma unroll
        for (int rr = 0; rr < 3; ++rr) {
            // pixel (local row lr, column 2*c2 + lg): top-left tap at slab row 2*lr, column 2*(2*c2 + lg) - 3 + 4
            const float* xb0 = Xs + (2 * (wave * 3 + rr)) * FD_XLD + 2 * lg + 1 + off[0];
            const float* xb1 = Xs + (2 * (wave * 3 + rr)) * FD_XLD + 2 * lg + 1 + off[1];
#pragma unroll
            for (int c2 = 0; c2 < C2; ++c2) {
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(areg[rr][c2], xb0[4 * c2], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(areg[rr][c2], xb1[4 * c2], acc[1], 0, 0, 0);
            }
        }
        // C layout of the 32x32 tile: column (tap) = lane & 31, row (channel) = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) red[((wave * 2 + n) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lg) * 32 + li] = acc[n][r];
        __syncthreads();
        for (int idx = tid; idx < 24 * 49; idx += 256) {
            const int co = idx / 49, tap = idx - co * 49, n = tap >> 5, col = tap & 31;
            const float* q = red + (n * 32 + co) * 32 + col;
            part[co * 735 + slab * 49 + tap] = (q[0] + q[2048]) + (q[4096] + q[6144]);
        }
    }
}
__global__ __launch_bounds__(256) void copy2d_kernel(const float* __restrict__ src, int ld_src, int so, int cs, float* __restrict__ dst, int ld_dst, int dof, int64_t rows, int cols,
                                                     int accumulate) {
    // dst[r*ld_dst + dof + c] (+)= src[r*ld_src + so + c*cs]
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < rows * cols; i += (int64_t)gridDim.x * 256) {
        const int c = i % cols; const int64_t r = i / cols;
        const float v = src[r * ld_src + so + (int64_t)c * cs];
        float* d = dst + r * ld_dst + dof + c;
        *d = accumulate ? *d + v : v;
    }
}

// ------------------------------------------------------------------------------------------------------------ backward driver
static int64_t enc_bwd_ws_floats(int B, int T, int H) {
    const int64_t NF = (int64_t)B * T, Hc = H / 2, Hp = H / 4;
    const int64_t maxact = std::max<int64_t>(NF * Hp * Hp * 24, NF * ((Hp + 1) / 2) * ((Hp + 1) / 2) * 116);   // largest unit map
    const int64_t maxhalf = NF * Hp * Hp * 58;                                                                    // largest t1
    int64_t n = 2 * maxact + 4 * maxhalf + NF * 9 * LAST_CH * 2;      // dy/dx ping-pong; g, dt, gd, db1; dlast, gconv
    n += NF * Hc * Hc * 24;                                            // dconv of the front-end
    n += NF * ((Hc + FD_CR - 1) / FD_CR) * FD_PART;                    // per-block partials of the Conv3d weight gradient
    n += (int64_t)64 * LAST_CH * STAGE_CH[3] + (int64_t)24 * 736;      // split-K partials (largest: conv_last with <= 64 splits ... bounded below), dW staging
    n += (int64_t)DW_RS * 9 * 512 + (int64_t)AB_RS * 3 * 1024 + NF * fb_strips((int)Hc) * 3 * 24 + 2 * 1024;
    return n + 64 * 32;
}

static int encoder_train_bwd(l2s_model* m, const float* video, int B, int T, int H, int W, const float* dfeat, int ld_df, float* tape_base, void* ws, int64_t ws_bytes,
                             hipStream_t s) {
    const Weights& w = m->w;
    EncTape tp; enc_tape_layout(&tp, tape_base, B, T, H);
    const int NF = tp.NF, Hc = H / 2, Hp = H / 4;
    const std::string E = "encoder.";
    Bump bp(ws, ws_bytes);
    const int64_t maxact = std::max<int64_t>((int64_t)NF * Hp * Hp * 24, (int64_t)NF * ((Hp + 1) / 2) * ((Hp + 1) / 2) * 116);
    const int64_t maxhalf = (int64_t)NF * Hp * Hp * 58;
    float* dA = bp.f(maxact); float* dB = bp.f(maxact);
    float* g = bp.f(maxhalf); float* dt = bp.f(maxhalf); float* gd = bp.f(maxhalf); float* dt1 = bp.f(maxhalf);
    float* dlast = bp.f((int64_t)NF * 9 * LAST_CH); float* gconv = bp.f((int64_t)NF * 9 * LAST_CH);
    float* dconv = bp.f((int64_t)NF * Hc * Hc * 24);
    const int fd_strips = (Hc + FD_CR - 1) / FD_CR;
    float* fdp = bp.f((int64_t)NF * fd_strips * FD_PART);
    const int64_t splitk_cap = (int64_t)64 * LAST_CH * STAGE_CH[3];     // floats; 256 slices of a <= 232 x 232 gradient fit as well
    float* skp = bp.f(splitk_cap);
    float* dwp = bp.f((int64_t)DW_RS * 9 * 512); float* abp = bp.f((int64_t)AB_RS * 3 * 1024); float* fbp = bp.f((int64_t)NF * fb_strips(Hc) * 3 * 24);
    float* totals = bp.f(2 * 1024);
    L2S_REQUIRE(!bp.overflow, "encoder training backward workspace too small");
    auto G = [&](const std::string& k) { return m->grad(E + k); };
    auto Cn = [&](const std::string& k) { return m->canon(E + k); };

    auto dX = [&](const float* dz, int ldz, int nout, const float* Wf, float* out, int ldo, int cin, int64_t rows, bool acc) -> int {
        return launch_gemm_bwd(bwd_dx(dz, ldz, Wf, out, ldo, 1, (int)rows, (int)rows, nout, cin, 1, 0, acc), s, "train_bwd_encoder_dx");
    };
    // both gradients of a 1x1 conv in one launch: they read the same dz, are independent, and neither fills the chip at 8 clips per GPU
    auto dWX = [&](const float* dz, int ldz, int nout, const float* x, int ldx, int xcin, float* gw, const float* Wf, float* dxo, int ldo, int cin, int64_t rows) -> int {
        if (!gw) return dX(dz, ldz, nout, Wf, dxo, ldo, cin, rows, false);
        BwdGemmP pw = bwd_dw(dz, ldz, x, ldx, gw, 1, (int)rows, (int)rows, nout, xcin, 1, 1, 0, false);
        const int tiles = ((nout + 63) / 64) * ((xcin + 63) / 64);
        int splits = std::max(1, std::min(256, 2048 / tiles));        // 58-channel layers: one tile, 33 408 rows -> 256 slices of four K steps each
        while (splits > 1 && gemm_bwd_splitk_floats(pw, splits) > splitk_cap) --splits;
        return launch_gemm_bwd_dw_dx(pw, splits, skp, bwd_dx(dz, ldz, Wf, dxo, ldo, 1, (int)rows, (int)rows, nout, cin, 1, 0, false), s, "train_bwd_encoder_dw_dx");
    };
    // epilogue backward of "conv (+BN) (+ReLU)": dy/z may sit at shuffled channel positions
    const bool bnb = m->bn_batch;
    auto epi = [&](const float* dy, int ldy, int csy, int coy, const float* z, int ldz, int csz, int coz, float* out, int64_t rows, int C, int act, const float* scale,
                   const std::string& bn, int id) -> int {
        ActBwdP a{}; a.dy = dy; a.ld_dy = ldy; a.cs_dy = csy; a.co_dy = coy; a.z = z; a.ld_z = ldz; a.cs_z = csz; a.co_z = coz; a.dconv = out; a.ld_dconv = C;
        a.rows = rows; a.C = C; a.act = act; a.scale = bnb ? bn_scale(tp, id) : scale; a.gamma = Cn(bn + ".weight"); a.beta = Cn(bn + ".bias"); a.partials = abp;
        L2S_REQUIRE(a.gamma && a.beta, "encoder parameters not bound (l2s_train_bind)");
        if (act_bwd(a, G(bn + ".bias"), G(bn + ".weight"), nullptr, nullptr, false, s, bnb ? totals : nullptr)) return 1;
        if (bnb) return bn_train_fix(out, C, z, ldz, csz, coz, a.gamma, a.beta, a.scale, totals, rows, C, s);      // batch statistics depend on the input too
        return 0;
    };
    auto dwconv_bwd = [&](const float* gdz, const float* x, int ldx, int xoff, int hi, int ho, int C, int stride, const float* w9, float* dx, int ld_dx, int dxoff,
                          bool acc, float* gw) -> int {
        const int64_t total = (int64_t)NF * hi * hi * C;
        {
            ProfScope ps("train_bwd_dwconv_dx", s);
            L2S_REQUIRE(total < (int64_t)1 << 31 && (int64_t)NF * hi * hi * ld_dx < (int64_t)1 << 31, "depthwise backward: map too large for 32-bit indexing");
            const dim3 grid((unsigned)std::min<int64_t>((total + 255) / 256, 8192));
            if (stride == 2) hipLaunchKernelGGL(dwconv_bwd_dx_kernel<2>, grid, dim3(256), 0, s, gdz, NF, ho, ho, C, w9, dx, hi, hi, ld_dx, dxoff, acc ? 1 : 0);
            else hipLaunchKernelGGL(dwconv_bwd_dx_kernel<1>, grid, dim3(256), 0, s, gdz, NF, ho, ho, C, w9, dx, hi, hi, ld_dx, dxoff, acc ? 1 : 0);
        }
        if (gw) {
            ProfScope ps("train_bwd_dwconv_dw", s);
            hipLaunchKernelGGL(dwconv_bwd_dw_kernel, dim3((C + 63) / 64, DW_RS), dim3(256), 0, s, gdz, x, NF, hi, hi, ldx, xoff, ho, ho, C, stride, dwp);
            hipLaunchKernelGGL(reduce_partials_kernel, dim3((9 * C + 63) / 64), dim3(1024), 0, s, dwp, DW_RS, 9 * C, 9, C, gw, 1, 9, 0);     // canonical (C,1,3,3)
        }
        L2S_CHECK_HIP(hipGetLastError());
        return 0;
    };
    auto blocks = [](int64_t n) { return dim3((unsigned)std::min<int64_t>((n + 255) / 256, 8192)); };

    // ---- normalise + AvgPool + conv_last
    const int hl = tp.h[N_UNITS], P = hl * hl;
    const int64_t pxl = (int64_t)NF * P;
    {
        ProfScope ps("train_bwd_pool_norm", s);
        hipLaunchKernelGGL(pool_norm_bwd_kernel, dim3(NF), dim3(256), 0, s, tp.last, P, LAST_CH, dfeat, ld_df, dlast);
    }
    if (epi(dlast, LAST_CH, 1, 0, bnb ? tp.lastz : tp.last, LAST_CH, 1, 0, gconv, pxl, LAST_CH, ACT_RELU, w.conv_last.scale, "trunk.1.1", 81)) return 1;
    float* dy = dA; float* dx = dB;
    if (dWX(gconv, LAST_CH, LAST_CH, tp.x[N_UNITS], STAGE_CH[3], STAGE_CH[3], G("trunk.1.0.weight"), w.conv_last.W, dy, STAGE_CH[3], STAGE_CH[3], pxl)) return 1;

    // ---- ShuffleNet units, last to first
    for (int u = N_UNITS - 1; u >= 0; --u) {
        const UnitW& U = w.unit[u];
        const int half = U.half, cout = 2 * half, h = tp.h[u], ho = tp.h[u + 1];
        const int64_t in_px = (int64_t)NF * h * h, out_px = (int64_t)NF * ho * ho;
        const std::string p = "trunk.0." + std::to_string(u) + ".";
        const float* y = bnb ? tp.yz[u] : tp.x[u + 1];          // pre-ReLU values in batch-statistics mode, post-ReLU (same sign test) otherwise
        const float* t1v = bnb ? tp.t1z[u] : tp.t1[u];
        const int id = 1 + 5 * u;
        // banch2 tail: pw2 (+BN+ReLU) at the odd output channels, then the depthwise conv (+BN)
        if (epi(dy, cout, 2, 1, y, cout, 2, 1, g, out_px, half, ACT_RELU, U.pw2.scale, p + "banch2.6", id + 4)) return 1;
        if (dWX(g, half, half, tp.t2[u], half, half, G(p + "banch2.5.weight"), U.pw2.W, dt, half, half, out_px)) return 1;
        if (epi(dt, half, 1, 0, tp.t2[u], half, 1, 0, gd, out_px, half, ACT_NONE, U.dw.scale, p + "banch2.4", id + 3)) return 1;
        if (dwconv_bwd(gd, tp.t1[u], half, 0, h, ho, half, U.stride2 ? 2 : 1, U.dw.w9, dt1, half, 0, false, G(p + "banch2.3.weight"))) return 1;
        if (epi(dt1, half, 1, 0, t1v, half, 1, 0, g, in_px, half, ACT_RELU, U.pw1.scale, p + "banch2.1", id + 2)) return 1;
        if (U.stride2) {
            const int cin = U.cin;
            if (dWX(g, half, half, tp.x[u], cin, cin, G(p + "banch2.0.weight"), U.pw1.W, dx, cin, cin, in_px)) return 1;
            // banch1: dw (+BN) -> pw (+BN+ReLU) at the even output channels
            if (epi(dy, cout, 2, 0, y, cout, 2, 0, g, out_px, half, ACT_RELU, U.b1_pw.scale, p + "banch1.3", id + 1)) return 1;
            if (dWX(g, half, half, tp.b1[u], cin, cin, G(p + "banch1.2.weight"), U.b1_pw.W, dt, cin, cin, out_px)) return 1;
            if (epi(dt, cin, 1, 0, tp.b1[u], cin, 1, 0, gd, out_px, cin, ACT_NONE, U.b1_dw.scale, p + "banch1.1", id + 0)) return 1;
            if (dwconv_bwd(gd, tp.x[u], cin, 0, h, ho, cin, 2, U.b1_dw.w9, dx, cin, 0, true, G(p + "banch1.0.weight"))) return 1;
        } else {
            if (dWX(g, half, half, tp.x[u] + half, cout, half, G(p + "banch2.0.weight"), U.pw1.W, dx + half, cout, half, in_px)) return 1;
            // passthrough half: out[2k] = x1[k]
            hipLaunchKernelGGL(copy2d_kernel, blocks(in_px * half), dim3(256), 0, s, dy, cout, 0, 2, dx, cout, 0, in_px, half, 0);
        }
        std::swap(dy, dx);
    }

    // ---- front-end: MaxPool + PReLU + BN backward, then the Conv3d weight gradient
    {
        const float* gamma = Cn("frontend3D.1.weight"); const float* beta = Cn("frontend3D.1.bias");
        L2S_REQUIRE(gamma && beta, "encoder parameters not bound (l2s_train_bind)");
        ProfScope ps("train_bwd_frontend_pool_prelu_bn", s);
        const float* fscale = bnb ? bn_scale(tp, 0) : w.fe.scale;
        const int fb_blocks = NF * fb_strips(Hc);
        if (Hc == 48) hipLaunchKernelGGL(frontend_bwd_kernel<48>, dim3(fb_strips(Hc), NF), dim3(256), 0, s, tp.z0, dy, NF, w.fe.slope, fscale, gamma, beta, dconv, fbp);
        else hipLaunchKernelGGL(frontend_bwd_kernel<44>, dim3(fb_strips(Hc), NF), dim3(256), 0, s, tp.z0, dy, NF, w.fe.slope, fscale, gamma, beta, dconv, fbp);
        float* outs[3] = {G("frontend3D.1.bias"), G("frontend3D.1.weight"), G("frontend3D.2.weight")};
        for (int k = 0; k < 3; ++k) {
            float* dst = k < 2 ? totals + k * 24 : outs[2];                  // r0, r1 are needed by the batch-statistics correction as well
            if (dst) hipLaunchKernelGGL(reduce_partials_kernel, dim3(1), dim3(1024), 0, s, fbp + k * 24, fb_blocks, 72, 1, 24, dst, 0, 1, 0);
            if (k < 2 && outs[k]) L2S_CHECK_HIP(hipMemcpyAsync(outs[k], totals + k * 24, 24 * sizeof(float), hipMemcpyDeviceToDevice, s));
        }
        if (bnb) { if (bn_train_fix(dconv, 24, tp.z0, 24, 1, 0, gamma, beta, fscale, totals, (int64_t)NF * Hc * Hc, 24, s)) return 1; }
    }
    L2S_CHECK_HIP(hipGetLastError());
    if (float* gw = G("frontend3D.0.weight")) {
        {
            ProfScope ps("train_bwd_frontend_dw", s);
            if (H == 96) hipLaunchKernelGGL(frontend_dw_kernel<96>, dim3(fd_strips, NF), dim3(256), 0, s, video, dconv, T, fdp);
            else hipLaunchKernelGGL(frontend_dw_kernel<88>, dim3(fd_strips, NF), dim3(256), 0, s, video, dconv, T, fdp);
        }
        ProfScope ps("train_bwd_frontend_dw_reduce", s);
        hipLaunchKernelGGL(reduce_partials_kernel, dim3((FD_PART + 63) / 64), dim3(1024), 0, s, fdp, NF * fd_strips, FD_PART, 1, FD_PART, gw, 0, 1, 0);
    }
    L2S_CHECK_HIP(hipGetLastError());
    return 0;
}

}  // namespace l2s

// ================================================================================================ C ABI
using namespace l2s;

extern "C" {

int64_t l2s_train_encoder_tape_floats(int B, int T, int H) { return enc_tape_layout(nullptr, nullptr, B, T, H); }
int64_t l2s_train_encoder_ws_bytes(int B, int T, int H) { return (enc_bwd_ws_floats(B, T, H) + 1024) * (int64_t)sizeof(float); }

int l2s_train_encoder_fwd(l2s_model* m, const float* video, int B, int T, int H, int W, const float* emb, float* vis, float* feat, float* tape, void* stream) {
    L2S_REQUIRE(m && m->finalized && m->has_enc && video && tape && (vis || feat), "bad arguments");
    L2S_REQUIRE(B >= 1 && T >= 1 && H == W && (H == 96 || H == 88), "sizes");
    L2S_REQUIRE(!vis || emb, "the visual sequence needs the speaker embedding");
    Bf16Scope bf16scope(m->opt.train_bf16);      // GEMMs / Conv1d stacks with bf16 operands when the model asks for it (fp32 accumulation)
    return encoder_train_fwd(m, video, B, T, H, W, emb, vis, feat, tape, (hipStream_t)stream);
}

int l2s_train_encoder_bwd(l2s_model* m, const float* video, int B, int T, int H, int W, const float* dfeat, int ld_dfeat, float* tape, void* ws, int64_t ws_bytes,
                          void* stream) {
    L2S_REQUIRE(m && m->finalized && m->has_enc && video && dfeat && tape && ws, "bad arguments");
    L2S_REQUIRE(B >= 1 && T >= 1 && H == W && (H == 96 || H == 88) && ld_dfeat >= LAST_CH, "sizes");
    Bf16Scope bf16scope(m->opt.train_bf16);      // GEMMs / Conv1d stacks with bf16 operands when the model asks for it (fp32 accumulation)
    return encoder_train_bwd(m, video, B, T, H, W, dfeat, ld_dfeat, tape, ws, ws_bytes, (hipStream_t)stream);
}

}  // extern "C"
