// Visual-encoder kernels (reference/model/modules/video.py:68-87, shufflenetv2.py:42-152).
//
//  frontend3d_kernel : Conv3d(3->24, k 5x7x7, s 1x2x2, p 2x3x3) + BN(eval) + PReLU + MaxPool(1x3x3, s 1x2x2, p 0x1x1)
//                      fused, reading (B,3,T,H,W) frames with coalesced float4 row loads and writing the pooled
//                      map channel-last (B*T, H/4, W/4, 24) - the conv intermediate (205 MB at B=32) never
//                      leaves the CU.  Implicit GEMM on v_mfma_f32_32x32x2_f32: M = conv pixels of a strip,
//                      N = 24 channels (padded to 32), K = 15 (ci,kt) slabs x 49 taps (padded to 50).
//  dwconv3x3_kernel  : depthwise 3x3 (+BN) on channel-last maps.
//  copy_cols_kernel  : channel passthrough of the stride-1 ShuffleNet unit, writing the shuffled position.
//  pool_norm_cat     : AvgPool(3x3) + L2 normalise + concatenation with the tiled speaker embedding
//                      (video.py:81-85, model.py:52-55).
#include "l2s_common.h"

namespace l2s {

// ------------------------------------------------------------------------------------------------ frontend
constexpr int FE_PR = 6;                    // pooled rows per block
constexpr int FE_CR = 2 * FE_PR + 1;        // conv rows per block (one halo row above)
constexpr int FE_XROWS = 2 * (FE_CR - 1) + 7 + 1;   // input rows per slab (+1 spare zero row for the padded tap 49)
constexpr int FE_XLD = 104;                 // LDS row stride (floats); data column x lives at x+4
constexpr int FE_KP = 50;                   // taps per slab, padded (kh*7+kw; tap 49 has zero weight)
constexpr int FE_CO = 24;

template <int HW>
__global__ __launch_bounds__(256, 2) void frontend3d_kernel(const FrontendW w, const float* __restrict__ video,
                                                            int T, float* __restrict__ out) {
    constexpr int H = HW, W = HW, Hc = H / 2, Wc = W / 2, Hp = Hc / 2, Wp = Wc / 2;
    constexpr int P = FE_CR * Wc;                    // conv pixels per strip
    constexpr int NT = (P + 31) / 32;                // 32-pixel MFMA row tiles
    constexpr int TPW = (NT + 3) / 4;                // tiles per wave
    constexpr int XS = FE_XROWS * FE_XLD;            // floats in the input slab
    constexpr int WS = FE_KP * 32;                   // floats in the weight slab
    constexpr int CS = P * FE_CO;                    // floats in the conv tile (aliases the slabs)
    constexpr int SMEM = (XS + WS) > CS ? (XS + WS) : CS;
    __shared__ __attribute__((aligned(16))) float smem[SMEM];
    float* Xs = smem;
    float* Ws = smem + XS;

    const int f = blockIdx.y;                        // frame index b*T + t
    const int b = f / T, t = f - b * T;
    const int p0 = blockIdx.x * FE_PR;               // first pooled row of this strip
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int li = lane & 31, lg = lane >> 5;

    // zero the whole slab once: column pads (x+4 outside [4, W+4)) and the spare row stay zero forever
    for (int i = tid; i < XS; i += 256) Xs[i] = 0.f;

    // per-tile base address of this lane's conv pixel inside the slab: (2*lr)*XLD + 2*c + 1
    int base[TPW];
#pragma unroll
    for (int j = 0; j < TPW; ++j) {
        int p = (wave + 4 * j) * 32 + li;
        p = p < P ? p : P - 1;
        const int lr = p / Wc, c = p - lr * Wc;
        base[j] = (2 * lr) * FE_XLD + 2 * c + 1;
    }
    f32x16 acc[TPW];
#pragma unroll
    for (int j = 0; j < TPW; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

    const int gy0 = 4 * p0 - 5;                      // global input row of slab row 0
    for (int slab = 0; slab < 15; ++slab) {
        const int ci = slab / 5, kt = slab - ci * 5;
        const int tt = t + kt - 2;
        if (tt < 0 || tt >= T) continue;             // temporal zero padding: the slab contributes nothing
        __syncthreads();                             // previous slab fully consumed
        const float* src = video + ((int64_t)(b * 3 + ci) * T + tt) * (H * W);
        for (int i = tid; i < (FE_XROWS - 1) * (W / 4); i += 256) {
            const int row = i / (W / 4), q = i - row * (W / 4);
            const int gy = gy0 + row;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (gy >= 0 && gy < H) v = *reinterpret_cast<const float4*>(src + gy * W + 4 * q);
            *reinterpret_cast<float4*>(&Xs[row * FE_XLD + 4 + 4 * q]) = v;
        }
        for (int i = tid; i < WS / 4; i += 256)
            *reinterpret_cast<float4*>(&Ws[4 * i]) = *reinterpret_cast<const float4*>(w.w + (int64_t)slab * WS + 4 * i);
        __syncthreads();
#pragma unroll
        for (int s = 0; s < FE_KP / 2; ++s) {
            // k = 2s + lg; tap offsets are compile-time for both lane groups
            const int k0 = 2 * s, k1 = 2 * s + 1;
            const int off0 = (k0 / 7) * FE_XLD + (k0 % 7);
            const int off1 = (k1 / 7) * FE_XLD + (k1 % 7);
            const int off = lg ? off1 : off0;
            const float bw = Ws[(2 * s + lg) * 32 + li];
#pragma unroll
            for (int j = 0; j < TPW; ++j) {
                if (wave + 4 * j < NT) {             // wave-uniform
                    const float a = Xs[base[j] + off];
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bw, acc[j], 0, 0, 0);
                }
            }
        }
    }
    __syncthreads();                                 // slabs dead; reuse LDS as the conv tile

    // BN + PReLU, conv tile Cs[pixel][24]
    float* Cs = smem;
    if (li < FE_CO) {
        const float sc = w.scale[li], sh = w.shift[li], sl = w.slope[li];
#pragma unroll
        for (int j = 0; j < TPW; ++j) {
            if (wave + 4 * j < NT) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int p = (wave + 4 * j) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lg;
                    if (p < P) {
                        float v = acc[j][r] * sc + sh;
                        v = v >= 0.f ? v : sl * v;
                        Cs[p * FE_CO + li] = v;
                    }
                }
            }
        }
    }
    __syncthreads();

    // 3x3 / stride 2 / pad 1 max pool (padding never wins: -inf) -> channel-last output
    for (int i = tid; i < FE_PR * Wp * FE_CO; i += 256) {
        const int ch = i % FE_CO;
        const int pw = (i / FE_CO) % Wp;
        const int prl = i / (FE_CO * Wp);
        const int pr = p0 + prl;
        if (pr >= Hp) continue;
        float m = -INFINITY;
#pragma unroll
        for (int dr = 0; dr < 3; ++dr) {
            const int crow = 2 * pr - 1 + dr;        // global conv row
            if (crow < 0 || crow >= Hc) continue;
            const int lrow = 2 * prl + dr;           // local conv row (local row 0 = conv row 2*p0-1)
#pragma unroll
            for (int dc = -1; dc <= 1; ++dc) {
                const int cc = 2 * pw + dc;
                if (cc < 0 || cc >= Wc) continue;
                m = fmaxf(m, Cs[(lrow * Wc + cc) * FE_CO + ch]);
            }
        }
        out[(((int64_t)f * Hp + pr) * Wp + pw) * FE_CO + ch] = m;
    }
}

int launch_frontend(const FrontendW& w, const float* video, int B, int T, int H, int W, float* out, hipStream_t s) {
    L2S_REQUIRE(H == W && (H == 96 || H == 88), "frontend supports 96x96 and 88x88 mouth crops");
    L2S_REQUIRE((reinterpret_cast<uintptr_t>(video) & 15u) == 0, "video must be 16-byte aligned");
    const int Hp = H / 4;
    dim3 grid((Hp + FE_PR - 1) / FE_PR, B * T);
    ProfScope ps("frontend3d_conv_bn_prelu_pool", s);
    if (H == 96)
        hipLaunchKernelGGL(frontend3d_kernel<96>, grid, dim3(256), 0, s, w, video, T, out);
    else
        hipLaunchKernelGGL(frontend3d_kernel<88>, grid, dim3(256), 0, s, w, video, T, out);
    L2S_CHECK_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------------ depthwise 3x3
__global__ __launch_bounds__(256) void dwconv3x3_kernel(const float* __restrict__ in, int N, int Hi, int Wi, int ldi,
                                                        int ci_off, int C, int stride, const float* __restrict__ w9,
                                                        const float* __restrict__ scale, const float* __restrict__ shift,
                                                        float* __restrict__ out, int Ho, int Wo, int ldo, int co_off) {
    const int64_t total = (int64_t)N * Ho * Wo * C;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int c = idx % C;
        int64_t r = idx / C;
        const int ow = r % Wo;
        r /= Wo;
        const int oh = r % Ho;
        const int n = r / Ho;
        float acc = 0.f;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const int ih = oh * stride + kh - 1;
            if (ih < 0 || ih >= Hi) continue;
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int iw = ow * stride + kw - 1;
                if (iw < 0 || iw >= Wi) continue;
                acc = fmaf(in[(((int64_t)n * Hi + ih) * Wi + iw) * ldi + ci_off + c], w9[(kh * 3 + kw) * C + c], acc);
            }
        }
        out[(((int64_t)n * Ho + oh) * Wo + ow) * ldo + co_off + c] = acc * scale[c] + shift[c];
    }
}

int launch_dwconv(const float* in, int N, int Hi, int Wi, int ldi, int ci_off, int C, int stride, const float* w9,
                  const float* scale, const float* shift, float* out, int ldo, int co_off, hipStream_t s) {
    const int Ho = (Hi + 2 - 3) / stride + 1, Wo = (Wi + 2 - 3) / stride + 1;
    const int64_t total = (int64_t)N * Ho * Wo * C;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 8192) blocks = 8192;
    ProfScope ps("dwconv3x3_bn", s);
    hipLaunchKernelGGL(dwconv3x3_kernel, dim3(blocks), dim3(256), 0, s, in, N, Hi, Wi, ldi, ci_off, C, stride, w9, scale,
                       shift, out, Ho, Wo, ldo, co_off);
    L2S_CHECK_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------------ column copy
__global__ __launch_bounds__(256) void copy_cols_kernel(const float* __restrict__ in, int ldi, int off_i,
                                                        float* __restrict__ out, int ldo, int off_o, int cs_o,
                                                        int64_t rows, int cols) {
    const int64_t total = rows * cols;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int c = idx % cols;
        const int64_t r = idx / cols;
        out[r * ldo + off_o + (int64_t)c * cs_o] = in[r * ldi + off_i + c];
    }
}

int launch_copy_cols(const float* in, int ldi, int off_i, float* out, int ldo, int off_o, int cs_o, int64_t rows,
                     int cols, hipStream_t s) {
    const int64_t total = rows * cols;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 8192) blocks = 8192;
    if (blocks < 1) blocks = 1;
    ProfScope ps("copy_cols", s);
    hipLaunchKernelGGL(copy_cols_kernel, dim3(blocks), dim3(256), 0, s, in, ldi, off_i, out, ldo, off_o, cs_o, rows, cols);
    L2S_CHECK_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------------ pool + norm + cat
// one block per frame: x (P pixels, C channels) -> mean over P -> / max(||.||2, 1e-12) -> vis[f][0:C]; emb -> vis[f][C:C+E]
__global__ __launch_bounds__(256) void pool_norm_cat_kernel(const float* __restrict__ x, int P, int C,
                                                            const float* __restrict__ emb, int E, int T,
                                                            float* __restrict__ vis, int ldv, float* __restrict__ feat) {
    const int f = blockIdx.x, tid = threadIdx.x;
    __shared__ float red[4];
    float vals[4];                                   // C <= 1024
    float ss = 0.f;
    int cnt = 0;
    for (int c = tid; c < C; c += 256, ++cnt) {
        float acc = 0.f;
        for (int p = 0; p < P; ++p) acc += x[((int64_t)f * P + p) * C + c];
        acc = acc / (float)P;
        vals[cnt] = acc;
        ss += acc * acc;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
    if ((tid & 63) == 0) red[tid >> 6] = ss;
    __syncthreads();
    const float tot = red[0] + red[1] + red[2] + red[3];
    const float denom = fmaxf(sqrtf(tot), 1e-12f);
    cnt = 0;
    for (int c = tid; c < C; c += 256, ++cnt) {
        const float v = vals[cnt] / denom;
        if (vis) vis[(int64_t)f * ldv + c] = v;
        if (feat) feat[(int64_t)f * C + c] = v;
    }
    if (vis && emb) {
        const int b = f / T;
        for (int e = tid; e < E; e += 256) vis[(int64_t)f * ldv + C + e] = emb[(int64_t)b * E + e];
    }
}

int launch_pool_norm_cat(const float* x, int NF, int P, int C, const float* emb, int E, int T, float* vis, int ldv,
                         float* feat, hipStream_t s) {
    L2S_REQUIRE(C <= 1024, "pool_norm_cat: C <= 1024");
    ProfScope ps("avgpool_l2norm_cat", s);
    hipLaunchKernelGGL(pool_norm_cat_kernel, dim3(NF), dim3(256), 0, s, x, P, C, emb, E, T, vis, ldv, feat);
    L2S_CHECK_HIP(hipGetLastError());
    return 0;
}

}  // namespace l2s
