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
#include <type_traits>

namespace l2s {

// ------------------------------------------------------------------------------------------------ frontend
constexpr int FE_PR = 6;                    // pooled rows per block
constexpr int FE_CR = 2 * FE_PR + 1;        // conv rows per block (one halo row above)
constexpr int FE_XROWS = 2 * (FE_CR - 1) + 7 + 1;   // input rows per slab (+1 spare zero row for the padded tap 49)
constexpr int FE_XLD = 104;                 // LDS row stride (floats); data column x lives at x+4
constexpr int FE_KP = 50;                   // taps per slab, padded (kh*7+kw; tap 49 has zero weight)
constexpr int FE_CO = 24;

// MODE 0: inference; 1: also save the pre-PReLU map (training tape); 2: batch-statistics pass (training): nothing is stored but the
// per-channel sum / sum of squares of the RAW conv output over this block's own conv rows -> zout[(block*2 + k)*24 + ch]
template <int HW, int MODE>
__global__ __launch_bounds__(256, 3) void frontend3d_kernel(const FrontendW w, const FrameSrc vsrc,
                                                            int T, float* __restrict__ out, float* __restrict__ zout) {
    constexpr bool SAVE_Z = MODE == 1;
    constexpr int H = HW, W = HW, Hc = H / 2, Wc = W / 2, Hp = Hc / 2, Wp = Wc / 2;
    constexpr int P = FE_CR * Wc;                    // conv pixels per strip
    constexpr int NT = (P + 31) / 32;                // 32-pixel MFMA row tiles
    constexpr int TPW = (NT + 3) / 4;                // tiles per wave
    constexpr int XS = FE_XROWS * FE_XLD;            // floats in the input slab
    constexpr int WS = FE_KP * 32;                   // floats in the weight slab
    constexpr int CS = MODE == 2 ? 0 : P * (FE_CO / 2);    // floats in the conv tile of HALF the channels (aliases the slabs; pooling in two passes; none in the statistics pass)
    constexpr int SMEM = (XS + WS) > CS ? (XS + WS) : CS;  // 20 KB of operands or 30 KB of tile: three blocks per CU (the 24-channel tile held it at two)
    __shared__ __attribute__((aligned(16))) float smem[SMEM];
    float* Xs = smem;
    float* Ws = smem + XS;

    const int f = blockIdx.y;                        // frame index b*T + t
    const int bg = f / T, t = f - bg * T;
    const int grp = bg / vsrc.per, b = bg - grp * vsrc.per;     // clip b of the grp-th batch tensor (block-uniform)
    const float* __restrict__ video = vsrc.p[grp];
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

    if (MODE == 2) {
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int j = 0; j < TPW; ++j) {
            if (wave + 4 * j < NT) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int p = (wave + 4 * j) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lg;
                    const int lr = p / Wc, crow = 2 * p0 - 1 + lr;
                    if (p < P && lr >= 1 && crow < Hc) {                                                      // the halo row belongs to the strip above
                        s1 += acc[j][r]; s2 = fmaf(acc[j][r], acc[j][r], s2);       // explicit: the contraction must not depend on the code around it
                        // one pass over the conv: the raw map is parked where the tape keeps the pre-PReLU map; launch_bn_apply + frontend_pool finish it
                        if (out && li < FE_CO) out[(((int64_t)f * Hc + crow) * Wc + (p - lr * Wc)) * FE_CO + li] = acc[j][r];
                    }
                }
            }
        }
        float* red = smem;                                   // [2][wave 4][lg 2][32]
        red[((0 * 4 + wave) * 2 + lg) * 32 + li] = s1;
        red[((1 * 4 + wave) * 2 + lg) * 32 + li] = s2;
        __syncthreads();
        if (tid < 2 * FE_CO) {
            const int k = tid / FE_CO, ch = tid - k * FE_CO;
            float t = 0.f;
            for (int q = 0; q < 8; ++q) t += red[(k * 8 + q) * 32 + ch];
            zout[((int64_t)(blockIdx.y * gridDim.x + blockIdx.x) * 2 + k) * FE_CO + ch] = t;
        }
        return;
    }
    // BN + PReLU -> conv tile Cs[pixel][12] -> 3x3 / stride 2 / pad 1 max pool (padding never wins: -inf) -> channel-last output, twelve channels at a time
    float* Cs = smem;
    constexpr int CH = FE_CO / 2;
    const float sc = li < FE_CO ? w.scale[li] : 0.f, sh = li < FE_CO ? w.shift[li] : 0.f, sl = li < FE_CO ? w.slope[li] : 0.f;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        if (li >= half * CH && li < (half + 1) * CH) {
#pragma unroll
            for (int j = 0; j < TPW; ++j) {
                if (wave + 4 * j < NT) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int p = (wave + 4 * j) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lg;
                        if (p < P) {
                            float v = acc[j][r] * sc + sh;
                            if (SAVE_Z) {            // training tape: pre-PReLU map (B*T, H/2, W/2, 24); the halo row belongs to the strip above
                                const int lr = p / Wc, crow = 2 * p0 - 1 + lr;
                                if (lr >= 1 && crow < Hc) zout[(((int64_t)f * Hc + crow) * Wc + (p - lr * Wc)) * FE_CO + li] = v;
                            }
                            v = v >= 0.f ? v : sl * v;
                            Cs[p * CH + (li - half * CH)] = v;
                        }
                    }
                }
            }
        }
        __syncthreads();
        for (int i = tid; i < FE_PR * Wp * CH; i += 256) {
            const int ch = i % CH;
            const int pw = (i / CH) % Wp;
            const int prl = i / (CH * Wp);
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
                    m = fmaxf(m, Cs[(lrow * Wc + cc) * CH + ch]);
                }
            }
            out[(((int64_t)f * Hp + pr) * Wp + pw) * FE_CO + half * CH + ch] = m;
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------ frontend, split-bf16 matrix path
// The same fused Conv3d + BN + PReLU + MaxPool block (one frame x strip of 6 pooled rows) on the bf16 matrix cores: every fp32 input
// value and weight is split EXACTLY into three bf16 planes (x = hi + mid + lo, see gemm_x3.hip) and a conv product is the six partial
// products of weight >= 2^-16 - six v_mfma_f32_32x32x16_bf16 per 16 taps instead of eight v_mfma_f32_32x32x2_f32 per 16 taps at 2x the
// cost each.  K order per (ci,kt) slab: 4 steps of 16 = two kernel rows (kh = 2s + lane>>5) x 8 columns: a ZERO tap in front of the 7 real
// ones (the operand of a conv pixel is then the 8 consecutive bf16 starting at LDS column 2c, a 4-byte aligned address), and kernel row 7 is
// a zero row.  The input slab is split by the threads that stage it (once per slab per block), the weights arrive pre-split from the packer
// in operand order ([slab][step][plane][32 channels][16 taps, 48-byte rows: conflict-free ds_read_b128]).
// bf16 per LDS row: data column x at x + 4, zero columns in front and behind.  The row length in dwords is = Wc/2... chosen so that TWICE the
// row (one conv row down) shifts the banks by exactly Wc mod 32: a 32-pixel tile that straddles two conv rows then still touches 32
// different banks (96x96: 48 columns, 56 dwords per row; 88x88: 44 columns, 54 dwords) - with 52-dword rows 21 % of the LDS cycles were conflicts
template <int HW> struct FxGeom { static constexpr int XLD = HW == 96 ? 112 : 108, PLANE = FE_XROWS * XLD * 2; };
constexpr int FX_WROW = 48;                          // bytes per weight row (16 bf16 + pad)

typedef __bf16 fx_bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void fx_split4(const float4& v, uint2& hi, uint2& mid, uint2& lo) {
    const float f[4] = {v.x, v.y, v.z, v.w};
    unsigned h[4], m[4], l[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const unsigned xb = __float_as_uint(f[e]);
        const float r1 = f[e] - __uint_as_float(xb & 0xFFFF0000u);                          // exact
        const float r2 = r1 - __uint_as_float(__float_as_uint(r1) & 0xFFFF0000u);           // exact, <= 8 significant bits
        h[e] = xb; m[e] = __float_as_uint(r1); l[e] = __float_as_uint(r2);
    }
    hi = make_uint2(__builtin_amdgcn_perm(h[1], h[0], 0x07060302u), __builtin_amdgcn_perm(h[3], h[2], 0x07060302u));
    mid = make_uint2(__builtin_amdgcn_perm(m[1], m[0], 0x07060302u), __builtin_amdgcn_perm(m[3], m[2], 0x07060302u));
    lo = make_uint2(__builtin_amdgcn_perm(l[1], l[0], 0x07060302u), __builtin_amdgcn_perm(l[3], l[2], 0x07060302u));
}

// TERMS = 3: the exact split above.  TERMS = 1 (model option "infer_bf16", the bf16 leg): ONE plane - inputs rounded to nearest even by the
// staging threads, weights pre-rounded by the packer (FrontendW::w1) - and one MFMA per step: a sixth of the matrix work, a third of the LDS.
typedef __bf16 fx_bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned fx_rne2(float a, float b) { fx_bf16x2 v = {(__bf16)a, (__bf16)b}; return __builtin_bit_cast(unsigned, v); }

template <int HW, int TERMS>
__global__ __launch_bounds__(256, 3) void frontend3d_x3_kernel(const FrontendW w, const FrameSrc vsrc, int T, float* __restrict__ out) {
    constexpr int FX_WSLAB = 4 * TERMS * 32 * FX_WROW;   // bytes of weights per slab: 18 432 with three planes, 6 144 with one
    constexpr int H = HW, W = HW, Hc = H / 2, Wc = W / 2, Hp = Hc / 2, Wp = Wc / 2;
    constexpr int P = FE_CR * Wc;                    // conv pixels per strip
    constexpr int NT = (P + 31) / 32;                // 32-pixel MFMA row tiles
    constexpr int TPW = (NT + 3) / 4;                // tiles per wave
    constexpr int XLD = FxGeom<HW>::XLD, PLANE = FxGeom<HW>::PLANE;
    static_assert((2 * (XLD / 2)) % 32 == Wc % 32 && XLD >= W + 8, "row pitch: conflict-free straddling tiles");
    constexpr int XS = TERMS * PLANE;                // bytes: input planes
    constexpr int CS = P * (FE_CO / 2) * 4;          // bytes: conv tile of HALF the channels (aliases the operand area; the pooling runs in two passes)
    constexpr int SMEM = (XS + FX_WSLAB) > CS ? (XS + FX_WSLAB) : CS;
    constexpr int NLD = ((FE_XROWS - 1) * (W / 4) + 255) / 256;      // input float4 per thread per slab
    constexpr int NWL = (FX_WSLAB / 16 + 255) / 256;                 // weight uint4 per thread per slab
    __shared__ __attribute__((aligned(16))) unsigned char smem[SMEM];
    unsigned char* const Xs = smem;
    unsigned char* const Ws = smem + XS;

    const int f = blockIdx.y;                        // frame index b*T + t
    const int bg = f / T, t = f - bg * T;
    const int grp = bg / vsrc.per, b = bg - grp * vsrc.per;     // clip b of the grp-th batch tensor (block-uniform)
    const float* __restrict__ video = vsrc.p[grp];
    const int p0 = blockIdx.x * FE_PR;               // first pooled row of this strip
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int li = lane & 31, lg = lane >> 5;

    // zero the planes once: the column pads, the rows outside the image and the spare row stay zero for every slab
    for (int i = tid; i < XS / 16; i += 256) reinterpret_cast<uint4*>(Xs)[i] = make_uint4(0u, 0u, 0u, 0u);

    // byte offset of this lane's operand inside a plane, per tile: row 2*lr + (lane>>5), column 2*c
    int base[TPW];
#pragma unroll
    for (int j = 0; j < TPW; ++j) {
        int p = (wave + 4 * j) * 32 + li;
        p = p < P ? p : P - 1;
        const int lr = p / Wc, c = p - lr * Wc;
        base[j] = ((2 * lr + lg) * XLD + 2 * c) * 2;
    }
    f32x16 acc[TPW];
#pragma unroll
    for (int j = 0; j < TPW; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

    const int gy0 = 4 * p0 - 5;                      // global input row of slab row 0
    // the (ci,kt) slabs whose frame t+kt-2 exists (temporal zero padding contributes nothing), in order
    auto valid = [&](int slab) { const int tt = t + (slab % 5) - 2; return tt >= 0 && tt < T; };
    auto next_valid = [&](int slab) { while (slab < 15 && !valid(slab)) ++slab; return slab; };

    float4 rin[NLD];
    uint4 rw[NWL];
    auto fetch = [&](int slab) {
        const int ci = slab / 5, kt = slab - ci * 5;
        const float* src = video + ((int64_t)(b * 3 + ci) * T + (t + kt - 2)) * (H * W);
#pragma unroll
        for (int q = 0; q < NLD; ++q) {
            const int i = tid + 256 * q;
            const int row = i / (W / 4), x4 = i - row * (W / 4);
            const int gy = gy0 + row;
            rin[q] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (i < (FE_XROWS - 1) * (W / 4) && gy >= 0 && gy < H) rin[q] = *reinterpret_cast<const float4*>(src + gy * W + 4 * x4);
        }
        const uint4* wsrc = reinterpret_cast<const uint4*>(TERMS == 3 ? w.w3 : w.w1) + (int64_t)slab * (FX_WSLAB / 16);
#pragma unroll
        for (int q = 0; q < NWL; ++q) {
            const int i = tid + 256 * q;
            rw[q] = i < FX_WSLAB / 16 ? wsrc[i] : make_uint4(0u, 0u, 0u, 0u);
        }
    };
    auto stage = [&]() {
#pragma unroll
        for (int q = 0; q < NLD; ++q) {
            const int i = tid + 256 * q;
            if (i < (FE_XROWS - 1) * (W / 4)) {
                const int row = i / (W / 4), x4 = i - row * (W / 4);
                unsigned char* d = Xs + (row * XLD + 4 + 4 * x4) * 2;
                if constexpr (TERMS == 3) {
                    uint2 hi, mid, lo;
                    fx_split4(rin[q], hi, mid, lo);
                    *reinterpret_cast<uint2*>(d) = hi; *reinterpret_cast<uint2*>(d + PLANE) = mid; *reinterpret_cast<uint2*>(d + 2 * PLANE) = lo;
                } else {
                    *reinterpret_cast<uint2*>(d) = make_uint2(fx_rne2(rin[q].x, rin[q].y), fx_rne2(rin[q].z, rin[q].w));
                }
            }
        }
#pragma unroll
        for (int q = 0; q < NWL; ++q) {
            const int i = tid + 256 * q;
            if (i < FX_WSLAB / 16) reinterpret_cast<uint4*>(Ws)[i] = rw[q];
        }
    };

    int slab = next_valid(0);
    if (slab < 15) fetch(slab);
    while (slab < 15) {
        __syncthreads();                             // previous slab fully consumed (first pass: the zero fill is complete)
        stage();
        const int nxt = next_valid(slab + 1);
        if (nxt < 15) fetch(nxt);                    // lands under this slab's MFMAs
        __syncthreads();
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            // weights of this step: lane (channel li, tap half lg) reads 8 bf16 of each plane
            const unsigned char* wp = Ws + ((s * TERMS) * 32 + li) * FX_WROW + lg * 16;
            const fx_bf16x8 bh = *reinterpret_cast<const fx_bf16x8*>(wp);
            if constexpr (TERMS == 1) {
#pragma unroll
                for (int j = 0; j < TPW; ++j) {
                    if (wave + 4 * j < NT) {         // wave-uniform
                        const unsigned* ap = reinterpret_cast<const unsigned*>(Xs + base[j] + s * (2 * XLD * 2));
                        const fx_bf16x8 ah = __builtin_bit_cast(fx_bf16x8, make_uint4(ap[0], ap[1], ap[2], ap[3]));
                        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[j], 0, 0, 0);
                    }
                }
            } else {
            const fx_bf16x8 bm = *reinterpret_cast<const fx_bf16x8*>(wp + 32 * FX_WROW);
            const fx_bf16x8 bl = *reinterpret_cast<const fx_bf16x8*>(wp + 64 * FX_WROW);
#pragma unroll
            for (int j = 0; j < TPW; ++j) {
                if (wave + 4 * j < NT) {             // wave-uniform
                    const unsigned* ap = reinterpret_cast<const unsigned*>(Xs + base[j] + s * (2 * XLD * 2));
                    const unsigned* am_ = reinterpret_cast<const unsigned*>(Xs + base[j] + s * (2 * XLD * 2) + PLANE);
                    const unsigned* al_ = reinterpret_cast<const unsigned*>(Xs + base[j] + s * (2 * XLD * 2) + 2 * PLANE);
                    const fx_bf16x8 ah = __builtin_bit_cast(fx_bf16x8, make_uint4(ap[0], ap[1], ap[2], ap[3]));
                    const fx_bf16x8 am = __builtin_bit_cast(fx_bf16x8, make_uint4(am_[0], am_[1], am_[2], am_[3]));
                    const fx_bf16x8 al = __builtin_bit_cast(fx_bf16x8, make_uint4(al_[0], al_[1], al_[2], al_[3]));
                    // smallest partial products first (interleaving the terms across tiles measured 5 % slower: more operand registers live)
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[j], 0, 0, 0);
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[j], 0, 0, 0);
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, acc[j], 0, 0, 0);
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, acc[j], 0, 0, 0);
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, acc[j], 0, 0, 0);
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[j], 0, 0, 0);
                }
            }
            }
        }
        slab = nxt;
    }
    __syncthreads();                                 // operands dead; reuse LDS as the conv tile

    // BN + PReLU -> conv tile Cs[pixel][12] -> 3x3 / stride 2 / pad 1 max pool (padding never wins: -inf) -> channel-last output, twelve channels at a
    // time: the whole 24-channel tile (60 KB) was what kept the block at two per CU; with half of it the operand area (41 KB) sets the LDS size
    // and three blocks fit, i.e. one more block's MFMAs to cover another's staging.  Per output the arithmetic is unchanged.
    float* Cs = reinterpret_cast<float*>(smem);
    constexpr int CH = FE_CO / 2;
    const float sc = li < FE_CO ? w.scale[li] : 0.f, sh = li < FE_CO ? w.shift[li] : 0.f, sl = li < FE_CO ? w.slope[li] : 0.f;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        if (li >= half * CH && li < (half + 1) * CH) {
#pragma unroll
            for (int j = 0; j < TPW; ++j) {
                if (wave + 4 * j < NT) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int p = (wave + 4 * j) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lg;
                        if (p < P) {
                            float v = acc[j][r] * sc + sh;
                            v = v >= 0.f ? v : sl * v;
                            Cs[p * CH + (li - half * CH)] = v;
                        }
                    }
                }
            }
        }
        __syncthreads();
        for (int i = tid; i < FE_PR * Wp * CH; i += 256) {
            const int ch = i % CH;
            const int pw = (i / CH) % Wp;
            const int prl = i / (CH * Wp);
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
                    m = fmaxf(m, Cs[(lrow * Wc + cc) * CH + ch]);
                }
            }
            out[(((int64_t)f * Hp + pr) * Wp + pw) * FE_CO + half * CH + ch] = m;
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------ frontend, two output frames per block
// The same operator with the temporal reuse a Conv3d offers: input frame tau feeds output frame t through tap kt = tau - t + 2, so a block that
// produces TWO consecutive output frames (t0, t0 + 1) stages every input slab ONCE for both (6 slabs per input channel instead of 2 x 5) and lets
// their 2 x 24 output channels share the N axis: 48 = three 16-wide tiles of v_mfma_f32_16x16x32_bf16, exactly - where one frame's 24 channels
// pad a 32-wide tile by a third.  K step = 32 = four kernel rows x 8 column slots (rows 0-3, then rows 4-6 + the zero row).  Per input channel:
// 16 column tiles (edge slabs carry one frame: two tiles) where 15 is ideal; padded MFMA work 1.88x -> 1.5x of the algorithm.
// Column q of the N axis = (output frame o = q / 24, channel q % 24); the weight operand of a slab is assembled in LDS from the two temporal taps
// it serves - rows taken straight from the packed planes of frontend3d_x3_kernel ([slab][4 steps of two kernel rows][3 planes][32 co][48-byte
// rows]: kernel row r = 2 s' + half is the 16-byte half of step s'), an absent tap as zero rows.  120 accumulator registers: two blocks per CU.
// Accumulation order per output differs from the one-frame kernel only inside the MFMA (32 k per instruction instead of 16): rounding-level.
template <int HW>
__global__ __launch_bounds__(256, 2) void frontend3d_x3p_kernel(const FrontendW w, const FrameSrc vsrc, int T, float* __restrict__ out) {
    constexpr int H = HW, W = HW, Hc = H / 2, Wc = W / 2, Hp = Hc / 2, Wp = Wc / 2;
    constexpr int P = FE_CR * Wc;                    // conv pixels per strip
    constexpr int PT = (P + 15) / 16;                // 16-pixel MFMA row tiles
    constexpr int TPW = (PT + 3) / 4;                // tiles per wave
    // input row pitch 112 bf16 = 56 dwords: the A operand is read with ds_read2_b32 (32 banks, lanes 0-31 and 32-63 as groups), a group holds the 16
    // pixels of TWO k-groups - placed two input rows apart (112 dwords = 16 mod 32) they cover the 32 banks exactly; one row apart (24 mod 32) eight banks
    // were hit twice and every operand read cost double (SQ_LDS_BANK_CONFLICT: 24 % of the kernel's CU cycles)
    constexpr int XLD = 112, PLANE = FE_XROWS * XLD * 2;
    static_assert(XLD >= W + 8 && (XLD / 2) % 32 == 24, "row pitch");
    constexpr int XS = 3 * PLANE;                    // bytes: input planes
    constexpr int WROW = 48, LROW = 32;              // weight row in the packed planes (16 bf16 + pad) and in LDS (no pad: a third fewer DMA pieces)
    constexpr int WSP = 24 * LROW;                   // bytes per (step, plane) of one output frame: 24 channel rows
    constexpr int WSO = 12 * WSP;                    // bytes per output frame: 4 steps x 3 planes
    constexpr int WS = 2 * WSO;                      // 18 432
    constexpr int CS = P * (FE_CO / 2) * 4;          // conv tile of half the channels of one output frame (aliases the operand area)
    constexpr int SMEM = (XS + 2 * WS) > CS ? (XS + 2 * WS) : CS;     // two weight buffers: 58.4 KB, two blocks per CU
    constexpr int NLD = ((FE_XROWS - 1) * (W / 4) + 255) / 256;      // input float4 per thread per slab
    constexpr int NWU = WS / 16, NWC = NWU / 64;                    // weight uint4 per slab; 1-KB pieces of the weight operand
    static_assert(NWU % 64 == 0, "the weight operand is a whole number of wave-wide 16-byte pieces");
    constexpr int SRC_SP = 32 * WROW / 16;           // uint4 per (step, plane) in the packed source (32 channel rows)
    __shared__ __attribute__((aligned(16))) unsigned char smem[SMEM];
    unsigned char* const Xs = smem;
    unsigned char* const Ws0 = smem + XS;

    const int NPAIR = (T + 1) / 2;
    const int bg = blockIdx.y / NPAIR, t0 = 2 * (blockIdx.y - bg * NPAIR);
    const bool has1 = t0 + 1 < T;                    // block-uniform
    const int grp = bg / vsrc.per, b = bg - grp * vsrc.per;
    const float* __restrict__ video = vsrc.p[grp];
    const int p0 = blockIdx.x * FE_PR;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int li = lane & 15, kg = lane >> 4;
    const int kr = ((kg & 1) << 1) | (kg >> 1);       // kernel row (of the four of a K step) this lane's k-group carries: k-groups 0,1,2,3 = rows 0,2,1,3

    for (int i = tid; i < XS / 16; i += 256) reinterpret_cast<uint4*>(Xs)[i] = make_uint4(0u, 0u, 0u, 0u);

    int base[TPW];
#pragma unroll
    for (int j = 0; j < TPW; ++j) {
        int p = (wave + 4 * j) * 16 + li;
        p = p < P ? p : P - 1;
        const int lr = p / Wc, c = p - lr * Wc;
        base[j] = ((2 * lr + kr) * XLD + 2 * c) * 2;
    }
    // weight operand of column tile nt: column q = 16 nt + li -> (frame o, channel ch); this lane's kernel row inside a K step is kr
    int wof[3];
#pragma unroll
    for (int nt = 0; nt < 3; ++nt) {
        const int q = nt * 16 + li, o = q >= FE_CO ? 1 : 0, ch = q - FE_CO * o;
        wof[nt] = o * WSO + (kr >> 1) * (3 * WSP) + ch * LROW + (kr & 1) * 16;
    }
    f32x4 acc[TPW][3];
#pragma unroll
    for (int j = 0; j < TPW; ++j)
#pragma unroll
        for (int nt = 0; nt < 3; ++nt) acc[j][nt] = {0.f, 0.f, 0.f, 0.f};

    const int gy0 = 4 * p0 - 5;
    // slab sl = ci * 6 + d: input frame tau = t0 - 2 + d; frame t0 takes it through tap d (d <= 4), frame t0 + 1 through tap d - 1 (d >= 1)
    auto valid = [&](int sl) { const int d = sl % 6, tau = t0 - 2 + d; return tau >= 0 && tau < T && (d <= 4 || has1); };
    auto next_valid = [&](int sl) { while (sl < 18 && !valid(sl)) ++sl; return sl; };

    float4 rin[NLD];
    auto fetch_piece = [&](int sl, int q) {
        const int ci = sl / 6, d = sl - ci * 6;
        const float* src = video + ((int64_t)(b * 3 + ci) * T + (t0 - 2 + d)) * (H * W);
        const int i = tid + 256 * q;
        const int row = i / (W / 4), x4 = i - row * (W / 4);
        const int gy = gy0 + row;
        rin[q] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i < (FE_XROWS - 1) * (W / 4) && gy >= 0 && gy < H) rin[q] = *reinterpret_cast<const float4*>(src + gy * W + 4 * x4);
    };
    auto fetch = [&](int sl) {
#pragma unroll
        for (int q = 0; q < NLD; ++q) fetch_piece(sl, q);
    };
    // the weight operand of slab sl, straight into LDS buffer `buf` (global_load_lds: a wave's 64 lanes fill one contiguous 1-KB piece; no staging
    // registers, no ds_write pass) - issued at the head of the PREVIOUS slab's MFMA phase, landed by the barrier that ends it.
    auto dma_piece = [&](int sl, int buf, int kp) {                          // piece c = wave + 4 kp of the NWC 1-KB pieces
        const int c = wave + 4 * kp;
        if (c < NWC) {                                                       // wave-uniform
            const int ci = sl / 6, d = sl - ci * 6;
            const uint4* w3 = reinterpret_cast<const uint4*>(w.w3);
            unsigned char* const Wd = Ws0 + buf * WS;
            const int i = c * 64 + lane;
            const int o = i >= NWU / 2 ? 1 : 0, r = i - o * (NWU / 2);
            const int sp = r / (WSP / 16), u = r - sp * (WSP / 16);
            const int kt = d - o;                                             // the tap through which frame t0 + o sees this input frame
            const bool on = kt >= 0 && kt <= 4 && (o == 0 || has1);
            // a frame the slab does not feed takes its rows from a zero chunk of the packed planes (output-channel row 24 of 32 is padding: zeros) -
            // as ordinary zero stores those lanes cost an s_waitcnt vmcnt(0) each (a store to LDS behind a pending LDS-DMA), i.e. a full drain of
            // the requests in flight in the middle of the MFMA phase of every edge slab
            const uint4* src = on ? w3 + (int64_t)(ci * 5 + kt) * (12 * SRC_SP) + sp * SRC_SP + (u >> 1) * (WROW / 16) + (u & 1) : w3 + (24 * WROW) / 16;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(Wd + c * 1024), 16, 0, 0);
        }
    };
    constexpr int NDP = (NWC + 3) / 4;                                       // pieces per wave
    auto dma_weights = [&](int sl, int buf) {
#pragma unroll
        for (int kp = 0; kp < NDP; ++kp) dma_piece(sl, buf, kp);
    };
    auto stage = [&]() {
#pragma unroll
        for (int q = 0; q < NLD; ++q) {
            const int i = tid + 256 * q;
            if (i < (FE_XROWS - 1) * (W / 4)) {
                const int row = i / (W / 4), x4 = i - row * (W / 4);
                unsigned char* dd = Xs + (row * XLD + 4 + 4 * x4) * 2;
                uint2 hi, mid, lo;
                fx_split4(rin[q], hi, mid, lo);
                *reinterpret_cast<uint2*>(dd) = hi; *reinterpret_cast<uint2*>(dd + PLANE) = mid; *reinterpret_cast<uint2*>(dd + 2 * PLANE) = lo;
            }
        }
    };

    int sl = next_valid(0), nslab = 0;
    if (sl < 18) { fetch(sl); dma_weights(sl, 0); }
    while (sl < 18) {
        __syncthreads();                             // previous slab consumed; its successor's frame rows (registers) and weights (LDS) have landed
        stage();
        const int d = sl % 6;
        const bool on0 = d <= 4, on1 = d >= 1 && has1;       // block-uniform: which column tiles carry weights (tile 1 always does)
        const int nxt = next_valid(sl + 1);
        __syncthreads();
        // the next slab's weight pieces and frame rows are requested BETWEEN this slab's pixel tiles (one request per tile): a
        // request costs the wave 60-100 clk to issue, and ten of them in front of the first MFMA left the pipe idle that long per slab
        const bool more = nxt < 18;
        static_assert(NDP + NLD <= 2 * TPW, "one request per pixel tile");
        const unsigned char* const Ws = Ws0 + (nslab & 1) * WS;
#pragma unroll
        for (int S = 0; S < 2; ++S) {
            fx_bf16x8 bh[3], bm[3], bl[3];
#pragma unroll
            for (int nt = 0; nt < 3; ++nt) {
                const unsigned char* wp = Ws + wof[nt] + S * (2 * 3 * WSP);
                bh[nt] = *reinterpret_cast<const fx_bf16x8*>(wp);
                bm[nt] = *reinterpret_cast<const fx_bf16x8*>(wp + WSP);
                bl[nt] = *reinterpret_cast<const fx_bf16x8*>(wp + 2 * WSP);
            }
#pragma unroll
            for (int j = 0; j < TPW; ++j) {
                if (more) {
                    const int pos = S * TPW + j;                              // compile-time
                    if (pos < NDP) dma_piece(nxt, (nslab + 1) & 1, pos);
                    else if (pos < NDP + NLD) fetch_piece(nxt, pos - NDP);
                }
                {                                    // (no `wave + 4 j < PT` guard: the last wave's tile past the strip repeats its last pixel - base[] is
                                                     //  clamped - and is dropped by the epilogue; without the branch 1.955 -> 1.88 ms per 128 clips)
                    const unsigned* ap = reinterpret_cast<const unsigned*>(Xs + base[j] + S * (4 * XLD * 2));
                    const unsigned* am_ = reinterpret_cast<const unsigned*>(Xs + base[j] + S * (4 * XLD * 2) + PLANE);
                    const unsigned* al_ = reinterpret_cast<const unsigned*>(Xs + base[j] + S * (4 * XLD * 2) + 2 * PLANE);
                    const fx_bf16x8 ah = __builtin_bit_cast(fx_bf16x8, make_uint4(ap[0], ap[1], ap[2], ap[3]));
                    const fx_bf16x8 am = __builtin_bit_cast(fx_bf16x8, make_uint4(am_[0], am_[1], am_[2], am_[3]));
                    const fx_bf16x8 al = __builtin_bit_cast(fx_bf16x8, make_uint4(al_[0], al_[1], al_[2], al_[3]));
                    // one column tile after the other, its six partial products smallest first; the frame a slab does not feed is skipped under a
                    // block-uniform branch.  Tried: the three slab shapes (both frames / first only / second only) as three straight-line code paths with
                    // the terms interleaved across tiles - 213 spilled registers, 3.8 ms per 128 clips; all three tiles always, interleaved - no
                    // spills, 12 % more MFMA work: 2.32 ms against this form's 2.13
#pragma unroll
                    for (int nt = 0; nt < 3; ++nt) {
                        if ((nt == 0 && !on0) || (nt == 2 && !on1)) continue;       // block-uniform
                        f32x4 a = acc[j][nt];
                        a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh[nt], a, 0, 0, 0);
                        a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl[nt], a, 0, 0, 0);
                        a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bm[nt], a, 0, 0, 0);
                        a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bh[nt], a, 0, 0, 0);
                        a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bm[nt], a, 0, 0, 0);
                        a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh[nt], a, 0, 0, 0);
                        acc[j][nt] = a;
                    }
                }
            }
        }
        sl = nxt; ++nslab;
    }
    __syncthreads();

    // BN + PReLU -> conv tile Cs[pixel][12] -> 3x3 / stride 2 / pad 1 max pool -> channel-last output, twelve channels of one output frame at a time
    float* Cs = reinterpret_cast<float*>(smem);
    constexpr int CH = FE_CO / 2;
    float sc[3], sh[3], slp[3];
#pragma unroll
    for (int nt = 0; nt < 3; ++nt) {
        const int q = nt * 16 + li, ch = q >= FE_CO ? q - FE_CO : q;
        sc[nt] = w.scale[ch]; sh[nt] = w.shift[ch]; slp[nt] = w.slope[ch];
    }
    for (int pass = 0; pass < (has1 ? 4 : 2); ++pass) {
        const int o = pass >> 1, half = pass & 1;
#pragma unroll
        for (int nt = 0; nt < 3; ++nt) {
            const int q = nt * 16 + li, qo = q >= FE_CO ? 1 : 0, ch = q - FE_CO * qo;
            if (qo == o && ch >= half * CH && ch < (half + 1) * CH) {
#pragma unroll
                for (int j = 0; j < TPW; ++j) {
                    if (wave + 4 * j < PT) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int p = (wave + 4 * j) * 16 + 4 * kg + r;
                            if (p < P) {
                                float v = acc[j][nt][r] * sc[nt] + sh[nt];
                                v = v >= 0.f ? v : slp[nt] * v;
                                Cs[p * CH + (ch - half * CH)] = v;
                            }
                        }
                    }
                }
            }
        }
        __syncthreads();
        const int f = bg * T + t0 + o;
        for (int i = tid; i < FE_PR * Wp * CH; i += 256) {
            const int ch = i % CH;
            const int pw = (i / CH) % Wp;
            const int prl = i / (CH * Wp);
            const int pr = p0 + prl;
            if (pr >= Hp) continue;
            float m = -INFINITY;
#pragma unroll
            for (int dr = 0; dr < 3; ++dr) {
                const int crow = 2 * pr - 1 + dr;
                if (crow < 0 || crow >= Hc) continue;
                const int lrow = 2 * prl + dr;
#pragma unroll
                for (int dc = -1; dc <= 1; ++dc) {
                    const int cc = 2 * pw + dc;
                    if (cc < 0 || cc >= Wc) continue;
                    m = fmaxf(m, Cs[(lrow * Wc + cc) * CH + ch]);
                }
            }
            out[(((int64_t)f * Hp + pr) * Wp + pw) * FE_CO + half * CH + ch] = m;
        }
        __syncthreads();
    }
}

// PIPE (option "frontend_x3" = 3): the same block with the NEXT slab's staging (split into planes + LDS stores) interleaved between the MFMA groups of the
// current slab instead of in a phase of its own between two barriers: the input planes are double-buffered (2 x 21.5 KB: 79.9 KB of LDS, still two blocks
// per CU), the next slab's frame rows are requested at the HEAD of the MFMA phase (before its weight pieces) and split at its TAIL, when they have landed,
// into the other plane buffer - the split's VALU runs in the shadow of the MFMAs and a slab needs ONE barrier.  Same operand bits, same MFMA order: same
// bits out.
template <int HW>
__global__ __launch_bounds__(256, 2) void frontend3d_x3q_kernel(const FrontendW w, const FrameSrc vsrc, int T, float* __restrict__ out) {
    constexpr int H = HW, W = HW, Hc = H / 2, Wc = W / 2, Hp = Hc / 2, Wp = Wc / 2;
    constexpr int P = FE_CR * Wc;                    // conv pixels per strip
    constexpr int PT = (P + 15) / 16;                // 16-pixel MFMA row tiles
    constexpr int TPW = (PT + 3) / 4;                // tiles per wave
    // input row pitch 112 bf16 = 56 dwords: the A operand is read with ds_read2_b32 (32 banks, lanes 0-31 and 32-63 as groups), a group holds the 16
    // pixels of TWO k-groups - placed two input rows apart (112 dwords = 16 mod 32) they cover the 32 banks exactly; one row apart (24 mod 32) eight banks
    // were hit twice and every operand read cost double (SQ_LDS_BANK_CONFLICT: 24 % of the kernel's CU cycles)
    constexpr int XLD = 112, PLANE = FE_XROWS * XLD * 2;
    static_assert(XLD >= W + 8 && (XLD / 2) % 32 == 24, "row pitch");
    constexpr int XS = 3 * PLANE;                    // bytes: input planes
    constexpr int WROW = 48, LROW = 32;              // weight row in the packed planes (16 bf16 + pad) and in LDS (no pad: a third fewer DMA pieces)
    constexpr int WSP = 24 * LROW;                   // bytes per (step, plane) of one output frame: 24 channel rows
    constexpr int WSO = 12 * WSP;                    // bytes per output frame: 4 steps x 3 planes
    constexpr int WS = 2 * WSO;                      // 18 432
    constexpr int CS = P * (FE_CO / 2) * 4;          // conv tile of half the channels of one output frame (aliases the operand area)
    constexpr int SMEM = (2 * XS + 2 * WS) > CS ? (2 * XS + 2 * WS) : CS;     // two input-plane buffers + two weight buffers: 79.9 KB, two blocks per CU
    constexpr int NLD = ((FE_XROWS - 1) * (W / 4) + 255) / 256;      // input float4 per thread per slab
    constexpr int NWU = WS / 16, NWC = NWU / 64;                    // weight uint4 per slab; 1-KB pieces of the weight operand
    static_assert(NWU % 64 == 0, "the weight operand is a whole number of wave-wide 16-byte pieces");
    constexpr int SRC_SP = 32 * WROW / 16;           // uint4 per (step, plane) in the packed source (32 channel rows)
    __shared__ __attribute__((aligned(16))) unsigned char smem[SMEM];
    unsigned char* const Xs0 = smem;
    unsigned char* const Ws0 = smem + 2 * XS;

    const int NPAIR = (T + 1) / 2;
    const int bg = blockIdx.y / NPAIR, t0 = 2 * (blockIdx.y - bg * NPAIR);
    const bool has1 = t0 + 1 < T;                    // block-uniform
    const int grp = bg / vsrc.per, b = bg - grp * vsrc.per;
    const float* __restrict__ video = vsrc.p[grp];
    const int p0 = blockIdx.x * FE_PR;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int li = lane & 15, kg = lane >> 4;
    const int kr = ((kg & 1) << 1) | (kg >> 1);       // kernel row (of the four of a K step) this lane's k-group carries: k-groups 0,1,2,3 = rows 0,2,1,3

    for (int i = tid; i < 2 * XS / 16; i += 256) reinterpret_cast<uint4*>(Xs0)[i] = make_uint4(0u, 0u, 0u, 0u);      // both plane buffers: pads and outside rows stay zero

    int base[TPW];
#pragma unroll
    for (int j = 0; j < TPW; ++j) {
        int p = (wave + 4 * j) * 16 + li;
        p = p < P ? p : P - 1;
        const int lr = p / Wc, c = p - lr * Wc;
        base[j] = ((2 * lr + kr) * XLD + 2 * c) * 2;
    }
    // weight operand of column tile nt: column q = 16 nt + li -> (frame o, channel ch); this lane's kernel row inside a K step is kr
    int wof[3];
#pragma unroll
    for (int nt = 0; nt < 3; ++nt) {
        const int q = nt * 16 + li, o = q >= FE_CO ? 1 : 0, ch = q - FE_CO * o;
        wof[nt] = o * WSO + (kr >> 1) * (3 * WSP) + ch * LROW + (kr & 1) * 16;
    }
    f32x4 acc[TPW][3];
#pragma unroll
    for (int j = 0; j < TPW; ++j)
#pragma unroll
        for (int nt = 0; nt < 3; ++nt) acc[j][nt] = {0.f, 0.f, 0.f, 0.f};

    const int gy0 = 4 * p0 - 5;
    // slab sl = ci * 6 + d: input frame tau = t0 - 2 + d; frame t0 takes it through tap d (d <= 4), frame t0 + 1 through tap d - 1 (d >= 1)
    auto valid = [&](int sl) { const int d = sl % 6, tau = t0 - 2 + d; return tau >= 0 && tau < T && (d <= 4 || has1); };
    auto next_valid = [&](int sl) { while (sl < 18 && !valid(sl)) ++sl; return sl; };

    float4 rin[NLD];
    auto fetch_piece = [&](int sl, int q) {
        const int ci = sl / 6, d = sl - ci * 6;
        const float* src = video + ((int64_t)(b * 3 + ci) * T + (t0 - 2 + d)) * (H * W);
        const int i = tid + 256 * q;
        const int row = i / (W / 4), x4 = i - row * (W / 4);
        const int gy = gy0 + row;
        rin[q] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i < (FE_XROWS - 1) * (W / 4) && gy >= 0 && gy < H) rin[q] = *reinterpret_cast<const float4*>(src + gy * W + 4 * x4);
    };
    auto fetch = [&](int sl) {
#pragma unroll
        for (int q = 0; q < NLD; ++q) fetch_piece(sl, q);
    };
    // the weight operand of slab sl, straight into LDS buffer `buf` (global_load_lds: a wave's 64 lanes fill one contiguous 1-KB piece; no staging
    // registers, no ds_write pass) - issued at the head of the PREVIOUS slab's MFMA phase, landed by the barrier that ends it.
    auto dma_piece = [&](int sl, int buf, int kp) {                          // piece c = wave + 4 kp of the NWC 1-KB pieces
        const int c = wave + 4 * kp;
        if (c < NWC) {                                                       // wave-uniform
            const int ci = sl / 6, d = sl - ci * 6;
            const uint4* w3 = reinterpret_cast<const uint4*>(w.w3);
            unsigned char* const Wd = Ws0 + buf * WS;
            const int i = c * 64 + lane;
            const int o = i >= NWU / 2 ? 1 : 0, r = i - o * (NWU / 2);
            const int sp = r / (WSP / 16), u = r - sp * (WSP / 16);
            const int kt = d - o;                                             // the tap through which frame t0 + o sees this input frame
            const bool on = kt >= 0 && kt <= 4 && (o == 0 || has1);
            // a frame the slab does not feed takes its rows from a zero chunk of the packed planes (output-channel row 24 of 32 is padding: zeros) -
            // as ordinary zero stores those lanes cost an s_waitcnt vmcnt(0) each (a store to LDS behind a pending LDS-DMA), i.e. a full drain of
            // the requests in flight in the middle of the MFMA phase of every edge slab
            const uint4* src = on ? w3 + (int64_t)(ci * 5 + kt) * (12 * SRC_SP) + sp * SRC_SP + (u >> 1) * (WROW / 16) + (u & 1) : w3 + (24 * WROW) / 16;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(Wd + c * 1024), 16, 0, 0);
        }
    };
    constexpr int NDP = (NWC + 3) / 4;                                       // pieces per wave
    auto dma_weights = [&](int sl, int buf) {
#pragma unroll
        for (int kp = 0; kp < NDP; ++kp) dma_piece(sl, buf, kp);
    };
    auto stage_piece = [&](unsigned char* Xd, int q) {
        const int i = tid + 256 * q;
        if (i < (FE_XROWS - 1) * (W / 4)) {
            const int row = i / (W / 4), x4 = i - row * (W / 4);
            unsigned char* dd = Xd + (row * XLD + 4 + 4 * x4) * 2;
            uint2 hi, mid, lo;
            fx_split4(rin[q], hi, mid, lo);
            *reinterpret_cast<uint2*>(dd) = hi; *reinterpret_cast<uint2*>(dd + PLANE) = mid; *reinterpret_cast<uint2*>(dd + 2 * PLANE) = lo;
        }
    };

    int sl = next_valid(0), nslab = 0;
    if (sl < 18) {
        fetch(sl); dma_weights(sl, 0);
        __syncthreads();                             // the zero fill is complete
#pragma unroll
        for (int q = 0; q < NLD; ++q) stage_piece(Xs0, q);
    }
    static_assert(NLD + NDP + NLD <= 2 * TPW, "one request / one staging piece per pixel tile");
    while (sl < 18) {
        __syncthreads();                             // this slab's planes (staged under the previous slab's MFMAs) and weights (LDS-DMA) are in place; the previous slab is consumed
        const int d = sl % 6;
        const bool on0 = d <= 4, on1 = d >= 1 && has1;       // block-uniform: which column tiles carry weights (tile 1 always does)
        const int nxt = next_valid(sl + 1);
        const bool more = nxt < 18;
        const unsigned char* const Xs = Xs0 + (nslab & 1) * XS;
        unsigned char* const Xn = Xs0 + ((nslab + 1) & 1) * XS;
        const unsigned char* const Ws = Ws0 + (nslab & 1) * WS;
#pragma unroll
        for (int S = 0; S < 2; ++S) {
            fx_bf16x8 bh[3], bm[3], bl[3];
#pragma unroll
            for (int nt = 0; nt < 3; ++nt) {
                const unsigned char* wp = Ws + wof[nt] + S * (2 * 3 * WSP);
                bh[nt] = *reinterpret_cast<const fx_bf16x8*>(wp);
                bm[nt] = *reinterpret_cast<const fx_bf16x8*>(wp + WSP);
                bl[nt] = *reinterpret_cast<const fx_bf16x8*>(wp + 2 * WSP);
            }
#pragma unroll
            for (int j = 0; j < TPW; ++j) {
                if (more) {
                    const int pos = S * TPW + j;                              // compile-time
                    if (pos < NLD) fetch_piece(nxt, pos);                     // frame rows first: they are split at the tail of this phase
                    else if (pos < NLD + NDP) dma_piece(nxt, (nslab + 1) & 1, pos - NLD);
                    else if (pos >= 2 * TPW - NLD) stage_piece(Xn, pos - (2 * TPW - NLD));
                }
                {
                    const unsigned* ap = reinterpret_cast<const unsigned*>(Xs + base[j] + S * (4 * XLD * 2));
                    const unsigned* am_ = reinterpret_cast<const unsigned*>(Xs + base[j] + S * (4 * XLD * 2) + PLANE);
                    const unsigned* al_ = reinterpret_cast<const unsigned*>(Xs + base[j] + S * (4 * XLD * 2) + 2 * PLANE);
                    const fx_bf16x8 ah = __builtin_bit_cast(fx_bf16x8, make_uint4(ap[0], ap[1], ap[2], ap[3]));
                    const fx_bf16x8 am = __builtin_bit_cast(fx_bf16x8, make_uint4(am_[0], am_[1], am_[2], am_[3]));
                    const fx_bf16x8 al = __builtin_bit_cast(fx_bf16x8, make_uint4(al_[0], al_[1], al_[2], al_[3]));
#pragma unroll
                    for (int nt = 0; nt < 3; ++nt) {
                        if ((nt == 0 && !on0) || (nt == 2 && !on1)) continue;       // block-uniform
                        f32x4 a = acc[j][nt];
                        a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh[nt], a, 0, 0, 0);
                        a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl[nt], a, 0, 0, 0);
                        a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bm[nt], a, 0, 0, 0);
                        a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bh[nt], a, 0, 0, 0);
                        a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bm[nt], a, 0, 0, 0);
                        a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh[nt], a, 0, 0, 0);
                        acc[j][nt] = a;
                    }
                }
            }
        }
        sl = nxt; ++nslab;
    }
    __syncthreads();

    // BN + PReLU -> conv tile Cs[pixel][12] -> 3x3 / stride 2 / pad 1 max pool -> channel-last output, twelve channels of one output frame at a time
    float* Cs = reinterpret_cast<float*>(smem);
    constexpr int CH = FE_CO / 2;
    float sc[3], sh[3], slp[3];
#pragma unroll
    for (int nt = 0; nt < 3; ++nt) {
        const int q = nt * 16 + li, ch = q >= FE_CO ? q - FE_CO : q;
        sc[nt] = w.scale[ch]; sh[nt] = w.shift[ch]; slp[nt] = w.slope[ch];
    }
    for (int pass = 0; pass < (has1 ? 4 : 2); ++pass) {
        const int o = pass >> 1, half = pass & 1;
#pragma unroll
        for (int nt = 0; nt < 3; ++nt) {
            const int q = nt * 16 + li, qo = q >= FE_CO ? 1 : 0, ch = q - FE_CO * qo;
            if (qo == o && ch >= half * CH && ch < (half + 1) * CH) {
#pragma unroll
                for (int j = 0; j < TPW; ++j) {
                    if (wave + 4 * j < PT) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int p = (wave + 4 * j) * 16 + 4 * kg + r;
                            if (p < P) {
                                float v = acc[j][nt][r] * sc[nt] + sh[nt];
                                v = v >= 0.f ? v : slp[nt] * v;
                                Cs[p * CH + (ch - half * CH)] = v;
                            }
                        }
                    }
                }
            }
        }
        __syncthreads();
        const int f = bg * T + t0 + o;
        for (int i = tid; i < FE_PR * Wp * CH; i += 256) {
            const int ch = i % CH;
            const int pw = (i / CH) % Wp;
            const int prl = i / (CH * Wp);
            const int pr = p0 + prl;
            if (pr >= Hp) continue;
            float m = -INFINITY;
#pragma unroll
            for (int dr = 0; dr < 3; ++dr) {
                const int crow = 2 * pr - 1 + dr;
                if (crow < 0 || crow >= Hc) continue;
                const int lrow = 2 * prl + dr;
#pragma unroll
                for (int dc = -1; dc <= 1; ++dc) {
                    const int cc = 2 * pw + dc;
                    if (cc < 0 || cc >= Wc) continue;
                    m = fmaxf(m, Cs[(lrow * Wc + cc) * CH + ch]);
                }
            }
            out[(((int64_t)f * Hp + pr) * Wp + pw) * FE_CO + half * CH + ch] = m;
        }
        __syncthreads();
    }
}

int launch_frontend(const FrontendW& w, const FrameSrc& video, int B, int T, int H, int W, float* out, hipStream_t s, float* zout) {
    L2S_REQUIRE(H == W && (H == 96 || H == 88), "frontend supports 96x96 and 88x88 mouth crops");
    L2S_REQUIRE(video.per >= 1 && (B + video.per - 1) / video.per <= MAX_GROUP, "too many frame tensors in one launch");
    for (int g = 0; g < (B + video.per - 1) / video.per; ++g)
        L2S_REQUIRE(video.p[g] && (reinterpret_cast<uintptr_t>(video.p[g]) & 15u) == 0, "video must be non-null and 16-byte aligned");
    const int Hp = H / 4;
    dim3 grid((Hp + FE_PR - 1) / FE_PR, B * T);
    ProfScope ps("frontend3d_conv_bn_prelu_pool", s);
    if (!zout && w.w1) {                       // the bf16 leg (option "infer_bf16"): one bf16 plane
        if (H == 96) hipLaunchKernelGGL((frontend3d_x3_kernel<96, 1>), grid, dim3(256), 0, s, w, video, T, out);
        else hipLaunchKernelGGL((frontend3d_x3_kernel<88, 1>), grid, dim3(256), 0, s, w, video, T, out);
    } else if (!zout && w.w3 && w.pair) {      // inference on the split-bf16 matrix path, two output frames per block (option "frontend_x3" = 2)
        dim3 gp((Hp + FE_PR - 1) / FE_PR, B * ((T + 1) / 2));
        if (w.pipe) {                          // the next slab's staging interleaved with the MFMAs (option "frontend_x3" = 3): 79.9 KB of LDS, two blocks per CU
            // w.solo (diagnostic "frontend_solo"): a pad of dynamic LDS on top, so that only ONE block fits a CU and the other half of every CU (78 KB of
            // LDS, 256 registers per lane on one wave per SIMD) stays free for the step kernels of other launch chains
            const unsigned pad = w.solo ? 4096u : 0u;
            if (H == 96) hipLaunchKernelGGL((frontend3d_x3q_kernel<96>), gp, dim3(256), pad, s, w, video, T, out);
            else hipLaunchKernelGGL((frontend3d_x3q_kernel<88>), gp, dim3(256), pad, s, w, video, T, out);
        } else if (H == 96) hipLaunchKernelGGL((frontend3d_x3p_kernel<96>), gp, dim3(256), 0, s, w, video, T, out);
        else hipLaunchKernelGGL((frontend3d_x3p_kernel<88>), gp, dim3(256), 0, s, w, video, T, out);
    } else if (!zout && w.w3) {                // inference on the split-bf16 matrix path (option "frontend_x3")
        if (H == 96) hipLaunchKernelGGL((frontend3d_x3_kernel<96, 3>), grid, dim3(256), 0, s, w, video, T, out);
        else hipLaunchKernelGGL((frontend3d_x3_kernel<88, 3>), grid, dim3(256), 0, s, w, video, T, out);
    } else if (zout) {
        if (H == 96) hipLaunchKernelGGL((frontend3d_kernel<96, 1>), grid, dim3(256), 0, s, w, video, T, out, zout);
        else hipLaunchKernelGGL((frontend3d_kernel<88, 1>), grid, dim3(256), 0, s, w, video, T, out, zout);
    } else {
        if (H == 96) hipLaunchKernelGGL((frontend3d_kernel<96, 0>), grid, dim3(256), 0, s, w, video, T, out, zout);
        else hipLaunchKernelGGL((frontend3d_kernel<88, 0>), grid, dim3(256), 0, s, w, video, T, out, zout);
    }
    L2S_CHECK_HIP(hipGetLastError());
    return 0;
}

// PReLU + MaxPool(1,3,3)/s(1,2,2)/p(0,1,1) over the pre-PReLU map z (NF,Hc,Wc,24) -> (NF,Hc/2,Wc/2,24): the tail of the fused front-end as a
// kernel of its own (training, batch statistics: the conv ran once, in the statistics pass)
__global__ __launch_bounds__(256) void frontend_pool_kernel(const float* __restrict__ z, const float* __restrict__ slope, int NF, int Hc, int Wc, float* __restrict__ out) {
    const int Hp = Hc / 2, Wp = Wc / 2;
    const unsigned total = (unsigned)NF * Hp * Wp * FE_CO;
    for (unsigned idx = blockIdx.x * 256u + threadIdx.x; idx < total; idx += gridDim.x * 256u) {
        const int ch = idx % FE_CO;
        unsigned r = idx / FE_CO;
        const int pw = r % Wp; r /= Wp;
        const int pr = r % Hp;
        const unsigned f = r / Hp;
        const float sl = slope[ch];
        float m = -INFINITY;
#pragma unroll
        for (int dr = -1; dr <= 1; ++dr) {
            const int crow = 2 * pr + dr;
            if (crow < 0 || crow >= Hc) continue;
#pragma unroll
            for (int dc = -1; dc <= 1; ++dc) {
                const int cc = 2 * pw + dc;
                if (cc < 0 || cc >= Wc) continue;
                float v = z[((size_t)(f * Hc + crow) * Wc + cc) * FE_CO + ch];
                v = v >= 0.f ? v : sl * v;
                m = fmaxf(m, v);
            }
        }
        out[idx] = m;
    }
}
int launch_frontend_pool(const float* z, const float* slope, int NF, int Hc, int Wc, float* out, hipStream_t s) {
    const int64_t total = (int64_t)NF * (Hc / 2) * (Wc / 2) * FE_CO;
    L2S_REQUIRE(total < (int64_t)1 << 31, "front-end pool: map too large for 32-bit indexing");
    ProfScope ps("train_frontend_prelu_pool", s);
    hipLaunchKernelGGL(frontend_pool_kernel, dim3((unsigned)std::min<int64_t>((total + 255) / 256, 16384)), dim3(256), 0, s, z, slope, NF, Hc, Wc, out);
    L2S_CHECK_HIP(hipGetLastError());
    return 0;
}

int launch_frontend_stats(const FrontendW& w, const float* video1, int B, int T, int H, int W, float* partials, int* nblocks, hipStream_t s, float* raw_out) {
    L2S_REQUIRE(H == W && (H == 96 || H == 88), "frontend supports 96x96 and 88x88 mouth crops");
    FrameSrc video{}; video.p[0] = video1; video.per = B;
    const int Hp = H / 4;
    dim3 grid((Hp + FE_PR - 1) / FE_PR, B * T);
    *nblocks = (int)(grid.x * grid.y);
    ProfScope ps("train_frontend3d_stats", s);
    if (H == 96) hipLaunchKernelGGL((frontend3d_kernel<96, 2>), grid, dim3(256), 0, s, w, video, T, raw_out, partials);
    else hipLaunchKernelGGL((frontend3d_kernel<88, 2>), grid, dim3(256), 0, s, w, video, T, raw_out, partials);
    L2S_CHECK_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------------ data boundary: normalise + pad
// The host side of the reference turns every decoded clip into fp32 before it reaches the model: `im.float() / 255.0`, then
// `Normalize(mean, std)` per channel (datasets/lrw/dataset.py:83-86), then the collate zero-pads every clip to the batch's longest and
// permutes to (B,3,T,H,W) (datasets/__init__.py:7-46) - 102.6 MB of fp32 per B=32 batch built by the CPU and pushed over PCIe.  Here the
// decoded uint8 frames (25.7 MB) cross the bus and ONE kernel does the rest, with the same three fp32 operations in the same order
// (IEEE division, no fma contraction), so the batch is bit-identical to the host collate's.
struct ClipTab { int64_t off[MAX_COLLATE_CLIPS]; int frames[MAX_COLLATE_CLIPS]; };

__global__ __launch_bounds__(256) void normalise_pad_kernel(const uint8_t* __restrict__ packed, const ClipTab tab, int b0, int T, int HW4,
                                                            float* __restrict__ video) {
    // block (x: chunks of 256 pixel quads of one frame, y: frame t, z: clip); thread = 4 consecutive pixels = 12 bytes in, 3 x float4 out
    const int q = blockIdx.x * 256 + threadIdx.x, t = blockIdx.y, bl = blockIdx.z;
    if (q >= HW4) return;
    const int b = b0 + bl;
    const int64_t plane = (int64_t)HW4 * 4;
    float* out = video + (((int64_t)b * 3) * T + t) * plane + (int64_t)q * 4;
    float4 r = make_float4(0.f, 0.f, 0.f, 0.f), g = r, bl4 = r;
    if (t < tab.frames[bl]) {
        const uint32_t* src = reinterpret_cast<const uint32_t*>(packed + tab.off[bl] + ((int64_t)t * plane + (int64_t)q * 4) * 3);
        const uint32_t w0 = src[0], w1 = src[1], w2 = src[2];                 // R0 G0 B0 R1 | G1 B1 R2 G2 | B2 R3 G3 B3
        const float mean[3] = {0.485f, 0.456f, 0.406f}, stdv[3] = {0.229f, 0.224f, 0.225f};
        auto nrm = [&](uint32_t byte, int c) { return __fdiv_rn(__fsub_rn(__fdiv_rn((float)byte, 255.0f), mean[c]), stdv[c]); };
        r = make_float4(nrm(w0 & 255u, 0), nrm(w0 >> 24, 0), nrm((w1 >> 16) & 255u, 0), nrm((w2 >> 8) & 255u, 0));
        g = make_float4(nrm((w0 >> 8) & 255u, 1), nrm(w1 & 255u, 1), nrm(w1 >> 24, 1), nrm((w2 >> 16) & 255u, 1));
        bl4 = make_float4(nrm((w0 >> 16) & 255u, 2), nrm((w1 >> 8) & 255u, 2), nrm(w2 & 255u, 2), nrm(w2 >> 24, 2));
    }
    *reinterpret_cast<float4*>(out) = r;
    *reinterpret_cast<float4*>(out + (int64_t)T * plane) = g;
    *reinterpret_cast<float4*>(out + 2 * (int64_t)T * plane) = bl4;
}

int launch_normalise_pad(const uint8_t* packed, const int64_t* offsets, const int* frames, int B, int T, int H, int W, float* video, hipStream_t s) {
    L2S_REQUIRE((H * W) % 4 == 0 && (reinterpret_cast<uintptr_t>(packed) & 3u) == 0 && (reinterpret_cast<uintptr_t>(video) & 15u) == 0,
                "frames: H*W must be a multiple of 4, buffers aligned");
    const int HW4 = H * W / 4;
    ProfScope ps("normalise_pad_frames", s);
    for (int b0 = 0; b0 < B; b0 += MAX_COLLATE_CLIPS) {
        const int nb = B - b0 < MAX_COLLATE_CLIPS ? B - b0 : MAX_COLLATE_CLIPS;
        ClipTab tab{};
        for (int i = 0; i < nb; ++i) {
            L2S_REQUIRE(frames[b0 + i] >= 0 && frames[b0 + i] <= T && offsets[b0 + i] % 4 == 0, "clip table: 0 <= frames <= T, 4-byte aligned offsets");
            tab.off[i] = offsets[b0 + i]; tab.frames[i] = frames[b0 + i];
        }
        hipLaunchKernelGGL(normalise_pad_kernel, dim3((HW4 + 255) / 256, T, nb), dim3(256), 0, s, packed, tab, b0, T, HW4, video);
    }
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

// ------------------------------------------------------------------------------------------------ fused stride-1 unit
// One ShuffleNetV2 stride-1 unit (shufflenetv2.py:53-65,96-104) in ONE launch: for F frames per block
//   x2 -> LDS -> pw1 (MFMA, +BN+ReLU) -> LDS -> dw3x3 (+BN) -> LDS -> pw2 (MFMA, +BN+ReLU) -> LDS -> interleaved store with the
//   x1 passthrough (channel_shuffle: out[2k] = x1[k], out[2k+1] = branch[k]).
// The unit reads x once and writes out once (the unfused form moved 6 activation-sized tensors through HBM).
// GEMMs: M = F*h*h pixels (16-row tiles), N = K = half channels; A fragments from LDS with the K-permuted
// ds_read_b128, B fragments straight from L2 in the frag16 layout (weights of stage 4 do not fit LDS).

__device__ __forceinline__ f32x4 su_mfma4(const float4& a, const float4& w, f32x4 acc) {
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, w.x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, w.y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, w.z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, w.w, acc, 0, 0, 0);
    return acc;
}

// Every extent of the fused units is a template constant (stage geometry is fixed by the architecture): with run-time
// extents the kernels were bound by the SCALAR unit - 1 200-2 800 s_ instructions per wave of software integer division and
// index arithmetic against 300-600 VALU/MFMA (rocprofv3 counters, profiles/r01_fused_units_pmc.txt).
constexpr int su_pad16(int v) { return (v + 15) / 16 * 16; }
constexpr int su_group(int mt) { return (mt + (mt + 4) / 5 - 1) / ((mt + 4) / 5); }   // row tiles per work item: <= 5, groups as even as possible

// One pointwise conv of a fused unit as a GEMM over an LDS tile: N columns, K = NC*16 (zero padded), MT 16-row tiles.
template <int N_, int NC_, int MT_>
struct SuGemm {
    static constexpr int N = N_, NC = NC_, MT = MT_, G = su_group(MT_);
    static constexpr int NT = (N + 15) / 16, MG = (MT + G - 1) / G, ITEMS = NT * MG, IT = (ITEMS + 7) / 8;
};

// buf[m][n] = relu((buf[m][:] . W[n][:]) * scale[n] + shift[n]) for all 16-row tiles of the block, IN PLACE: every wave
// keeps the accumulators of its (<= IT) work items in registers across a block barrier, so one LDS buffer serves as both
// operand and result. Work item = (column tile, group of G row tiles); the last group is shifted back so every item
// has exactly G tiles (an overlapped tile is computed twice with identical results) - no data-dependent guards around
// the MFMAs. One buffer instead of two is what lets two blocks share a CU.
template <class GM, int LDA>
__device__ __forceinline__ void su_gemm(float* __restrict__ buf, const float* __restrict__ Wf,
                                        const float* __restrict__ scale, const float* __restrict__ shift) {
    constexpr int NC = GM::NC, G = GM::G, IT = GM::IT, NT = GM::NT;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int li = lane & 15, lg = lane >> 4;
    f32x4 acc[IT][G];
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const int item = wave + 8 * it;
#pragma unroll
        for (int j = 0; j < G; ++j) acc[it][j] = {0.f, 0.f, 0.f, 0.f};
        if (item < GM::ITEMS) {
            const int nt = item % NT, mg = item / NT;
            const int mt0 = min(mg * G, GM::MT - G);
            const float4* wb = reinterpret_cast<const float4*>(Wf) + nt * (NC * 64) + lane;
            const float* ab = buf + (mt0 * 16 + li) * LDA + 4 * lg;
            float4 b4[NC];                               // the whole K strip of this column tile: one round trip to L2
#pragma unroll
            for (int c = 0; c < NC; ++c) b4[c] = wb[c * 64];
#pragma unroll
            for (int c = 0; c < NC; ++c) {
#pragma unroll
                for (int j = 0; j < G; ++j) {
                    const float4 a4 = *reinterpret_cast<const float4*>(ab + j * 16 * LDA + 16 * c);
                    acc[it][j] = su_mfma4(a4, b4[c], acc[it][j]);
                }
                __builtin_amdgcn_sched_barrier(0);      // keep the LDS operand reads of later chunks from being hoisted (VGPRs)
            }
        }
    }
    __syncthreads();                                     // every wave has read its operand rows
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const int item = wave + 8 * it;
        if (item < GM::ITEMS) {
            const int nt = item % NT, mg = item / NT;
            const int mt0 = min(mg * G, GM::MT - G);
            const int n = nt * 16 + li;
            if (n < GM::N) {
                const float sc = scale[n], sh = shift[n];
#pragma unroll
                for (int j = 0; j < G; ++j) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float v = acc[it][j][r] * sc + sh;
                        buf[((mt0 + j) * 16 + 4 * lg + r) * LDA + n] = v > 0.f ? v : 0.f;
                    }
                }
            }
        }
    }
    __syncthreads();
}

// Fused stride-1 unit: H = spatial size of the stage, HALF = channels per branch, F = frames per block.
template <int H, int HALF, int F>
struct S1Geo {
    static constexpr int HH = H * H, C = 2 * HALF, KP = su_pad16(HALF), LDA = KP + 4, PX = F * HH;
    static constexpr int MT = (PX + 15) / 16, ROWS = MT * 16, PPW = (PX + 7) / 8, CIT = (C + 63) / 64, CH = (HALF + 63) / 64;
    using GM = SuGemm<HALF, KP / 16, MT>;
    static constexpr size_t SMEM = (size_t)ROWS * LDA * sizeof(float);
};

// measurement hook (tools/fused_unit_timeline.py): non-null -> the stamped build; thread 0 of every block writes 8 x 64-bit stamps
// of the 100 MHz wall clock plus its XCC / CU id
#ifdef L2S_DIAG      // libl2s_diag.so only (include/l2s_diag.h l2s_op_fused_unit_timeline); the product launches TIMED = false and never instantiates the stamped builds
static unsigned long long* g_su_ts = nullptr;
static int g_su_ts_h = 0;                                 // stamp only the units of this spatial size
void shuffle_set_timeline(unsigned long long* ts, int h) { g_su_ts = ts; g_su_ts_h = h; }
#endif
#define SU_STAMP(k) do { if (TIMED && threadIdx.x == 0) ts[blockIdx.x * 10 + (k)] = wall_clock64(); } while (0)

template <int H, int HALF, int F, bool TIMED>
__global__ __launch_bounds__(512, 4) void shuffle_s1_kernel(const ShuffleS1P p, unsigned long long* __restrict__ ts) {
    using Q = S1Geo<H, HALF, F>;
    constexpr int HH = Q::HH, C = Q::C, KP = Q::KP, LDA = Q::LDA, ROWS = Q::ROWS, PPW = Q::PPW, CIT = Q::CIT;
    extern __shared__ __attribute__((aligned(16))) float su_smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    SU_STAMP(0);
    const int f0 = blockIdx.x * F;
    const int Mv = min(F, p.NF - f0) * HH;                // valid pixel rows of this block
    float* buf = su_smem;
    const float* xb = p.x + (int64_t)f0 * HH * C;
    float* ob = p.out + (int64_t)f0 * HH * C;

    // phase 0: the block's whole input, ONE round of loads: wave -> pixels m = wave + 8*i, lane -> channels c = lane + 64*j.
    // Buffer loads (T8): scalar descriptor + scalar per-pixel offset + one per-lane byte offset per j, so the 36-40 loads in
    // flight cost no address VGPRs, and the descriptor's byte count zero-fills everything past the block's valid pixels.
    // The second channel half goes to LDS (GEMM operand); the first half (passthrough) stays in registers.
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xb), 0, Mv * C * 4, 0x00020000);
    int voff[CIT];
#pragma unroll
    for (int j = 0; j < CIT; ++j) voff[j] = (lane + 64 * j < C) ? (lane + 64 * j) * 4 : 0x7ffffff0;   // past the end: reads 0
    float xr[PPW][CIT];
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        const int soff = (wave + 8 * i) * C * 4;
#pragma unroll
        for (int j = 0; j < CIT; ++j)
            xr[i][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, voff[j], soff, 0));
    }
    // zero the K padding columns and the padded rows (the GEMMs read them; 0 * weight-padding must stay 0)
    if (KP > HALF) {
        for (int idx = tid; idx < ROWS * (KP - HALF); idx += 512) {
            const int m = idx / (KP - HALF), k = HALF + idx - m * (KP - HALF);
            buf[m * LDA + k] = 0.f;
        }
    }
    for (int idx = tid; idx < (ROWS - Mv) * HALF; idx += 512) {
        const int m = Mv + idx / HALF, k = idx - (m - Mv) * HALF;
        buf[m * LDA + k] = 0.f;
    }
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        const int m = wave + 8 * i;
#pragma unroll
        for (int j = 0; j < CIT; ++j) {
            const int c = lane + 64 * j;
            if (m < Mv && c >= HALF && c < C) buf[m * LDA + (c - HALF)] = xr[i][j];
        }
    }
    SU_STAMP(1);                                          // input landed, LDS written
    __syncthreads();
    SU_STAMP(2);
    // phase 1: pw1 + BN + ReLU, in place
    su_gemm<typename Q::GM, LDA>(buf, p.w1f, p.s1, p.b1);
    SU_STAMP(3);
    // phase 2: depthwise 3x3 (pad 1) + BN, in place.  A wave takes whole pixel ROWS (row R = wave + 8*r over the block's F*H rows) and slides a 3x3
    // window along x: every input value of the three rows is read from LDS ONCE (3 reads per output instead of 9), the x borders are compile-time
    // (their taps are simply not issued), the y borders are two wave-uniform flags that zero the upper / lower tap weights.  Per output the FMA chain
    // is the tap-ordered one of the pixel-per-wave form it replaces: a skipped or zero-weight tap adds exactly +-0 to an accumulator that is never -0,
    // so the results are the same bits.  The CU is bound by VALU + MFMA issue over its two blocks (phase timelines: 16-byte input loads / output stores,
    // a quarter of the instructions, cut the input phase from 5.9 to 3.3 us and only moved the wait - block lifetime 23.2 -> 23.0 us - at 10 more
    // registers; not kept), and this phase was 1 100 VALU / LDS instructions per wave of ~36 per output.  From 116 channels per branch on a lane takes TWO
    // channels (8-byte LDS reads, float2 FMAs); with 58 the pair form only idles half the lanes and a lane keeps one channel.
    constexpr bool PAIRS = HALF > 64;
    constexpr int NROW = F * H, RPW = (NROW + 7) / 8;
    typedef float su_f2 __attribute__((ext_vector_type(2)));
    using DV = std::conditional_t<PAIRS, su_f2, float>;
    constexpr int DW = PAIRS ? 2 : 1;                         // channels per lane
    constexpr int CHD = (HALF + 64 * DW - 1) / (64 * DW);     // channel passes
    DV dv[CHD][RPW][H];
    const DV zero = DV{};
#pragma unroll
    for (int jc = 0; jc < CHD; ++jc) {
        const int c = min(DW * lane + 64 * DW * jc, HALF - DW);
        DV wk[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) wk[t] = *reinterpret_cast<const DV*>(p.wd + t * HALF + c);
        const DV sd = *reinterpret_cast<const DV*>(p.sd + c), bd = *reinterpret_cast<const DV*>(p.bd + c);
#pragma unroll
        for (int r = 0; r < RPW; ++r) {
            const int R = wave + 8 * r;                       // wave-uniform
            if (R * H < Mv) {
                const int y = R % H;
                const bool up = y > 0, dn = y < H - 1;
                const float* rm = buf + (R * H) * LDA + c;    // this row; the rows above / below (or this one again, under zero weights)
                const float* ru = rm - (up ? H * LDA : 0);
                const float* rd = rm + (dn ? H * LDA : 0);
                DV wu[3], wdn[3];
#pragma unroll
                for (int k = 0; k < 3; ++k) { wu[k] = up ? wk[k] : zero; wdn[k] = dn ? wk[6 + k] : zero; }
                DV a0 = zero, a1 = zero, a2 = zero;
                DV b0 = *reinterpret_cast<const DV*>(ru), b1 = *reinterpret_cast<const DV*>(rm), b2 = *reinterpret_cast<const DV*>(rd);
                DV c0 = zero, c1 = zero, c2 = zero;
                if constexpr (H > 1) { c0 = *reinterpret_cast<const DV*>(ru + LDA); c1 = *reinterpret_cast<const DV*>(rm + LDA); c2 = *reinterpret_cast<const DV*>(rd + LDA); }
#pragma unroll
                for (int x = 0; x < H; ++x) {
                    DV acc = zero;
                    if (x > 0) acc = __builtin_elementwise_fma(a0, wu[0], acc);
                    acc = __builtin_elementwise_fma(b0, wu[1], acc);
                    if (x < H - 1) acc = __builtin_elementwise_fma(c0, wu[2], acc);
                    if (x > 0) acc = __builtin_elementwise_fma(a1, wk[3], acc);
                    acc = __builtin_elementwise_fma(b1, wk[4], acc);
                    if (x < H - 1) acc = __builtin_elementwise_fma(c1, wk[5], acc);
                    if (x > 0) acc = __builtin_elementwise_fma(a2, wdn[0], acc);
                    acc = __builtin_elementwise_fma(b2, wdn[1], acc);
                    if (x < H - 1) acc = __builtin_elementwise_fma(c2, wdn[2], acc);
                    dv[jc][r][x] = acc * sd + bd;
                    a0 = b0; a1 = b1; a2 = b2; b0 = c0; b1 = c1; b2 = c2;
                    if (x + 2 < H) {
                        c0 = *reinterpret_cast<const DV*>(ru + (x + 2) * LDA); c1 = *reinterpret_cast<const DV*>(rm + (x + 2) * LDA);
                        c2 = *reinterpret_cast<const DV*>(rd + (x + 2) * LDA);
                    }
                }
            }
        }
    }
    SU_STAMP(4);                                          // depthwise taps computed
    __syncthreads();
#pragma unroll
    for (int jc = 0; jc < CHD; ++jc) {
        const int c = DW * lane + 64 * DW * jc;
#pragma unroll
        for (int r = 0; r < RPW; ++r) {
            const int R = wave + 8 * r;
            if (R * H < Mv && c < HALF) {
#pragma unroll
                for (int x = 0; x < H; ++x) *reinterpret_cast<DV*>(buf + (R * H + x) * LDA + c) = dv[jc][r][x];
            }
        }
    }
    __syncthreads();
    SU_STAMP(5);
    // phase 3: pw2 + BN + ReLU, in place
    su_gemm<typename Q::GM, LDA>(buf, p.w2f, p.s2, p.b2);
    SU_STAMP(6);
    // phase 4: channel_shuffle store: out[2k] = x1[k] (register), out[2k+1] = branch[k] (LDS) as one 8-byte store
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        const int m = wave + 8 * i;
#pragma unroll
        for (int j = 0; j < CIT; ++j) {
            const int c = lane + 64 * j;
            if (m < Mv && c < HALF) {
                float2 o;
                o.x = xr[i][j];
                o.y = buf[m * LDA + c];
                *reinterpret_cast<float2*>(ob + (int64_t)m * C + 2 * c) = o;
            }
        }
    }
    if (TIMED) {
        __builtin_amdgcn_s_waitcnt(0);                    // stores drained
        SU_STAMP(7);
        unsigned hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        if (threadIdx.x == 0) { ts[blockIdx.x * 10 + 8] = hw; ts[blockIdx.x * 10 + 9] = xcc; }
    }
}

template <int H, int HALF, int F>
static int launch_s1_inst(const ShuffleS1P& p, hipStream_t s) {
    using Q = S1Geo<H, HALF, F>;
    static_assert(Q::SMEM <= 80 * 1024, "two blocks per CU");
    static bool attr_set = false;
    if (!attr_set) {
        L2S_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(shuffle_s1_kernel<H, HALF, F, false>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)Q::SMEM));
#ifdef L2S_DIAG
        L2S_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(shuffle_s1_kernel<H, HALF, F, true>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)Q::SMEM));
#endif
        attr_set = true;
    }
#ifdef L2S_DIAG
    if (g_su_ts && g_su_ts_h == H) hipLaunchKernelGGL((shuffle_s1_kernel<H, HALF, F, true>), dim3((p.NF + F - 1) / F), dim3(512), Q::SMEM, s, p, g_su_ts);
    else
#endif
    hipLaunchKernelGGL((shuffle_s1_kernel<H, HALF, F, false>), dim3((p.NF + F - 1) / F), dim3(512), Q::SMEM, s, p, (unsigned long long*)nullptr);
    return 0;
}

// ------------------------------------------------------------------------------------------------ fused units on the bf16 matrix cores
// The pointwise convs of the fused units through the exact three-way split of the dense kernels (x = hi + mid + lo, six bf16 products per fp32
// product, fp32 accumulation: `v_mfma_f32_16x16x32_bf16`, 96 clk per 16x16x32 where `v_mfma_f32_16x16x4_f32` takes 256).  An activation is split ONCE,
// by the thread that writes it to LDS (three bf16 planes, rows of KP32 * 2 + 32 bytes: 32 x odd, conflict-free for the operand's ds_read_b128 lane
// groups), not once per column tile by the waves that read it; the weights arrive pre-split in operand order (su_planes_kernel: [column tile][32-k chunk]
// [plane][lane] 16 bytes, derived on the device from the packed [N][K] matrix at load and after every refresh).  The planes (6 bytes per value) and
// the fp32 map the depthwise conv reads (4 bytes) never live at the same time - a GEMM holds its accumulators across the barrier in front of its
// in-place epilogue, the depthwise holds its outputs across the barrier in front of its stores - so ONE region of 3 planes serves both and two
// blocks still share a CU (69-78 KB).
typedef __bf16 su_bf16x8 __attribute__((ext_vector_type(8)));
constexpr int su_pad32(int v) { return (v + 31) / 32 * 32; }
constexpr int su_pitch(int kp32) { return kp32 * 2 + 32; }                  // bytes per plane row
// two adjacent k as {lo16 = even k, hi16 = odd k} in each of the three planes
__device__ __forceinline__ void su_split2(float a, float b, unsigned& hi, unsigned& mid, unsigned& lo) {
    const unsigned xa = __float_as_uint(a), xb = __float_as_uint(b);
    const float ra = a - __uint_as_float(xa & 0xFFFF0000u), rb = b - __uint_as_float(xb & 0xFFFF0000u);                                  // exact
    const float qa = ra - __uint_as_float(__float_as_uint(ra) & 0xFFFF0000u), qb = rb - __uint_as_float(__float_as_uint(rb) & 0xFFFF0000u);   // exact, <= 8 bits
    hi = __builtin_amdgcn_perm(xb, xa, 0x07060302u);
    mid = __builtin_amdgcn_perm(__float_as_uint(rb), __float_as_uint(ra), 0x07060302u);
    lo = __builtin_amdgcn_perm(__float_as_uint(qb), __float_as_uint(qa), 0x07060302u);
}

// weights [N][K] fp32 -> operand planes: out[((nt * NCH + ch) * 3 + plane) * 64 + lane] (16 bytes) = bf16 plane of W[nt * 16 + (lane & 15)][ch * 32 + 8 * (lane >> 4) + 0..7]
__global__ __launch_bounds__(256) void su_planes_kernel(const float* __restrict__ W, int N, int K, int NT, int NCH, uint4* __restrict__ out) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= NT * NCH * 64) return;
    const int lane = idx & 63, ch = (idx >> 6) % NCH, nt = (idx >> 6) / NCH;
    const int n = nt * 16 + (lane & 15), k0 = ch * 32 + 8 * (lane >> 4);
    unsigned h[4], m[4], l[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int k = k0 + 2 * e;
        const float a = (n < N && k < K) ? W[(int64_t)n * K + k] : 0.f, b = (n < N && k + 1 < K) ? W[(int64_t)n * K + k + 1] : 0.f;
        su_split2(a, b, h[e], m[e], l[e]);
    }
    uint4* o = out + ((int64_t)(nt * NCH + ch) * 3) * 64 + lane;
    o[0] = make_uint4(h[0], h[1], h[2], h[3]);
    o[64] = make_uint4(m[0], m[1], m[2], m[3]);
    o[128] = make_uint4(l[0], l[1], l[2], l[3]);
}
int64_t su_planes_bytes(int N, int K) { return (int64_t)((N + 15) / 16) * (su_pad32(K) / 32) * 3 * 1024; }
int launch_su_planes(const float* W, int N, int K, void* out, hipStream_t s) {
    const int NT = (N + 15) / 16, NCH = su_pad32(K) / 32;
    ProfScope ps("shuffle_unit_weight_planes", s);
    hipLaunchKernelGGL(su_planes_kernel, dim3((NT * NCH * 64 + 255) / 256), dim3(256), 0, s, W, N, K, NT, NCH, reinterpret_cast<uint4*>(out));
    L2S_CHECK_HIP(hipGetLastError());
    return 0;
}

// out[m][n] = relu((A[m][:] . W[n][:]) * scale[n] + shift[n]) for all 16-row tiles of the block: A from the three bf16 planes at `pl` (rows of PITCH
// bytes, planes PLANE bytes apart), the result as fp32 rows of LDA floats at the SAME address (accumulators held across the barrier).  Work items as
// su_gemm.  A column tile's weight chunks are requested BD at a time (all of them up to K = 128; three in flight beyond).
// OPQ: the thread index through an opaque copy - inside a loop over units (shuffle_s1xc_kernel) the operand addresses are loop-invariant, and hoisted out
// of the loop they would stay in registers across every phase of every unit (spills at the 128-register budget of two blocks per CU)
template <class GM, int NCH, int PITCH, int PLANE, int LDA, bool OPQ = false>
__device__ __forceinline__ void su_gemm_x3(unsigned char* __restrict__ pl, const uint4* __restrict__ Wp,
                                           const float* __restrict__ scale, const float* __restrict__ shift, int mrows = 1 << 30) {
    constexpr int G = GM::G, IT = GM::IT, NT = GM::NT, BD = NCH <= 4 ? NCH : 3;      // (two chunks in flight at K = 128: 3 % slower)
    int tx = threadIdx.x;
    if constexpr (OPQ) asm volatile("" : "+v"(tx));
    const int lane = tx & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int li = lane & 15, lg = lane >> 4;
    float* const buf = reinterpret_cast<float*>(pl);
    f32x4 acc[IT][G];
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const int item = wave + 8 * it;
#pragma unroll
        for (int j = 0; j < G; ++j) acc[it][j] = {0.f, 0.f, 0.f, 0.f};
        if (item < GM::ITEMS) {
            const int nt = item % NT, mg = item / NT;
            const int mt0 = min(mg * G, GM::MT - G);
            const uint4* wb = Wp + (int64_t)nt * NCH * 192 + lane;
            const unsigned char* ab = pl + (mt0 * 16 + li) * PITCH + 16 * lg;
            uint4 b[BD][3];
#pragma unroll
            for (int c = 0; c < BD; ++c) {
#pragma unroll
                for (int q = 0; q < 3; ++q) b[c][q] = wb[(c * 3 + q) * 64];
            }
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                const int sl = c % BD;
                const su_bf16x8 bh = __builtin_bit_cast(su_bf16x8, b[sl][0]), bm = __builtin_bit_cast(su_bf16x8, b[sl][1]), bl = __builtin_bit_cast(su_bf16x8, b[sl][2]);
#pragma unroll
                for (int j = 0; j < G; ++j) {
                    const unsigned char* ap = ab + j * 16 * PITCH + 64 * c;
                    const su_bf16x8 ah = *reinterpret_cast<const su_bf16x8*>(ap), am = *reinterpret_cast<const su_bf16x8*>(ap + PLANE),
                                    al = *reinterpret_cast<const su_bf16x8*>(ap + 2 * PLANE);
                    f32x4 a = acc[it][j];      // smallest partial products first (term-major over two, three or all row tiles: no faster)
                    a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, a, 0, 0, 0);
                    a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, a, 0, 0, 0);
                    a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bm, a, 0, 0, 0);
                    a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bh, a, 0, 0, 0);
                    a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bm, a, 0, 0, 0);
                    a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, a, 0, 0, 0);
                    acc[it][j] = a;
                }
                __builtin_amdgcn_sched_barrier(0);      // keep the operand reads of later chunks from being hoisted (VGPRs; without it: 3 % slower)
                if (c + BD < NCH) {
#pragma unroll
                    for (int q = 0; q < 3; ++q) b[sl][q] = wb[((c + BD) * 3 + q) * 64];
                }
            }
        }
    }
    __syncthreads();                                     // every wave has read its operand rows
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const int item = wave + 8 * it;
        if (item < GM::ITEMS) {
            const int nt = item % NT, mg = item / NT;
            const int mt0 = min(mg * G, GM::MT - G);
            const int n = nt * 16 + li;
            if (n < GM::N) {
                const float sc = scale[n], sh = shift[n];
#pragma unroll
                for (int j = 0; j < G; ++j) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float v = acc[it][j][r] * sc + sh;
                        const int row = (mt0 + j) * 16 + 4 * lg + r;
                        if (row < mrows) buf[row * LDA + n] = v > 0.f ? v : 0.f;      // mrows: the rows the region was sized for (the tile's other rows are nobody's)
                    }
                }
            }
        }
    }
    __syncthreads();
}

template <int H, int HALF, int F>
struct S1XGeo {
    static constexpr int HH = H * H, C = 2 * HALF, KP = su_pad16(HALF), LDA = KP + 4, PX = F * HH;
    static constexpr int MT = (PX + 15) / 16, ROWS = MT * 16;
    static constexpr int KP32 = su_pad32(HALF), NCH = KP32 / 32, PITCH = su_pitch(KP32), PLANE = ROWS * PITCH;
    static constexpr int NPAIR = HALF / 2, KPAIR = KP32 / 2;               // channel pairs of a branch: real, and with the K padding
    static constexpr int PPI = KPAIR <= 32 ? 2 : 1;                        // pixels per wave instruction of the input phase
    static constexpr int QIT = PPI == 2 ? 1 : KPAIR / 64, PPW = (PX + 8 * PPI - 1) / (8 * PPI);
    using GM = SuGemm<HALF, KP / 16, MT>;
    static constexpr size_t SMEM = 3 * (size_t)PLANE > (size_t)ROWS * LDA * 4 ? 3 * (size_t)PLANE : (size_t)ROWS * LDA * 4;
    static_assert(HALF % 2 == 0 && (PPI == 2 || KPAIR % 64 == 0), "channel pairs; a pixel's pairs fill whole wave instructions");
};

template <int H, int HALF, int F, bool TIMED>
__global__ __launch_bounds__(512, 4) void shuffle_s1x_kernel(const ShuffleS1P p, unsigned long long* __restrict__ ts) {
    using Q = S1XGeo<H, HALF, F>;
    constexpr int HH = Q::HH, C = Q::C, LDA = Q::LDA, PPW = Q::PPW, QIT = Q::QIT, PPI = Q::PPI, PITCH = Q::PITCH, PLANE = Q::PLANE;
    extern __shared__ __attribute__((aligned(16))) float su_smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    SU_STAMP(0);
    const int f0 = blockIdx.x * F;
    const int Mv = min(F, p.NF - f0) * HH;                // valid pixel rows of this block
    float* buf = su_smem;
    unsigned char* const pl = reinterpret_cast<unsigned char*>(su_smem);
    const float* xb = p.x + (int64_t)f0 * HH * C;
    float* ob = p.out + (int64_t)f0 * HH * C;

    // phase 0: the block's whole input in one round of 8-byte loads: a lane takes channel PAIR q of both halves of pixel m - the passthrough pair stays
    // in registers, the branch pair is split and goes to the three planes as one 4-byte store each (pairs past HALF: zeros - the K padding)
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xb), 0, Mv * C * 4, 0x00020000);
    const int sub = PPI == 2 ? lane >> 5 : 0;
    typedef unsigned su_u2 __attribute__((ext_vector_type(2)));
    float2 xr[PPW][QIT];
    su_u2 br[PPW][QIT];
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        const int soff = (wave + 8 * i) * PPI * C * 4;
#pragma unroll
        for (int j = 0; j < QIT; ++j) {
            const int q = PPI == 2 ? (lane & 31) : lane + 64 * j;
            const int voff = q < Q::NPAIR ? (sub * C + 2 * q) * 4 : 0x7ffffff0;      // past the end: reads 0
            const su_u2 a = __builtin_amdgcn_raw_buffer_load_b64(rsrc, voff, soff, 0);
            br[i][j] = __builtin_amdgcn_raw_buffer_load_b64(rsrc, q < Q::NPAIR ? voff + HALF * 4 : 0x7ffffff0, soff, 0);
            xr[i][j] = make_float2(__uint_as_float(a.x), __uint_as_float(a.y));
        }
    }
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        const int m = (wave + 8 * i) * PPI + sub;
#pragma unroll
        for (int j = 0; j < QIT; ++j) {
            const int q = PPI == 2 ? (lane & 31) : lane + 64 * j;
            unsigned hi, mid, lo;
            su_split2(__uint_as_float(br[i][j].x), __uint_as_float(br[i][j].y), hi, mid, lo);
            if (m < Q::ROWS) {
                unsigned* d = reinterpret_cast<unsigned*>(pl + m * PITCH + q * 4);
                d[0] = hi; d[PLANE / 4] = mid; d[2 * (PLANE / 4)] = lo;
            }
        }
    }
    SU_STAMP(1);                                          // input landed, LDS written
    __syncthreads();
    SU_STAMP(2);
    // phase 1: pw1 + BN + ReLU: planes -> fp32 map
    su_gemm_x3<typename Q::GM, Q::NCH, PITCH, PLANE, LDA>(pl, reinterpret_cast<const uint4*>(p.w1p), p.s1, p.b1);
    SU_STAMP(3);
    // phase 2: depthwise 3x3 (pad 1) + BN over the fp32 map, the sliding window of shuffle_s1_kernel (same taps, same order); its outputs go back as planes
    constexpr bool PAIRS = HALF > 64;
    constexpr int NROW = F * H, RPW = (NROW + 7) / 8;
    typedef float su_f2 __attribute__((ext_vector_type(2)));
    using DV = std::conditional_t<PAIRS, su_f2, float>;
    constexpr int DW = PAIRS ? 2 : 1;                         // channels per lane
    constexpr int CHD = Q::KP32 / (64 * DW);                  // channel passes (over the padded width: the pad lanes write zeros)
    DV dv[CHD][RPW][H];
    const DV zero = DV{};
#pragma unroll
    for (int jc = 0; jc < CHD; ++jc) {
        const int c = min(DW * lane + 64 * DW * jc, HALF - DW);
        DV wk[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) wk[t] = *reinterpret_cast<const DV*>(p.wd + t * HALF + c);
        const DV sd = *reinterpret_cast<const DV*>(p.sd + c), bd = *reinterpret_cast<const DV*>(p.bd + c);
#pragma unroll
        for (int r = 0; r < RPW; ++r) {
            const int R = wave + 8 * r;                       // wave-uniform
            if (R * H < Mv) {
                const int y = R % H;
                const bool up = y > 0, dn = y < H - 1;
                const float* rm = buf + (R * H) * LDA + c;    // this row; the rows above / below (or this one again, under zero weights)
                const float* ru = rm - (up ? H * LDA : 0);
                const float* rd = rm + (dn ? H * LDA : 0);
                DV wu[3], wdn[3];
#pragma unroll
                for (int k = 0; k < 3; ++k) { wu[k] = up ? wk[k] : zero; wdn[k] = dn ? wk[6 + k] : zero; }
                DV a0 = zero, a1 = zero, a2 = zero;
                DV b0 = *reinterpret_cast<const DV*>(ru), b1 = *reinterpret_cast<const DV*>(rm), b2 = *reinterpret_cast<const DV*>(rd);
                DV c0 = zero, c1 = zero, c2 = zero;
                if constexpr (H > 1) { c0 = *reinterpret_cast<const DV*>(ru + LDA); c1 = *reinterpret_cast<const DV*>(rm + LDA); c2 = *reinterpret_cast<const DV*>(rd + LDA); }
#pragma unroll
                for (int x = 0; x < H; ++x) {
                    DV acc = zero;
                    if (x > 0) acc = __builtin_elementwise_fma(a0, wu[0], acc);
                    acc = __builtin_elementwise_fma(b0, wu[1], acc);
                    if (x < H - 1) acc = __builtin_elementwise_fma(c0, wu[2], acc);
                    if (x > 0) acc = __builtin_elementwise_fma(a1, wk[3], acc);
                    acc = __builtin_elementwise_fma(b1, wk[4], acc);
                    if (x < H - 1) acc = __builtin_elementwise_fma(c1, wk[5], acc);
                    if (x > 0) acc = __builtin_elementwise_fma(a2, wdn[0], acc);
                    acc = __builtin_elementwise_fma(b2, wdn[1], acc);
                    if (x < H - 1) acc = __builtin_elementwise_fma(c2, wdn[2], acc);
                    dv[jc][r][x] = acc * sd + bd;
                    a0 = b0; a1 = b1; a2 = b2; b0 = c0; b1 = c1; b2 = c2;
                    if (x + 2 < H) {
                        c0 = *reinterpret_cast<const DV*>(ru + (x + 2) * LDA); c1 = *reinterpret_cast<const DV*>(rm + (x + 2) * LDA);
                        c2 = *reinterpret_cast<const DV*>(rd + (x + 2) * LDA);
                    }
                }
            }
        }
    }
    SU_STAMP(4);                                          // depthwise taps computed
    __syncthreads();
#pragma unroll
    for (int jc = 0; jc < CHD; ++jc) {
        const int c = DW * lane + 64 * DW * jc;           // < KP32: every lane writes (zeros in the K padding)
#pragma unroll
        for (int r = 0; r < RPW; ++r) {
            const int R = wave + 8 * r;
            if (R * H < Mv) {
#pragma unroll
                for (int x = 0; x < H; ++x) {
                    unsigned char* d = pl + (R * H + x) * PITCH + c * 2;
                    if constexpr (PAIRS) {
                        unsigned hi, mid, lo;
                        su_split2(c < HALF ? dv[jc][r][x][0] : 0.f, c < HALF ? dv[jc][r][x][1] : 0.f, hi, mid, lo);
                        *reinterpret_cast<unsigned*>(d) = hi; *reinterpret_cast<unsigned*>(d + PLANE) = mid; *reinterpret_cast<unsigned*>(d + 2 * PLANE) = lo;
                    } else {
                        const float v = c < HALF ? dv[jc][r][x] : 0.f;
                        const float r1 = v - __uint_as_float(__float_as_uint(v) & 0xFFFF0000u);
                        const float r2 = r1 - __uint_as_float(__float_as_uint(r1) & 0xFFFF0000u);
                        *reinterpret_cast<unsigned short*>(d) = (unsigned short)(__float_as_uint(v) >> 16);
                        *reinterpret_cast<unsigned short*>(d + PLANE) = (unsigned short)(__float_as_uint(r1) >> 16);
                        *reinterpret_cast<unsigned short*>(d + 2 * PLANE) = (unsigned short)(__float_as_uint(r2) >> 16);
                    }
                }
            }
        }
    }
    __syncthreads();
    SU_STAMP(5);
    // phase 3: pw2 + BN + ReLU: planes -> fp32 map
    su_gemm_x3<typename Q::GM, Q::NCH, PITCH, PLANE, LDA>(pl, reinterpret_cast<const uint4*>(p.w2p), p.s2, p.b2);
    SU_STAMP(6);
    // phase 4: channel_shuffle store: out[2k] = x1[k] (register), out[2k+1] = branch[k] (LDS): a lane's pair as one 16-byte store
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        const int m = (wave + 8 * i) * PPI + sub;
#pragma unroll
        for (int j = 0; j < QIT; ++j) {
            const int q = PPI == 2 ? (lane & 31) : lane + 64 * j;
            if (m < Mv && q < Q::NPAIR) {
                const float2 bv = *reinterpret_cast<const float2*>(buf + m * LDA + 2 * q);
                *reinterpret_cast<float4*>(ob + (int64_t)m * C + 4 * q) = make_float4(xr[i][j].x, bv.x, xr[i][j].y, bv.y);
            }
        }
    }
    if (TIMED) {
        __builtin_amdgcn_s_waitcnt(0);                    // stores drained
        SU_STAMP(7);
        unsigned hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        if (threadIdx.x == 0) { ts[blockIdx.x * 10 + 8] = hw; ts[blockIdx.x * 10 + 9] = xcc; }
    }
}

template <int H, int HALF, int F>
static int launch_s1x_inst(const ShuffleS1P& p, hipStream_t s) {
    using Q = S1XGeo<H, HALF, F>;
    static_assert(Q::SMEM <= 80 * 1024, "two blocks per CU");
    static bool attr_set = false;
    if (!attr_set) {
        L2S_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(shuffle_s1x_kernel<H, HALF, F, false>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)Q::SMEM));
#ifdef L2S_DIAG
        L2S_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(shuffle_s1x_kernel<H, HALF, F, true>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)Q::SMEM));
#endif
        attr_set = true;
    }
#ifdef L2S_DIAG
    if (g_su_ts && g_su_ts_h == H) hipLaunchKernelGGL((shuffle_s1x_kernel<H, HALF, F, true>), dim3((p.NF + F - 1) / F), dim3(512), Q::SMEM, s, p, g_su_ts);
    else
#endif
    hipLaunchKernelGGL((shuffle_s1x_kernel<H, HALF, F, false>), dim3((p.NF + F - 1) / F), dim3(512), Q::SMEM, s, p, (unsigned long long*)nullptr);
    return 0;
}

// ------------------------------------------------------------------------------------------------ a whole stage's stride-1 units in ONE launch
// The stride-1 units of a stage (3 at 12x12, 7 at 6x6, 3 at 3x3; shufflenetv2.py:135-148) all work on the same map, and a block of the fused unit
// already holds its frames' whole map on chip - the passthrough half in registers, the branch half in LDS.  One launch per unit still wrote that map
// to HBM and read it back, unit after unit (each unit: the map in + the map out, 0.7-0.9 of the launch at ~5 TB/s).  Here a block walks ALL the units
// of its stage: between two units the output channel order out[2k] = x1[k], out[2k+1] = branch[k] (channel_shuffle) is re-cut into the next unit's
// halves ON CHIP - a lane holds out[4q .. 4q+3] of its pixel (its passthrough pair and the branch pair it reads back from LDS); the channels below
// HALF are the next passthrough half: lane q' takes its pair from lane q' >> 1 of the same wave instruction (ds_bpermute, no LDS round trip); the
// channels from HALF on are the next branch input: split and stored as operand planes where the fp32 branch map was.  Same regions, same registers
// as one unit; per pixel the arithmetic of the single-unit kernel, operation for operation: same bits.  Only the units' weights stream.
struct S1UnitW {
    const void* w1p; const float* s1; const float* b1;      // pw1 operand planes, BN scale / shift
    const float* wd; const float* sd; const float* bd;      // depthwise [9][half], BN scale / shift
    const void* w2p; const float* s2; const float* b2;      // pw2
};
struct ShuffleS1ChainP {
    const float* x; float* out;                              // (NF, h, h, 2*half) channel-last: input of the first unit, output of the last
    int NF, n;
    S1UnitW u[S1_CHAIN_MAX];
};

template <int H, int HALF, int F>
__global__ __launch_bounds__(512, 4) void shuffle_s1xc_kernel(const ShuffleS1ChainP p) {
    using Q = S1XGeo<H, HALF, F>;
    constexpr int HH = Q::HH, C = Q::C, LDA = Q::LDA, PPW = Q::PPW, QIT = Q::QIT, PPI = Q::PPI, PITCH = Q::PITCH, PLANE = Q::PLANE;
    static_assert(HALF % 2 == 0 && (Q::KPAIR - Q::NPAIR) <= 32, "a lane's four output channels are two whole pairs; the K padding is at most 32 pairs");
    extern __shared__ __attribute__((aligned(16))) float su_smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int f0 = blockIdx.x * F;
    const int Mv = min(F, p.NF - f0) * HH;                // valid pixel rows of this block
    float* buf = su_smem;
    unsigned char* const pl = reinterpret_cast<unsigned char*>(su_smem);
    const float* xb = p.x + (int64_t)f0 * HH * C;
    float* ob = p.out + (int64_t)f0 * HH * C;
    const int nu = p.n;

    // phase 0 (once): the block's whole input, as in shuffle_s1x_kernel
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xb), 0, Mv * C * 4, 0x00020000);
    const int sub = PPI == 2 ? lane >> 5 : 0;
    typedef unsigned su_u2 __attribute__((ext_vector_type(2)));
    float2 xr[PPW][QIT];
    {
        su_u2 br[PPW][QIT];
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            const int soff = (wave + 8 * i) * PPI * C * 4;
#pragma unroll
            for (int j = 0; j < QIT; ++j) {
                const int q = PPI == 2 ? (lane & 31) : lane + 64 * j;
                const int voff = q < Q::NPAIR ? (sub * C + 2 * q) * 4 : 0x7ffffff0;      // past the end: reads 0
                const su_u2 a = __builtin_amdgcn_raw_buffer_load_b64(rsrc, voff, soff, 0);
                br[i][j] = __builtin_amdgcn_raw_buffer_load_b64(rsrc, q < Q::NPAIR ? voff + HALF * 4 : 0x7ffffff0, soff, 0);
                xr[i][j] = make_float2(__uint_as_float(a.x), __uint_as_float(a.y));
            }
        }
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            const int m = (wave + 8 * i) * PPI + sub;
#pragma unroll
            for (int j = 0; j < QIT; ++j) {
                const int q = PPI == 2 ? (lane & 31) : lane + 64 * j;
                unsigned hi, mid, lo;
                su_split2(__uint_as_float(br[i][j].x), __uint_as_float(br[i][j].y), hi, mid, lo);
                if (m < Q::ROWS) {
                    unsigned* d = reinterpret_cast<unsigned*>(pl + m * PITCH + q * 4);
                    d[0] = hi; d[PLANE / 4] = mid; d[2 * (PLANE / 4)] = lo;
                }
            }
        }
    }
    __syncthreads();

    for (int u = 0; u < nu; ++u) {
        const S1UnitW& U = p.u[u];
        const void* const w1p = U.w1p; const float* const s1 = U.s1; const float* const b1 = U.b1;
        const float* const wd = U.wd; const float* const sdp = U.sd; const float* const bdp = U.bd;
        const void* const w2p = U.w2p; const float* const s2 = U.s2; const float* const b2 = U.b2;
        // phase 1: pw1 + BN + ReLU: planes -> fp32 map
        su_gemm_x3<typename Q::GM, Q::NCH, PITCH, PLANE, LDA, true>(pl, reinterpret_cast<const uint4*>(w1p), s1, b1);
        // phase 2: depthwise 3x3 (pad 1) + BN over the fp32 map (the sliding window of shuffle_s1x_kernel); its outputs go back as planes
        {
            constexpr bool PAIRS = HALF > 64;
            constexpr int NROW = F * H, RPW = (NROW + 7) / 8;
            typedef float su_f2 __attribute__((ext_vector_type(2)));
            using DV = std::conditional_t<PAIRS, su_f2, float>;
            constexpr int DW = PAIRS ? 2 : 1;                         // channels per lane
            constexpr int CHD = Q::KP32 / (64 * DW);                  // channel passes (over the padded width: the pad lanes write zeros)
            DV dv[CHD][RPW][H];
            const DV zero = DV{};
            int lane_d = lane;
            asm volatile("" : "+v"(lane_d));                  // (addresses of this phase must not be hoisted out of the unit loop: registers)
#pragma unroll
            for (int jc = 0; jc < CHD; ++jc) {
                const int c = min(DW * lane_d + 64 * DW * jc, HALF - DW);
                DV wk[9];
#pragma unroll
                for (int t = 0; t < 9; ++t) wk[t] = *reinterpret_cast<const DV*>(wd + t * HALF + c);
                const DV sd = *reinterpret_cast<const DV*>(sdp + c), bd = *reinterpret_cast<const DV*>(bdp + c);
#pragma unroll
                for (int r = 0; r < RPW; ++r) {
                    const int R = wave + 8 * r;                       // wave-uniform
                    if (R * H < Mv) {
                        const int y = R % H;
                        const bool up = y > 0, dn = y < H - 1;
                        const float* rm = buf + (R * H) * LDA + c;
                        const float* ru = rm - (up ? H * LDA : 0);
                        const float* rd = rm + (dn ? H * LDA : 0);
                        DV wu[3], wdn[3];
#pragma unroll
                        for (int k = 0; k < 3; ++k) { wu[k] = up ? wk[k] : zero; wdn[k] = dn ? wk[6 + k] : zero; }
                        DV a0 = zero, a1 = zero, a2 = zero;
                        DV b0 = *reinterpret_cast<const DV*>(ru), b1v = *reinterpret_cast<const DV*>(rm), b2v = *reinterpret_cast<const DV*>(rd);
                        DV c0 = zero, c1 = zero, c2 = zero;
                        if constexpr (H > 1) { c0 = *reinterpret_cast<const DV*>(ru + LDA); c1 = *reinterpret_cast<const DV*>(rm + LDA); c2 = *reinterpret_cast<const DV*>(rd + LDA); }
#pragma unroll
                        for (int x = 0; x < H; ++x) {
                            DV acc = zero;
                            if (x > 0) acc = __builtin_elementwise_fma(a0, wu[0], acc);
                            acc = __builtin_elementwise_fma(b0, wu[1], acc);
                            if (x < H - 1) acc = __builtin_elementwise_fma(c0, wu[2], acc);
                            if (x > 0) acc = __builtin_elementwise_fma(a1, wk[3], acc);
                            acc = __builtin_elementwise_fma(b1v, wk[4], acc);
                            if (x < H - 1) acc = __builtin_elementwise_fma(c1, wk[5], acc);
                            if (x > 0) acc = __builtin_elementwise_fma(a2, wdn[0], acc);
                            acc = __builtin_elementwise_fma(b2v, wdn[1], acc);
                            if (x < H - 1) acc = __builtin_elementwise_fma(c2, wdn[2], acc);
                            dv[jc][r][x] = acc * sd + bd;
                            a0 = b0; a1 = b1v; a2 = b2v; b0 = c0; b1v = c1; b2v = c2;
                            if (x + 2 < H) {
                                c0 = *reinterpret_cast<const DV*>(ru + (x + 2) * LDA); c1 = *reinterpret_cast<const DV*>(rm + (x + 2) * LDA);
                                c2 = *reinterpret_cast<const DV*>(rd + (x + 2) * LDA);
                            }
                        }
                    }
                }
            }
            __syncthreads();
#pragma unroll
            for (int jc = 0; jc < CHD; ++jc) {
                const int c = DW * lane_d + 64 * DW * jc;         // < KP32: every lane writes (zeros in the K padding)
#pragma unroll
                for (int r = 0; r < RPW; ++r) {
                    const int R = wave + 8 * r;
                    if (R * H < Mv) {
#pragma unroll
                        for (int x = 0; x < H; ++x) {
                            unsigned char* d = pl + (R * H + x) * PITCH + c * 2;
                            if constexpr (PAIRS) {
                                unsigned hi, mid, lo;
                                su_split2(c < HALF ? dv[jc][r][x][0] : 0.f, c < HALF ? dv[jc][r][x][1] : 0.f, hi, mid, lo);
                                *reinterpret_cast<unsigned*>(d) = hi; *reinterpret_cast<unsigned*>(d + PLANE) = mid; *reinterpret_cast<unsigned*>(d + 2 * PLANE) = lo;
                            } else {
                                const float v = c < HALF ? dv[jc][r][x] : 0.f;
                                const float r1 = v - __uint_as_float(__float_as_uint(v) & 0xFFFF0000u);
                                const float r2 = r1 - __uint_as_float(__float_as_uint(r1) & 0xFFFF0000u);
                                *reinterpret_cast<unsigned short*>(d) = (unsigned short)(__float_as_uint(v) >> 16);
                                *reinterpret_cast<unsigned short*>(d + PLANE) = (unsigned short)(__float_as_uint(r1) >> 16);
                                *reinterpret_cast<unsigned short*>(d + 2 * PLANE) = (unsigned short)(__float_as_uint(r2) >> 16);
                            }
                        }
                    }
                }
            }
            __syncthreads();
        }
        // phase 3: pw2 + BN + ReLU: planes -> fp32 map
        su_gemm_x3<typename Q::GM, Q::NCH, PITCH, PLANE, LDA, true>(pl, reinterpret_cast<const uint4*>(w2p), s2, b2);
        // phase 4: channel_shuffle: out[2k] = x1[k] (register), out[2k+1] = branch[k] (LDS): a lane's pair q gives out[4q .. 4q+3] of its pixel
        if (u + 1 == nu) {                                    // last unit: the map leaves the chip, 16 bytes per lane
#pragma unroll
            for (int i = 0; i < PPW; ++i) {
                const int m = (wave + 8 * i) * PPI + sub;
#pragma unroll
                for (int j = 0; j < QIT; ++j) {
                    const int q = PPI == 2 ? (lane & 31) : lane + 64 * j;
                    if (m < Mv && q < Q::NPAIR) {
                        const float2 bv = *reinterpret_cast<const float2*>(buf + m * LDA + 2 * q);
                        *reinterpret_cast<float4*>(ob + (int64_t)m * C + 4 * q) = make_float4(xr[i][j].x, bv.x, xr[i][j].y, bv.y);
                    }
                }
            }
        } else {                                              // re-cut into the next unit's halves on chip
            // (index arithmetic from an opaque copy of the lane id: hoisted out of the unit loop it would hold ~40 address registers across every phase)
            int lane_v = lane;
            asm volatile("" : "+v"(lane_v));
            const int sub_v = PPI == 2 ? lane_v >> 5 : 0;
            float2 bv[PPW][QIT];                              // this lane's branch pair: with its passthrough pair, out[4q .. 4q+3] = (x1.x, br.x, x1.y, br.y)
#pragma unroll
            for (int i = 0; i < PPW; ++i) {
                const int m = (wave + 8 * i) * PPI + sub_v;
#pragma unroll
                for (int j = 0; j < QIT; ++j) {
                    const int q = PPI == 2 ? (lane_v & 31) : lane_v + 64 * j;
                    bv[i][j] = make_float2(0.f, 0.f);
                    if (m < Q::ROWS && q < Q::NPAIR) bv[i][j] = *reinterpret_cast<const float2*>(buf + m * LDA + 2 * q);
                }
            }
            __syncthreads();                                  // every lane has read its branch pair: the fp32 map may be overwritten by planes
#pragma unroll
            for (int i = 0; i < PPW; ++i) {
                const int m = (wave + 8 * i) * PPI + sub_v;
#pragma unroll
                for (int j = 0; j < QIT; ++j) {
                    const int q = PPI == 2 ? (lane_v & 31) : lane_v + 64 * j;
                    // channels >= HALF: the next branch input, k = c - HALF (even: a lane's two pairs never straddle HALF)
                    if (m < Q::ROWS && q < Q::NPAIR) {
#pragma unroll
                        for (int e = 0; e < 2; ++e) {
                            const int c = 4 * q + 2 * e;
                            if (c >= HALF) {
                                unsigned hi, mid, lo;
                                su_split2(e ? xr[i][j].y : xr[i][j].x, e ? bv[i][j].y : bv[i][j].x, hi, mid, lo);
                                unsigned* d = reinterpret_cast<unsigned*>(pl + m * PITCH + (c - HALF) * 2);
                                d[0] = hi; d[PLANE / 4] = mid; d[2 * (PLANE / 4)] = lo;
                            }
                        }
                    }
                    // the K padding of the planes (pairs NPAIR .. KPAIR-1) was overwritten by the fp32 map: zeros again, by the first lanes of the row
                    if (j == 0 && m < Q::ROWS && q < Q::KPAIR - Q::NPAIR) {
                        unsigned* d = reinterpret_cast<unsigned*>(pl + m * PITCH + (Q::NPAIR + q) * 4);
                        d[0] = 0u; d[PLANE / 4] = 0u; d[2 * (PLANE / 4)] = 0u;
                    }
                }
                // channels < HALF: the next passthrough half: pair q of a lane = out[2q], out[2q+1] = one (x1, branch) column of lane q >> 1, register j = 0
                float2 nx[QIT];
#pragma unroll
                for (int j = 0; j < QIT; ++j) {
                    const int q = PPI == 2 ? (lane_v & 31) : lane_v + 64 * j;
                    const int srcq = q >> 1;
                    const int src = PPI == 2 ? ((lane_v & 32) | srcq) : srcq;       // < 64: the passthrough half comes from the first HALF / 4 lanes
                    const float ex = __shfl(xr[i][0].x, src), ey = __shfl(bv[i][0].x, src), ez = __shfl(xr[i][0].y, src), ew = __shfl(bv[i][0].y, src);
                    nx[j] = (q & 1) ? make_float2(ez, ew) : make_float2(ex, ey);
                }
#pragma unroll
                for (int j = 0; j < QIT; ++j) xr[i][j] = nx[j];
            }
            __syncthreads();                                  // the next unit's operand planes are in place
        }
    }
}

template <int H, int HALF, int F>
static int launch_s1xc_inst(const ShuffleS1ChainP& p, hipStream_t s) {
    using Q = S1XGeo<H, HALF, F>;
    static_assert(Q::SMEM <= 80 * 1024, "two blocks per CU");
    static bool attr_set = false;
    if (!attr_set) {
        L2S_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(shuffle_s1xc_kernel<H, HALF, F>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)Q::SMEM));
        attr_set = true;
    }
    hipLaunchKernelGGL((shuffle_s1xc_kernel<H, HALF, F>), dim3((p.NF + F - 1) / F), dim3(512), Q::SMEM, s, p);
    return 0;
}

// n consecutive stride-1 units of one stage (units[0].x -> units[n-1].out) as ONE launch; every unit must carry operand planes (trunk_x3)
int launch_shuffle_s1_chain(const ShuffleS1P* units, int n, hipStream_t s) {
    L2S_REQUIRE(n >= 1 && n <= S1_CHAIN_MAX, "shuffle_s1 chain: 1..7 units");
    ShuffleS1ChainP c{};
    c.x = units[0].x; c.out = units[n - 1].out; c.NF = units[0].NF; c.n = n;
    const int h = units[0].h, half = units[0].half;
    for (int i = 0; i < n; ++i) {
        const ShuffleS1P& p = units[i];
        L2S_REQUIRE(p.w1p && p.w2p && p.h == h && p.half == half && p.NF == c.NF && p.Kpad == su_pad16(p.half), "shuffle_s1 chain: units of one stage with operand planes");
        c.u[i] = S1UnitW{p.w1p, p.s1, p.b1, p.wd, p.sd, p.bd, p.w2p, p.s2, p.b2};
    }
    ProfScope ps(h >= 11 ? "shuffle_stage_s1_chain_h12" : h == 6 ? "shuffle_stage_s1_chain_h6" : "shuffle_stage_s1_chain_h3", s);
    int rc = 1;
    if (h == 12 && half == 58) rc = launch_s1xc_inst<12, 58, 1>(c, s);
    else if (h == 11 && half == 58) rc = launch_s1xc_inst<11, 58, 1>(c, s);
    else if (h == 6 && half == 116) rc = launch_s1xc_inst<6, 116, 2>(c, s);
    else if (h == 3 && half == 232) rc = c.NF >= 2048 ? launch_s1xc_inst<3, 232, 5>(c, s) : launch_s1xc_inst<3, 232, 2>(c, s);
    else set_error("shuffle_s1 chain: unsupported unit geometry");
    if (rc) return 1;
    L2S_CHECK_HIP(hipGetLastError());
    return 0;
}

// frames per block (measured, profiles/r01_fused_units_pmc.txt): 1 at 12x12 / 11x11, 2 at 6x6 and 3x3
int launch_shuffle_s1(const ShuffleS1P& p, hipStream_t s) {
    L2S_REQUIRE(p.Kpad == su_pad16(p.half), "shuffle_s1: weight fragments are packed for K = pad16(half)");
    ProfScope ps(p.h >= 11 ? "shuffle_unit_s1_fused_h12" : p.h == 6 ? "shuffle_unit_s1_fused_h6" : "shuffle_unit_s1_fused_h3", s);
    int rc = 1;
    if (p.w1p && p.w2p) {      // the pointwise convs on the bf16 matrix cores (option "trunk_x3"): same frames per block
        if (p.h == 12 && p.half == 58) rc = launch_s1x_inst<12, 58, 1>(p, s);
        else if (p.h == 11 && p.half == 58) rc = launch_s1x_inst<11, 58, 1>(p, s);
        else if (p.h == 6 && p.half == 116) rc = launch_s1x_inst<6, 116, 2>(p, s);
        else if (p.h == 3 && p.half == 232) rc = p.NF >= 2048 ? launch_s1x_inst<3, 232, 5>(p, s) : launch_s1x_inst<3, 232, 2>(p, s);
        else set_error("shuffle_s1: unsupported unit geometry");
    }
    else if (p.h == 12 && p.half == 58) rc = launch_s1_inst<12, 58, 1>(p, s);
    else if (p.h == 11 && p.half == 58) rc = launch_s1_inst<11, 58, 1>(p, s);      // 88x88 crops
    // frames per block: with many frames in the launch (grouped batches) more frames share one pass over the unit's weights, which every block
    // streams from L2 (3x3 stage: 2 x 215 KB per block against 2 x 9 pixels of work) and the 16-row tiles fill better (18 of 32 rows -> 45 of 48):
    // 273 -> 197 us per unit at 256 clips with five frames (four: 226, seven: 231 - registers).  A pixel's arithmetic does not depend on its
    // block's other pixels, so the results are the same bits whatever the grouping
    else if (p.h == 6 && p.half == 116) rc = launch_s1_inst<6, 116, 2>(p, s);         // three frames per block: 128 VGPRs + spills, 238 -> 267 us
    else if (p.h == 3 && p.half == 232) rc = p.NF >= 2048 ? launch_s1_inst<3, 232, 5>(p, s) : launch_s1_inst<3, 232, 2>(p, s);
    else set_error("shuffle_s1: unsupported unit geometry");
    if (rc) return 1;
    L2S_CHECK_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------------ fused stride-2 unit
// H = input spatial size, CIN = input channels, HALF = channels per output branch, RO = output rows per block.
template <int H, int CIN, int HALF, int RO>
struct S2Geo {
    static constexpr int HO = (H + 1) / 2, KIN = su_pad16(CIN), KH = su_pad16(HALF), LDA = (KIN > KH ? KIN : KH) + 4;
    static constexpr int RIN = 2 * RO + 1, MTI = (RIN * H + 15) / 16, MTO = (RO * HO + 15) / 16, STRIPS = (HO + RO - 1) / RO;
    using G1 = SuGemm<HALF, KIN / 16, MTI>;              // banch2 pw1 at input resolution
    using G2 = SuGemm<HALF, KIN / 16, MTO>;              // banch1 pw
    using G3 = SuGemm<HALF, KH / 16, MTO>;               // banch2 pw2
    static constexpr size_t SMEM = (size_t)(MTI + 2 * MTO) * 16 * LDA * sizeof(float);
    static_assert(CIN % 4 == 0 && CIN <= HALF, "float4 tile rows; pw1 runs in place over the input tile");
};

// depthwise 3x3 / stride 2 / pad 1 (+BN) from an LDS tile of input rows (tile row 0 = frame row iy0) to an LDS tile of output
// pixels; lanes = channels, waves = output pixels. Frame borders are skipped, not read: the padding is of THIS map (for banch2
// that is relu(bn(pw1(x))), whose value on a zero pixel is not zero).
// PITCH > 0: the outputs go to `dst` as three bf16 planes (rows of PITCH bytes, planes PLANE bytes apart; the split-bf16 units) instead of fp32 rows.
template <int H, int HO, int LDA, int CN, int PITCH = 0, int PLANE = 0>
__device__ __forceinline__ void su_dw_s2(const float* __restrict__ src, float* __restrict__ dst, int iy0, int outv,
                                         const float* __restrict__ w9, const float* __restrict__ sc, const float* __restrict__ sh) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if constexpr (CN > 64) {
        // two channels per lane (8-byte LDS accesses, float2 FMAs): one pass over the pixels per 128 channels - see the stride-1 unit
        typedef float su_f2 __attribute__((ext_vector_type(2)));
        static_assert(CN % 2 == 0 && LDA % 2 == 0, "channel pairs");
#pragma unroll
        for (int jc = 0; jc < (CN + 127) / 128; ++jc) {
            const int c = 2 * lane + 128 * jc;
            if (c < CN) {
                su_f2 wk[9];
#pragma unroll
                for (int t = 0; t < 9; ++t) wk[t] = *reinterpret_cast<const su_f2*>(w9 + t * CN + c);
                const su_f2 s = *reinterpret_cast<const su_f2*>(sc + c), b = *reinterpret_cast<const su_f2*>(sh + c);
                const su_f2 zero2 = {0.f, 0.f};
                for (int m = wave; m < outv; m += 8) {            // wave-uniform
                    const int oyl = m / HO, ox = m - oyl * HO;
                    su_f2 acc = zero2;
#pragma unroll
                    for (int ky = 0; ky < 3; ++ky) {
#pragma unroll
                        for (int kx = 0; kx < 3; ++kx) {
                            const int r = 2 * oyl + ky, iy = iy0 + r, ix = 2 * ox + kx - 1;
                            const bool in = (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)H;
                            const su_f2 v = *reinterpret_cast<const su_f2*>(src + (in ? r * H + ix : (2 * oyl + 1) * H + 2 * ox) * LDA + c);
                            acc = __builtin_elementwise_fma(v, in ? wk[ky * 3 + kx] : zero2, acc);
                        }
                    }
                    const su_f2 o = acc * s + b;
                    if constexpr (PITCH > 0) {
                        unsigned hi, mid, lo;
                        su_split2(o[0], o[1], hi, mid, lo);
                        unsigned char* d = reinterpret_cast<unsigned char*>(dst) + m * PITCH + c * 2;
                        *reinterpret_cast<unsigned*>(d) = hi; *reinterpret_cast<unsigned*>(d + PLANE) = mid; *reinterpret_cast<unsigned*>(d + 2 * PLANE) = lo;
                    } else {
                        *reinterpret_cast<su_f2*>(dst + m * LDA + c) = o;
                    }
                }
            }
        }
        return;
    }
#pragma unroll
    for (int jc = 0; jc < (CN + 63) / 64; ++jc) {
        const int c = lane + 64 * jc;
        if (c < CN) {
            float wk[9];
#pragma unroll
            for (int t = 0; t < 9; ++t) wk[t] = w9[t * CN + c];
            const float s = sc[c], b = sh[c];
            for (int m = wave; m < outv; m += 8) {            // wave-uniform
                const int oyl = m / HO, ox = m - oyl * HO;
                float acc = 0.f;
                // border taps are predicated (weight 0, the always-valid centre address), not branched around
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        const int r = 2 * oyl + ky, iy = iy0 + r, ix = 2 * ox + kx - 1;
                        const bool in = (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)H;
                        acc = fmaf(src[(in ? r * H + ix : (2 * oyl + 1) * H + 2 * ox) * LDA + c], in ? wk[ky * 3 + kx] : 0.f, acc);
                    }
                }
                const float o = acc * s + b;
                if constexpr (PITCH > 0) {
                    const float r1 = o - __uint_as_float(__float_as_uint(o) & 0xFFFF0000u);
                    const float r2 = r1 - __uint_as_float(__float_as_uint(r1) & 0xFFFF0000u);
                    unsigned char* d = reinterpret_cast<unsigned char*>(dst) + m * PITCH + c * 2;
                    *reinterpret_cast<unsigned short*>(d) = (unsigned short)(__float_as_uint(o) >> 16);
                    *reinterpret_cast<unsigned short*>(d + PLANE) = (unsigned short)(__float_as_uint(r1) >> 16);
                    *reinterpret_cast<unsigned short*>(d + 2 * PLANE) = (unsigned short)(__float_as_uint(r2) >> 16);
                } else {
                    dst[m * LDA + c] = o;
                }
            }
        }
    }
}

#define S2_STAMP(k) do { if (TIMED && threadIdx.x == 0) ts[(blockIdx.y * gridDim.x + blockIdx.x) * 10 + (k)] = wall_clock64(); } while (0)
template <int H, int CIN, int HALF, int RO, bool TIMED>
__global__ __launch_bounds__(512, 4) void shuffle_s2_kernel(const ShuffleS2P p, unsigned long long* __restrict__ ts) {
    using Q = S2Geo<H, CIN, HALF, RO>;
    constexpr int HO = Q::HO, KIN = Q::KIN, KH = Q::KH, LDA = Q::LDA, MTI = Q::MTI, MTO = Q::MTO, CIN4 = CIN / 4;
    extern __shared__ __attribute__((aligned(16))) float su_smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int f = blockIdx.y;
    const int oy0 = blockIdx.x * RO;
    const int ro = min(RO, HO - oy0);
    const int iy0 = 2 * oy0 - 1;                           // frame row of tile row 0
    const int outv = ro * HO;                              // valid output pixels of this block
    float* X = su_smem;
    float* D1 = X + MTI * 16 * LDA;
    float* D2 = D1 + MTO * 16 * LDA;
    S2_STAMP(0);

    // the tile's rows inside the frame are one contiguous run of (pixel, channel) floats in HBM: float4 loads, LDS row per pixel.
    // Tile rows outside the frame are never written and never read by the depthwise taps; the GEMM turns them into garbage
    // rows that nobody reads (GEMM rows are independent).
    const int iy_lo = max(iy0, 0), iy_hi = min(iy0 + 2 * ro + 1, H);
    const int n4 = (iy_hi - iy_lo) * (H * CIN4);
    const float4* src4 = reinterpret_cast<const float4*>(p.x + ((int64_t)f * H + iy_lo) * (H * CIN));
    float* xt = X + (iy_lo - iy0) * (H * LDA);
    // every request before the first LDS write: as a rolled load -> store loop this was one memory round trip per trip (stamped: 6.4 us of a 26 us
    // block at stage 3, four trips)
    constexpr int NLD = (Q::RIN * H * CIN4 + 511) / 512;
    float4 xin[NLD];
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        const int idx = tid + 512 * i;
        xin[i] = idx < n4 ? src4[idx] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        const int idx = tid + 512 * i;
        if (idx < n4) {
            const int px = idx / CIN4, c4 = idx - px * CIN4;
            *reinterpret_cast<float4*>(xt + px * LDA + 4 * c4) = xin[i];
        }
    }
    // zero the K padding columns (0 * weight padding must stay 0, and LDS garbage may be NaN)
    if (KIN > CIN) {
        for (int idx = tid; idx < (MTI + MTO) * 16 * (KIN - CIN); idx += 512) {          // X and D1 are adjacent
            const int m = idx / (KIN - CIN), k = CIN + idx - m * (KIN - CIN);
            X[m * LDA + k] = 0.f;
        }
    }
    if (KH > HALF) {
        for (int idx = tid; idx < MTO * 16 * (KH - HALF); idx += 512) {
            const int m = idx / (KH - HALF), k = HALF + idx - m * (KH - HALF);
            D2[m * LDA + k] = 0.f;
        }
    }
    S2_STAMP(1);
    __syncthreads();
    S2_STAMP(2);
    // banch1: depthwise s2 + BN of x -> D1 (must read x before pw1 overwrites it in place)
    su_dw_s2<H, HO, LDA, CIN>(X, D1, iy0, outv, p.wd1, p.sd1, p.bd1);
    S2_STAMP(3);
    // banch2: pw1 + BN + ReLU at full resolution, in place (its first barrier also orders the D1 reads of x before the overwrite)
    su_gemm<typename Q::G1, LDA>(X, p.w1f, p.s1, p.b1);
    S2_STAMP(4);
    // banch2: depthwise s2 + BN -> D2
    su_dw_s2<H, HO, LDA, HALF>(X, D2, iy0, outv, p.wd, p.sd, p.bd);
    __syncthreads();
    S2_STAMP(5);
    // the two output-resolution pointwise convs + BN + ReLU, in place
    su_gemm<typename Q::G2, LDA>(D1, p.wb1f, p.sb1, p.bb1);
    S2_STAMP(6);
    su_gemm<typename Q::G3, LDA>(D2, p.w2f, p.s2, p.b2);
    S2_STAMP(7);
    // channel_shuffle store: out[2k] = banch1[k], out[2k+1] = banch2[k]
    float* ob = p.out + ((int64_t)f * HO + oy0) * (HO * 2 * HALF);
    for (int m = wave; m < outv; m += 8) {
#pragma unroll
        for (int jc = 0; jc < (HALF + 63) / 64; ++jc) {
            const int c = lane + 64 * jc;
            if (c < HALF) {
                float2 o;
                o.x = D1[m * LDA + c];
                o.y = D2[m * LDA + c];
                *reinterpret_cast<float2*>(ob + m * (2 * HALF) + 2 * c) = o;
            }
        }
    }
    if (TIMED) { __builtin_amdgcn_s_waitcnt(0); S2_STAMP(8); }
}

// The stride-2 unit with its three pointwise convs on the bf16 matrix cores (stages 2 and 3; stage 4's unit streams 3 x 223 KB of weights per 36-pixel
// block and is bound by them - as planes they would be 1.5x the bytes).  The input tile is needed twice: as fp32 by banch1's depthwise, as planes by
// pw1 - the threads keep the float4s they loaded and write the planes over the fp32 tile once the depthwise has read it.  Both depthwise convs write
// their outputs as planes; every GEMM leaves fp32 rows in place of its operand planes.
template <int H, int CIN, int HALF, int RO>
struct S2XGeo {
    using B = S2Geo<H, CIN, HALF, RO>;
    static constexpr int HO = B::HO, LDA = B::LDA, MTI = B::MTI, MTO = B::MTO, RX = B::RIN * H;
    static constexpr int KI32 = su_pad32(CIN), KH32 = su_pad32(HALF), PIN = su_pitch(KI32), PH = su_pitch(KH32);
    static constexpr int RV = RO * HO;                     // output pixels of a block: the rows the output-resolution regions are sized for
    // valid rows only: a tile's other rows are read on into whatever follows (garbage rows nobody reads) and never written (su_gemm_x3's row limit)
    static constexpr int XPLANE = RX * PIN, D1PLANE = RV * PIN, D2PLANE = RV * PH;
    static constexpr int XBYTES = 3 * XPLANE > MTI * 16 * LDA * 4 ? 3 * XPLANE : MTI * 16 * LDA * 4;
    static constexpr int TAIL = (MTO * 16 - RV) * PIN + 16;      // the last region's over-read stays inside the allocation
    static constexpr size_t SMEM = (size_t)XBYTES + 3 * D2PLANE + 3 * D1PLANE + TAIL;      // [X | D2 | D1 | tail]: stage 2 fits three blocks per CU (54 016 B)
    static_assert(3 * D1PLANE >= RV * LDA * 4 && 3 * D2PLANE >= RV * LDA * 4, "a GEMM's fp32 rows fit in place of its operand planes");
    static_assert(XBYTES % 16 == 0 && D1PLANE % 16 == 0 && D2PLANE % 16 == 0 && TAIL % 16 == 0, "16-byte aligned regions");
    using G1 = SuGemm<HALF, 1, MTI>;
    using G2 = SuGemm<HALF, 1, MTO>;
};

template <int H, int CIN, int HALF, int RO, bool TIMED>
__global__ __launch_bounds__(512, 4) void shuffle_s2x_kernel(const ShuffleS2P p, unsigned long long* __restrict__ ts) {
    using Q = S2XGeo<H, CIN, HALF, RO>;
    constexpr int HO = Q::HO, LDA = Q::LDA, CIN4 = CIN / 4, PIN = Q::PIN, PH = Q::PH;
    extern __shared__ __attribute__((aligned(16))) float su_smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int f = blockIdx.y;
    const int oy0 = blockIdx.x * RO;
    const int ro = min(RO, HO - oy0);
    const int iy0 = 2 * oy0 - 1;                           // frame row of tile row 0
    const int outv = ro * HO;                              // valid output pixels of this block
    float* X = su_smem;
    unsigned char* const XP = reinterpret_cast<unsigned char*>(su_smem);
    unsigned char* const D2P = XP + Q::XBYTES;
    unsigned char* const D1P = D2P + 3 * Q::D2PLANE;
    float* D1 = reinterpret_cast<float*>(D1P);
    float* D2 = reinterpret_cast<float*>(D2P);
    S2_STAMP(0);

    const int iy_lo = max(iy0, 0), iy_hi = min(iy0 + 2 * ro + 1, H);
    const int n4 = (iy_hi - iy_lo) * (H * CIN4);
    const int row0 = (iy_lo - iy0) * H;                    // first tile row inside the frame
    const float4* src4 = reinterpret_cast<const float4*>(p.x + ((int64_t)f * H + iy_lo) * (H * CIN));
    float* xt = X + row0 * LDA;
    constexpr int NLD = (Q::RX * CIN4 + 511) / 512;
    float4 xin[NLD];
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        const int idx = tid + 512 * i;
        xin[i] = idx < n4 ? src4[idx] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    // the output-resolution operand planes start as zeros: their K padding must be, and their unused rows must not hold NaN patterns
    for (int idx = tid; idx < (3 * Q::D1PLANE + 3 * Q::D2PLANE + Q::TAIL) / 16; idx += 512) reinterpret_cast<uint4*>(D2P)[idx] = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        const int idx = tid + 512 * i;
        if (idx < n4) {
            const int px = idx / CIN4, c4 = idx - px * CIN4;
            *reinterpret_cast<float4*>(xt + px * LDA + 4 * c4) = xin[i];
        }
    }
    S2_STAMP(1);
    __syncthreads();
    S2_STAMP(2);
    // banch1: depthwise s2 + BN of x -> D1 planes
    su_dw_s2<H, HO, LDA, CIN, PIN, Q::D1PLANE>(X, D1, iy0, outv, p.wd1, p.sd1, p.bd1);
    __syncthreads();                                       // the fp32 tile has been read: its planes take its place
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        const int idx = tid + 512 * i;
        if (idx < n4) {
            const int px = idx / CIN4, c4 = idx - px * CIN4;
            unsigned h0, m0, l0, h1, m1, l1;
            su_split2(xin[i].x, xin[i].y, h0, m0, l0);
            su_split2(xin[i].z, xin[i].w, h1, m1, l1);
            unsigned char* d = XP + (row0 + px) * PIN + c4 * 8;
            *reinterpret_cast<uint2*>(d) = make_uint2(h0, h1);
            *reinterpret_cast<uint2*>(d + Q::XPLANE) = make_uint2(m0, m1);
            *reinterpret_cast<uint2*>(d + 2 * Q::XPLANE) = make_uint2(l0, l1);
        }
    }
    if constexpr (Q::KI32 > CIN) {                         // the K padding of the input planes
        constexpr int PS = (Q::KI32 - CIN) / 4;
        for (int idx = tid; idx < Q::RX * PS; idx += 512) {
            const int m = idx / PS, k4 = idx - m * PS;
            unsigned char* d = XP + m * PIN + (CIN4 + k4) * 8;
            *reinterpret_cast<uint2*>(d) = make_uint2(0u, 0u);
            *reinterpret_cast<uint2*>(d + Q::XPLANE) = make_uint2(0u, 0u);
            *reinterpret_cast<uint2*>(d + 2 * Q::XPLANE) = make_uint2(0u, 0u);
        }
    }
    S2_STAMP(3);
    __syncthreads();
    // banch2: pw1 + BN + ReLU at full resolution: planes -> fp32 tile
    su_gemm_x3<typename Q::G1, Q::KI32 / 32, PIN, Q::XPLANE, LDA>(XP, reinterpret_cast<const uint4*>(p.w1p), p.s1, p.b1);
    S2_STAMP(4);
    // banch2: depthwise s2 + BN -> D2 planes
    su_dw_s2<H, HO, LDA, HALF, PH, Q::D2PLANE>(X, D2, iy0, outv, p.wd, p.sd, p.bd);
    __syncthreads();
    S2_STAMP(5);
    // the two output-resolution pointwise convs + BN + ReLU: planes -> fp32 rows in place
    su_gemm_x3<typename Q::G2, Q::KI32 / 32, PIN, Q::D1PLANE, LDA>(D1P, reinterpret_cast<const uint4*>(p.wb1p), p.sb1, p.bb1, Q::RV);
    S2_STAMP(6);
    su_gemm_x3<typename Q::G2, Q::KH32 / 32, PH, Q::D2PLANE, LDA>(D2P, reinterpret_cast<const uint4*>(p.w2p), p.s2, p.b2, Q::RV);
    S2_STAMP(7);
    // channel_shuffle store: out[2k] = banch1[k], out[2k+1] = banch2[k]
    float* ob = p.out + ((int64_t)f * HO + oy0) * (HO * 2 * HALF);
    for (int m = wave; m < outv; m += 8) {
#pragma unroll
        for (int jc = 0; jc < (HALF + 63) / 64; ++jc) {
            const int c = lane + 64 * jc;
            if (c < HALF) {
                float2 o;
                o.x = D1[m * LDA + c];
                o.y = D2[m * LDA + c];
                *reinterpret_cast<float2*>(ob + m * (2 * HALF) + 2 * c) = o;
            }
        }
    }
    if (TIMED) { __builtin_amdgcn_s_waitcnt(0); S2_STAMP(8); }
}

template <int H, int CIN, int HALF, int RO>
static int launch_s2x_inst(const ShuffleS2P& p, hipStream_t s) {
    using Q = S2XGeo<H, CIN, HALF, RO>;
    static_assert(Q::SMEM <= 80 * 1024, "two blocks per CU");
    static bool attr_set = false;
    if (!attr_set) {
        L2S_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(shuffle_s2x_kernel<H, CIN, HALF, RO, false>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)Q::SMEM));
#ifdef L2S_DIAG
        L2S_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(shuffle_s2x_kernel<H, CIN, HALF, RO, true>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)Q::SMEM));
#endif
        attr_set = true;
    }
#ifdef L2S_DIAG
    if (g_su_ts && g_su_ts_h == -H) hipLaunchKernelGGL((shuffle_s2x_kernel<H, CIN, HALF, RO, true>), dim3(Q::B::STRIPS, p.NF), dim3(512), Q::SMEM, s, p, g_su_ts);
    else
#endif
    hipLaunchKernelGGL((shuffle_s2x_kernel<H, CIN, HALF, RO, false>), dim3(Q::B::STRIPS, p.NF), dim3(512), Q::SMEM, s, p, (unsigned long long*)nullptr);
    return 0;
}

template <int H, int CIN, int HALF, int RO>
static int launch_s2_inst(const ShuffleS2P& p, hipStream_t s) {
    using Q = S2Geo<H, CIN, HALF, RO>;
    static_assert(Q::SMEM <= 80 * 1024, "two blocks per CU");
    static bool attr_set = false;
    if (!attr_set) {
        L2S_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(shuffle_s2_kernel<H, CIN, HALF, RO, false>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)Q::SMEM));
#ifdef L2S_DIAG
        L2S_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(shuffle_s2_kernel<H, CIN, HALF, RO, true>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)Q::SMEM));
#endif
        attr_set = true;
    }
    // measurement hook: l2s_op_fused_unit_timeline(ts, -H) stamps the stride-2 unit whose INPUT is H x H
#ifdef L2S_DIAG
    if (g_su_ts && g_su_ts_h == -H) hipLaunchKernelGGL((shuffle_s2_kernel<H, CIN, HALF, RO, true>), dim3(Q::STRIPS, p.NF), dim3(512), Q::SMEM, s, p, g_su_ts);
    else
#endif
    hipLaunchKernelGGL((shuffle_s2_kernel<H, CIN, HALF, RO, false>), dim3(Q::STRIPS, p.NF), dim3(512), Q::SMEM, s, p, (unsigned long long*)nullptr);
    return 0;
}

// output rows per block (measured): 2 at stages 2 and 3, the whole 3x3 frame at stage 4
int launch_shuffle_s2(const ShuffleS2P& p, hipStream_t s) {
    L2S_REQUIRE(p.Kin == su_pad16(p.cin) && p.Kh == su_pad16(p.half) && p.ho == (p.h + 1) / 2, "shuffle_s2: fragment packing / geometry mismatch");
    L2S_REQUIRE((reinterpret_cast<uintptr_t>(p.x) & 15u) == 0, "shuffle_s2 input must be 16-byte aligned");
    ProfScope ps(p.cin == 24 ? "shuffle_unit_s2_fused_st2" : p.cin == 116 ? "shuffle_unit_s2_fused_st3" : "shuffle_unit_s2_fused_st4", s);
    int rc = 1;
    // stages 2 and 3 on the bf16 matrix cores (option "trunk_x3").  Stage 2 (K = 24: the matrix pipe is 38 % of the f32 unit) only pays with its regions sized
    // for the valid rows - 54 KB, three blocks per CU like the f32 unit: 676 -> 652 us per 256 clips (with full 16-row tiles, two blocks per CU: 715); stage 4 is
    // bound by its weight stream (planes would be 1.5x the bytes) and stays on f32 MFMA
    const bool x3 = p.w1p && p.w2p && p.wb1p && p.cin != 232;
    if (x3 && p.h == 24 && p.cin == 24 && p.half == 58) rc = launch_s2x_inst<24, 24, 58, 2>(p, s);
    else if (x3 && p.h == 22 && p.cin == 24 && p.half == 58) rc = launch_s2x_inst<22, 24, 58, 2>(p, s);
    else if (x3 && p.h == 12 && p.cin == 116 && p.half == 116) rc = launch_s2x_inst<12, 116, 116, 2>(p, s);
    else if (x3 && p.h == 11 && p.cin == 116 && p.half == 116) rc = launch_s2x_inst<11, 116, 116, 2>(p, s);
    else if (p.h == 24 && p.cin == 24 && p.half == 58) rc = launch_s2_inst<24, 24, 58, 2>(p, s);
    else if (p.h == 22 && p.cin == 24 && p.half == 58) rc = launch_s2_inst<22, 24, 58, 2>(p, s);      // 88x88 crops
    else if (p.h == 12 && p.cin == 116 && p.half == 116) rc = launch_s2_inst<12, 116, 116, 2>(p, s);
    else if (p.h == 11 && p.cin == 116 && p.half == 116) rc = launch_s2_inst<11, 116, 116, 2>(p, s);
    else if (p.h == 6 && p.cin == 232 && p.half == 232) rc = launch_s2_inst<6, 232, 232, 3>(p, s);
    else set_error("shuffle_s2: unsupported unit geometry");
    if (rc) return 1;
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
