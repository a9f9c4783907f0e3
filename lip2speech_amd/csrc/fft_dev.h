// Wave-level FFTs for the vocoder / metric kernels (vocoder.hip): one 64-lane wave transforms 512 complex points - a 1024-point real
// transform, the n_fft of the reference's mel pipeline (hparams.py: filter_length 1024) - as a Stockham autosort FFT of three radix-8
// stages.  Lane j owns the eight points j + 64 r of a stage's input; between stages the points travel through a 520-float2 LDS scratch
// of the wave's own (LDS instructions of one wave execute in order: a wave-scope fence orders the compiler, no block barrier).
#pragma once
#include "l2s_common.h"

namespace l2s {

__device__ __forceinline__ float2 cf_add(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 cf_sub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 cf_mul(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ float2 cf_conj(float2 a) { return make_float2(a.x, -a.y); }
// a * (SIGN * i)
template <int SIGN>
__device__ __forceinline__ float2 cf_mul_si(float2 a) { return SIGN > 0 ? make_float2(-a.y, a.x) : make_float2(a.y, -a.x); }

// stores of one lane become visible to the other lanes' later loads: LDS ops of a wave are issued and executed in order
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// X[k] = sum_n v[n] W^(n k), W = exp(SIGN * 2 pi i / 8); decimation in frequency, in place, natural order in and out
template <int SIGN>
__device__ __forceinline__ void dft8(float2 (&v)[8]) {
    constexpr float C = 0.70710678118654752440f;
    float2 u[4], d[4];
#pragma unroll
    for (int n = 0; n < 4; ++n) { u[n] = cf_add(v[n], v[n + 4]); d[n] = cf_sub(v[n], v[n + 4]); }
    // d[n] *= W^n: W^1 = C (1 + SIGN i), W^2 = SIGN i, W^3 = C (-1 + SIGN i)
    d[1] = SIGN > 0 ? make_float2(C * (d[1].x - d[1].y), C * (d[1].x + d[1].y)) : make_float2(C * (d[1].x + d[1].y), C * (d[1].y - d[1].x));
    d[2] = cf_mul_si<SIGN>(d[2]);
    d[3] = SIGN > 0 ? make_float2(-C * (d[3].x + d[3].y), C * (d[3].x - d[3].y)) : make_float2(C * (d[3].y - d[3].x), -C * (d[3].x + d[3].y));
    auto dft4 = [](const float2 (&y)[4], float2& Y0, float2& Y1, float2& Y2, float2& Y3) {
        const float2 p0 = cf_add(y[0], y[2]), p1 = cf_sub(y[0], y[2]), q0 = cf_add(y[1], y[3]), q1 = cf_mul_si<SIGN>(cf_sub(y[1], y[3]));
        Y0 = cf_add(p0, q0); Y2 = cf_sub(p0, q0); Y1 = cf_add(p1, q1); Y3 = cf_sub(p1, q1);
    };
    dft4(u, v[0], v[2], v[4], v[6]);
    dft4(d, v[1], v[3], v[5], v[7]);
}

// per-lane twiddle constants of the 512-point transform (computed once per kernel): stage 1 exp(i 2 pi r (j % 8) / 64), stage 2
// exp(i 2 pi r j / 512), and the real-transform split exp(i 2 pi (j + 64 r) / 1024); POSITIVE angles - the forward transform conjugates
struct Fft512Tw {
    float2 s1[8], s2[8], rs[8];
    __device__ __forceinline__ void init(int lane) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            float s, c;
            sincospif((float)(r * (lane & 7)) / 32.0f, &s, &c); s1[r] = make_float2(c, s);
            sincospif((float)(r * lane) / 256.0f, &s, &c); s2[r] = make_float2(c, s);
            sincospif((float)(lane + 64 * r) / 512.0f, &s, &c); rs[r] = make_float2(c, s);
        }
    }
};

// 512-point complex FFT of the wave: v[r] = point (lane + 64 r) on entry and on exit; unnormalised; SIGN = -1 forward, +1 inverse.
// `sc`: 512 float2 of LDS scratch owned by this wave (its previous contents are dead).
template <int SIGN>
__device__ __forceinline__ void fft512(float2 (&v)[8], float2* sc, int lane, const Fft512Tw& tw) {
    // stage 0 (Ns = 1): no twiddles; out[8 j + r]
    dft8<SIGN>(v);
#pragma unroll
    for (int r = 0; r < 8; r += 2) *reinterpret_cast<float4*>(sc + 8 * lane + r) = make_float4(v[r].x, v[r].y, v[r + 1].x, v[r + 1].y);
    wave_lds_sync();
    // stage 1 (Ns = 8): in[j + 64 r] * W64^(r (j % 8)); out[(j / 8) 64 + (j % 8) + 8 r]
#pragma unroll
    for (int r = 0; r < 8; ++r) v[r] = sc[lane + 64 * r];
    wave_lds_sync();
#pragma unroll
    for (int r = 1; r < 8; ++r) v[r] = cf_mul(v[r], SIGN > 0 ? tw.s1[r] : cf_conj(tw.s1[r]));
    dft8<SIGN>(v);
    {
        const int j0 = (lane >> 3) * 64 + (lane & 7);
#pragma unroll
        for (int r = 0; r < 8; ++r) sc[j0 + 8 * r] = v[r];
    }
    wave_lds_sync();
    // stage 2 (Ns = 64): in[j + 64 r] * W512^(r j); out[j + 64 r] stays in registers
#pragma unroll
    for (int r = 0; r < 8; ++r) v[r] = sc[lane + 64 * r];
    wave_lds_sync();
#pragma unroll
    for (int r = 1; r < 8; ++r) v[r] = cf_mul(v[r], SIGN > 0 ? tw.s2[r] : cf_conj(tw.s2[r]));
    dft8<SIGN>(v);
}

// Real 1024-point transform from the packed 512-point one.  On entry v[r] = Z[lane + 64 r] with z[n] = x[2n] + i x[2n+1]; on exit v[r] =
// X[lane + 64 r] (k = 0..511) and the return value is X[512] (valid in lane 0).  Uses the scratch for the mirrored read.
__device__ __forceinline__ float rfft1024_post(float2 (&v)[8], float2* sc, int lane, const Fft512Tw& tw) {
#pragma unroll
    for (int r = 0; r < 8; ++r) sc[lane + 64 * r] = v[r];
    wave_lds_sync();
    float2 m[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) m[r] = cf_conj(sc[(512 - lane - 64 * r) & 511]);
    wave_lds_sync();
    const float nyq = v[0].x - v[0].y;               // lane 0: X[512] = Re Z[0] - Im Z[0]
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const float2 s = cf_add(v[r], m[r]), d = cf_sub(v[r], m[r]);
        const float2 t = cf_mul(cf_conj(tw.rs[r]), d);          // exp(-2 pi i k / 1024) (Z[k] - conj Z[512 - k])
        v[r] = make_float2(0.5f * (s.x + t.y), 0.5f * (s.y - t.x));   // 0.5 (s - i t)
    }
    return nyq;
}

// Inverse of the above.  On entry v[r] = S[lane + 64 r] (k = 0..511) of a half spectrum S[0..512] and `s512` = S[512] (any lane's value is
// ignored except lane 0's); the imaginary parts of S[0] and S[512] are ignored like a C2R transform ignores them.  On exit v[r] = the packed
// input Z'[lane + 64 r] of the 512-point INVERSE transform whose output z[n] satisfies x[2n] = Re z[n] / 1024, x[2n+1] = Im z[n] / 1024.
__device__ __forceinline__ void irfft1024_pre(float2 (&v)[8], float2 s512, float2* sc, int lane, const Fft512Tw& tw) {
    if (lane == 0) { v[0].y = 0.f; sc[512] = make_float2(s512.x, 0.f); }
#pragma unroll
    for (int r = 0; r < 8; ++r) sc[lane + 64 * r] = v[r];
    wave_lds_sync();
    float2 m[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) m[r] = cf_conj(sc[512 - lane - 64 * r]);
    wave_lds_sync();
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const float2 s = cf_add(v[r], m[r]), d = cf_sub(v[r], m[r]);
        const float2 t = cf_mul(tw.rs[r], d);                    // exp(+2 pi i k / 1024) (S[k] - conj S[512 - k])
        v[r] = make_float2(s.x - t.y, s.y + t.x);                // s + i t
    }
}

}  // namespace l2s
