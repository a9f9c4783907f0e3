// Device pieces shared by the tiled GEMM kernels (gemm_nt.hip: f32 MFMA; gemm_x3.hip: split-bf16 MFMA): the fused epilogue.
#pragma once
#include "l2s_common.h"

namespace l2s {

__device__ __forceinline__ float apply_act(float v, int act, const float* actw, int col) {
    if (act == ACT_RELU) return v > 0.f ? v : 0.f;
    if (act == ACT_SILU) return v / (1.0f + expf(-v));
    if (act == ACT_PSINE) return sinf(v) * actw[col];
    return v;
}

// one output element: scale/shift, activation, masks, addends, and the plain / windowed / strided / channel-first store
__device__ __forceinline__ void gemm_store(const GemmP& p, int row, int col, float accv, float sc, float sh) {
    if (p.win_T > 0) { const int wb = row / p.Tout; row = wb * p.win_T + p.win_off + (row - wb * p.Tout); }
    float v = accv * sc + sh;
    if (p.Zout) p.Zout[(int64_t)row * p.ldc + (int64_t)col * p.c_cstride] = v;
    v = apply_act(v, p.act, p.actw, col);
    if (p.mask && p.mask_pre) v *= p.mask[(int64_t)row * p.ldmask + col];
    if (p.R1) v += p.R1[(int64_t)(p.r1_mod ? row % p.r1_mod : row) * p.ldr1 + col];
    if (p.R2) v += p.R2[(int64_t)(p.r2_div ? row / p.r2_div : (p.r2_mod ? row % p.r2_mod : row)) * p.ldr2 + col];
    if (p.mask && !p.mask_pre) v *= p.mask[(int64_t)row * p.ldmask + col];
    if (p.c_tr_T > 0) {
        const int b = row / p.c_tr_T, t = row - b * p.c_tr_T;
        p.C[((int64_t)b * p.N + col) * p.c_tr_T + t] = v;
    } else {
        p.C[(int64_t)row * p.ldc + (int64_t)col * p.c_cstride] = v;
    }
}

}  // namespace l2s
