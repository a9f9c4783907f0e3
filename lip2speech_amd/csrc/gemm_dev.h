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

// ---- bf16-operand mode of the training GEMMs (option "train_bf16"): operands rounded to bf16 (round-to-nearest-even, v_cvt_pk_bf16_f32) on
// their way into LDS, v_mfma_f32_32x32x16_bf16 with fp32 accumulation.  LDS tile = [64 rows][32 k] bf16, rows of 80 bytes (64 + 16 pad:
// conflict-free ds_read_b128 for the 32x32x16 operand layout: lane l reads the 8 consecutive k of row l&31 at k offset 8*(l>>5)).
typedef __bf16 gd_bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 gd_bf16x8 __attribute__((ext_vector_type(8)));
constexpr int GD_BROW = 80;            // bytes per bf16 LDS row
__device__ __forceinline__ unsigned gd_pack2(float a, float b) { gd_bf16x2 v = {(__bf16)a, (__bf16)b}; return __builtin_bit_cast(unsigned, v); }
__device__ __forceinline__ uint2 gd_pack4(const float4& v) { return make_uint2(gd_pack2(v.x, v.y), gd_pack2(v.z, v.w)); }
__device__ __forceinline__ unsigned short gd_bf16(float a) { return __builtin_bit_cast(unsigned short, (__bf16)a); }
// the two K steps of a 64x64x32 tile for this wave's 32x32 sub-tile
__device__ __forceinline__ f32x16 gd_mma_tile_bf16(const unsigned char* As, const unsigned char* Bs, int wm, int wn, int li, int lg, f32x16 acc) {
    const unsigned char* ap = As + (wm * 32 + li) * GD_BROW + lg * 16;
    const unsigned char* bp = Bs + (wn * 32 + li) * GD_BROW + lg * 16;
#pragma unroll
    for (int st = 0; st < 2; ++st) {
        const gd_bf16x8 a = *reinterpret_cast<const gd_bf16x8*>(ap + st * 32);
        const gd_bf16x8 b = *reinterpret_cast<const gd_bf16x8*>(bp + st * 32);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
    }
    return acc;
}

}  // namespace l2s
