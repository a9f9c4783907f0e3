// Backward GEMMs of the dense / Conv1d layers (fp32 MFMA, same 64x64x32 tile and K-permuted operand fetch as gemm_nt.hip):
//
//   BWD_DX : dX[m][j]  = sum_r dZcol(m, r) * Wp[n(r)][tap(r)*Cin + j]        - input gradient.  For a Conv1d (stride 1) the rows of
//            dZ are gathered with the flipped-tap implicit addressing, r = (tap', n); the weight operand is read in its forward
//            layout [Nout][taps*Cin] and transposed on its way into LDS.
//   BWD_DW : dWp[n][j] = sum_m dZ[m][n] * Xcol(m, j),  j = (tap, ci)          - weight gradient; Xcol is the forward implicit
//            im2col of the input; both operands are transposed on their way into LDS (the reduction runs over rows).
//
// Reference semantics: autograd of nn.Linear / nn.Conv1d in /root/reference/model/modules/decoder.py (train.py:184 loss.backward()).
#include "l2s_common.h"
#include "gemm_dev.h"

#include <algorithm>

namespace l2s {

constexpr int TB = 64, TK = 32, TLD = TK + 4;

// stage a [32 reduction rows][64 cols] global tile transposed into S[col][row]; this thread owns rows r0, r0+16 and cols c4..c4+3
__device__ __forceinline__ void stage_transposed(float* S, int r0, int c4, const float4& v0, const float4& v1) {
    S[(c4 + 0) * TLD + r0] = v0.x; S[(c4 + 1) * TLD + r0] = v0.y; S[(c4 + 2) * TLD + r0] = v0.z; S[(c4 + 3) * TLD + r0] = v0.w;
    S[(c4 + 0) * TLD + r0 + 16] = v1.x; S[(c4 + 1) * TLD + r0 + 16] = v1.y; S[(c4 + 2) * TLD + r0 + 16] = v1.z; S[(c4 + 3) * TLD + r0 + 16] = v1.w;
}

// bf16 form of the transposed staging: S[col][row] as bf16, rows of GD_BROW bytes
__device__ __forceinline__ void stage_transposed_bf16(unsigned char* S, int r0, int c4, const float4& v0, const float4& v1) {
    const float a[4] = {v0.x, v0.y, v0.z, v0.w}, b[4] = {v1.x, v1.y, v1.z, v1.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        *reinterpret_cast<unsigned short*>(S + (c4 + e) * GD_BROW + r0 * 2) = gd_bf16(a[e]);
        *reinterpret_cast<unsigned short*>(S + (c4 + e) * GD_BROW + (r0 + 16) * 2) = gd_bf16(b[e]);
    }
}

// one 64x64 output tile (bx, by) of K slice bz; As / Bs: the block's two LDS tiles
template <int MODE, bool BF16, bool VEC>
__device__ __forceinline__ void gemm_bwd_block(const BwdGemmP& p, const int bx, const int by, const int bz, float* As, float* Bs) {
    const int m0 = by * TB, n0 = bx * TB;
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63, wm = wave >> 1, wn = wave & 1, li = lane & 31, lg = lane >> 5;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const int nkt_all = (p.K + TK - 1) / TK;
    int kt_begin = 0, kt_end = nkt_all;
    float* Cout = p.C;
    if (p.ksplit > 1) {                                // split-K: slice z of the reduction -> its own partial matrix
        const int per = (nkt_all + p.ksplit - 1) / p.ksplit;
        kt_begin = bz * per;
        kt_end = kt_begin + per < nkt_all ? kt_begin + per : nkt_all;
        Cout += (int64_t)bz * p.c_split_stride;
    }
    // VEC (a template constant: with a run-time flag the compiler merged both forms into four 4-byte FLAT loads per quad, and kept the zero quad in
    // scratch to select between pointers): dimensions and strides are multiples of 4 and the pointers 16-byte aligned -> one 16-byte global load.
    // Out-of-range quads load from the (always valid) base of their matrix and are zeroed afterwards: a value select, not a pointer select.
    auto ld4 = [&](const float* q, const float* base, bool ok, int nvalid) -> float4 {
        float4 v;
        if constexpr (VEC) {
            v = *reinterpret_cast<const float4*>(ok ? q : base);
        } else {
            const float* qq = ok ? q : base;
            const int nv = ok ? nvalid : 0;
            v.x = nv > 0 ? qq[0] : 0.f;
            v.y = nv > 1 ? qq[1] : 0.f;
            v.z = nv > 2 ? qq[2] : 0.f;
            v.w = nv > 3 ? qq[3] : 0.f;
        }
        v.x = ok ? v.x : 0.f; v.y = ok ? v.y : 0.f; v.z = ok ? v.z : 0.f; v.w = ok ? v.w : 0.f;
        return v;
    };

    // ---- per-mode loaders ------------------------------------------------------------------------------------------------
    // DX: A = dZ gathered (row-major staging: thread -> row lr (+32), k-quad kq), B = Wp transposed staging
    // DW: A = dZ transposed staging (reduction = rows m), B = Xcol transposed staging
    const int lr = tid >> 3, kq = (tid & 7) * 4;      // row-major staging coordinates
    const int r0 = tid >> 4, c4 = (tid & 15) * 4;     // transposed staging coordinates

    auto load_dx_a = [&](int j, int k) -> float4 {     // dZcol(m, r..r+3), r = tap'*Nout + n, 4 consecutive n
        const int m = m0 + lr + 32 * j;
        bool ok = m < p.M && k < p.K;
        const int mm = ok ? m : 0, kk = ok ? k : 0;
        const int b = mm / p.Tx, t = mm - b * p.Tx;
        int tap = 0, n = kk;
        if (p.taps > 1) { tap = kk / p.Nout; n = kk - tap * p.Nout; }
        const int tz = t + tap - p.padp;
        ok = ok && tz >= 0 && tz < p.Tz;
        return ld4(p.A + ((int64_t)b * p.Tz + (ok ? tz : 0)) * p.lda + n, p.A, ok, p.K - kk);
    };
    auto load_dx_b = [&](int j, int k0) -> float4 {    // Wp[n(r)][(taps-1-tap')*Cin + col..col+3] for reduction row r = k0 + r0 + 16j
        const int r = k0 + r0 + 16 * j, col = n0 + c4;
        const bool ok = r < p.K && col < p.N;
        const int rr = ok ? r : 0, cc = ok ? col : 0;
        int tap = 0, n = rr;
        if (p.taps > 1) { tap = rr / p.Nout; n = rr - tap * p.Nout; }
        return ld4(p.B + (int64_t)n * p.ldb + (p.taps - 1 - tap) * p.Cin + cc, p.B, ok, p.N - cc);
    };
    auto load_dw_a = [&](int j, int k0) -> float4 {    // dZ[m][n0 + c4 ..], m = k0 + r0 + 16j
        const int m = k0 + r0 + 16 * j, col = m0 + c4;
        const bool ok = m < p.K && col < p.M;
        const int mm = ok ? m : 0, cc = ok ? col : 0;
        return ld4(p.A + (int64_t)mm * p.lda + cc, p.A, ok, p.M - cc);
    };
    auto load_dw_b = [&](int j, int k0) -> float4 {    // Xcol(m, jcol..jcol+3), jcol = (tap, ci)
        const int m = k0 + r0 + 16 * j, col = n0 + c4;
        bool ok = m < p.K && col < p.N;
        const int mm = ok ? m : 0, cc = ok ? col : 0;
        const int b = mm / p.Tz, t = mm - b * p.Tz;
        int tap = 0, ci = cc;
        if (p.taps > 1) { tap = cc / p.Cin; ci = cc - tap * p.Cin; }
        const int tx = t * p.stride + tap - p.pad;
        ok = ok && tx >= 0 && tx < p.Tx;
        return ld4(p.B + ((int64_t)b * p.Tx + (ok ? tx : 0)) * p.ldb + ci, p.B, ok, p.N - cc);
    };

    float4 ra[2], rb[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        if (MODE == BWD_DX) { ra[j] = load_dx_a(j, kt_begin * TK + kq); rb[j] = load_dx_b(j, kt_begin * TK); }
        else { ra[j] = load_dw_a(j, kt_begin * TK); rb[j] = load_dw_b(j, kt_begin * TK); }
    }
    for (int kt = kt_begin; kt < kt_end; ++kt) {
        if (BF16) {
            unsigned char* Ab = reinterpret_cast<unsigned char*>(As);
            if (MODE == BWD_DX) {
#pragma unroll
                for (int j = 0; j < 2; ++j) *reinterpret_cast<uint2*>(Ab + (lr + 32 * j) * GD_BROW + kq * 2) = gd_pack4(ra[j]);
            } else {
                stage_transposed_bf16(Ab, r0, c4, ra[0], ra[1]);
            }
            stage_transposed_bf16(reinterpret_cast<unsigned char*>(Bs), r0, c4, rb[0], rb[1]);
        } else {
            if (MODE == BWD_DX) {
#pragma unroll
                for (int j = 0; j < 2; ++j) *reinterpret_cast<float4*>(&As[(lr + 32 * j) * TLD + kq]) = ra[j];
            } else {
                stage_transposed(As, r0, c4, ra[0], ra[1]);
            }
            stage_transposed(Bs, r0, c4, rb[0], rb[1]);
        }
        __syncthreads();
        if (kt + 1 < kt_end) {
            const int k0 = (kt + 1) * TK;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                if (MODE == BWD_DX) { ra[j] = load_dx_a(j, k0 + kq); rb[j] = load_dx_b(j, k0); }
                else { ra[j] = load_dw_a(j, k0); rb[j] = load_dw_b(j, k0); }
            }
        }
        if (BF16) {
            acc = gd_mma_tile_bf16(reinterpret_cast<const unsigned char*>(As), reinterpret_cast<const unsigned char*>(Bs), wm, wn, li, lg, acc);
        } else {
            const float* ap = &As[(wm * 32 + li) * TLD + 4 * lg];
            const float* bp = &Bs[(wn * 32 + li) * TLD + 4 * lg];
#pragma unroll
            for (int c = 0; c < TK / 8; ++c) {
                const float4 a4 = *reinterpret_cast<const float4*>(ap + 8 * c);
                const float4 b4 = *reinterpret_cast<const float4*>(bp + 8 * c);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, b4.x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, b4.y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, b4.z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, b4.w, acc, 0, 0, 0);
            }
        }
        __syncthreads();
    }
    const int col = n0 + wn * 32 + li;
    if (col >= p.N) return;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lg;
        if (row >= p.M) continue;
        int64_t off;
        if (p.c_T > 0) { const int b = row / p.c_T; off = (int64_t)b * p.c_seq_stride + (int64_t)(row - b * p.c_T) * p.ldc + col; }
        else off = (int64_t)row * p.ldc + col;
        const float v = acc[r] * p.alpha;
        Cout[off] = p.accumulate ? Cout[off] + v : v;
    }
}

static bool al16(const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15u) == 0; }

template <int MODE, bool BF16, bool VEC>
__global__ __launch_bounds__(256) void gemm_bwd_kernel(const BwdGemmP p) {
    __shared__ __attribute__((aligned(16))) float As[TB * TLD];
    __shared__ __attribute__((aligned(16))) float Bs[TB * TLD];
    gemm_bwd_block<MODE, BF16, VEC>(p, blockIdx.x, blockIdx.y, blockIdx.z, As, Bs);
}
// The weight gradient and the input gradient of ONE layer in one launch: they read the same dZ and are independent of each other, and at 8 clips
// per GPU neither fills the chip - blocks [0, wx*wy*wz) are the (split-K) weight-gradient tiles, the rest the input-gradient tiles.
template <bool BF16, bool VEC>
__global__ __launch_bounds__(256) void gemm_bwd_pair_kernel(const BwdGemmP pw, const BwdGemmP px, int wx, int wy, int wz, int xx) {
    __shared__ __attribute__((aligned(16))) float As[TB * TLD];
    __shared__ __attribute__((aligned(16))) float Bs[TB * TLD];
    int id = blockIdx.x;
    const int nw = wx * wy * wz;
    if (id < nw) gemm_bwd_block<BWD_DW, BF16, VEC>(pw, id % wx, (id / wx) % wy, id / (wx * wy), As, Bs);
    else { id -= nw; gemm_bwd_block<BWD_DX, BF16, VEC>(px, id % xx, id / xx, 0, As, Bs); }
}

static bool bwd_vec_ok(const BwdGemmP& p) {
    bool vec = al16(p.A) && al16(p.B) && p.lda % 4 == 0 && p.ldb % 4 == 0;
    if (p.mode == BWD_DX) vec = vec && p.Nout % 4 == 0 && p.N % 4 == 0 && p.Cin % 4 == 0;
    else vec = vec && p.M % 4 == 0 && p.Cin % 4 == 0 && p.N % 4 == 0;
    return vec;
}

int launch_gemm_bwd(const BwdGemmP& p, hipStream_t s, const char* name) {
    L2S_REQUIRE(p.M > 0 && p.N > 0 && p.K > 0, "bwd gemm dims");
    BwdGemmP q = p;
    bool vec = al16(p.A) && al16(p.B) && p.lda % 4 == 0 && p.ldb % 4 == 0;
    if (p.mode == BWD_DX) vec = vec && p.Nout % 4 == 0 && p.N % 4 == 0 && p.Cin % 4 == 0;
    else vec = vec && p.M % 4 == 0 && p.Cin % 4 == 0 && p.N % 4 == 0;
    q.vec = vec ? 1 : 0;
    L2S_REQUIRE(vec || p.taps == 1, "bwd gemm: unaligned operands are supported for 1x1 layers only");
    L2S_REQUIRE(p.ksplit <= 1 || (!p.accumulate && p.c_T == 0), "split-K writes plain partial matrices (reduced by launch_gemm_bwd_splitk)");
    dim3 grid((p.N + TB - 1) / TB, (p.M + TB - 1) / TB, p.ksplit > 1 ? p.ksplit : 1);
    ProfScope ps(name, s);
    const bool bf16 = gemm_bf16_mode() != 0;                 // option "train_bf16"
#define L2S_BWD_LAUNCH(MODE_, BF_, VEC_) hipLaunchKernelGGL((gemm_bwd_kernel<MODE_, BF_, VEC_>), grid, dim3(256), 0, s, q)
    if (p.mode == BWD_DX) { if (bf16) { if (vec) L2S_BWD_LAUNCH(BWD_DX, true, true); else L2S_BWD_LAUNCH(BWD_DX, true, false); }
                            else { if (vec) L2S_BWD_LAUNCH(BWD_DX, false, true); else L2S_BWD_LAUNCH(BWD_DX, false, false); } }
    else { if (bf16) { if (vec) L2S_BWD_LAUNCH(BWD_DW, true, true); else L2S_BWD_LAUNCH(BWD_DW, true, false); }
           else { if (vec) L2S_BWD_LAUNCH(BWD_DW, false, true); else L2S_BWD_LAUNCH(BWD_DW, false, false); } }
#undef L2S_BWD_LAUNCH
    L2S_CHECK_HIP(hipGetLastError());
    return 0;
}

// C[m][n] (+)= sum_z partial[z][m][n]
__global__ __launch_bounds__(256) void reduce_splits_kernel(const float* __restrict__ partials, int nsplit, int64_t stride, int M, int N, int ldp,
                                                            float* __restrict__ C, int ldc, int accumulate) {
    const int64_t total = (int64_t)M * N;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int n = idx % N; const int64_t m = idx / N;
        float a = 0.f;
#pragma unroll 8
        for (int z = 0; z < nsplit; ++z) a += partials[(int64_t)z * stride + m * ldp + n];
        float* dst = C + m * ldc + n;
        *dst = accumulate ? *dst + a : a;
    }
}

int64_t gemm_bwd_splitk_floats(const BwdGemmP& p, int splits) { return (int64_t)splits * p.M * p.N; }

// backward GEMM with the (long) reduction split over gridDim.z; partials [splits][M][N] are summed in a fixed order. dW mode: the
// reduction runs over the rows of the batch; dX mode: over (tap, output channel) - the B*T-row input gradients of the wide
// Conv1d layers have 32 output tiles and K up to 5632, so without the split they run on an eighth of the chip.
int launch_gemm_bwd_splitk(const BwdGemmP& p, int splits, float* partials, hipStream_t s, const char* name) {
    L2S_REQUIRE(p.c_T == 0, "split-K: plain row-major output only");
    const int nkt = (p.K + TK - 1) / TK;
    if (splits > nkt) splits = nkt;
    if (splits <= 1) return launch_gemm_bwd(p, s, name);
    BwdGemmP q = p;
    q.C = partials; q.ldc = p.N; q.ksplit = splits; q.c_split_stride = (int64_t)p.M * p.N; q.accumulate = 0;
    if (launch_gemm_bwd(q, s, name)) return 1;
    const int64_t total = (int64_t)p.M * p.N;
    hipLaunchKernelGGL(reduce_splits_kernel, dim3((unsigned)std::min<int64_t>((total + 255) / 256, 4096)), dim3(256), 0, s, partials, splits, q.c_split_stride, p.M, p.N,
                       p.N, p.C, p.ldc, p.accumulate);
    L2S_CHECK_HIP(hipGetLastError());
    return 0;
}

int launch_gemm_bwd_dw_dx(const BwdGemmP& pdw, int splits, float* partials, const BwdGemmP& pdx, hipStream_t s, const char* name) {
    L2S_REQUIRE(pdw.mode == BWD_DW && pdx.mode == BWD_DX && pdw.c_T == 0 && pdx.ksplit <= 1 && pdw.taps == 1 && pdx.taps == 1, "dW + dX pair: 1x1 layers, plain outputs");
    const int nkt = (pdw.K + TK - 1) / TK;
    if (splits > nkt) splits = nkt;
    if (splits < 1) splits = 1;
    BwdGemmP qw = pdw, qx = pdx;
    if (splits > 1) { qw.C = partials; qw.ldc = pdw.N; qw.ksplit = splits; qw.c_split_stride = (int64_t)pdw.M * pdw.N; qw.accumulate = 0; }
    const bool vec = bwd_vec_ok(pdw) && bwd_vec_ok(pdx);          // one instance for the pair: 16-byte loads only when both halves allow them
    qw.vec = vec ? 1 : 0; qx.vec = qw.vec;
    const int wx = (pdw.N + TB - 1) / TB, wy = (pdw.M + TB - 1) / TB, xx = (pdx.N + TB - 1) / TB, xy = (pdx.M + TB - 1) / TB;
    {
        ProfScope ps(name, s);
        const dim3 grid(wx * wy * splits + xx * xy);
        const bool bf16 = gemm_bf16_mode() != 0;
        if (bf16 && vec) hipLaunchKernelGGL((gemm_bwd_pair_kernel<true, true>), grid, dim3(256), 0, s, qw, qx, wx, wy, splits, xx);
        else if (bf16) hipLaunchKernelGGL((gemm_bwd_pair_kernel<true, false>), grid, dim3(256), 0, s, qw, qx, wx, wy, splits, xx);
        else if (vec) hipLaunchKernelGGL((gemm_bwd_pair_kernel<false, true>), grid, dim3(256), 0, s, qw, qx, wx, wy, splits, xx);
        else hipLaunchKernelGGL((gemm_bwd_pair_kernel<false, false>), grid, dim3(256), 0, s, qw, qx, wx, wy, splits, xx);
    }
    if (splits > 1) {
        const int64_t total = (int64_t)pdw.M * pdw.N;
        hipLaunchKernelGGL(reduce_splits_kernel, dim3((unsigned)std::min<int64_t>((total + 255) / 256, 4096)), dim3(256), 0, s, partials, splits, qw.c_split_stride, pdw.M,
                           pdw.N, pdw.N, pdw.C, pdw.ldc, pdw.accumulate);
    }
    L2S_CHECK_HIP(hipGetLastError());
    return 0;
}

// dX of a Conv1d / Linear layer: dZ (B,Tz,Nout) -> dX (B,Tx,Cin); Wp [Nout][taps*Cin] (forward layout); stride 1
BwdGemmP bwd_dx(const float* dZ, int ldz, const float* Wp, float* dX, int ldx, int B, int Tz, int Tx, int Nout, int Cin, int taps, int pad,
                bool accumulate) {
    BwdGemmP p{};
    p.mode = BWD_DX; p.A = dZ; p.lda = ldz; p.B = Wp; p.ldb = taps * Cin; p.C = dX; p.ldc = ldx;
    p.M = B * Tx; p.N = Cin; p.K = taps * Nout;
    p.Tx = Tx; p.Tz = Tz; p.taps = taps; p.stride = 1; p.pad = pad; p.padp = taps - 1 - pad; p.Nout = Nout; p.Cin = Cin;
    p.alpha = 1.f; p.accumulate = accumulate ? 1 : 0;
    return p;
}
// dWp [Nout][taps*Cin] of a Conv1d / Linear layer from dZ (B,Tz,Nout) and X (B,Tx,Cin)
BwdGemmP bwd_dw(const float* dZ, int ldz, const float* X, int ldx, float* dWp, int B, int Tz, int Tx, int Nout, int Cin, int taps, int stride,
                int pad, bool accumulate) {
    BwdGemmP p{};
    p.mode = BWD_DW; p.A = dZ; p.lda = ldz; p.B = X; p.ldb = ldx; p.C = dWp; p.ldc = taps * Cin;
    p.M = Nout; p.N = taps * Cin; p.K = B * Tz;
    p.Tx = Tx; p.Tz = Tz; p.taps = taps; p.stride = stride; p.pad = pad; p.Nout = Nout; p.Cin = Cin;
    p.alpha = 1.f; p.accumulate = accumulate ? 1 : 0;
    return p;
}

}  // namespace l2s
