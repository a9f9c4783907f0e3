// Training-side primitives of the data-parallel step (reference: /root/reference/train.py:102-104,172-193 and
// train_utils/losses.py:35-79): the 4-term loss with its gradients, the global gradient norm for clipping, and a fused
// AdamW(amsgrad) update that applies the clip coefficient and the 1/world averaging of the all-reduced gradient in the
// same pass.  All reductions are two-stage (per-block partials in fp64, then one block) so results are run-to-run
// deterministic - no floating-point atomics.
#include "l2s_common.h"

#include <algorithm>

namespace l2s {

constexpr int RED_BLOCKS = 1024;

__device__ __forceinline__ double block_sum_d(double x, double* sh) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = x;
    __syncthreads();
    return (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

__global__ __launch_bounds__(256) void sumsq_partial_kernel(const float* __restrict__ x, int64_t n, double* __restrict__ partials) {
    __shared__ double sh[4];
    double acc = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const double v = x[i];
        acc += v * v;
    }
    acc = block_sum_d(acc, sh);
    if (threadIdx.x == 0) partials[blockIdx.x] = acc;
}
__global__ __launch_bounds__(256) void sumsq_final_kernel(const double* __restrict__ partials, int np, float* __restrict__ norm_out) {
    __shared__ double sh[4];
    double acc = 0.0;
    for (int i = threadIdx.x; i < np; i += 256) acc += partials[i];
    acc = block_sum_d(acc, sh);
    if (threadIdx.x == 0) norm_out[0] = (float)sqrt(acc);
}

int launch_l2_norm(const float* x, int64_t n, double* partials /*[RED_BLOCKS]*/, float* norm_out, hipStream_t s) {
    ProfScope ps("grad_l2_norm", s);
    hipLaunchKernelGGL(sumsq_partial_kernel, dim3(RED_BLOCKS), dim3(256), 0, s, x, n, partials);
    hipLaunchKernelGGL(sumsq_final_kernel, dim3(1), dim3(256), 0, s, partials, RED_BLOCKS, norm_out);
    L2S_CHECK_HIP(hipGetLastError());
    return 0;
}

// torch.optim.AdamW(amsgrad=True) on a flat parameter range; g_eff = g * grad_mul * min(1, max_norm / (norm + 1e-6))
// (torch.nn.utils.clip_grad_norm_ semantics; `norm` is the norm of g * grad_mul, i.e. *norm_dev * grad_mul).
__global__ __launch_bounds__(256) void adamw_amsgrad_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                            float* __restrict__ v, float* __restrict__ vmax, int64_t n, float lr, float b1,
                                                            float b2, float eps, float wd, float bc1, float bc2_sqrt,
                                                            const float* __restrict__ norm_dev, float grad_mul, float max_norm) {
    float scale = grad_mul;
    if (norm_dev && max_norm > 0.f) {
        const float total = norm_dev[0] * grad_mul;
        const float coef = max_norm / (total + 1e-6f);
        if (coef < 1.f) scale *= coef;
    }
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float gi = g[i] * scale;
        float pi = p[i];
        pi *= 1.f - lr * wd;
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        const float vm = fmaxf(vmax[i], vi);
        const float denom = sqrtf(vm) / bc2_sqrt + eps;
        pi -= (lr / bc1) * (mi / denom);
        p[i] = pi; m[i] = mi; v[i] = vi; vmax[i] = vm;
    }
}

int launch_adamw(float* p, const float* g, float* m, float* v, float* vmax, int64_t n, float lr, float b1, float b2, float eps, float wd,
                 int step, const float* norm_dev, float grad_mul, float max_norm, hipStream_t s) {
    const float bc1 = 1.f - powf(b1, (float)step), bc2_sqrt = sqrtf(1.f - powf(b2, (float)step));
    int blocks = (int)((n + 255) / 256);
    if (blocks > 16384) blocks = 16384;
    ProfScope ps("adamw_amsgrad_clip", s);
    hipLaunchKernelGGL(adamw_amsgrad_kernel, dim3(blocks), dim3(256), 0, s, p, g, m, v, vmax, n, lr, b1, b2, eps, wd, bc1, bc2_sqrt, norm_dev,
                       grad_mul, max_norm);
    L2S_CHECK_HIP(hipGetLastError());
    return 0;
}

// ---- loss (train_utils/losses.py:69-77): mel MSE + 10 * post-net mel MSE + BCE-with-logits on the gate + KLD(content_dis || uniform)
// term: 0 = sum (a-b)^2 over mel, 1 = same for mel_post, 2 = BCE sum, 3 = KLD sum.  Gradients are written in the same pass.
struct LossP {
    const float* mel; const float* mel_post; const float* mel_tgt;    // (B,80,S) channel-first, like the reference's outputs
    const float* stop; const float* gate;                            // (B,S)
    const float* dis;                                                // (R,V)
    float* dmel; float* dmel_post; float* dstop; float* ddis;        // gradients of the summed loss (may be null)
    int64_t n_mel, n_stop, n_dis; int R, V;
};

__global__ __launch_bounds__(256) void loss_partial_kernel(const LossP p, double* __restrict__ partials /*[4][RED_BLOCKS]*/) {
    __shared__ double sh[4];
    double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
    const int64_t stride = (int64_t)gridDim.x * 256, start = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const float inv_mel = 1.f / (float)p.n_mel, inv_stop = 1.f / (float)p.n_stop, inv_R = 1.f / (float)p.R;
    for (int64_t i = start; i < p.n_mel; i += stride) {
        const float t = p.mel_tgt[i];
        const float d0 = p.mel[i] - t, d1 = p.mel_post[i] - t;
        a0 += (double)d0 * d0;
        a1 += (double)d1 * d1;
        if (p.dmel) p.dmel[i] = 2.f * d0 * inv_mel;
        if (p.dmel_post) p.dmel_post[i] = 20.f * d1 * inv_mel;
    }
    for (int64_t i = start; i < p.n_stop; i += stride) {
        const float x = p.stop[i], y = p.gate[i];
        // BCEWithLogits: max(x,0) - x*y + log(1 + exp(-|x|))
        a2 += (double)(fmaxf(x, 0.f) - x * y + log1pf(expf(-fabsf(x))));
        if (p.dstop) p.dstop[i] = (1.f / (1.f + expf(-x)) - y) * inv_stop;
    }
    for (int64_t i = start; i < p.n_dis; i += stride) {
        const float q = p.dis[i];
        const float arg = q * (float)p.V + 1e-20f;
        const float lr = logf(arg);
        a3 += (double)(q * lr);
        if (p.ddis) p.ddis[i] = (lr + q * (float)p.V / arg) * inv_R;
    }
    a0 = block_sum_d(a0, sh); a1 = block_sum_d(a1, sh); a2 = block_sum_d(a2, sh); a3 = block_sum_d(a3, sh);
    if (threadIdx.x == 0) {
        partials[0 * RED_BLOCKS + blockIdx.x] = a0; partials[1 * RED_BLOCKS + blockIdx.x] = a1;
        partials[2 * RED_BLOCKS + blockIdx.x] = a2; partials[3 * RED_BLOCKS + blockIdx.x] = a3;
    }
}
__global__ __launch_bounds__(256) void loss_final_kernel(const double* __restrict__ partials, const LossP p, float* __restrict__ out /*[5]*/) {
    __shared__ double sh[4];
    double tot[4];
    for (int t = 0; t < 4; ++t) {
        double acc = 0.0;
        for (int i = threadIdx.x; i < RED_BLOCKS; i += 256) acc += partials[t * RED_BLOCKS + i];
        tot[t] = block_sum_d(acc, sh);
    }
    if (threadIdx.x == 0) {
        out[0] = (float)(tot[0] / (double)p.n_mel);                 // mel_loss
        out[1] = (float)(10.0 * tot[1] / (double)p.n_mel);          // postnet_mel_loss
        out[2] = (float)(tot[2] / (double)p.n_stop);                // gate_loss
        out[3] = (float)(tot[3] / (double)p.R);                     // KLD
        out[4] = out[0] + out[1] + out[2] + out[3];
    }
}

int launch_loss(const LossP& p, double* partials /*[4*RED_BLOCKS]*/, float* out5, hipStream_t s) {
    ProfScope ps("loss_fwd_bwd", s);
    hipLaunchKernelGGL(loss_partial_kernel, dim3(RED_BLOCKS), dim3(256), 0, s, p, partials);
    hipLaunchKernelGGL(loss_final_kernel, dim3(1), dim3(256), 0, s, partials, p, out5);
    L2S_CHECK_HIP(hipGetLastError());
    return 0;
}

}  // namespace l2s

// ================================================================================================ C ABI
#include "../../include/l2s.h"
namespace l2s {

// ------------------------------------------------------------------------------------------------ batch-statistics BatchNorm
// partials[blk*blk_stride + k*C + c], k = 0: sum, 1: sum of squares of the raw conv output over the rows of block blk
// one block = 16 channels x 64 lanes striding over the partial rows (up to ~8 000 of them for the widest maps), fp64 tree in LDS; with 64
// channels x 16 lanes the 58-channel layers ran as ONE block walking 33 dependent trips
__global__ __launch_bounds__(1024) void bn_stats_final_kernel(const float* __restrict__ partials, int nblk, int blk_stride, double inv_n, double unbias,
                                                              const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ rmean,
                                                              float* __restrict__ rvar, const float* __restrict__ conv_bias, float momentum, int C,
                                                              float* __restrict__ scale, float* __restrict__ shift) {
    __shared__ double sh[2][64][16];
    const int cl = threadIdx.x & 15, lane = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + cl;
    double s1 = 0.0, s2 = 0.0;
    if (c < C) {
#pragma unroll 4
        for (int b = lane; b < nblk; b += 64) { s1 += partials[(int64_t)b * blk_stride + c]; s2 += partials[(int64_t)b * blk_stride + C + c]; }
    }
    sh[0][lane][cl] = s1; sh[1][lane][cl] = s2;
    __syncthreads();
    if (lane != 0 || c >= C) return;
    s1 = 0.0; s2 = 0.0;
    for (int l = 0; l < 64; ++l) { s1 += sh[0][l][cl]; s2 += sh[1][l][cl]; }
    const double mean = s1 * inv_n;
    double var = s2 * inv_n - mean * mean;                   // biased (the normaliser); fp64 combination of fp32 tile sums
    var = var > 0.0 ? var : 0.0;
    const float rstd = (float)(1.0 / sqrt(var + (double)1e-5f));
    const float sc = gamma[c] * rstd;
    scale[c] = sc;
    shift[c] = beta[c] - (float)mean * sc;                  // a conv bias cancels against its own mean
    if (rmean) rmean[c] = (1.f - momentum) * rmean[c] + momentum * ((float)mean + (conv_bias ? conv_bias[c] : 0.f));
    if (rvar) rvar[c] = (1.f - momentum) * rvar[c] + momentum * (float)(var * unbias);
}
int bn_stats_finalize(const float* partials, int nblk, int blk_stride, int64_t count, const BnLayer& L, float momentum, hipStream_t s) {
    L2S_REQUIRE(L.gamma && L.beta && L.scale && L.shift && count > 1, "batch-norm layer not bound (l2s_train_bind incl. running statistics)");
    ProfScope ps("train_bn_stats_finalize", s);
    hipLaunchKernelGGL(bn_stats_final_kernel, dim3((L.C + 15) / 16), dim3(1024), 0, s, partials, nblk, blk_stride, 1.0 / (double)count,
                       (double)count / (double)(count - 1), L.gamma, L.beta, L.rmean, L.rvar, L.conv_bias, momentum, L.C, L.scale, L.shift);
    L2S_CHECK_HIP(hipGetLastError());
    return 0;
}

// dconv[r][c] -= scale[c] * (r0[c]/n + xhat[r][c] * r1[c]/n),  xhat = (z - beta)/gamma   (totals = [r0 | r1])
// block = 64 channels x 4 row lanes over one of `nsplit` row ranges: the per-channel constants sit in registers, no index division per element
__global__ __launch_bounds__(256) void bn_train_fix_kernel(float* __restrict__ dconv, int ld_dconv, const float* __restrict__ z, int ld_z, int cs_z, int co_z,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ scale,
                                                           const float* __restrict__ totals, float inv_n, int64_t rows, int C, int nsplit) {
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), rl = threadIdx.x >> 6, rs = blockIdx.y;
    if (c >= C) return;
    const int64_t chunk = (rows + nsplit - 1) / nsplit;
    const int64_t r_begin = rs * chunk, r_end = r_begin + chunk < rows ? r_begin + chunk : rows;
    const float be = beta[c], ga = gamma[c], sc = scale[c], t0 = totals[c], t1 = totals[C + c];
    const float* zp = z + co_z + (int64_t)c * cs_z;
    float* dp = dconv + c;
    constexpr int U = 4;
    for (int64_t r0 = r_begin + rl; r0 < r_end; r0 += 4 * U) {
        float zz[U], dd[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t r = r0 + 4 * u < r_end ? r0 + 4 * u : r_begin + rl;
            zz[u] = zp[r * ld_z]; dd[u] = dp[r * ld_dconv];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t r = r0 + 4 * u;
            if (r < r_end) {
                const float xh = (zz[u] - be) / ga;
                dp[r * ld_dconv] = dd[u] - sc * (t0 + xh * t1) * inv_n;
            }
        }
    }
}
int bn_train_fix(float* dconv, int ld_dconv, const float* z, int ld_z, int cs_z, int co_z, const float* gamma, const float* beta, const float* scale,
                 const float* totals, int64_t rows, int C, hipStream_t s) {
    ProfScope ps("train_bn_batchstat_bwd", s);
    const int nsplit = (int)std::min<int64_t>(1024, std::max<int64_t>(1, (rows + 63) / 64));
    hipLaunchKernelGGL(bn_train_fix_kernel, dim3((C + 63) / 64, nsplit), dim3(256), 0, s, dconv, ld_dconv ? ld_dconv : C, z, ld_z ? ld_z : C,
                       cs_z ? cs_z : 1, co_z, gamma, beta, scale, totals, 1.f / (float)rows, rows, C, nsplit);
    L2S_CHECK_HIP(hipGetLastError());
    return 0;
}

// depthwise 3x3 (pad 1) statistics pass over channel-last maps: raw conv recomputed per output element, two-stage column sums
__global__ __launch_bounds__(256) void dwconv_stats_kernel(const float* __restrict__ in, int N, int Hi, int Wi, int ldi, int ci_off, int C, int stride,
                                                           const float* __restrict__ w9, int Ho, int Wo, float* __restrict__ partials,
                                                           float* __restrict__ raw_out, int ldo, int co_off) {
    __shared__ float sh[2][4][64];
    const int col = blockIdx.x * 64 + (threadIdx.x & 63), rl = threadIdx.x >> 6, rs = blockIdx.y;
    const int64_t rows = (int64_t)N * Ho * Wo, chunk = (rows + DWS_RS - 1) / DWS_RS;
    const int64_t r_begin = rs * chunk, r_end = r_begin + chunk < rows ? r_begin + chunk : rows;
    float s1 = 0.f, s2 = 0.f;
    if (col < C) {
        float wk[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) wk[k] = w9[k * C + col];
        // U rows per trip with their 9 x U loads in flight together (one row per trip was a chain of dependent round trips: 32 us for the
        // 12x12 maps); rows ascending within the thread as before, so the sums are the same numbers
        constexpr int U = 4;
        for (int64_t r0 = r_begin + rl; r0 < r_end; r0 += 4 * U) {
            float xv[U][9];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t r = r0 + 4 * u;
                const bool live = r < r_end;
                const int64_t rr = live ? r : r_begin;
                const int ow = rr % Wo; const int64_t q = rr / Wo;
                const int oh = q % Ho, n = q / Ho;
#pragma unroll
                for (int kh = 0; kh < 3; ++kh) {
                    const int ih = oh * stride + kh - 1;
#pragma unroll
                    for (int kw = 0; kw < 3; ++kw) {
                        const int iw = ow * stride + kw - 1;
                        const bool in_ = live && ih >= 0 && ih < Hi && iw >= 0 && iw < Wi;
                        xv[u][kh * 3 + kw] = in_ ? in[(((int64_t)n * Hi + ih) * Wi + iw) * ldi + ci_off + col] : 0.f;
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (r0 + 4 * u < r_end) {
                    float acc = 0.f;
#pragma unroll
                    for (int k = 0; k < 9; ++k) acc = fmaf(xv[u][k], wk[k], acc);
                    s1 += acc; s2 += acc * acc;
                    if (raw_out) raw_out[(r0 + 4 * u) * ldo + co_off + col] = acc;     // parked for bn_apply_kernel: one pass over the conv
                }
            }
        }
    }
    sh[0][rl][threadIdx.x & 63] = s1; sh[1][rl][threadIdx.x & 63] = s2;
    __syncthreads();
    if (rl == 0 && col < C) {
        const int c = threadIdx.x & 63;
        partials[((int64_t)rs * 2 + 0) * C + col] = (sh[0][0][c] + sh[0][1][c]) + (sh[0][2][c] + sh[0][3][c]);
        partials[((int64_t)rs * 2 + 1) * C + col] = (sh[1][0][c] + sh[1][1][c]) + (sh[1][2][c] + sh[1][3][c]);
    }
}
int launch_dwconv_stats(const float* in, int N, int Hi, int Wi, int ldi, int ci_off, int C, int stride, const float* w9, float* partials, hipStream_t s,
                        float* raw_out, int ldo, int co_off) {
    const int Ho = (Hi + 2 - 3) / stride + 1, Wo = (Wi + 2 - 3) / stride + 1;
    ProfScope ps("train_dwconv_stats", s);
    hipLaunchKernelGGL(dwconv_stats_kernel, dim3((C + 63) / 64, DWS_RS), dim3(256), 0, s, in, N, Hi, Wi, ldi, ci_off, C, stride, w9, Ho, Wo, partials, raw_out, ldo, co_off);
    L2S_CHECK_HIP(hipGetLastError());
    return 0;
}
// x[r][co_off + c] = x[r][co_off + c] * scale[c] + shift[c] over a channel-last map: the BatchNorm of a depthwise conv whose raw output the
// statistics pass parked in place (the expression of dwconv3x3_kernel's last line)
template <bool FUSED>        // FUSED: one rounding (fmaf) - what the front-end's fused epilogue compiles `acc * sc + sh` to; else product and sum rounded
                             // separately - what dwconv3x3_kernel's last line compiles to (tools/hash_train_step.py tells them apart)
__global__ __launch_bounds__(256) void bn_apply_kernel(float* __restrict__ x, int64_t rows, int C, int ld, int co_off, const float* __restrict__ scale,
                                                       const float* __restrict__ shift, int nsplit) {
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), rl = threadIdx.x >> 6, rs = blockIdx.y;
    if (c >= C) return;
    const int64_t chunk = (rows + nsplit - 1) / nsplit;
    const int64_t r_begin = rs * chunk, r_end = r_begin + chunk < rows ? r_begin + chunk : rows;
    const float sc = scale[c], sh = shift[c];
    float* xp = x + co_off + c;
    constexpr int U = 4;
    for (int64_t r0 = r_begin + rl; r0 < r_end; r0 += 4 * U) {
        float v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = xp[(r0 + 4 * u < r_end ? r0 + 4 * u : r_begin + rl) * ld];
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (r0 + 4 * u < r_end) { const float acc = v[u]; xp[(r0 + 4 * u) * ld] = FUSED ? fmaf(acc, sc, sh) : __fadd_rn(__fmul_rn(acc, sc), sh); }
    }
}
int launch_bn_apply(float* x, int64_t rows, int C, int ld, int co_off, const float* scale, const float* shift, hipStream_t s, bool fused) {
    ProfScope ps("train_bn_apply", s);
    const int nsplit = (int)std::min<int64_t>(1024, std::max<int64_t>(1, (rows + 63) / 64));
    if (fused) hipLaunchKernelGGL(bn_apply_kernel<true>, dim3((C + 63) / 64, nsplit), dim3(256), 0, s, x, rows, C, ld, co_off, scale, shift, nsplit);
    else hipLaunchKernelGGL(bn_apply_kernel<false>, dim3((C + 63) / 64, nsplit), dim3(256), 0, s, x, rows, C, ld, co_off, scale, shift, nsplit);
    L2S_CHECK_HIP(hipGetLastError());
    return 0;
}

}  // namespace l2s

using namespace l2s;

extern "C" {

int64_t l2s_train_scratch_bytes(void) { return (int64_t)4 * RED_BLOCKS * sizeof(double); }

int l2s_grad_norm(const float* grads, int64_t n, void* scratch, float* norm_out, void* stream) {
    L2S_REQUIRE(grads && scratch && norm_out && n > 0, "bad arguments");
    return launch_l2_norm(grads, n, (double*)scratch, norm_out, (hipStream_t)stream);
}

int l2s_adamw_amsgrad_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, float* max_exp_avg_sq, int64_t n,
                           float lr, float beta1, float beta2, float eps, float weight_decay, int step, const float* grad_norm,
                           float grad_mul, float max_norm, void* stream) {
    L2S_REQUIRE(params && grads && exp_avg && exp_avg_sq && max_exp_avg_sq && n > 0 && step >= 1, "bad arguments");
    return launch_adamw(params, grads, exp_avg, exp_avg_sq, max_exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay, step, grad_norm,
                        grad_mul, max_norm, (hipStream_t)stream);
}

int l2s_loss(const float* mel, const float* mel_post, const float* mel_target, const float* stop, const float* gate_target,
             const float* content_dis, int B, int S, int R, float* losses5, float* dmel, float* dmel_post, float* dstop, float* ddis,
             void* scratch, void* stream) {
    L2S_REQUIRE(mel && mel_post && mel_target && stop && gate_target && content_dis && losses5 && scratch, "bad arguments");
    LossP p{};
    p.mel = mel; p.mel_post = mel_post; p.mel_tgt = mel_target; p.stop = stop; p.gate = gate_target; p.dis = content_dis;
    p.dmel = dmel; p.dmel_post = dmel_post; p.dstop = dstop; p.ddis = ddis;
    p.n_mel = (int64_t)B * L2S_N_MELS * S; p.n_stop = (int64_t)B * S; p.R = R; p.V = L2S_VOCAB; p.n_dis = (int64_t)R * L2S_VOCAB;
    return launch_loss(p, (double*)scratch, losses5, (hipStream_t)stream);
}

}  // extern "C"
