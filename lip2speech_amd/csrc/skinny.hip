// Batch-row ("skinny") kernels of the autoregressive decode step and of the BiLSTM recurrence
// (reference/model/modules/decoder.py:412-429, 353-375, 389; SURVEY.md Appendix A "Step i").
//
// Every dense op inside a step has M = batch rows (<= 32 at the benchmark size) and a weight matrix that
// is read exactly once, so there is nothing to tile through LDS: weights and activations are kept in the
// "frag16" layout (l2s_common.h) in which one coalesced 1-KiB float4 load per wave is exactly the operand
// set of four v_mfma_f32_16x16x4_f32.  A block = 4 waves owns one 16-column output tile for one 16-row
// batch tile; the waves split K, partial tiles are summed through LDS and the fused epilogue (PSine,
// SiLU, LSTM cell, mel/stop store) runs on the reduced tile.  Tiles of several independent ops of the
// same step phase share one launch (groups), and the attention role runs in the same launch as the
// second prenet layer (step_attn_kernel).
#include "l2s_common.h"

namespace l2s {

__device__ __forceinline__ float act_apply(float v, int act, const float* actw, int n) {
    if (act == ACT_RELU) return v > 0.f ? v : 0.f;
    if (act == ACT_SILU) return v / (1.0f + expf(-v));
    if (act == ACT_PSINE) return sinf(v) * actw[n];
    return v;
}
__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// One block: output tile `tile` (16 columns) x batch tile `mt` (16 rows).
__device__ __forceinline__ void skinny_block(const SkinnyP& p, int tile, int mt, float* red /*[4][16][17] + [16][17]*/) {
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int NC = p.K >> 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const float4* wbase = reinterpret_cast<const float4*>(p.W) + (int64_t)tile * NC * 64 + lane;

    int c_lo = 0;
    for (int sidx = 0; sidx < p.nseg; ++sidx) {
        const int n = p.seg[sidx].nchunks;
        const float4* abase = reinterpret_cast<const float4*>(p.seg[sidx].a) + (int64_t)mt * n * 64 + lane;
        // this wave takes the chunks c of the segment with (c_lo + c) % 4 == wave
        int first = (wave - (c_lo & 3) + 4) & 3;
#pragma unroll 4
        for (int c = first; c < n; c += 4) {
            const float4 a4 = abase[(int64_t)c * 64];
            const float4 w4 = wbase[(int64_t)(c_lo + c) * 64];
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.x, w4.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.y, w4.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.z, w4.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.w, w4.w, acc, 0, 0, 0);
        }
        c_lo += n;
    }
    // D layout: col = lane&15, row = 4*(lane>>4) + r
    {
        const int col = lane & 15, rb = 4 * (lane >> 4);
#pragma unroll
        for (int r = 0; r < 4; ++r) red[(wave * 16 + rb + r) * 17 + col] = acc[r];
    }
    __syncthreads();
    const int row = tid >> 4, col = tid & 15;
    const int b = mt * 16 + row;
    const int np = tile * 16 + col;                       // (permuted) weight row
    float v = red[(0 * 16 + row) * 17 + col] + red[(1 * 16 + row) * 17 + col] + red[(2 * 16 + row) * 17 + col] +
              red[(3 * 16 + row) * 17 + col];
    if (p.bias) v += p.bias[np];

    if (p.epi == SK_LSTM) {
        float* gt = red + 4 * 16 * 17;                    // reduced gate tile [16][17]
        const int u = col >> 2, gate = col & 3;
        const int unit = tile * 4 + u;
        if (p.pre && b < p.B) v += p.pre[(int64_t)b * p.ld_pre + gate * p.H + unit];
        gt[row * 17 + col] = v;
        __syncthreads();
        if (tid < 64) {
            const int r2 = tid >> 2, u2 = tid & 3;
            const int b2 = mt * 16 + r2, unit2 = tile * 4 + u2;
            if (b2 < p.B) {
                const float gi = gt[r2 * 17 + 4 * u2 + 0], gf = gt[r2 * 17 + 4 * u2 + 1];
                const float gg = gt[r2 * 17 + 4 * u2 + 2], go = gt[r2 * 17 + 4 * u2 + 3];
                const int64_t ci = frag16_index(b2, unit2, p.H);
                const float cprev = p.c_in[ci];
                const float cn = sigmoidf_(gf) * cprev + sigmoidf_(gi) * tanhf(gg);
                const float hn = sigmoidf_(go) * tanhf(cn);
                p.c_out[ci] = cn;
                p.h_out[frag16_index(b2, p.h_out_off + unit2, p.h_out_K)] = hn;
                if (p.h_seq) p.h_seq[(int64_t)b2 * p.ld_hseq + unit2] = hn;
                if (p.h_plain) p.h_plain[(int64_t)b2 * p.ld_hplain + unit2] = hn;
            }
        }
        return;
    }
    if (b >= p.B) return;
    if (p.epi == SK_MEL) {
        if (np < 80) {
            p.mel[(int64_t)b * p.ld_mel_b + np] = v;
            p.yfrag[frag16_index(b, np, 80)] = v;
        } else if (np == 80) {
            p.stop[(int64_t)b * p.ld_stop_b] = v + p.stop_const[b];
        }
        return;
    }
    if (np >= p.N) return;
    v = act_apply(v, p.act, p.actw, np);
    if (p.add) v += p.add[(int64_t)b * p.ld_add + np];
    if (p.addrow) v += p.addrow[np];
    if (p.epi == SK_FRAG)
        p.out[frag16_index(b, np, p.ldo)] = v;
    else
        p.out[(int64_t)b * p.ldo + np] = v;
}

__global__ __launch_bounds__(256) void skinny_kernel(const SkinnyBatch batch) {
    __shared__ float red[5 * 16 * 17];
    const int g = blockIdx.z;
    if ((int)blockIdx.x >= batch.ntiles[g]) return;
    skinny_block(batch.p[g], blockIdx.x, blockIdx.y, red);
}

int launch_skinny(const SkinnyBatch& b, hipStream_t s, const char* name) {
    L2S_REQUIRE(b.count >= 1 && b.count <= SKINNY_MAX_GROUP, "skinny group size");
    int maxt = 0, mts = 0;
    for (int i = 0; i < b.count; ++i) {
        const SkinnyP& p = b.p[i];
        int k = 0;
        for (int j = 0; j < p.nseg; ++j) k += 16 * p.seg[j].nchunks;
        L2S_REQUIRE(k == p.K && p.K % 16 == 0, "skinny K segments");
        L2S_REQUIRE(p.B >= 1, "skinny B");
        maxt = b.ntiles[i] > maxt ? b.ntiles[i] : maxt;
        int m = (p.B + 15) / 16;
        L2S_REQUIRE(mts == 0 || mts == m, "skinny groups must share B");
        mts = m;
    }
    ProfScope ps(name, s);
    hipLaunchKernelGGL(skinny_kernel, dim3(maxt, mts, b.count), dim3(256), 0, s, b);
    L2S_CHECK_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------------------------
// Attention role (decoder.py:414-419,421 and Content.forward :262-271), one block per batch row:
//   a = softmax_T(q*tau . k),  av = a @ v ;  alpha = softmax_m(qc*tau_c . key),  cc = alpha @ value
// The 512-/256-long dot products are accumulated in fp64 (free at this size), so the logits - whose magnitude
// reaches several thousand because tau multiplies - carry only the fp32 rounding of their inputs.
// Launched together with the second prenet layer (independent of it) in one grid.
struct StepB {
    AttnP at;
    SkinnyP pre2;
    int pre2_tiles;
    int mts;
};

__device__ __forceinline__ double wave_sum_d(double x) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o);
    return x;
}
__device__ __forceinline__ float wave_max_f(float x) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x = fmaxf(x, __shfl_xor(x, o));
    return x;
}
__device__ __forceinline__ float wave_sum_f(float x) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o);
    return x;
}

constexpr int ATT_MAXT = 320;

__device__ __forceinline__ float block_max(float x, float* scratch) {
    x = wave_max_f(x);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = x;
    __syncthreads();
    return fmaxf(fmaxf(scratch[0], scratch[1]), fmaxf(scratch[2], scratch[3]));
}
__device__ __forceinline__ float block_sum(float x, float* scratch) {
    x = wave_sum_f(x);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = x;
    __syncthreads();
    return (scratch[0] + scratch[1]) + (scratch[2] + scratch[3]);
}

__device__ __forceinline__ void attention_block(const AttnP& p, int b, float* sm) {
    float* qs = sm;                 // 512
    float* sc = sm + 512;           // ATT_MAXT
    float* scratch = sc + ATT_MAXT; // 8
    float* csc = scratch + 8;       // 16
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const float tau = p.tau[0];
    for (int j = tid; j < 512; j += 256) qs[j] = p.q[(int64_t)b * p.ldq + j] * tau;
    __syncthreads();
    const float* kb = p.k + (int64_t)b * p.T * 512;
    for (int t = wave; t < p.T; t += 4) {
        const float4 k0 = *reinterpret_cast<const float4*>(kb + (int64_t)t * 512 + lane * 8);
        const float4 k1 = *reinterpret_cast<const float4*>(kb + (int64_t)t * 512 + lane * 8 + 4);
        const float* qq = qs + lane * 8;
        double d = (double)qq[0] * k0.x + (double)qq[1] * k0.y + (double)qq[2] * k0.z + (double)qq[3] * k0.w +
                   (double)qq[4] * k1.x + (double)qq[5] * k1.y + (double)qq[6] * k1.z + (double)qq[7] * k1.w;
        d = wave_sum_d(d);
        if (lane == 0) sc[t] = (float)d;
    }
    __syncthreads();
    // softmax over T
    float mx = -INFINITY;
    for (int t = tid; t < p.T; t += 256) mx = fmaxf(mx, sc[t]);
    mx = block_max(mx, scratch);
    float part = 0.f;
    float ex[2] = {0.f, 0.f};
    int cnt = 0;
    for (int t = tid; t < p.T; t += 256, ++cnt) {
        if (p.attn_out && p.attn_logits) p.attn_out[(int64_t)b * p.ld_attn_b + t] = sc[t];
        ex[cnt] = expf(sc[t] - mx);
        part += ex[cnt];
    }
    const float tot = block_sum(part, scratch);
    cnt = 0;
    for (int t = tid; t < p.T; t += 256, ++cnt) {
        const float a = ex[cnt] / tot;
        sc[t] = a;
        if (p.attn_out && !p.attn_logits) p.attn_out[(int64_t)b * p.ld_attn_b + t] = a;
    }
    __syncthreads();
    // av = a @ v : thread owns columns tid and tid+256
    {
        const float* vb = p.v + (int64_t)b * p.T * 512;
        float a0 = 0.f, a1 = 0.f;
        for (int t = 0; t < p.T; ++t) {
            const float a = sc[t];
            a0 = fmaf(a, vb[(int64_t)t * 512 + tid], a0);
            a1 = fmaf(a, vb[(int64_t)t * 512 + 256 + tid], a1);
        }
        p.av_frag[frag16_index(b, tid, 512)] = a0;
        p.av_frag[frag16_index(b, tid + 256, 512)] = a1;
    }
    // content attention (m <= 16 slots)
    __syncthreads();
    const float tau_c = p.tau_c[0];
    qs[tid] = p.qc[(int64_t)b * p.ldqc + tid] * tau_c;
    __syncthreads();
    const float* keyb = p.ckey + (int64_t)b * p.m * 256;
    for (int i = wave; i < p.m; i += 4) {
        const float4 k0 = *reinterpret_cast<const float4*>(keyb + (int64_t)i * 256 + lane * 4);
        const float* qq = qs + lane * 4;
        double d = (double)qq[0] * k0.x + (double)qq[1] * k0.y + (double)qq[2] * k0.z + (double)qq[3] * k0.w;
        d = wave_sum_d(d);
        if (lane == 0) csc[i] = d;
    }
    __syncthreads();
    float cmx = -INFINITY;
    for (int i = 0; i < p.m; ++i) cmx = fmaxf(cmx, csc[i]);
    float csum = 0.f;
    for (int i = 0; i < p.m; ++i) csum += expf(csc[i] - cmx);
    const float* valb = p.cval + (int64_t)b * p.m * 256;
    float o = 0.f;
    for (int i = 0; i < p.m; ++i) o = fmaf(expf(csc[i] - cmx) / csum, valb[(int64_t)i * 256 + tid], o);
    p.cc_frag[frag16_index(b, tid, 256)] = o;
}

__global__ __launch_bounds__(256) void step_attn_kernel(const StepB sb) {
    __shared__ __attribute__((aligned(16))) float sm[512 + ATT_MAXT + 8 + 16 + 5 * 16 * 17];
    const int nb = sb.at.B;
    if ((int)blockIdx.x < nb) {
        attention_block(sb.at, blockIdx.x, sm);
    } else {
        const int j = blockIdx.x - nb;
        const int tile = j % sb.pre2_tiles, mt = j / sb.pre2_tiles;
        skinny_block(sb.pre2, tile, mt, sm);
    }
}

int launch_step_attn(const AttnP& at, const SkinnyP& pre2, int pre2_tiles, hipStream_t s) {
    L2S_REQUIRE(at.T <= ATT_MAXT && at.m <= 16, "attention sizes");
    StepB sb;
    sb.at = at;
    sb.pre2 = pre2;
    sb.pre2_tiles = pre2_tiles;
    sb.mts = (at.B + 15) / 16;
    ProfScope ps("step_attention_prenet2", s);
    hipLaunchKernelGGL(step_attn_kernel, dim3(at.B + pre2_tiles * sb.mts), dim3(256), 0, s, sb);
    L2S_CHECK_HIP(hipGetLastError());
    return 0;
}

}  // namespace l2s
