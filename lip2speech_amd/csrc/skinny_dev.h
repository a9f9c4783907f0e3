// Device code of the batch-row ("skinny") blocks, shared by the inference kernels (skinny.hip) and the training-forward
// kernels (train_decoder.hip).  TRAIN=true adds the stores the backward pass needs (pre-activations, LSTM gates, plain
// copies); with TRAIN=false those paths are compiled out and the code is the tuned inference code.
#pragma once
#include "l2s_common.h"

#include <type_traits>
#include <utility>

namespace l2s {

struct SkinnyTrain {               // training-side stores of one skinny group (any pointer may be null)
    float* zsave; int ld_z;            // pre-activation value (after bias / pre-gates), plain [b*ld_z + n]
    float* gates; int64_t ld_gates;    // SK_LSTM: post-nonlinearity i,f,g,o at [b*ld_gates + gate*H + unit]
    float* c_new; int64_t ld_c;        // SK_LSTM: plain copy of the new cell state [b*ld_c + unit]
    float* out_plain; int ld_out;      // SK_FRAG: plain copy of the output [b*ld_out + n]
    const float* out_mask; int ld_mask; // dropout multiplier (0 or 1/(1-p)) applied to the activated output [b*ld_mask + n]
    float* h_drop; int h_drop_K; const float* h_mask; int ld_hmask;   // SK_LSTM: second frag16 copy of the new hidden state times a dropout mask
    // SK_PLAIN, backward loop: the LSTM-cell backward fused into the GEMM that produces dh (columns n < lb_H are hidden units). dh = the
    // epilogue's value, or lb_dha[b][n] + value * lb_mask[b][n] when lb_dha is set; gate gradients go out as frag16 (K = 4*lb_H) + stack
    const float* lb_gates; const float* lb_cprev; const float* lb_cnew; float* lb_dc; const float* lb_dha; int lb_ld_a; const float* lb_mask;
    float* lb_frag; float* lb_stack; int lb_H; int lb_ld_dc;     // lb_ld_dc: row pitch of lb_dc (0 = lb_H)
    float* lb_stack2; int64_t lb_ld_stack2;                       // a second copy of the gate gradients with its own row pitch (may be null)
    // SK_PLAIN, backward loop: columns n >= add_hi_from take their addend from add_hi[b*ld_add + n] instead of add
    const float* add_hi; int add_hi_from;
    // SK_PLAIN, backward loop: side outputs on the column block [sd_lo, sd_hi), C = sd_hi - sd_lo, c = n - sd_lo: d = value (* sd_mask[b*C + c])
    // -> sd_stack[b*C + c] and frag16 sd_frag_d (K = C; may be null); through a PSine, d * cos(sd_z[b*C + c]) * sd_w[c] -> frag16 sd_frag_dz
    // (what du_dz2_kernel and the second half of carry_dz1_kernel did in launches of their own)
    const float* sd_z; const float* sd_w; const float* sd_mask; float* sd_stack; float* sd_frag_d; float* sd_frag_dz; int sd_lo, sd_hi;
    float* sd_stack_dz;                // plain copy of the PSine-backward values [b*C + c] (may be null)
};
struct AttnTrain {
    const float* logit_mask; int ld_lmask;   // dropout multiplier on the attention logits [b*ld_lmask + t] (decoder.py:363)
    float* alpha; int ld_alpha;        // content attention weights [b*ld_alpha + j]
    float* av_plain;                   // [b*512 + c]
    float* cc_plain;                   // [b*256 + c]
};

// compile-time loop: f(std::integral_constant<int, I>) for I in [B, E)
template <int B, int E, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (B < E) {
        f(std::integral_constant<int, B>{});
        static_for<B + 1, E>(f);
    }
}

__device__ __forceinline__ float act_apply(float v, int act, const float* actw, int n) {
    if (act == ACT_RELU) return v > 0.f ? v : 0.f;
    if (act == ACT_SILU) return v / (1.0f + expf(-v));
    if (act == ACT_PSINE) return sinf(v) * actw[n];
    return v;
}
__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// One block (8 waves): output tile `tile` (16 columns) x batch tile `mt` (16 rows).  Wave w owns the K chunks
// c = w, w+8, w+16, ...; ALL of its operand loads (<= 12 A + 12 W float4 per lane) are issued before the first MFMA so
// the whole K slice is one round trip to L2/HBM instead of a load->MFMA->load chain.
constexpr int SK_WAVES = 8;
constexpr int SK_MAXC = 12;                 // chunks per wave: K <= 16 * 8 * 12 = 1536

__device__ __forceinline__ f32x4 mfma4(const float4& a, const float4& w, f32x4 acc) {
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, w.x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, w.y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, w.z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, w.w, acc, 0, 0, 0);
    return acc;
}

// Pin a set of wave-uniform kernel parameters in SGPRs at this point.  Without it hipcc fetches each parameter from the kernarg
// segment lazily, right before its first use, behind its own `s_waitcnt lgkmcnt(0)`: the block start becomes a chain of ~15
// dependent scalar-cache round trips (the kernarg was just written by the host, so they miss) in front of the first operand load.
#define L2S_PIN_S(...) asm volatile("" ::__VA_ARGS__)

// TIMED: a measurement build of the same block that drops 100 MHz wall-clock stamps of its phases into ts[0..7] (tools/skinny_timeline.py)
#define L2S_STAMP(i) do { if (TIMED) { if (threadIdx.x == 0) ts[i] = wall_clock64(); } } while (0)
// K-segment layout known at compile time (chunks per segment, every boundary a multiple of the 8 waves): which segment a wave's j-th
// chunk belongs to is then a constant, and the block's load-issue phase loses its per-chunk compare / branch / 64-bit address chains
// (24 loads used to sit behind ~250 scalar instructions and ~50 branches per wave).  SegRuntime keeps the general path.
struct SegRuntime { static constexpr bool STATIC = false; static constexpr int n0 = 0, n1 = 0, n2 = 0, n3 = 0, NC = 0; };
template <int A0, int A1, int A2, int A3, bool SUM1_ = false>
struct SegLay {
    static constexpr bool STATIC = true;
    static constexpr bool SUM1 = SUM1_;          // segment 1 has a second source (SkinnyP::a_sum) added by the loader
    static constexpr int n0 = A0, n1 = A1, n2 = A2, n3 = A3, NC = A0 + A1 + A2 + A3;
    static_assert(A0 % SK_WAVES == 0 && A1 % SK_WAVES == 0 && A2 % SK_WAVES == 0 && A3 % SK_WAVES == 0, "segment boundaries on wave multiples");
};

template <bool TRAIN = false, int MAXC = SK_MAXC, bool TIMED = false, class LAY = SegRuntime, int SPLIT = 1>
__device__ __forceinline__ void skinny_block(const SkinnyP& p, int tile, int mt, float* red /*[8][16][17] + [16][17]*/, int ntiles = 1 << 30,
                                             const SkinnyTrain* tr = nullptr, unsigned long long* ts = nullptr) {
    L2S_STAMP(0);
    // ---- the parameters the operand loads need first; the rest is fetched while those loads are in flight (a single-group kernel with these
    // arguments preloaded into SGPRs by the command processor, -amdgpu-kernarg-preload-count, moved the first load 0.2 us earlier and the
    // kernel end not at all: the block is bound by its 128-192 KB of operand traffic through one CU's vector-memory path)
    const float* const W = p.W;
    const float* const sa0 = p.seg[0].a; const float* const sa1 = p.seg[1].a; const float* const sa2 = p.seg[2].a; const float* const sa3 = p.seg[3].a;
    const int n0 = p.seg[0].nchunks, n1 = p.seg[1].nchunks, n2 = p.seg[2].nchunks, n3 = p.seg[3].nchunks;
    const int K = p.K;
    L2S_PIN_S("s"(W), "s"(sa0), "s"(sa1), "s"(sa2), "s"(sa3), "s"(n0), "s"(n1), "s"(n2), "s"(n3), "s"(K), "s"(ntiles));
    if (tile >= ntiles) return;                      // block-uniform: grid x is sized for the widest group of the launch
    L2S_STAMP(1);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int NC = LAY::STATIC ? LAY::NC : (K >> 4);
    const float4* wbase = reinterpret_cast<const float4*>(W) + (int64_t)tile * NC * 64 + lane;
    const int e0 = n0, e1 = e0 + n1, e2 = e1 + n2;

    // ---- main operand loads first: the whole K slice of this wave in one round trip
    float4 a[MAXC], w[MAXC];
    auto load_chunk = [&](int j) {
        const int c = wave + SK_WAVES * j;          // wave-uniform
        if (c < NC) {
            const float* ab = sa0; int lc = c, nn = n0;
            if (c >= e2) { ab = sa3; lc = c - e2; nn = n3; }
            else if (c >= e1) { ab = sa2; lc = c - e1; nn = n2; }
            else if (c >= e0) { ab = sa1; lc = c - e0; nn = n1; }
            a[j] = reinterpret_cast<const float4*>(ab)[((int64_t)mt * nn + lc) * 64 + lane];
            w[j] = wbase[(int64_t)c * 64];      // default cache policy: the tile is read by both row-tile blocks of its XCD (nt: +8 % per step)
        }
    };
    if (LAY::STATIC) {
        // one scalar base per segment (row tile mt, this wave's first chunk of it); everything per chunk is a constant after unrolling
        const float4* sb0 = reinterpret_cast<const float4*>(sa0) + ((int64_t)mt * LAY::n0 + wave) * 64 + lane;
        const float4* sb1 = reinterpret_cast<const float4*>(sa1) + ((int64_t)mt * LAY::n1 + wave) * 64 + lane;
        const float4* sb2 = reinterpret_cast<const float4*>(sa2) + ((int64_t)mt * LAY::n2 + wave) * 64 + lane;
        const float4* sb3 = reinterpret_cast<const float4*>(sa3) + ((int64_t)mt * LAY::n3 + wave) * 64 + lane;
        const float4* wb = wbase + (int64_t)wave * 64;
        constexpr int E0 = LAY::n0, E1 = E0 + LAY::n1, E2 = E1 + LAY::n2;
#pragma unroll
        for (int j = 0; j < MAXC; ++j) {
            const int cj = SK_WAVES * j;
            if (cj < LAY::NC) {
                if (cj >= E2) a[j] = sb3[(cj - E2) * 64];
                else if (cj >= E1) a[j] = sb2[(cj - E1) * 64];
                else if (cj >= E0) a[j] = sb1[(cj - E0) * 64];
                else a[j] = sb0[cj * 64];
                w[j] = wb[cj * 64];
            }
        }
    } else {
        // SPLIT > 1: only the first 1/SPLIT of the wave's chunks now, the next batch after this batch's MFMAs (SPLIT round trips, 1/SPLIT of the
        // operand registers: an instance that fits three or four blocks per CU instead of two - for batches in flight)
#pragma unroll
        for (int j = 0; j < MAXC / SPLIT; ++j) load_chunk(j);
    }
    // ---- everything else the block needs, fetched in one batch of scalar loads that overlaps the operand loads already in flight
    const int epi = p.epi, nB = p.B, N = p.N, H = p.H, act = p.act;
    const float* const bias = p.bias; const float* const pre = p.pre; const int64_t ld_pre = p.ld_pre;
    const float* const c_in = p.c_in; const float* const add = p.add; const int ld_add = p.ld_add; const float* const addrow = p.addrow;
    L2S_PIN_S("s"(epi), "s"(nB), "s"(N), "s"(H), "s"(act), "s"(bias), "s"(pre), "s"(ld_pre), "s"(c_in), "s"(add), "s"(ld_add), "s"(addrow));
    // ---- epilogue operands have launch-time addresses too: fetch them under the same round trip
    const int e_row = tid >> 4, e_col = tid & 15;
    const int e_b = mt * 16 + e_row, e_np = tile * 16 + e_col;
    float pf_bias = 0.f, pf_extra = 0.f, pf_c = 0.f;
    if (tid < 256) {
        if (bias) pf_bias = bias[e_np];
        if (epi == SK_LSTM) {
            if (pre && e_b < nB) pf_extra = pre[(int64_t)e_b * ld_pre + (e_col & 3) * H + tile * 4 + (e_col >> 2)];
        } else if (epi != SK_MEL && e_b < nB && e_np < N) {
            if (add) {
                const float* ad = add;
                if constexpr (TRAIN) { if (tr->add_hi && e_np >= tr->add_hi_from) ad = tr->add_hi; }
                pf_extra = ad[(int64_t)e_b * ld_add + e_np];
            }
            if (addrow) pf_extra += addrow[e_np];
        }
    }
    if (epi == SK_LSTM && tid < 64) {
        const int b2 = mt * 16 + (tid >> 2);
        if (b2 < nB) pf_c = c_in[frag16_index(b2, tile * 4 + (tid & 3), H)];
    }
    float lb_g[4] = {0.f, 0.f, 0.f, 0.f}, lb_cn = 0.f, lb_cp = 0.f, lb_dcv = 0.f, lb_a = 0.f, lb_m = 1.f;
    bool lb_on = false;
    if constexpr (TRAIN) {
        lb_on = tr->lb_gates != nullptr && tid < 256 && e_b < nB && e_np < tr->lb_H;
        if (lb_on) {                                   // the cell's tape values, fetched with everything else
            const int LH = tr->lb_H;
            const int64_t ei = (int64_t)e_b * LH + e_np;
            const float* g = tr->lb_gates + (int64_t)e_b * 4 * LH + e_np;
            lb_g[0] = g[0]; lb_g[1] = g[LH]; lb_g[2] = g[2 * LH]; lb_g[3] = g[3 * LH];
            lb_cn = tr->lb_cnew[ei]; lb_cp = tr->lb_cprev[ei]; lb_dcv = tr->lb_dc[(int64_t)e_b * (tr->lb_ld_dc ? tr->lb_ld_dc : LH) + e_np];
            if (tr->lb_dha) lb_a = tr->lb_dha[(int64_t)e_b * tr->lb_ld_a + e_np];
            if (tr->lb_mask) lb_m = tr->lb_mask[ei];
        }
    }
    // side outputs of the backward loop: their tape operands too are fetched under the operand round trip
    float sd_zv = 0.f, sd_wv = 0.f, sd_mv = 1.f;
    bool sd_on = false;
    if constexpr (TRAIN) {
        const bool live = tid < 256 && e_b < nB && e_np < N && epi == SK_PLAIN;
        sd_on = live && tr->sd_stack != nullptr && e_np >= tr->sd_lo && e_np < tr->sd_hi;
        if (sd_on) {
            const int C = tr->sd_hi - tr->sd_lo, c = e_np - tr->sd_lo;
            sd_zv = tr->sd_z[(int64_t)e_b * C + c]; sd_wv = tr->sd_w[c];
            if (tr->sd_mask) sd_mv = tr->sd_mask[(int64_t)e_b * C + c];
        }
    }
    L2S_STAMP(2);
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < MAXC; ++j) {
        if (SPLIT > 1 && !LAY::STATIC && j > 0 && j % (MAXC / SPLIT) == 0) {     // next batch of loads: the previous batch's registers are free again
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j2 = j; j2 < j + MAXC / SPLIT; ++j2) load_chunk(j2);
            __builtin_amdgcn_sched_barrier(0);
        }
        const int c = wave + SK_WAVES * j;
        if (LAY::STATIC ? (SK_WAVES * j < LAY::NC) : (c < NC)) {
            if (j & 1) acc1 = mfma4(a[j], w[j], acc1);
            else acc0 = mfma4(a[j], w[j], acc0);
        }
        if (TIMED && j == 0) { asm volatile("s_nop 0" :: "v"(acc0[0])); L2S_STAMP(3); }      // first operands have landed
    }
    if (TIMED) { asm volatile("s_nop 0" :: "v"(acc0[0]), "v"(acc1[0])); }
    L2S_STAMP(4);
    // D layout: col = lane&15, row = 4*(lane>>4) + r
    {
        const int col = lane & 15, rb = 4 * (lane >> 4);
#pragma unroll
        for (int r = 0; r < 4; ++r) red[(wave * 16 + rb + r) * 17 + col] = acc0[r] + acc1[r];
    }
    __syncthreads();
    L2S_STAMP(5);
    float* gt = red + SK_WAVES * 16 * 17;               // reduced tile [16][17]
    const int row = tid >> 4, col = tid & 15;          // valid for tid < 256
    const int b = mt * 16 + row;
    const int np = tile * 16 + col;                     // (permuted) weight row
    float v = 0.f;
    if (tid < 256) {
#pragma unroll
        for (int wv = 0; wv < SK_WAVES; ++wv) v += red[(wv * 16 + row) * 17 + col];
        v += pf_bias;
    }

    if (epi == SK_LSTM) {
        if (tid < 256) {
            const int u = col >> 2, gate = col & 3;
            const int unit = tile * 4 + u;
            (void)unit; (void)gate;
            v += pf_extra;
            gt[row * 17 + col] = v;
        }
        __syncthreads();
        L2S_STAMP(6);
        if (tid < 64) {
            const int r2 = tid >> 2, u2 = tid & 3;
            const int b2 = mt * 16 + r2, unit2 = tile * 4 + u2;
            if (b2 < nB) {
                const float gi = gt[r2 * 17 + 4 * u2 + 0], gf = gt[r2 * 17 + 4 * u2 + 1];
                const float gg = gt[r2 * 17 + 4 * u2 + 2], go = gt[r2 * 17 + 4 * u2 + 3];
                const int64_t ci = frag16_index(b2, unit2, H);
                const float cprev = pf_c;
                const float cn = sigmoidf_(gf) * cprev + sigmoidf_(gi) * tanhf(gg);
                const float hn = sigmoidf_(go) * tanhf(cn);
                p.c_out[ci] = cn;
                p.h_out[frag16_index(b2, p.h_out_off + unit2, p.h_out_K)] = hn;
                if constexpr (TRAIN) {
                    if (tr->gates) {
                        float* gs = tr->gates + (int64_t)b2 * tr->ld_gates + unit2;
                        gs[0] = sigmoidf_(gi); gs[H] = sigmoidf_(gf); gs[2 * H] = tanhf(gg); gs[3 * H] = sigmoidf_(go);
                    }
                    if (tr->c_new) tr->c_new[(int64_t)b2 * tr->ld_c + unit2] = cn;
                    if (tr->h_drop) tr->h_drop[frag16_index(b2, unit2, tr->h_drop_K)] = tr->h_mask ? hn * tr->h_mask[(int64_t)b2 * tr->ld_hmask + unit2] : hn;
                }
                if (p.h_seq) p.h_seq[(int64_t)b2 * p.ld_hseq + unit2] = hn;
                if (p.h_plain) p.h_plain[(int64_t)b2 * p.ld_hplain + unit2] = hn;
            }
        }
        if (TIMED) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
        L2S_STAMP(7);
        return;
    }
    if (tid >= 256 || b >= nB) return;
    if (epi == SK_MEL) {
        if (np < 80) {
            p.mel[(int64_t)b * p.ld_mel_b + np] = v;
            if (p.yfrag) p.yfrag[frag16_index(b, np, 80)] = v;
        } else if (np == 80) {
            p.stop[(int64_t)b * p.ld_stop_b] = v + p.stop_const[b];
        }
        return;
    }
    if (np >= N) return;
    if constexpr (TRAIN) { if (tr->zsave) tr->zsave[(int64_t)b * tr->ld_z + np] = v; }
    v = act_apply(v, act, p.actw, np);
    v += pf_extra;
    if constexpr (TRAIN) { if (tr->out_mask) v *= tr->out_mask[(int64_t)b * tr->ld_mask + np]; }
    if constexpr (TRAIN) { if (tr->out_plain) tr->out_plain[(int64_t)b * tr->ld_out + np] = v; }
    if constexpr (TRAIN) {
        if (lb_on) {        // LSTM cell backward of unit np (train_decoder.hip lstm_bwd_kernel, same expressions in the same order)
            const int LH = tr->lb_H;
            float dh = v;
            if (tr->lb_dha) dh = lb_a + v * lb_m;
            const float gi = lb_g[0], gf = lb_g[1], gg = lb_g[2], go = lb_g[3];
            const float tc = tanhf(lb_cn);
            const float dc = lb_dcv + dh * go * (1.f - tc * tc);
            const float vals[4] = {dc * gg * gi * (1.f - gi), dc * lb_cp * gf * (1.f - gf), dc * gi * (1.f - gg * gg), dh * tc * go * (1.f - go)};
            tr->lb_dc[(int64_t)b * (tr->lb_ld_dc ? tr->lb_ld_dc : LH) + np] = dc * gf;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                tr->lb_frag[frag16_index(b, k * LH + np, 4 * LH)] = vals[k];
                tr->lb_stack[(int64_t)b * 4 * LH + k * LH + np] = vals[k];
                if (tr->lb_stack2) tr->lb_stack2[(int64_t)b * tr->lb_ld_stack2 + k * LH + np] = vals[k];
            }
        }
    }
    if constexpr (TRAIN) {
        if (sd_on) {
            const int C = tr->sd_hi - tr->sd_lo, c = np - tr->sd_lo;
            const float d = tr->sd_mask ? v * sd_mv : v;
            tr->sd_stack[(int64_t)b * C + c] = d;
            if (tr->sd_frag_d) tr->sd_frag_d[frag16_index(b, c, C)] = d;
            const float dz = d * cosf(sd_zv) * sd_wv;
            tr->sd_frag_dz[frag16_index(b, c, C)] = dz;
            if (tr->sd_stack_dz) tr->sd_stack_dz[(int64_t)b * C + c] = dz;
        }
    }
    if (epi == SK_FRAG)
        p.out[frag16_index(b, np, p.ldo)] = v;
    else
        p.out[(int64_t)b * p.ldo + np] = v;
}

constexpr int SK_RED_FLOATS = (SK_WAVES + 1) * 16 * 17;

// ---------------------------------------------------------------------------------------------------------------------------------------
// Register-blocked form for many batch rows (grouped decode: G independent B=32 batches advance in ONE launch chain, M = 32*G rows).
// One block (8 waves) owns RT row tiles x CT column tiles of 16x16; the waves split K exactly as in skinny_block (wave w owns chunks
// w, w+8, ...) and every operand fragment a wave fetches feeds CT (activations) or RT (weights) MFMA groups, so a 2x2 block moves
// (32+32)*K*4 bytes for four tiles where four 1x1 blocks move 4*(16+16)*K*4 - half the traffic through the CU's vector-memory path, which
// is what bounds the 1x1 form - and at M = 128 the launch becomes MFMA-bound (805 MFLOP per LSTM0 launch = 5.1 us at the fp32
// matrix peak).  Operands arrive in batches of JB chunks, two batches in flight, the next one requested right after a batch's MFMAs.
// Per output element the arithmetic is the 1x1 kernel's, operation for operation (even chunks into one accumulator, odd chunks into a
// second one, x,y,z,w in order; acc0+acc1; waves 0..7 in order; then bias, pre-gates), so a row's result does not depend on how many
// rows share the launch: a grouped pass is bit-identical to the same batches run one by one.
template <int RT, int CT>
struct SkRc { static constexpr int NT = RT * CT, RED_FLOATS = (SK_WAVES + 1) * NT * 16 * 17; };

// DEPTH = operand batches in flight: a wave requests batch b + DEPTH right after the MFMAs of batch b, so the round trip of a batch may take
// DEPTH - 1 batch-compute times before the matrix pipe waits for it.  JB = 1 with DEPTH = 4 holds the same four chunks in registers as JB = 2
// with DEPTH = 2 but tolerates 1.5x the latency; a 4x2 block runs alone on its CU (8 waves, two per SIMD: 256 VGPRs each), so five chunks fit.
template <int RT, int CT, int MAXC, int JB, int DEPTH = 2>
__device__ __forceinline__ void skinny_block_rc(const SkinnyP& p, int tp, int mg, float* red, int ntiles, int mts) {
    static_assert(MAXC % JB == 0, "chunk batches");
    constexpr int NT = RT * CT, NBATCH = MAXC / JB, NQ = (NT + 1) / 2, PRE = NBATCH < DEPTH ? NBATCH : DEPTH;
    const float* const W = p.W;
    const float* const sa0 = p.seg[0].a; const float* const sa1 = p.seg[1].a; const float* const sa2 = p.seg[2].a; const float* const sa3 = p.seg[3].a;
    const int n0 = p.seg[0].nchunks, n1 = p.seg[1].nchunks, n2 = p.seg[2].nchunks, n3 = p.seg[3].nchunks;
    const int K = p.K;
    L2S_PIN_S("s"(W), "s"(sa0), "s"(sa1), "s"(sa2), "s"(sa3), "s"(n0), "s"(n1), "s"(n2), "s"(n3), "s"(K), "s"(ntiles), "s"(mts));
    if (tp * CT >= ntiles) return;                   // block-uniform: grid x is sized for the widest group of the launch
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int NC = K >> 4;
    const int e0 = n0, e1 = e0 + n1, e2 = e1 + n2;
    // tiles past the end of the group / batch are computed on the last valid tile's operands and dropped in the epilogue
    int ct_[CT], rt_[RT];
#pragma unroll
    for (int i = 0; i < CT; ++i) ct_[i] = min(tp * CT + i, ntiles - 1);
#pragma unroll
    for (int r = 0; r < RT; ++r) rt_[r] = min(mg * RT + r, mts - 1);
    const float4* wb[CT];
#pragma unroll
    for (int i = 0; i < CT; ++i) wb[i] = reinterpret_cast<const float4*>(W) + (int64_t)ct_[i] * NC * 64 + lane;

    float4 a[MAXC][RT], w[MAXC][CT];
    auto load_chunk = [&](int j) {
        const int c = wave + SK_WAVES * j;          // wave-uniform (starting the K walk at a different chunk per block - so that the blocks of an
                                                    // XCD do not ask its L2 for the same lines at the same time - measured no gain: 10.7 vs 10.9 us at
                                                    // 128 rows, 18.7 vs 17.8 at 256)
        if (c < NC) {
            const float* ab = sa0; int lc = c, nn = n0;
            if (c >= e2) { ab = sa3; lc = c - e2; nn = n3; }
            else if (c >= e1) { ab = sa2; lc = c - e1; nn = n2; }
            else if (c >= e0) { ab = sa1; lc = c - e0; nn = n1; }
#pragma unroll
            for (int r = 0; r < RT; ++r) a[j][r] = reinterpret_cast<const float4*>(ab)[((int64_t)rt_[r] * nn + lc) * 64 + lane];
#pragma unroll
            for (int i = 0; i < CT; ++i) w[j][i] = wb[i][(int64_t)c * 64];
        }
    };
#pragma unroll
    for (int j = 0; j < PRE * JB; ++j) load_chunk(j);
    // ---- everything else the block needs: one batch of scalar loads under the operand loads already in flight
    const int epi = p.epi, nB = p.B, N = p.N, H = p.H, act = p.act;
    const float* const bias = p.bias; const float* const pre = p.pre; const int64_t ld_pre = p.ld_pre;
    const float* const c_in = p.c_in; const float* const add = p.add; const int ld_add = p.ld_add; const float* const addrow = p.addrow;
    L2S_PIN_S("s"(epi), "s"(nB), "s"(N), "s"(H), "s"(act), "s"(bias), "s"(pre), "s"(ld_pre), "s"(c_in), "s"(add), "s"(ld_add), "s"(addrow));
    // epilogue roles: thread (half, l) finishes element (l>>4, l&15) of tiles q = half, half+2, ...; thread (q2, t64) runs the LSTM cell of
    // (row t64>>2, unit t64&3) of tile q2
    const int half = tid >> 8, l256 = tid & 255, e_row = l256 >> 4, e_col = l256 & 15;
    float pf_bias[NQ], pf_extra[NQ];
#pragma unroll
    for (int n = 0; n < NQ; ++n) {
        const int q = half + 2 * n;
        pf_bias[n] = 0.f; pf_extra[n] = 0.f;
        if (q < NT) {
            const int t = tp * CT + q % CT, rt = mg * RT + q / CT;
            const int e_b = rt * 16 + e_row, e_np = t * 16 + e_col;
            if (t < ntiles && rt < mts) {
                if (bias) pf_bias[n] = bias[e_np];
                if (epi == SK_LSTM) {
                    if (pre && e_b < nB) pf_extra[n] = pre[(int64_t)e_b * ld_pre + (e_col & 3) * H + t * 4 + (e_col >> 2)];
                } else if (epi != SK_MEL && e_b < nB && e_np < N) {
                    if (add) pf_extra[n] = add[(int64_t)e_b * ld_add + e_np];
                    if (addrow) pf_extra[n] += addrow[e_np];
                }
            }
        }
    }
    // LSTM cells: thread (q2 = tid>>6 (+8 per round), t64) owns (row t64>>2, unit t64&3) of tile q2; NT <= 8 tiles need one round, 16 two
    constexpr int NCR = (NT + 7) / 8;
    const int t64 = tid & 63;
    float pf_c[NCR];
    bool cell_on[NCR];
#pragma unroll
    for (int cr = 0; cr < NCR; ++cr) {
        const int q2 = (tid >> 6) + 8 * cr;
        const int t2 = tp * CT + q2 % CT, rt2 = mg * RT + q2 / CT;
        const int b2 = rt2 * 16 + (t64 >> 2), unit2 = t2 * 4 + (t64 & 3);
        cell_on[cr] = epi == SK_LSTM && q2 < NT && t2 < ntiles && rt2 < mts && b2 < nB;
        pf_c[cr] = cell_on[cr] ? c_in[frag16_index(b2, unit2, H)] : 0.f;
    }

    f32x4 acc[NT][2];
#pragma unroll
    for (int q = 0; q < NT; ++q) { acc[q][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[q][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
    for (int bt = 0; bt < NBATCH; ++bt) {
#pragma unroll
        for (int j = bt * JB; j < (bt + 1) * JB; ++j) {
            if (wave + SK_WAVES * j < NC) {
#pragma unroll
                for (int r = 0; r < RT; ++r)
#pragma unroll
                    for (int i = 0; i < CT; ++i) acc[r * CT + i][j & 1] = mfma4(a[j][r], w[j][i], acc[r * CT + i][j & 1]);
            }
        }
        if (bt + PRE < NBATCH) {                  // the registers of this batch are free again: request the batch DEPTH ahead
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = (bt + PRE) * JB; j < (bt + PRE + 1) * JB; ++j) load_chunk(j);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // D layout: col = lane&15, row = 4*(lane>>4) + r
    {
        const int col = lane & 15, rb = 4 * (lane >> 4);
#pragma unroll
        for (int q = 0; q < NT; ++q)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[((wave * NT + q) * 16 + rb + r) * 17 + col] = acc[q][0][r] + acc[q][1][r];
    }
    __syncthreads();
    float* gt = red + SK_WAVES * NT * 16 * 17;          // reduced tiles [NT][16][17]
#pragma unroll
    for (int n = 0; n < NQ; ++n) {
        const int q = half + 2 * n;
        if (q >= NT) continue;
        const int t = tp * CT + q % CT, rt = mg * RT + q / CT;
        if (t >= ntiles || rt >= mts) continue;
        float v = 0.f;
#pragma unroll
        for (int wv = 0; wv < SK_WAVES; ++wv) v += red[((wv * NT + q) * 16 + e_row) * 17 + e_col];
        v += pf_bias[n];
        const int b = rt * 16 + e_row, np = t * 16 + e_col;
        if (epi == SK_LSTM) {
            v += pf_extra[n];
            gt[(q * 16 + e_row) * 17 + e_col] = v;
            continue;
        }
        if (b >= nB) continue;
        if (epi == SK_MEL) {
            if (np < 80) {
                p.mel[(int64_t)b * p.ld_mel_b + np] = v;
                if (p.yfrag) p.yfrag[frag16_index(b, np, 80)] = v;
            } else if (np == 80) {
                p.stop[(int64_t)b * p.ld_stop_b] = v + p.stop_const[b];
            }
            continue;
        }
        if (np >= N) continue;
        v = act_apply(v, act, p.actw, np);
        v += pf_extra[n];
        if (epi == SK_FRAG) p.out[frag16_index(b, np, p.ldo)] = v;
        else p.out[(int64_t)b * p.ldo + np] = v;
    }
    if (epi != SK_LSTM) return;
    __syncthreads();
#pragma unroll
    for (int cr = 0; cr < NCR; ++cr) {
        if (!cell_on[cr]) continue;
        const int q2 = (tid >> 6) + 8 * cr;
        const int t2 = tp * CT + q2 % CT, rt2 = mg * RT + q2 / CT;
        const int b2 = rt2 * 16 + (t64 >> 2), unit2 = t2 * 4 + (t64 & 3);
        const int r2 = t64 >> 2, u2 = t64 & 3;
        const float* g4 = gt + (q2 * 16 + r2) * 17 + 4 * u2;
        const float gi = g4[0], gf = g4[1], gg = g4[2], go = g4[3];
        const float cn = sigmoidf_(gf) * pf_c[cr] + sigmoidf_(gi) * tanhf(gg);
        const float hn = sigmoidf_(go) * tanhf(cn);
        p.c_out[frag16_index(b2, unit2, H)] = cn;
        p.h_out[frag16_index(b2, p.h_out_off + unit2, p.h_out_K)] = hn;
        if (p.h_seq) p.h_seq[(int64_t)b2 * p.ld_hseq + unit2] = hn;
        if (p.h_plain) p.h_plain[(int64_t)b2 * p.ld_hplain + unit2] = hn;
    }
}

// The same block with NOTHING conditional between its operand loads and its MFMAs: the K-segment layout is a template constant (LAY, every
// boundary a multiple of the 8 waves, K a multiple of 128 so that every wave owns exactly LAY::NC / 8 chunks) and the epilogue's own operands
// are requested BEFORE the first operand batch.  Why it matters (ISA of the general form, round 3): its per-chunk `if (c < NC)` guards and
// segment-select branches sit between the loads and the MFMAs that consume them, the compiler cannot count the younger loads across those
// joins and waits `s_waitcnt vmcnt(0)` before EVERY chunk's MFMAs - the just-issued prefetch included - so the block ran load -> wait ->
// compute with nothing in flight: 36.5 GB/s per CU where a plain streaming kernel with the same operand reuse gets > 70 (tools/membw).  Here the
// waits are exact (`vmcnt(6 * (DEPTH - 1))`), DEPTH - 1 chunks stay in flight under every chunk's MFMAs.  Arithmetic per output element is
// unchanged: bit-identical.
// IS_LSTM: the launch's groups are all SK_LSTM (true) or none is (false) - a compile-time fact of the caller, so that the epilogue's operand
// requests are straight-line too (with the kind decided at run time the compiler guards each of them with its own `s_waitcnt vmcnt(0)`).
// NW = real waves of the block: 8, or 4 (one per SIMD: no partner on the matrix pipe, 512 VGPRs per lane - room for many chunks in flight); with 4
// every wave plays TWO of the eight K slices (w and w + 4) with their own accumulators, so the partial sums that reach the reduction - and
// every output bit - are the eight-wave block's.
// X3 (LSTM launches of the decode step, option "lstm_x3"): the products run on the BF16 matrix cores by the exact three-way split of the dense kernels
// (x = hi + mid + lo, six partial products per pair).  A K step is a PAIR of a slice's chunks (32 k): the activations' two fp32 quads are split by the
// wave that loaded them (44 VALU per row tile), the weights arrive pre-split (SkinnyP::W3: [tile][slice][pair][plane][lane] 16 bytes, made from the packed
// fp32 fragments by skx_planes_kernel) - six v_mfma_f32_16x16x32_bf16 (96 clk) where the f32 form needs eight v_mfma_f32_16x16x4_f32 (256 clk).  The
// K-slice structure, the even / odd accumulators (now per pair), the reduction order and the epilogue are unchanged; per output the arithmetic is the same
// whatever the block shape or grouping, and differs from the f32 form by rounding-level amounts like every split-bf16 kernel of this path.
typedef __bf16 skx_bf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ void skx_split4(const float4& v, uint2& hi, uint2& mid, uint2& lo) {
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
__device__ __forceinline__ void skx_split8(const float4& a, const float4& b, skx_bf16x8& hi, skx_bf16x8& mid, skx_bf16x8& lo) {
    uint2 h0, m0, l0, h1, m1, l1;
    skx_split4(a, h0, m0, l0); skx_split4(b, h1, m1, l1);
    hi = __builtin_bit_cast(skx_bf16x8, make_uint4(h0.x, h0.y, h1.x, h1.y));
    mid = __builtin_bit_cast(skx_bf16x8, make_uint4(m0.x, m0.y, m1.x, m1.y));
    lo = __builtin_bit_cast(skx_bf16x8, make_uint4(l0.x, l0.y, l1.x, l1.y));
}

// Where element (row, col) of partial tile `slot` sits in the reduction buffer of skinny_block_rcs: 256 floats per tile, no padding, row r at
// 16 * (r ^ bit 2 of r) with its columns XORed by bits 1-2 of r.  Both sides go through LDS as 32-bit accesses (32 banks, 32-lane groups): the
// MFMA D layout writes rows R and R + 4 from one group (they land in opposite 16-bank halves), the cell threads read rows 0-7 x the four columns
// {4u + g} (four rows per half, their low column bits made distinct by the XOR) - neither side shares a bank.  The [16][17] layout cost every
// write and every gate read a second LDS cycle (SQ_LDS_BANK_CONFLICT = half of SQ_LDS_IDX_ACTIVE in the LSTM launches).
__device__ __forceinline__ int sk_red_idx(int slot, int row, int col) { return slot * 256 + ((row ^ ((row >> 2) & 1)) << 4) + (col ^ ((row >> 1) & 3)); }
// SEQ (four waves): a wave plays its two K slices ONE AFTER THE OTHER instead of chunk by chunk in turn - the first slice's partial tiles go to the
// reduction buffer as soon as its chunks are through and the accumulators start again from zero for the second.  Per slice the chunks, their order
// and the two alternating accumulators are the interleaved form's, so every partial sum that reaches the reduction is the same bits; what changes
// is the register budget: one set of accumulators (64 at 4x2) instead of two, which puts the split-bf16 4x2 block under 256 registers - half a
// compute unit (skinny_rc4h_kernel).
template <int RT, int CT, class LAY, int DEPTH, bool IS_LSTM, bool TIMED = false, int NW = SK_WAVES, bool X3 = false, bool TRAIN = false, bool SEQ = false>
__device__ __forceinline__ void skinny_block_rcs(const SkinnyP& p, int tp, int mg, float* red, int ntiles, int mts, unsigned long long* ts = nullptr,
                                                 const SkinnyTrain* tr = nullptr) {
    static_assert(!TRAIN || IS_LSTM, "the training stores of this form: the LSTM cell's tape (gates, new cell, dropped-out hidden state)");
    static_assert(NW == 8 || NW == 4, "eight waves, or four that each play two");
    static_assert(!X3 || (IS_LSTM && (LAY::NC / SK_WAVES) % 2 == 0), "the split-bf16 form: LSTM blocks, whole chunk pairs per slice");
    static_assert(!SEQ || NW == 4, "slices one after the other: the four-wave form");
    constexpr int VW = SK_WAVES / NW, NH = NW / 4;       // K slices per real wave; 256-thread epilogue teams
    constexpr int VA = SEQ ? 1 : VW;                     // accumulator sets per wave
    L2S_STAMP(0);
    static_assert(LAY::STATIC && LAY::NC % SK_WAVES == 0, "static layout, every wave the same number of chunks");
    constexpr int NT = RT * CT, MAXC = LAY::NC / SK_WAVES, TC = MAXC * VW, NQ = (NT + NH - 1) / NH, PRE = TC < DEPTH ? TC : DEPTH, NC = LAY::NC;
    constexpr int E0 = LAY::n0, E1 = E0 + LAY::n1, E2 = E1 + LAY::n2;
    const float* const W = p.W;
    const float* const sa0 = p.seg[0].a; const float* const sa1 = p.seg[1].a; const float* const sa2 = p.seg[2].a; const float* const sa3 = p.seg[3].a;
    L2S_PIN_S("s"(W), "s"(sa0), "s"(sa1), "s"(sa2), "s"(sa3), "s"(ntiles), "s"(mts));
    if (tp * CT >= ntiles) return;                   // block-uniform: grid x is sized for the widest group of the launch
    L2S_STAMP(1);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // ---- operand streams: BUFFER loads - one scalar descriptor per stream (weights, the four K segments, the second source of segment 1), one 32-bit
    // per-lane offset per (segment, row tile) and per column tile, and the chunk as a compile-time SCALAR offset.  The flat form (a 64-bit base
    // pointer per stream in VGPRs plus a 64-bit add per request whose chunk offset does not fit the 12-bit immediate) took the wave ~100 clk per
    // request to issue: 1.1 us for the first 24 of an LSTM block, on the launch's critical path (the attention block's stamps showed the same,
    // 2.1 -> 0.6 us).  (Tiles past the end of the group / batch are computed on the last valid tile's operands and dropped in the epilogue.)
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(W), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_a0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(sa0), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_a1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(sa1), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_a2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(sa2), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_a3 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(sa3), 0, 0x7fffffff, 0x00020000);
    int wo[CT];
#pragma unroll
    for (int i = 0; i < CT; ++i) wo[i] = ((min(tp * CT + i, ntiles - 1) * NC + wave) * 64 + lane) * 16;
    int ao[4][RT];
#pragma unroll
    for (int r = 0; r < RT; ++r) {
        const int rt = min(mg * RT + r, mts - 1);
        ao[0][r] = ((rt * LAY::n0 + wave) * 64 + lane) * 16;
        ao[1][r] = ((rt * LAY::n1 + wave) * 64 + lane) * 16;
        ao[2][r] = ((rt * LAY::n2 + wave) * 64 + lane) * 16;
        ao[3][r] = ((rt * LAY::n3 + wave) * 64 + lane) * 16;
    }
    // the slots of segment 1 when it has a second source (LAY::SUM1): its fragments travel beside the first source's and are added right before the MFMAs
    constexpr int NS1 = LAY::SUM1 ? (LAY::n1 / SK_WAVES) * VW : 1;          // slots whose chunk lies in segment 1 (first slot: FS1)
    constexpr int FS1 = (E0 / SK_WAVES) * VW;
    const __amdgpu_buffer_rsrc_t rs_s1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(LAY::SUM1 ? p.a_sum : W), 0, 0x7fffffff, 0x00020000);
    constexpr int NPI = MAXC / 2;                                           // chunk pairs per K slice (X3)
    const __amdgpu_buffer_rsrc_t rs_w3 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(X3 ? p.W3 : (const void*)W), 0, 0x7fffffff, 0x00020000);
    int wo3[CT];
#pragma unroll
    for (int i = 0; i < CT; ++i) wo3[i] = (((min(tp * CT + i, ntiles - 1) * SK_WAVES + wave) * NPI * 3) * 64 + lane) * 16;
    float4 a[TC][RT], w[X3 ? 1 : TC][CT], a2[NS1][RT];
    uint4 w3[X3 ? TC / 2 : 1][CT][3];
    auto load_chunk = [&](auto jc) {          // slot t: K slice h = t % VW (wave + NW * h), its j-th chunk (j = t / VW)
        constexpr int t_ = decltype(jc)::value, j = t_;
        constexpr int jj_ = SEQ ? t_ % MAXC : t_ / VW, hs_ = SEQ ? t_ / MAXC : t_ % VW;      // chunk jj_ of K slice wave + NW * hs_
        constexpr int cj = SK_WAVES * jj_ + NW * hs_;
        constexpr int sg = cj >= E2 ? 3 : cj >= E1 ? 2 : cj >= E0 ? 1 : 0;
        constexpr int off = cj - (sg == 3 ? E2 : sg == 2 ? E1 : sg == 1 ? E0 : 0);
        const __amdgpu_buffer_rsrc_t rs = sg == 3 ? rs_a3 : sg == 2 ? rs_a2 : sg == 1 ? rs_a1 : rs_a0;
#pragma unroll
        for (int r = 0; r < RT; ++r) a[j][r] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, ao[sg][r], off * 1024, 0));
        if constexpr (LAY::SUM1 && sg == 1) {
#pragma unroll
            for (int r = 0; r < RT; ++r) a2[SEQ ? hs_ * (LAY::n1 / SK_WAVES) + jj_ - E0 / SK_WAVES : j - FS1][r] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs_s1, ao[1][r], off * 1024, 0));
        }
        if constexpr (X3) {
            if constexpr (jj_ % 2 == 0) {       // the first chunk of a pair brings the pair's three weight planes
                constexpr int ip = jj_ / 2, hs = hs_;
#pragma unroll
                for (int i = 0; i < CT; ++i)
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl)
                        w3[SEQ ? hs * NPI + ip : ip * VW + hs][i][pl] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rs_w3, wo3[i], ((hs * NW * NPI + ip) * 3 + pl) * 1024, 0));
            }
        } else {
#pragma unroll
            for (int i = 0; i < CT; ++i) w[j][i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs_w, wo[i], cj * 1024, 0));
        }
    };
    __builtin_amdgcn_sched_barrier(0);
    static_for<0, PRE>([&](auto jc) { load_chunk(jc); });
    __builtin_amdgcn_sched_barrier(0);               // every one of the first DEPTH chunks is requested before anything else

    // ---- the epilogue's operands, under the operand loads already in flight.  Every request is ONE load from an always-valid address (a null
    // table is replaced by W, indices are clamped into the table) whose value is kept or dropped by a select afterwards: no load sits inside a
    // branch, so the waits of the K loop below count the younger loads exactly
    const int epi = IS_LSTM ? (int)SK_LSTM : p.epi, nB = p.B, N = p.N, H = p.H, act = p.act;
    const float* const bias = p.bias; const float* const pre = p.pre; const int64_t ld_pre = p.ld_pre;
    const float* const c_in = p.c_in; const float* const add = p.add; const int ld_add = p.ld_add; const float* const addrow = p.addrow;
    L2S_PIN_S("s"(epi), "s"(nB), "s"(N), "s"(H), "s"(act), "s"(bias), "s"(pre), "s"(ld_pre), "s"(c_in), "s"(add), "s"(ld_add), "s"(addrow));
    const int half = tid >> 8, l256 = tid & 255, e_row = l256 >> 4, e_col = l256 & 15;
    constexpr int NQE = IS_LSTM ? 1 : NQ;            // LSTM blocks finish in their cell threads (below): no per-element epilogue operands
    float pf_bias[NQE], pf_add[NQE], pf_row[NQE];
    const float* const bias_p = bias ? bias : W;
    const float* const add_p = IS_LSTM ? (pre ? pre : W) : (add ? add : W);
    const float* const row_p = addrow ? addrow : W;
    if constexpr (!IS_LSTM) {
#pragma unroll
        for (int n = 0; n < NQ; ++n) {
            const int q = min(half + NH * n, NT - 1);
            const int t = min(tp * CT + q % CT, ntiles - 1), rt = min(mg * RT + q / CT, mts - 1);
            const int e_b = min(rt * 16 + e_row, nB - 1), e_np = t * 16 + e_col;
            const float vb = bias_p[bias ? e_np : 0];
            pf_bias[n] = bias ? vb : 0.f;
            const int e_nc = min(e_np, N - 1);       // SK_MEL groups read (and drop) element N - 1: they have neither table
            const float va = add_p[add ? (int64_t)e_b * ld_add + e_nc : 0];
            const float vr = row_p[addrow ? e_nc : 0];
            pf_add[n] = add ? va : 0.f;
            pf_row[n] = addrow ? vr : 0.f;
        }
    }
    // LSTM cells: thread (q2 = tid>>6 (+NW per round), t64) owns (row t64>>2, unit t64&3) of tile q2 and finishes it alone: the eight K-slice
    // partials of its four gate columns are summed in slice order, bias and pre-gates added, in the order of the element epilogue - same bits,
    // one LDS round and one barrier less than staging the reduced tile first
    constexpr int NCR = (NT + NW - 1) / NW;
    const int t64 = tid & 63;
    float pf_c[NCR], pf_gb[IS_LSTM ? NCR : 1][4], pf_gp[IS_LSTM ? NCR : 1][4];
    bool cell_on[NCR];
#pragma unroll
    for (int cr = 0; cr < NCR; ++cr) {
        const int q2 = (tid >> 6) + NW * cr;
        const int t2 = tp * CT + q2 % CT, rt2 = mg * RT + q2 / CT;
        const int b2 = rt2 * 16 + (t64 >> 2);
        cell_on[cr] = IS_LSTM && q2 < NT && t2 < ntiles && rt2 < mts && b2 < nB;
        if constexpr (IS_LSTM) {
            const int q2c = min(q2, NT - 1);
            const int t2c = min(tp * CT + q2c % CT, ntiles - 1), rt2c = min(mg * RT + q2c / CT, mts - 1);
            const int b2c = min(rt2c * 16 + (t64 >> 2), nB - 1), u2 = t64 & 3;
            pf_c[cr] = c_in[frag16_index(b2c, t2c * 4 + u2, H)];
            const float4 b4 = *reinterpret_cast<const float4*>(bias_p + (bias ? t2c * 16 + 4 * u2 : 0));      // the unit's four gate rows are adjacent (permuted weight rows)
            pf_gb[cr][0] = bias ? b4.x : 0.f; pf_gb[cr][1] = bias ? b4.y : 0.f; pf_gb[cr][2] = bias ? b4.z : 0.f; pf_gb[cr][3] = bias ? b4.w : 0.f;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float va = add_p[pre ? (int64_t)b2c * ld_pre + g * H + t2c * 4 + u2 : 0];
                pf_gp[cr][g] = pre ? va : 0.f;
            }
        } else {
            pf_c[cr] = 0.f;
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    L2S_STAMP(2);

    f32x4 acc[NT][2][VA];
#pragma unroll
    for (int q = 0; q < NT; ++q)
#pragma unroll
        for (int h = 0; h < VA; ++h) { acc[q][0][h] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[q][1][h] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    static_for<0, TC>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        constexpr int jj = SEQ ? j % MAXC : j / VW, hs = SEQ ? j / MAXC : j % VW;      // chunk jj of the wave's K slice hs
        constexpr int par = jj & 1, hh = SEQ ? 0 : hs, jp = SEQ ? 1 : VW;             // jp: slots back to the first chunk of a pair
        constexpr int cjk = SK_WAVES * jj + NW * hs;
        // element-major issue order: the four dependent MFMAs of a tile (x, y, z, w into one accumulator) are NT instructions apart, so none waits
        // for its predecessor's result; per accumulator the order of the additions is mfma4's
        {
            if constexpr (LAY::SUM1 && cjk >= E0 && cjk < E1) {
                constexpr int s1 = SEQ ? hs * (LAY::n1 / SK_WAVES) + jj - E0 / SK_WAVES : j - FS1;
#pragma unroll
                for (int r = 0; r < RT; ++r) {
                    a[j][r].x += a2[s1][r].x; a[j][r].y += a2[s1][r].y; a[j][r].z += a2[s1][r].z; a[j][r].w += a2[s1][r].w;
                }
            }
            if constexpr (X3) {
                if constexpr (jj % 2 == 1) {   // the second chunk of a pair: both quads are here (and summed): split, six partial products per tile pair
                    constexpr int ip = jj / 2, px = ip & 1, ws = SEQ ? hs * NPI + ip : ip * VW + hs;
                    skx_bf16x8 ah[RT], am[RT], al[RT];
#pragma unroll
                    for (int r = 0; r < RT; ++r) skx_split8(a[j - jp][r], a[j][r], ah[r], am[r], al[r]);
                    // smallest partial products first; tile-major inside a term, so that an accumulator's next MFMA is NT instructions away
#define L2S_SKX_TERM(A_, PL_)                                                                                                              \
                    _Pragma("unroll") for (int r = 0; r < RT; ++r)                                                                         \
                        _Pragma("unroll") for (int i = 0; i < CT; ++i)                                                                     \
                            acc[r * CT + i][px][hh] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A_[r], __builtin_bit_cast(skx_bf16x8, w3[ws][i][PL_]), acc[r * CT + i][px][hh], 0, 0, 0);
                    L2S_SKX_TERM(al, 0) L2S_SKX_TERM(ah, 2) L2S_SKX_TERM(am, 1) L2S_SKX_TERM(am, 0) L2S_SKX_TERM(ah, 1) L2S_SKX_TERM(ah, 0)
#undef L2S_SKX_TERM
                }
            } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int r = 0; r < RT; ++r)
#pragma unroll
                    for (int i = 0; i < CT; ++i) {
                        const float av = e == 0 ? a[j][r].x : e == 1 ? a[j][r].y : e == 2 ? a[j][r].z : a[j][r].w;
                        const float wv = e == 0 ? w[j][i].x : e == 1 ? w[j][i].y : e == 2 ? w[j][i].z : w[j][i].w;
                        acc[r * CT + i][par][hh] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, wv, acc[r * CT + i][par][hh], 0, 0, 0);
                    }
            }
        }
        if constexpr (j + PRE < TC) {                // the registers of this chunk are free again: request the chunk DEPTH ahead
            __builtin_amdgcn_sched_barrier(0);
            load_chunk(std::integral_constant<int, j + PRE>{});
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (SEQ && jj == MAXC - 1 && hs + 1 < VW) {      // this slice is through: its partial tiles to the reduction buffer, accumulators from zero again
            const int col = lane & 15, rb = 4 * (lane >> 4);
#pragma unroll
            for (int q = 0; q < NT; ++q) {
#pragma unroll
                for (int r = 0; r < 4; ++r) red[sk_red_idx((wave + NW * hs) * NT + q, rb + r, col)] = acc[q][0][0][r] + acc[q][1][0][r];
                acc[q][0][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[q][1][0] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
        if constexpr (TIMED && j == 0) { asm volatile("s_nop 0" :: "v"(acc[0][0][0][0])); L2S_STAMP(3); }      // first operands landed, first chunk computed
    });
    if constexpr (TIMED) { asm volatile("s_nop 0" :: "v"(acc[0][0][0][0]), "v"(acc[NT - 1][1][VA - 1][0])); }
    L2S_STAMP(4);
    // D layout: col = lane&15, row = 4*(lane>>4) + r
    {
        const int col = lane & 15, rb = 4 * (lane >> 4);
#pragma unroll
        for (int h = 0; h < VA; ++h)
#pragma unroll
            for (int q = 0; q < NT; ++q)
#pragma unroll
                for (int r = 0; r < 4; ++r) red[sk_red_idx((wave + NW * (SEQ ? VW - 1 : h)) * NT + q, rb + r, col)] = acc[q][0][h][r] + acc[q][1][h][r];
    }
    __syncthreads();
    L2S_STAMP(5);
    if constexpr (!IS_LSTM) {
#pragma unroll
        for (int n = 0; n < NQ; ++n) {
            const int q = half + NH * n;
            if (q >= NT) continue;
            const int t = tp * CT + q % CT, rt = mg * RT + q / CT;
            if (t >= ntiles || rt >= mts) continue;
            float v = 0.f;
#pragma unroll
            for (int wv = 0; wv < SK_WAVES; ++wv) v += red[sk_red_idx(wv * NT + q, e_row, e_col)];
            v += pf_bias[n];
            const int b = rt * 16 + e_row, np = t * 16 + e_col;
            if (b >= nB) continue;
            if (epi == SK_MEL) {
                if (np < 80) {
                    p.mel[(int64_t)b * p.ld_mel_b + np] = v;
                    if (p.yfrag) p.yfrag[frag16_index(b, np, 80)] = v;
                } else if (np == 80) {
                    p.stop[(int64_t)b * p.ld_stop_b] = v + p.stop_const[b];
                }
                continue;
            }
            if (np >= N) continue;
            v = act_apply(v, act, p.actw, np);
            v += pf_add[n] + pf_row[n];                  // the general form adds (add + addrow) as one pre-summed value: same order
            if (epi == SK_FRAG) p.out[frag16_index(b, np, p.ldo)] = v;
            else p.out[(int64_t)b * p.ldo + np] = v;
        }
        return;
    }
    L2S_STAMP(6);
#pragma unroll
    for (int cr = 0; cr < NCR; ++cr) {
        if (!cell_on[cr]) continue;
        const int q2 = (tid >> 6) + NW * cr;
        const int t2 = tp * CT + q2 % CT, rt2 = mg * RT + q2 / CT;
        const int b2 = rt2 * 16 + (t64 >> 2), unit2 = t2 * 4 + (t64 & 3);
        const int r2 = t64 >> 2, u2 = t64 & 3;
        float g4[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float v = 0.f;
#pragma unroll
            for (int wv = 0; wv < SK_WAVES; ++wv) v += red[sk_red_idx(wv * NT + q2, r2, 4 * u2 + g)];
            v += pf_gb[cr][g];
            v += pf_gp[cr][g];
            g4[g] = v;
        }
        const float gi = g4[0], gf = g4[1], gg = g4[2], go = g4[3];
        const float cn = sigmoidf_(gf) * pf_c[cr] + sigmoidf_(gi) * tanhf(gg);
        const float hn = sigmoidf_(go) * tanhf(cn);
        p.c_out[frag16_index(b2, unit2, H)] = cn;
        p.h_out[frag16_index(b2, p.h_out_off + unit2, p.h_out_K)] = hn;
        if (p.h_seq) p.h_seq[(int64_t)b2 * p.ld_hseq + unit2] = hn;
        if (p.h_plain) p.h_plain[(int64_t)b2 * p.ld_hplain + unit2] = hn;
        if constexpr (TRAIN) {      // the tape of the cell, the expressions of skinny_block<TRAIN>: same bits
            if (tr->gates) {
                float* gs = tr->gates + (int64_t)b2 * tr->ld_gates + unit2;
                gs[0] = sigmoidf_(gi); gs[H] = sigmoidf_(gf); gs[2 * H] = tanhf(gg); gs[3 * H] = sigmoidf_(go);
            }
            if (tr->c_new) tr->c_new[(int64_t)b2 * tr->ld_c + unit2] = cn;
            if (tr->h_drop) tr->h_drop[frag16_index(b2, unit2, tr->h_drop_K)] = tr->h_mask ? hn * tr->h_mask[(int64_t)b2 * tr->ld_hmask + unit2] : hn;
        }
    }
    if constexpr (TIMED) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
    L2S_STAMP(7);
}

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
constexpr int ATT_SM_FLOATS = 512 + ATT_MAXT + 16 + 16;
constexpr int ATT_VLDS_FLOATS = 32 * 256;            // a clip's projected values V' (<= 32 frames x 256) staged through LDS (attention_block<.., VLDS>)

__device__ __forceinline__ float block_max8(float x, float* scratch) {
    x = wave_max_f(x);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = x;
    __syncthreads();
    float m = scratch[0];
#pragma unroll
    for (int i = 1; i < 8; ++i) m = fmaxf(m, scratch[i]);
    return m;
}
__device__ __forceinline__ float block_sum8(float x, float* scratch) {
    x = wave_sum_f(x);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = x;
    __syncthreads();
    return ((scratch[0] + scratch[1]) + (scratch[2] + scratch[3])) + ((scratch[4] + scratch[5]) + (scratch[6] + scratch[7]));
}

// 512 threads (8 waves) per batch row.  Every global operand of the block (q, this wave's k rows, this thread's v
// column) has an address known at launch, so all loads are issued before the first dependent instruction: one memory
// round trip instead of five serialized ones.
#define L2S_ATT_STAMP(k) do { if constexpr (TIMED) { if (threadIdx.x == 0) ats[blockIdx.x * 8 + (k)] = wall_clock64(); } } while (0)
// VLDS (the inference step kernel; needs ATT_SM_FLOATS + ATT_VLDS_FLOATS of shared memory): with the projected values (256 columns) and T <= 32 the
// block's threads fetch V' as 16-byte rows - four requests per thread instead of 29 one-column dword requests for half of them - and pass it through
// LDS; a@V' then reads its column from there, t ascending as before (same bits).  The stamped build showed the 38 requests of a thread taking 2.1 us
// to ISSUE (~60 clk each), 40 % of the block's lifetime.
// SKIP0 (with VLDS): tau MULTIPLIES the logits (decoder.py:414), so they spread over thousands and the soft-max of all but one to three frames
// underflows to EXACTLY 0.0f.  A frame of weight 0 contributes fmaf(0, v, acc) = acc (V' is finite; acc is never -0): skipping it is the same bits.
// The block then fetches the projected values of the non-zero frames only, AFTER the soft-max (one more round trip in the block's life, a quarter
// fewer bytes through the fabric per launch: 7.6 of 29.7 MB at 256 rows) - for launches that run beside other chains' kernels, where the launch is
// bound by the fabric, not by its own latency chain.
template <bool TRAIN = false, bool TIMED = false, bool VLDS = false, bool SKIP0 = false>
__device__ __forceinline__ void attention_block(const AttnP& p, int b, float* sm, const AttnTrain* tr = nullptr, unsigned long long* ats = nullptr) {
    static_assert(!SKIP0 || VLDS, "the zero-weight skip rides on the buffer-load form of the inference step");
    L2S_ATT_STAMP(0);
    float* qs = sm;                  // 512
    float* sc = sm + 512;            // ATT_MAXT
    float* scratch = sc + ATT_MAXT;  // 16
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int T = p.T;
    // values: v (512 columns) or, when the caller hoisted attention_proj into the prologue, V' = V W_ap^T + b_ap (256 columns: waves 4-7 have none)
    const int vcols = p.vp ? 256 : 512;
    const float* const pq = p.q; const float* const pk = p.k; const float* const pv = p.vp ? p.vp : p.v; const float* const ptau = p.tau;
    float* const pav = p.av_frag; float* const pattn = p.attn_out;
    const int ldq = p.ldq, logits = p.attn_logits; const int64_t ld_attn = p.ld_attn_b;
    L2S_PIN_S("s"(T), "s"(pq), "s"(pk), "s"(pv), "s"(ptau), "s"(pav), "s"(pattn), "s"(ldq), "s"(logits), "s"(ld_attn), "s"(vcols));
    const float* kb = pk + (int64_t)b * T * 512 + lane * 8;
    const bool vlane = tid < vcols;                  // wave-uniform
    const float* vb = pv + (int64_t)b * T * vcols + (vlane ? tid : 0);
    // ---- loads
    const float qv = pq[(int64_t)b * ldq + tid];
    const float tau = ptau[0];
    float4 k0[4], k1[4];
    if constexpr (VLDS) {
        // buffer loads: one scalar descriptor per clip, the frame as a scalar byte offset, one per-lane offset for all eight requests - no 64-bit address
        // per request (the flat form kept ~20 registers of addresses alive and the block at 102: two blocks per CU, the launch's 768 blocks in 1.5 rounds);
        // frames past T read zero through the descriptor's byte count
        const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(pk + (int64_t)b * T * 512), 0, T * 2048, 0x00020000);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int so = __builtin_amdgcn_readfirstlane((wave + 8 * r) * 2048);
            k0[r] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rk, lane * 32, so, 0));
            k1[r] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rk, lane * 32, so + 16, 0));
        }
    } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int t = wave + 8 * r;
            if (t < T) {
                k0[r] = *reinterpret_cast<const float4*>(kb + (int64_t)t * 512);
                k1[r] = *reinterpret_cast<const float4*>(kb + (int64_t)t * 512 + 4);
            }
        }
    }
    constexpr bool vlds = VLDS && !SKIP0;                     // the launch picks the instance: VLDS only with projected values and T <= 32
    float* const vs = sm + ATT_SM_FLOATS;
    float vv[VLDS ? 1 : 32];
    float4 vq[vlds ? 4 : 1];
    if constexpr (SKIP0) {
        // nothing up front: the values of the frames that count are requested after the soft-max
    } else if constexpr (vlds) {
        const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(pv + (int64_t)b * T * 256), 0, T * 1024, 0x00020000);
#pragma unroll
        for (int i = 0; i < 4; ++i) vq[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rv, tid * 16, 8192 * i, 0));
    } else {
#pragma unroll
        for (int e = 0; e < 32; ++e) vv[e] = (vlane && e < T) ? vb[(int64_t)e * vcols] : 0.f;
    }
    L2S_ATT_STAMP(1);               // every request issued
    // ---- logits
    qs[tid] = qv * tau;
    __syncthreads();
    L2S_ATT_STAMP(2);               // q landed, visible to the block
    float qq[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) qq[e] = qs[lane * 8 + e];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int t = wave + 8 * r;
        if (t < T) {
            double d = (double)qq[0] * k0[r].x + (double)qq[1] * k0[r].y + (double)qq[2] * k0[r].z + (double)qq[3] * k0[r].w +
                       (double)qq[4] * k1[r].x + (double)qq[5] * k1[r].y + (double)qq[6] * k1[r].z + (double)qq[7] * k1[r].w;
            d = wave_sum_d(d);
            if (lane == 0) sc[t] = (float)d;
        }
    }
    for (int t = wave + 32; t < T; t += 8) {             // clips longer than 32 frames: remaining rows, one at a time
        const float4 a0 = *reinterpret_cast<const float4*>(kb + (int64_t)t * 512);
        const float4 a1 = *reinterpret_cast<const float4*>(kb + (int64_t)t * 512 + 4);
        double d = (double)qq[0] * a0.x + (double)qq[1] * a0.y + (double)qq[2] * a0.z + (double)qq[3] * a0.w +
                   (double)qq[4] * a1.x + (double)qq[5] * a1.y + (double)qq[6] * a1.z + (double)qq[7] * a1.w;
        d = wave_sum_d(d);
        if (lane == 0) sc[t] = (float)d;
    }
    if constexpr (vlds) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { const int idx = tid + 512 * i; if (idx < T * 64) reinterpret_cast<float4*>(vs)[idx] = vq[i]; }
    }
    L2S_ATT_STAMP(3);               // this wave's logits (its k rows have landed)
    __syncthreads();
    L2S_ATT_STAMP(4);
    // ---- softmax over T
    static_assert(!(VLDS && TRAIN), "the LDS-staged form is the inference step's");
    float awr = 0.f;                                 // VLDS form: this lane's attention weight (lane = frame), computed by EVERY wave
    if constexpr (VLDS) {
        // T <= 32: each wave runs the 29-lane softmax itself (same shuffles, same bits in every wave) and keeps the weights in a register - no LDS
        // round trip and no third block barrier; a@V' below broadcasts weight e with v_readlane
        const bool on = lane < T;
        const float x = on ? sc[lane] : -INFINITY;
        const float mx = wave_max_f(x);
        const float ex = on ? expf(x - mx) : 0.f;
        const float tot = wave_sum_f(ex);
        awr = on ? ex / tot : 0.f;
        if (wave == 0 && on && pattn) pattn[(int64_t)b * ld_attn + lane] = logits ? x : awr;
    } else if (T <= 64) {
        // one wave, shuffles only (LRW: T = 29): one block barrier instead of five
        if (wave == 0) {
            const bool on = lane < T;
            float x = on ? sc[lane] : -INFINITY;
            if constexpr (TRAIN) { if (on && tr->logit_mask) x *= tr->logit_mask[(int64_t)b * tr->ld_lmask + lane]; }
            const float mx = wave_max_f(x);
            const float ex = on ? expf(x - mx) : 0.f;
            const float tot = wave_sum_f(ex);
            if (on) {
                const float aw = ex / tot;
                if (pattn) pattn[(int64_t)b * ld_attn + lane] = logits ? x : aw;
                sc[lane] = aw;
            }
        }
        __syncthreads();
    } else {
        // T <= 320 < 512: one element per thread
        const bool on = tid < T;
        float x = on ? sc[tid] : -INFINITY;
        if constexpr (TRAIN) { if (on && tr->logit_mask) x *= tr->logit_mask[(int64_t)b * tr->ld_lmask + tid]; }
        const float mx = block_max8(x, scratch);
        const float ex = on ? expf(x - mx) : 0.f;
        const float tot = block_sum8(ex, scratch);
        if (on) {
            const float aw = ex / tot;
            if (pattn) pattn[(int64_t)b * ld_attn + tid] = logits ? x : aw;
            sc[tid] = aw;
        }
        __syncthreads();
    }
    L2S_ATT_STAMP(5);               // attention weights visible
    // ---- av = a @ v : one column per thread, t ascending
    float acc = 0.f;
    if constexpr (SKIP0) {
        const unsigned long long nz = __ballot(awr != 0.f);          // the same in every wave (each ran the same soft-max)
        if (wave < 4) {                                               // 256 value columns: waves 0-3, one column per thread
            const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(pv + (int64_t)b * T * 256), 0, T * 1024, 0x00020000);
            unsigned long long m = nz;
            while (m) {                                               // four frames per round: their loads are in flight together; t ascending as in the full form
                int tt[4]; float ww[4], vl[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const bool on = m != 0ull;
                    const int t = on ? (int)__builtin_ctzll(m) : 0;
                    tt[i] = __builtin_amdgcn_readfirstlane(t);
                    ww[i] = on ? __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, awr), tt[i])) : 0.f;
                    m = on ? (m & (m - 1ull)) : 0ull;
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) vl[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rv, tid * 4, tt[i] * 1024, 0));
#pragma unroll
                for (int i = 0; i < 4; ++i) acc = fmaf(ww[i], vl[i], acc);     // an unused slot adds fmaf(0, v, acc) = acc
            }
        }
    } else if constexpr (vlds) {
        const float* vc = vs + (tid & 255);
#pragma unroll
        for (int h = 0; h < 2; ++h) {                                         // sixteen unconditional reads in flight at a time (a guarded read per frame serialises them)
            float vl[16];
#pragma unroll
            for (int e = 0; e < 16; ++e) vl[e] = vc[(16 * h + e < T ? 16 * h + e : 0) * 256];
#pragma unroll
            for (int e = 0; e < 16; ++e)
                if (16 * h + e < T) acc = fmaf(__builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, awr), 16 * h + e)), vl[e], acc);
        }
    } else {
#pragma unroll
        for (int e = 0; e < 32; ++e)
            if (e < T) acc = fmaf(sc[e], vv[e], acc);
    }
    for (int t0 = 32; t0 < T; t0 += 16) {
        float v2[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) v2[e] = (vlane && t0 + e < T) ? vb[(int64_t)(t0 + e) * vcols] : 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e)
            if (t0 + e < T) acc = fmaf(sc[t0 + e], v2[e], acc);
    }
    if (vlane) pav[frag16_index(b, tid, vcols)] = acc;
    if constexpr (TRAIN) { if (tr->av_plain) tr->av_plain[(int64_t)b * 512 + tid] = acc; }
    if constexpr (TIMED) { __builtin_amdgcn_s_waitcnt(0); L2S_ATT_STAMP(6); }
}

// Content.forward (decoder.py:262-271) for one batch row: alpha = softmax_m(SiLU(..)*tau_c . key), cc = alpha @ value
template <bool TRAIN = false>
__device__ __forceinline__ void content_block(const AttnP& p, int b, float* sm, const AttnTrain* tr = nullptr) {
    float* qs = sm;                  // 256
    float* csc = sm + 512;           // 16
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = p.m;
    const int col = tid & 255;
    const float* const pqc = p.qc; const float* const pkey = p.ckey; const float* const pval = p.cval; const float* const ptc = p.tau_c;
    float* const pcc = p.cc_frag; const int ldqc = p.ldqc;
    L2S_PIN_S("s"(m), "s"(pqc), "s"(pkey), "s"(pval), "s"(ptc), "s"(pcc), "s"(ldqc));
    const float qv = pqc[(int64_t)b * ldqc + col];
    const float tau_c = ptc[0];
    const float* keyb = pkey + (int64_t)b * m * 256 + lane * 4;
    const float* valb = pval + (int64_t)b * m * 256 + col;
    float4 kk[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int i = wave + 8 * r;
        if (i < m) kk[r] = *reinterpret_cast<const float4*>(keyb + (int64_t)i * 256);
    }
    float vals[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) vals[i] = i < m ? valb[(int64_t)i * 256] : 0.f;
    if (tid < 256) qs[tid] = qv * tau_c;
    __syncthreads();
    const float* q4 = qs + lane * 4;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int i = wave + 8 * r;
        if (i < m) {
            double d = (double)q4[0] * kk[r].x + (double)q4[1] * kk[r].y + (double)q4[2] * kk[r].z + (double)q4[3] * kk[r].w;
            d = wave_sum_d(d);
            if (lane == 0) csc[i] = (float)d;
        }
    }
    __syncthreads();
    if (tid < 256) {
        float cmx = -INFINITY;
        for (int i = 0; i < m; ++i) cmx = fmaxf(cmx, csc[i]);
        float csum = 0.f;
        for (int i = 0; i < m; ++i) csum += expf(csc[i] - cmx);
        float o = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i)
            if (i < m) o = fmaf(expf(csc[i] - cmx) / csum, vals[i], o);
        pcc[frag16_index(b, tid, 256)] = o;
        if constexpr (TRAIN) {
            if (tr->cc_plain) tr->cc_plain[(int64_t)b * 256 + tid] = o;
            if (tr->alpha && tid < m) tr->alpha[(int64_t)b * tr->ld_alpha + tid] = expf(csc[tid] - cmx) / csum;
        }
    }
}

}  // namespace l2s
