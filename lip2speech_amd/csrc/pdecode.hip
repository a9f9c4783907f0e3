// The latency form of the autoregressive decode loop (reference/model/modules/decoder.py:412-435) for one or two clips (demo.py:60-90 runs batch_size = 1,
// BASELINE config 1 runs 2): ONE launch for all S steps, weight-stationary.
//
// At <= 32 rows the launch-per-phase loop (l2s_api.hip decode_launches) is bound by four dependent launches per step (~20-23 us: boundary, first-operand
// latency and 21 MB of step weights streamed from the Infinity Cache 300 times).  Here the 5.25 M step weights live in the REGISTERS of 128 resident
// workgroups per clip (162 VGPRs per thread): a workgroup owns four hidden units of both LSTM layers (16 gate columns each), four columns of Q, two of
// content-Q, of prenet1 o fc_out, of prenet2 and of fc_out+stop; thread t owns k = 2t, 2t+1 of every 512-long half of a column.  The per-step activations
// (h, c, q, qc, prenet: 3 072 floats per clip) cross the chip between the phases of a step as 8-byte {value, tag} granules, written with one write-through
// store each and polled by the threads that consume them - the data is the flag (cdna_hip_programming.md section 6 Guideline 16 R2); a thread reads
// exactly the granules of its own K slice, straight into registers.  Every phase consumes a full vector from all of the clip's producers, so a phase is
// also a barrier among them and single-buffered granules are safe.  Keys (registers), projected values and content values (LDS) of the clip sit in every
// one of its workgroups, so attention is computed whole and locally (T <= 32 frames).  Two clips share nothing: each has its half of the chip.
//
// Measured on MI355X (profiles/r04_persist_edge_probe.txt, r04_pdecode_timeline.txt, r04_latency_path.txt): one all-to-all edge costs ~1.2 us among 128
// workgroups and ~1.8 among 256 with the exchange buffer replicated per XCD (2.1-2.6 with all 256 polling one copy), growing linearly beyond 2 048 floats
// (9.5 us at 16 384): the form pays for one or two clips and is not used above.  Arithmetic: fp32 FMAs over the K slices, DPP / row-swap trees across the
// 256 threads - another order of the same sums as the launch path (both within 1e-3 of the reference; tests/test_gpu_parity.py::test_persistent_decode_*).
#include "l2s_common.h"
#include "l2s_model.h"
#include "pdecode.h"
#include "skinny_dev.h"

#include <algorithm>
#include <cstdlib>
#include <mutex>

namespace l2s {

constexpr int PD_WG = 256;                // workgroups of a full launch = compute units (128 per clip)
constexpr int PD_NT = 256;                // threads per workgroup
constexpr int PD_MAXT = 32, PD_MAXM = 16, PD_MAXB = 4;
constexpr int PD_LDS_MIN = 84 * 1024;     // at least 84 KB of LDS per workgroup: more than half of a CU's, so the workgroups sit one per CU
constexpr int PD_LDS_MAX = 159 * 1024;    // most dynamic LDS a launch may ask for (a few static bytes ride along)
// LDS of a workgroup, in floats: the fixed part, then its clip's projected values V' [T4][256] and content values [m4][256] (T4, m4 = T, m rounded up to 4;
// the rows past T / m are zero)
__host__ __device__ constexpr int pd_lds_fixed(int V) { return 512 + 256 + 48 + 2 * 4 * 8 * V + 2 * 4 * 12 * V + 4 * 32; }
__host__ __device__ inline int pd_lds_floats(int V, int T, int m) { return pd_lds_fixed(V) + (((T + 3) & ~3) + ((m + 3) & ~3)) * 256; }
// granule arrays of one replica, in u64 units, for NB clips: [h0 | h1 | c0 | c1 | q: NB x 512 each][qc, p1, p2: NB x 256 each]
__host__ __device__ constexpr int pd_off_h0(int NB) { return 0; }
__host__ __device__ constexpr int pd_off_h1(int NB) { return NB * 512; }
__host__ __device__ constexpr int pd_off_c0(int NB) { return 2 * NB * 512; }
__host__ __device__ constexpr int pd_off_c1(int NB) { return 3 * NB * 512; }
__host__ __device__ constexpr int pd_off_q(int NB) { return 4 * NB * 512; }
__host__ __device__ constexpr int pd_off_qc(int NB) { return 5 * NB * 512; }
__host__ __device__ constexpr int pd_off_p1(int NB) { return pd_off_qc(NB) + NB * 256; }
__host__ __device__ constexpr int pd_off_p2(int NB) { return pd_off_p1(NB) + NB * 256; }
__host__ __device__ constexpr int pd_granules(int NB) { return pd_off_p2(NB) + NB * 256; }
constexpr int PD_MAXREP = 8;
__host__ __device__ constexpr int pd_rstride(int NB) { return pd_granules(NB) + 520; }      // replicas 4 KB + 64 B off a power-of-two pitch

#define PD_RLX __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

template <int CTRL>
__device__ __forceinline__ float pd_dpp(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
}
// add the other three rows of the wave: v_permlane16_swap / v_permlane32_swap exchange whole rows of 16 lanes (gfx950)
__device__ __forceinline__ float pd_rows_sum(float v) {
    const int x = __builtin_bit_cast(int, v);
    auto a = __builtin_amdgcn_permlane16_swap(x, x, false, false);
    const float s1 = __builtin_bit_cast(float, (int)a[0]) + __builtin_bit_cast(float, (int)a[1]);
    const int y = __builtin_bit_cast(int, s1);
    auto b = __builtin_amdgcn_permlane32_swap(y, y, false, false);
    return __builtin_bit_cast(float, (int)b[0]) + __builtin_bit_cast(float, (int)b[1]);
}
__device__ __forceinline__ float pd_rows_max(float v) {
    const int x = __builtin_bit_cast(int, v);
    auto a = __builtin_amdgcn_permlane16_swap(x, x, false, false);
    const float s1 = fmaxf(__builtin_bit_cast(float, (int)a[0]), __builtin_bit_cast(float, (int)a[1]));
    const int y = __builtin_bit_cast(int, s1);
    auto b = __builtin_amdgcn_permlane32_swap(y, y, false, false);
    return fmaxf(__builtin_bit_cast(float, (int)b[0]), __builtin_bit_cast(float, (int)b[1]));
}
// sum / max over the 64 lanes of a wave, result in every lane
__device__ __forceinline__ float pd_wave_sum(float v) {
    v += pd_dpp<0xB1>(v);       // quad_perm [1,0,3,2]
    v += pd_dpp<0x4E>(v);       // quad_perm [2,3,0,1]
    v += pd_dpp<0x124>(v);      // row_ror:4
    v += pd_dpp<0x128>(v);      // row_ror:8
    return pd_rows_sum(v);
}
__device__ __forceinline__ float pd_wave_max(float v) {
    v = fmaxf(v, pd_dpp<0xB1>(v)); v = fmaxf(v, pd_dpp<0x4E>(v)); v = fmaxf(v, pd_dpp<0x124>(v)); v = fmaxf(v, pd_dpp<0x128>(v));
    return pd_rows_max(v);
}
// V per-thread values summed over the wave at once (V a multiple of 4): two transposing quad steps (a lane keeps half of its values and hands the other
// half to its partner), two row rotations, the row exchange - 1.9 V + 6 V / 4 instructions instead of 11 V.  Afterwards lane l holds the wave total of
// value 4 j + (l & 3) in v[j], j < V / 4.
template <int V>
__device__ __forceinline__ void pd_wave_sum_multi(float (&v)[V], int lane) {
    static_assert(V % 4 == 0, "values in fours");
    const bool b0 = lane & 1, b1 = lane & 2;
#pragma unroll
    for (int j = 0; j < V / 2; ++j) {
        const float keep = b0 ? v[2 * j + 1] : v[2 * j], send = b0 ? v[2 * j] : v[2 * j + 1];
        v[j] = keep + pd_dpp<0xB1>(send);
    }
#pragma unroll
    for (int j = 0; j < V / 4; ++j) {
        const float keep = b1 ? v[2 * j + 1] : v[2 * j], send = b1 ? v[2 * j] : v[2 * j + 1];
        float x = keep + pd_dpp<0x4E>(send);
        x += pd_dpp<0x124>(x);
        x += pd_dpp<0x128>(x);
        v[j] = pd_rows_sum(x);
    }
}
// the cells' gate non-linearities on the hardware exp2 / rcp (1 ulp each): absolute error < 3e-7 on values in [0, 1] / [-1, 1], a fifth of the
// instructions of expf / tanhf / an IEEE division (12.2 against 13.0 us per step at one clip; same deviation from the launch path)
__device__ __forceinline__ float pd_sigmoid(float x) { return __frcp_rn(1.0f + __expf(-x)); }
__device__ __forceinline__ float pd_tanh(float x) { return 1.0f - 2.0f * __frcp_rn(1.0f + __expf(2.0f * x)); }

// one 8-byte write-through store per replica: value and tag land together.  The exchange buffer exists PD_MAXREP times, pd_rstride granules apart:
// every producer writes all replicas (the replica as the buffer instruction's scalar offset), a consumer polls replica (workgroup % 8) - its XCD's own
// copy, so the 256 pollers of a granule line become 32 per copy (one copy: 16.7 us per step at one clip, two: 14.4, four: 13.2, eight: 13.1)
typedef unsigned pd_u2 __attribute__((ext_vector_type(2)));
typedef float pd_f2 __attribute__((ext_vector_type(2)));
template <int NB>
__device__ __forceinline__ void pd_publish(__amdgpu_buffer_rsrc_t rsall, int granule, unsigned tag, float v) {
    pd_u2 x; x.x = __float_as_uint(v); x.y = tag;
#pragma unroll
    for (int r = 0; r < PD_MAXREP; ++r) __builtin_amdgcn_raw_buffer_store_b64(x, rsall, granule * 8, r * pd_rstride(NB) * 8, 16);
}

// One poll pass = every load of the phase issued, then every tag compared; repeated until the whole WAVE has fresh granules, so the wave stays
// converged for the DPP reductions that follow.  A wave whose polls have made no progress for PD_GIVE_UP_TICKS of the 100 MHz wall clock (~2 s: the
// workgroups it waits for are not resident - a CU-masked or shared device; other streams' kernels delay residency by milliseconds, not seconds) raises
// status[0] and leaves, and so, each on its own clock, does every other wave that waits for it.  The clock is read every 256th failed pass only (the
// first read starts the budget), so a pass that succeeds - the normal case: a few passes per edge - never pays for it.  (Two passes in flight half a
// round trip apart - register sets taking turns, one exit branch so that the compiler waits for the older pass alone - were measured: 9.6-10.0 against
// 9.7 us per step; more polling is more contention, not a shorter edge.)
constexpr unsigned long long PD_GIVE_UP_TICKS = 200000000ull;      // 2 s of wall_clock64() without a fresh granule
struct PdPoll {
    unsigned* status; unsigned spins; bool dead; unsigned long long t0;
    __device__ __forceinline__ PdPoll(unsigned* st) : status(st), spins(0), dead(false), t0(0) {}
    __device__ __forceinline__ bool retry(bool ok) {         // true: go around again
        if (__all(ok)) { spins = 0; return false; }
        if ((++spins & 255u) == 0u) {
            const unsigned long long now = wall_clock64();
            if (spins == 256u) t0 = now;
            else if (now - t0 > PD_GIVE_UP_TICKS) { dead = true; return false; }
        }
        __builtin_amdgcn_s_sleep(1);
        asm volatile("" ::: "memory");
        return true;
    }
    __device__ __forceinline__ bool gave_up() {              // after a poll loop
        if (dead) __hip_atomic_store(status, 1u, PD_RLX);
        return dead;
    }
};
// two adjacent granules {value, tag} x 2 as ONE 16-byte L1-bypassing load (buffer_load_dwordx4 ... sc1) through the exchange buffer's descriptor
__device__ __forceinline__ uint4 pd_load16(__amdgpu_buffer_rsrc_t rs, int granule) {
    return __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rs, granule * 8, 0, 16));
}
__device__ __forceinline__ u64 pd_load8(__amdgpu_buffer_rsrc_t rs, int granule) {
    return __builtin_bit_cast(u64, __builtin_amdgcn_raw_buffer_load_b64(rs, granule * 8, 0, 16));
}
// frag16 address of W[n][k .. k+1] (k even) in a packed [N][K] matrix with NC = K / 16 chunks per 16-row tile
__device__ __forceinline__ float2 pd_w2(const float* W, int NC, int n, int k) {
    return *reinterpret_cast<const float2*>(W + ((int64_t)((n >> 4) * NC + (k >> 4)) * 64 + ((k >> 2) & 3) * 16 + (n & 15)) * 4 + (k & 3));
}

// NG = clips of the launch, V = "virtual" workgroups per workgroup.  Clip g is served by the 256 / V workgroups g * 256 / V ..., and such a workgroup
// stands for V workgroups jv = V * (j mod 256 / V) + b of the one-per-CU layout (units 2 jv, 2 jv + 1 of both layers, columns 2 jv, 2 jv + 1 of Q,
// column jv of content-Q / prenet1 o fc_out / prenet2 / fc_out+stop: 81 V weight registers per thread).  An all-to-all edge is cheaper the fewer
// workgroups take part (128: ~1.3 us, 256: ~1.8), more than the doubled per-thread work costs: V = 2 for one clip (half the chip idle) and for two.  The clips never meet:
// a workgroup polls and publishes its own clip's vectors only, holds its own clip's keys / values only - two clips run as two one-clip loops side by
// side on half the chip each (11.6 -> ~9.8 us per step against the form in which every workgroup served both clips and did both clips' attention).
// A step is four phases, and every phase polls ONLY what the phase before it produced (each vector crosses the chip once per step):
//   1: h1', c1' (of the previous step) -> the h1 / c1 halves of Q, content-Q; prenet1 o fc_out; W_hh1 h1'; mel frame + stop logit of the previous step
//   2: wave 3 alone turns prenet1 into this workgroup's prenet2 columns and publishes them at once; q, qc -> while prenet2 crosses the chip - attention and
//      content attention of the clip, whole, in every one of its workgroups (keys in registers, values in LDS): o and cc never leave the CU
//   3: prenet2 (+ the local o = u, local cc) -> W_ih0 [cc | u]; with W_hh0 h0 kept from phase 4 of the previous step: the layer-0 cells
//   4: h0', c0' -> W_ih1 h0'; with W_hh1 h1 kept from phase 1: the layer-1 cells; and for the next step the h0 / c0 halves of Q, content-Q and W_hh0 h0'
// Only what the NEXT phase waits for is summed across the threads before a phase publishes (4 NG values in phase 1, the 8 NG gates in phases 3 / 4); the
// sums a later phase needs (W_hh products, the mel frame) are reduced after the publish, while the vector is on its way.
// (Measured and not kept: prenet2 computed once PER XCD - eight columns per workgroup, a plain store that stays in the XCD's L2, its 32 workgroups polling
// it there, XCC ids and ranks taken at start - 9.24 against 9.33 us; the same loop on EIGHT waves, 512 threads with k = t - 9.41 against 9.48 at one clip,
// 14.3 against 12.7 at two: a step is its four edges, 4 x ~1.75 us, plus ~2.4 us.)
template <int NG, int V>
__global__ __launch_bounds__(PD_NT, 1) void pdecode_kernel(const PDecP p) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* const qs = sm;                                   // [512]              q * tau
    float* const qcs = qs + 512;                            // [256]              qc * tau_c
    float* const sc = qcs + 256;                            // [48]               attention logits [0, 32), content logits [32, 48)
    constexpr int RS = 8 * V, RD = 12 * V;                // floats per wave: a phase's critical sums, its deferred sums
    float* const red = sc + 48;                             // [2][4 waves][RS]   critical sums, ping-pong by phase
    float* const redB = red + 2 * 4 * RS;                   // [4 waves][RD]      phase 1's deferred sums: W_hh1 h1', fc
    float* const redC = redB + 4 * RD;                      // [4 waves][RD]      phase 4's deferred sums: W_hh0 h0', the h0 / c0 parts of q0, q1, qc
    float* const aws = redC + 4 * RD;                       // [4 waves][32]      each wave's softmax weights, for broadcast reads
    float* const vs = aws + 4 * 32;                         // [T4 + m4][256]     projected values V' and content values of the clip, rows past T / m zero
    const int T4 = (p.T + 3) & ~3, M4 = (p.m + 3) & ~3;

    constexpr int WPG = PD_WG / V;                          // workgroups per clip (the launch has NG * WPG of them)
    const int j = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const int g = j / WPG, jl = j - g * WPG;                // this workgroup's clip (of the launch's NG), its index among the clip's workgroups
    const int gc = p.b0 + g;                                // ... the clip's row in the caller's batch
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int T = p.T, M = p.m, S = p.S;
    u64* const X = p.xch;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(X + (int64_t)(j % PD_MAXREP) * pd_rstride(NG), 0, pd_granules(NG) * 8, 0x00020000);      // this workgroup's replica
    const __amdgpu_buffer_rsrc_t rsall = __builtin_amdgcn_make_buffer_rsrc(X, 0, PD_MAXREP * pd_rstride(NG) * 8, 0x00020000);
    // this clip's vectors inside a replica
    const int gH0 = pd_off_h0(NG) + g * 512, gH1 = pd_off_h1(NG) + g * 512, gC0 = pd_off_c0(NG) + g * 512, gC1 = pd_off_c1(NG) + g * 512;
    const int gQ = pd_off_q(NG) + g * 512, gQC = pd_off_qc(NG) + g * 256, gP1 = pd_off_p1(NG) + g * 256, gP2 = pd_off_p2(NG) + g * 256;
    PdPoll poll(p.status);
    unsigned long long* const ts = p.ts ? p.ts + (int64_t)j * 16 : nullptr;      // measurement: thread 0 of every workgroup stamps the phases of step p.ts_step
#define PD_STAMP(i) do { if (ts && s == p.ts_step && tid == 0) ts[i] = wall_clock64(); } while (0)

    // ---------------------------------------------------------------- weights into registers: thread t owns k = 2t, 2t + 1 of every 512-long half
    float2 wl0i[V][8], wl0h[V][8], wl1i[V][8], wl1h[V][8], wq0[V][2], wq1[V][2], wcq0[V], wcq1[V], wp1[V], wfc[V];
    float4 wp2[V];                                         // prenet2 column jv over k = 4 lane .. 4 lane + 3 (K = 256: one wave covers it; wave 3 does)
#pragma unroll
    for (int b = 0; b < V; ++b) {
        const int jv = V * jl + b;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            wl0i[b][c] = pd_w2(p.Wl0, 64, 8 * jv + c, 2 * tid); wl0h[b][c] = pd_w2(p.Wl0, 64, 8 * jv + c, 512 + 2 * tid);
            wl1i[b][c] = pd_w2(p.Wl1, 64, 8 * jv + c, 2 * tid); wl1h[b][c] = pd_w2(p.Wl1, 64, 8 * jv + c, 512 + 2 * tid);
        }
#pragma unroll
        for (int c = 0; c < 2; ++c) { wq0[b][c] = pd_w2(p.Wq, 64, 2 * jv + c, 2 * tid); wq1[b][c] = pd_w2(p.Wq, 64, 2 * jv + c, 512 + 2 * tid); }
        wcq0[b] = pd_w2(p.Wcq, 64, jv, 2 * tid); wcq1[b] = pd_w2(p.Wcq, 64, jv, 512 + 2 * tid);
        wp1[b] = pd_w2(p.Wp1f, 32, jv, 2 * tid);
        wfc[b] = jv < 96 ? pd_w2(p.Wfc, 32, jv, 2 * tid) : make_float2(0.f, 0.f);
        wp2[b] = *reinterpret_cast<const float4*>(p.Wp2 + ((int64_t)((jv >> 4) * 16 + (lane >> 2)) * 64 + (lane & 3) * 16 + (jv & 15)) * 4);
    }
    const float tau = p.tau[0], tau_c = p.tau_c[0];
    // epilogue constants of the threads that finish a phase (tid < 4 V in phase 1, lanes < V of wave 3 in phase 2, tid < 2 V in the cells, tid < V for the mel)
    const int fb = tid >> 2, fk = tid & 3;                  // phase-1 finisher: virtual workgroup fb, value fk (q0, q1, qc, p1)
    const int fjv = V * jl + (fb < V ? fb : V - 1);
    const float e_b1 = fk < 2 ? p.bq[2 * fjv + fk] : fk == 2 ? p.bcq[fjv] : p.bp1f[fjv];
    const float e_a1 = fk < 2 ? p.aq[2 * fjv + fk] : p.ap1[fjv];
    const int mjv = V * jl + (tid < V ? tid : V - 1);    // mel / stop: thread tid < V finishes column mjv (< 81)
    const float e_bfc = mjv < 96 ? p.bfc[mjv] : 0.f;
    const int pjv = V * jl + (lane < V ? lane : V - 1);  // prenet2: lane < V of wave 3 finishes column pjv
    const float e_b2 = p.bp2[pjv], e_a2 = p.ap2[pjv];
    const int cb = tid >> 1, cu = tid & 1;                  // cell thread: virtual workgroup cb, unit 2 jv + cu
    const int cjv = V * jl + (cb < V ? cb : V - 1);
    float e_bl0[4], e_bl1[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) { e_bl0[q] = p.bl0[8 * cjv + 4 * cu + q]; e_bl1[q] = p.bl1[8 * cjv + 4 * cu + q]; }
    float c0 = 0.f, c1 = 0.f;                               // cell states of the cell threads (decoder.py:406: zeros)
    const float stopc = p.stop_const[gc];

    // ---------------------------------------------------------------- the clip's keys / content keys into registers, its values into LDS
    const int kf = tid >> 3, kp = tid & 7;                  // attention logits: 8 threads per frame, 64 k each (k = 4 kp + 32 i + e)
    const int cf = tid >> 4, cp = tid & 15;                 // content logits: 16 threads per content frame, 16 k each
    const bool o_role = wave >= 2;                          // waves 2, 3 form u[2 (tid - 128) ..] = prenet2 + o; waves 0, 1 form cc[2 tid ..] (wave-uniform: a scalar)
    const int xc = o_role ? 2 * (tid - 128) : 2 * tid;      // this thread's pair of columns of o / cc
    float4 kreg[16], ckreg[4];
    const float* const vrow = vs + (o_role ? 0 : T4 * 256) + xc;      // this thread's pair of columns in LDS, row pitch 256
#pragma unroll
    for (int i = 0; i < 16; ++i)
        kreg[i] = kf < T ? *reinterpret_cast<const float4*>(p.k + ((int64_t)gc * T + kf) * 512 + 4 * kp + 32 * i) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int i = 0; i < 4; ++i)
        ckreg[i] = cf < M ? *reinterpret_cast<const float4*>(p.ckey + ((int64_t)gc * M + cf) * 256 + 4 * (cp + 16 * i)) : make_float4(0.f, 0.f, 0.f, 0.f);
    for (int i = tid; i < T4 * 64; i += PD_NT)
        *reinterpret_cast<float4*>(vs + 4 * i) = i < T * 64 ? *reinterpret_cast<const float4*>(p.vp + (int64_t)gc * T * 256 + 4 * i) : make_float4(0.f, 0.f, 0.f, 0.f);
    for (int i = tid; i < M4 * 64; i += PD_NT)
        *reinterpret_cast<float4*>(vs + T4 * 256 + 4 * i) = i < M * 64 ? *reinterpret_cast<const float4*>(p.cval + (int64_t)gc * M * 256 + 4 * i) : make_float4(0.f, 0.f, 0.f, 0.f);
    // prenet1 of the BOS frame (step 0 has no previous h1; decoder.py:407,413): the finisher's column (fk = 3 uses it)
    float p1_bos;
    {
        float bosv[V];
#pragma unroll
        for (int b = 0; b < V; ++b) {
            const int jv = V * jl + b;
            float a = 0.f;
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int k = lane + 64 * r;
                if (k < 80) a = fmaf(p.Wp1[(((int64_t)((jv >> 4) * 5 + (k >> 4)) * 64 + ((k >> 2) & 3) * 16 + (jv & 15)) * 4) + (k & 3)], p.bos[k], a);
            }
            bosv[b] = sinf(pd_wave_sum(a) + p.bp1[jv]) * p.ap1[jv];
        }
        p1_bos = bosv[0];
#pragma unroll
        for (int b = 1; b < V; ++b) if (fb == b) p1_bos = bosv[b];
    }

    // cross-thread sums: V values per thread -> wave totals (pd_wave_sum_multi: lane l < 4 ends with value 4 i + l in vals[i]) -> dst[wave][..]; after the
    // block's next barrier PD_SUM4(dst, stride, i) adds the four waves' totals of value i
    int rpar = 0;
#define PD_WSUM(vals, V, dst, stride)                                                                             \
    do {                                                                                                          \
        pd_wave_sum_multi<(V)>(vals, lane);                                                                       \
        if (lane < 4) { _Pragma("unroll") for (int _i = 0; _i < (V) / 4; ++_i) (dst)[wave * (stride) + 4 * _i + lane] = (vals)[_i]; } \
    } while (0)
#define PD_SUM4(src, stride, i) (((src)[(i)] + (src)[(stride) + (i)]) + ((src)[2 * (stride) + (i)] + (src)[3 * (stride) + (i)]))

    // what phase 4 of "step -1" would have left in redC: W_hh0 h0 and the h0 halves of Q from the prologue's h0 (c0 = 0: nothing for content-Q); h1 / c1 as granules
    float l0hh[4] = {0.f, 0.f, 0.f, 0.f}, l1hh[4] = {0.f, 0.f, 0.f, 0.f};      // cell threads: W_hh0 h0 / W_hh1 h1 of their four gates
    {
        const int64_t hb = (int64_t)((p.B + 15) & ~15) * 512;
        const float2 h = *reinterpret_cast<const float2*>(p.h_init + frag16_index(gc, 2 * tid, 512));
        float v[12 * V];
#pragma unroll
        for (int b = 0; b < V; ++b) {
#pragma unroll
            for (int c = 0; c < 8; ++c) v[12 * b + c] = fmaf(wl0h[b][c].y, h.y, wl0h[b][c].x * h.x);
            v[12 * b + 8] = fmaf(wq0[b][0].y, h.y, wq0[b][0].x * h.x);
            v[12 * b + 9] = fmaf(wq0[b][1].y, h.y, wq0[b][1].x * h.x);
            v[12 * b + 10] = 0.f; v[12 * b + 11] = 0.f;
        }
        PD_WSUM(v, 12 * V, redC, RD);
        if (tid < 2 * V) {
            const int unit = 2 * cjv + cu;
            pd_publish<NG>(rsall, gH1 + unit, 1u, p.h_init[hb + frag16_index(gc, unit, 512)]);
            pd_publish<NG>(rsall, gC1 + unit, 1u, 0.f);
        }
        __syncthreads();
    }

    for (int s = 0; s <= S; ++s) {
        const unsigned tag_prev = (unsigned)s + 1u, tag_now = (unsigned)s + 2u;
        PD_STAMP(0);
        // ------------------------------------------------------------ phase 1
        float2 hy, cy;
        {
            const float posv = (s < S && tid < 4 * V && fk < 2) ? p.pos[(int64_t)s * 512 + 2 * fjv + fk] : 0.f;      // requested before the poll, used after it
            do {
                const uint4 a = pd_load16(rs, gH1 + 2 * tid), c = pd_load16(rs, gC1 + 2 * tid);
                const bool ok = a.y == tag_prev && a.w == tag_prev && c.y == tag_prev && c.w == tag_prev;
                hy = make_float2(__uint_as_float(a.x), __uint_as_float(a.z));
                cy = make_float2(__uint_as_float(c.x), __uint_as_float(c.z));
                if (!poll.retry(ok)) break;
            } while (true);
            if (poll.gave_up()) return;
            PD_STAMP(1);
            if (s < S) {
                float v[4 * V];
#pragma unroll
                for (int b = 0; b < V; ++b) {
                    v[4 * b + 0] = fmaf(wq1[b][0].y, hy.y, wq1[b][0].x * hy.x);
                    v[4 * b + 1] = fmaf(wq1[b][1].y, hy.y, wq1[b][1].x * hy.x);
                    v[4 * b + 2] = fmaf(wcq1[b].y, cy.y, wcq1[b].x * cy.x);
                    v[4 * b + 3] = fmaf(wp1[b].y, hy.y, wp1[b].x * hy.x);
                }
                PD_WSUM(v, 4 * V, red + rpar * 4 * RS, RS);
                __syncthreads();
                PD_STAMP(2);
                if (tid < 4 * V) {                          // q0, q1, qc, p1 of virtual workgroup fb: one sinf and one expf for the four lanes, then a select
                    const float qh = fk < 3 ? PD_SUM4(redC, RD, 12 * fb + 8 + fk) : 0.f;      // the h0 / c0 part, from phase 4 of the previous step
                    const float x = PD_SUM4(red + rpar * 4 * RS, RS, 4 * fb + fk) + qh + e_b1;
                    const float sx = sinf(x) * e_a1, ex = x / (1.0f + expf(-x));
                    const float val = fk < 2 ? sx + posv : fk == 2 ? ex : (s == 0 ? p1_bos : sx);
                    const int gg = fk < 2 ? gQ + 2 * fjv + fk : fk == 2 ? gQC + fjv : gP1 + fjv;
                    pd_publish<NG>(rsall, gg, tag_now, val);
                }
                rpar ^= 1;
            }
            PD_STAMP(3);
            if (s < S && tid < 2 * V) {                     // W_hh0 h0 from phase 4 of the previous step, for phase 3 (redC is valid since this phase's barrier)
#pragma unroll
                for (int q = 0; q < 4; ++q) l0hh[q] = PD_SUM4(redC, RD, 12 * cb + 4 * cu + q);
            }
            {   // deferred: W_hh1 h1' for phase 4, the mel frame / stop logit of step s - 1
                float v[12 * V];
#pragma unroll
                for (int b = 0; b < V; ++b) {
#pragma unroll
                    for (int c = 0; c < 8; ++c) v[12 * b + c] = fmaf(wl1h[b][c].y, hy.y, wl1h[b][c].x * hy.x);
                    v[12 * b + 8] = fmaf(wfc[b].y, hy.y, wfc[b].x * hy.x);
                    v[12 * b + 9] = 0.f; v[12 * b + 10] = 0.f; v[12 * b + 11] = 0.f;
                }
                PD_WSUM(v, 12 * V, redB, RD);
            }
        }
        if (s == S) {                                        // the last mel frame
            __syncthreads();
            if (tid < V && mjv <= 80) {
                const float x = PD_SUM4(redB, RD, 12 * tid + 8) + e_bfc;
                if (mjv < 80) p.mel[((int64_t)gc * S + (S - 1)) * 80 + mjv] = x; else p.stop[(int64_t)gc * S + (S - 1)] = x + stopc;
            }
            break;
        }
        // ------------------------------------------------------------ phase 2
        float2 xin;                                          // this thread's pair of layer-0 inputs: cc (waves 0, 1) or o (waves 2, 3; prenet2 is added in phase 3)
        uint4 ppre;                                          // waves 2, 3: prenet2 requested ahead of the soft-max (published ~1.5 us earlier: it is usually there)
        {
            float2 qv; float qcv;
            float4 pv;
            do {
                const uint4 a = pd_load16(rs, gQ + 2 * tid);
                const u64 c = pd_load8(rs, gQC + tid);
                bool ok = a.y == tag_now && a.w == tag_now && (unsigned)(c >> 32) == tag_now;
                qv = make_float2(__uint_as_float(a.x), __uint_as_float(a.z));
                qcv = __uint_as_float((unsigned)c);
                if (wave == 3) {                             // prenet1, k = 4 lane .. 4 lane + 3
                    const uint4 x0 = pd_load16(rs, gP1 + 4 * lane), x1 = pd_load16(rs, gP1 + 4 * lane + 2);
                    ok &= x0.y == tag_now && x0.w == tag_now && x1.y == tag_now && x1.w == tag_now;
                    pv = make_float4(__uint_as_float(x0.x), __uint_as_float(x0.z), __uint_as_float(x1.x), __uint_as_float(x1.z));
                }
                if (!poll.retry(ok)) break;
            } while (true);
            if (poll.gave_up()) return;
            PD_STAMP(4);
            if (wave == 3) {                                 // prenet2, this workgroup's columns: the wave's own sums, out at once
                float mine = 0.f;
#pragma unroll
                for (int b = 0; b < V; ++b) {
                    const float t2 = pd_wave_sum(fmaf(wp2[b].w, pv.w, fmaf(wp2[b].z, pv.z, fmaf(wp2[b].y, pv.y, wp2[b].x * pv.x))));
                    if (lane == b) mine = t2;
                }
                if (lane < V) pd_publish<NG>(rsall, gP2 + pjv, tag_now, __sinf(mine + e_b2) * e_a2);
            }
            *reinterpret_cast<float2*>(qs + 2 * tid) = make_float2(qv.x * tau, qv.y * tau);
            qcs[tid] = qcv * tau_c;
            __syncthreads();                                 // qs / qcs visible to the block (and phase 1's deferred sums in redB)
            PD_STAMP(5);
            {   // logits: 8 threads per frame (keys in registers, q from LDS: the 8 frames of a wave read the same addresses), 16 threads per content frame
                pd_f2 a01 = {0.f, 0.f}, a23 = {0.f, 0.f};     // two packed accumulators: v_pk_fma_f32, two products per instruction
                const float* qr = qs + 4 * kp;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const float4 qq = *reinterpret_cast<const float4*>(qr + 32 * i);
                    a01 = __builtin_elementwise_fma(pd_f2{qq.x, qq.y}, pd_f2{kreg[i].x, kreg[i].y}, a01);
                    a23 = __builtin_elementwise_fma(pd_f2{qq.z, qq.w}, pd_f2{kreg[i].z, kreg[i].w}, a23);
                }
                float a = (a01.x + a01.y) + (a23.x + a23.y);
                a += pd_dpp<0xB1>(a); a += pd_dpp<0x4E>(a); a += pd_dpp<0x141>(a);      // the 8 lanes of a frame (row_half_mirror)
                if (kp == 0 && kf < T) sc[kf] = a;
                float c0a = 0.f, c1a = 0.f, c2a = 0.f, c3a = 0.f;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float4 qq = *reinterpret_cast<const float4*>(qcs + 4 * (cp + 16 * i));
                    c0a = fmaf(qq.x, ckreg[i].x, c0a); c1a = fmaf(qq.y, ckreg[i].y, c1a); c2a = fmaf(qq.z, ckreg[i].z, c2a); c3a = fmaf(qq.w, ckreg[i].w, c3a);
                }
                float ca = (c0a + c1a) + (c2a + c3a);
                ca += pd_dpp<0xB1>(ca); ca += pd_dpp<0x4E>(ca); ca += pd_dpp<0x124>(ca); ca += pd_dpp<0x128>(ca);      // the 16 lanes of a content frame
                if (cp == 0 && cf < M) sc[32 + cf] = ca;
            }
            __syncthreads();
            PD_STAMP(6);
            if (o_role) ppre = pd_load16(rs, gP2 + xc);
            {   // every wave: softmax of its role's logits (lane = frame), then its threads' two columns of a @ V' / alpha @ value (decoder.py:414-419, 262-271)
                const int n = o_role ? T : M;
                const bool on = lane < n;
                const float x = on ? sc[(o_role ? 0 : 32) + lane] : -INFINITY;
                const float mx = pd_wave_max(x);
                const float ex = on ? expf(x - mx) : 0.f;
                const float aw = ex * __frcp_rn(pd_wave_sum(ex));
                if (wave == 2 && jl == 0 && on && p.attn) p.attn[((int64_t)gc * S + s) * T + lane] = p.attn_logits ? x : aw;
                // the wave's weights through LDS (lanes past n hold 0, value rows past T / m are 0): four frames per trip, no per-frame predicate
                if (lane < 32) aws[wave * 32 + lane] = aw;
                __builtin_amdgcn_wave_barrier();
                const int n4 = (n + 3) & ~3;
                float ox = 0.f, oy = 0.f, ox2 = 0.f, oy2 = 0.f;
#pragma unroll 2
                for (int f = 0; f < n4; f += 4) {
                    const float4 w4 = *reinterpret_cast<const float4*>(aws + wave * 32 + f);
                    const float2 v0 = *reinterpret_cast<const float2*>(vrow + f * 256), v1 = *reinterpret_cast<const float2*>(vrow + (f + 1) * 256);
                    const float2 v2 = *reinterpret_cast<const float2*>(vrow + (f + 2) * 256), v3 = *reinterpret_cast<const float2*>(vrow + (f + 3) * 256);
                    ox = fmaf(w4.x, v0.x, ox); oy = fmaf(w4.x, v0.y, oy); ox2 = fmaf(w4.y, v1.x, ox2); oy2 = fmaf(w4.y, v1.y, oy2);
                    ox = fmaf(w4.z, v2.x, ox); oy = fmaf(w4.z, v2.y, oy); ox2 = fmaf(w4.w, v3.x, ox2); oy2 = fmaf(w4.w, v3.y, oy2);
                }
                xin = make_float2(ox + ox2, oy + oy2);
            }
            PD_STAMP(7);
        }
        // ------------------------------------------------------------ phase 3: layer-0 cells
        {
            if (o_role) {                                    // u = prenet + o (decoder.py:421): waves 2, 3 need prenet2 - first what was requested ahead
                bool first = true;
                do {
                    const uint4 a = first ? ppre : pd_load16(rs, gP2 + xc);
                    const bool ok = a.y == tag_now && a.w == tag_now;
                    first = false;
                    if (poll.retry(ok)) continue;            // stale: go around again
                    xin.x += __uint_as_float(a.x); xin.y += __uint_as_float(a.z);
                    break;
                } while (true);
            }
            if (poll.gave_up()) return;
            PD_STAMP(8);
            float v[8 * V];
#pragma unroll
            for (int b = 0; b < V; ++b)
#pragma unroll
                for (int c = 0; c < 8; ++c) v[8 * b + c] = fmaf(wl0i[b][c].y, xin.y, wl0i[b][c].x * xin.x);
            PD_WSUM(v, 8 * V, red + rpar * 4 * RS, RS);
            __syncthreads();
            PD_STAMP(9);
            if (tid < 2 * V) {
                const float* r = red + rpar * 4 * RS;
                const float gi = PD_SUM4(r, RS, 8 * cb + 4 * cu + 0) + l0hh[0] + e_bl0[0], gf = PD_SUM4(r, RS, 8 * cb + 4 * cu + 1) + l0hh[1] + e_bl0[1];
                const float gg = PD_SUM4(r, RS, 8 * cb + 4 * cu + 2) + l0hh[2] + e_bl0[2], go = PD_SUM4(r, RS, 8 * cb + 4 * cu + 3) + l0hh[3] + e_bl0[3];
                c0 = pd_sigmoid(gf) * c0 + pd_sigmoid(gi) * pd_tanh(gg);
                const float hn = pd_sigmoid(go) * pd_tanh(c0);
                pd_publish<NG>(rsall, gH0 + 2 * cjv + cu, tag_now, hn);
                pd_publish<NG>(rsall, gC0 + 2 * cjv + cu, tag_now, c0);
            }
            rpar ^= 1;
            PD_STAMP(10);
            // while h0' / c0' cross the chip: phase 1's deferred sums (valid since phase 2's first barrier) - the mel frame / stop logit of step s - 1
            // (decoder.py:423-428) and W_hh1 h1 for phase 4
            if (s > 0 && tid < V && mjv <= 80) {
                const float x = PD_SUM4(redB, RD, 12 * tid + 8) + e_bfc;
                if (mjv < 80) p.mel[((int64_t)gc * S + (s - 1)) * 80 + mjv] = x; else p.stop[(int64_t)gc * S + (s - 1)] = x + stopc;
            }
            if (tid < 2 * V) {
#pragma unroll
                for (int q = 0; q < 4; ++q) l1hh[q] = PD_SUM4(redB, RD, 12 * cb + 4 * cu + q);
            }
        }
        // ------------------------------------------------------------ phase 4: layer-1 cells; the h0 / c0 parts of the next step
        {
            float2 hx, cx;
            do {
                const uint4 a = pd_load16(rs, gH0 + 2 * tid), c = pd_load16(rs, gC0 + 2 * tid);
                const bool ok = a.y == tag_now && a.w == tag_now && c.y == tag_now && c.w == tag_now;
                hx = make_float2(__uint_as_float(a.x), __uint_as_float(a.z));
                cx = make_float2(__uint_as_float(c.x), __uint_as_float(c.z));
                if (!poll.retry(ok)) break;
            } while (true);
            if (poll.gave_up()) return;
            PD_STAMP(11);
            float v[8 * V];
#pragma unroll
            for (int b = 0; b < V; ++b)
#pragma unroll
                for (int c = 0; c < 8; ++c) v[8 * b + c] = fmaf(wl1i[b][c].y, hx.y, wl1i[b][c].x * hx.x);
            PD_WSUM(v, 8 * V, red + rpar * 4 * RS, RS);
            __syncthreads();
            PD_STAMP(12);
            if (tid < 2 * V) {
                const float* r = red + rpar * 4 * RS;
                const float gi = PD_SUM4(r, RS, 8 * cb + 4 * cu + 0) + l1hh[0] + e_bl1[0], gf = PD_SUM4(r, RS, 8 * cb + 4 * cu + 1) + l1hh[1] + e_bl1[1];
                const float gg = PD_SUM4(r, RS, 8 * cb + 4 * cu + 2) + l1hh[2] + e_bl1[2], go = PD_SUM4(r, RS, 8 * cb + 4 * cu + 3) + l1hh[3] + e_bl1[3];
                c1 = pd_sigmoid(gf) * c1 + pd_sigmoid(gi) * pd_tanh(gg);
                const float hn = pd_sigmoid(go) * pd_tanh(c1);
                pd_publish<NG>(rsall, gH1 + 2 * cjv + cu, tag_now, hn);
                pd_publish<NG>(rsall, gC1 + 2 * cjv + cu, tag_now, c1);
            }
            rpar ^= 1;
            PD_STAMP(13);
            {   // deferred, for the next step: W_hh0 h0' (phase 3) and the h0 / c0 parts of q0, q1, qc (phase 1)
                float w[12 * V];
#pragma unroll
                for (int b = 0; b < V; ++b) {
#pragma unroll
                    for (int c = 0; c < 8; ++c) w[12 * b + c] = fmaf(wl0h[b][c].y, hx.y, wl0h[b][c].x * hx.x);
                    w[12 * b + 8] = fmaf(wq0[b][0].y, hx.y, wq0[b][0].x * hx.x);
                    w[12 * b + 9] = fmaf(wq0[b][1].y, hx.y, wq0[b][1].x * hx.x);
                    w[12 * b + 10] = fmaf(wcq0[b].y, cx.y, wcq0[b].x * cx.x);
                    w[12 * b + 11] = 0.f;
                }
                PD_WSUM(w, 12 * V, redC, RD);
            }
        }
    }
#undef PD_WSUM
#undef PD_SUM4
#undef PD_STAMP
}

// ------------------------------------------------------------------------------------------------------------------------------------------------------
// The decoder prologue's BiLSTM recurrence (decoder.py:386-389) of one or two clips in the same form: T dependent steps of one 2048 x 512 product per
// direction were T launches at 6.9 us (+ ten launches of layout glue around them).  The chip is split between the 2 NB (direction, clip) pairs, 256 / (2 NB)
// workgroups each; a workgroup owns 4 NB hidden units of its direction (16 NB gate columns, k = 2t, 2t + 1 per thread: 32 NB weight registers); h crosses the
// group as tagged granules (ping-pong by step parity: a step reads and rewrites it), the input gates W_ih x + b come from the prologue's GEMM, the cell
// state stays in the cell threads.  Outputs as the launch path leaves them: rnn_out (B, T, 1024), the final h of both directions as the decoder's initial
// hidden state (frag16), the final c side by side for E_C.
// ------------------------------------------------------------------------------------------------------------------------------------------------------
__host__ __device__ constexpr int pb_granules(int NB) { return 2 * NB * 2 * 512; }      // [pair][parity][512]
__host__ __device__ constexpr int pb_rstride(int NB) { return pb_granules(NB) + 520; }
template <int NB>
__global__ __launch_bounds__(PD_NT, 1) void pbilstm_kernel(const PBiP p) {
    constexpr int NP = 2 * NB, WPG = PD_WG / NP, U = 512 / WPG, C = 4 * U;      // pairs, workgroups per pair, hidden units and gate columns per workgroup
    __shared__ float red[2][4][C];
    __shared__ float pad[PD_LDS_MIN / 4];                    // one workgroup per CU
    const int j = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pr = j / WPG, jl = j - pr * WPG, d = pr & 1, b = pr >> 1;      // this workgroup's (direction, clip), its index in the group
    const int T = p.T;
    if (tid == 0) pad[0] = 0.f;
    u64* const X = p.xch;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(X + (int64_t)(j % PD_MAXREP) * pb_rstride(NB), 0, pb_granules(NB) * 8, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsall = __builtin_amdgcn_make_buffer_rsrc(X, 0, PD_MAXREP * pb_rstride(NB) * 8, 0x00020000);
    PdPoll poll(p.status);
    const float* const Wd = d == 0 ? p.Whh0 : p.Whh1;
    float2 w[C];
#pragma unroll
    for (int c = 0; c < C; ++c) w[c] = pd_w2(Wd, 32, C * jl + c, 2 * tid);      // packed rows are (unit, gate): a workgroup's units are C adjacent rows
    const int cu = tid;                                      // cell thread (tid < U): unit U jl + cu
    const int unit = U * jl + (cu < U ? cu : U - 1);
    float cst = p.s_e[(int64_t)b * 512 + unit];              // h0 = c0 = the encoder-site embedding (decoder.py:386)
    auto publish = [&](int par, unsigned tag, float v) {
        pd_u2 x; x.x = __float_as_uint(v); x.y = tag;
#pragma unroll
        for (int r = 0; r < PD_MAXREP; ++r) __builtin_amdgcn_raw_buffer_store_b64(x, rsall, ((pr * 2 + par) * 512 + unit) * 8, r * pb_rstride(NB) * 8, 16);
    };
    if (tid < U) publish(0, 1u, cst);
    int rpar = 0;
    float hlast = cst;
    for (int s = 0; s < T; ++s) {
        const int t = d == 0 ? s : T - 1 - s;
        // this step's input gates (bias folded in by the GEMM), requested before the poll
        float pre[4];
        if (tid < U) {
            const float* g = p.gin + ((int64_t)b * T + t) * 4096 + d * 2048 + unit;
#pragma unroll
            for (int q = 0; q < 4; ++q) pre[q] = g[q * 512];
        }
        float2 h;
        do {
            const uint4 a = pd_load16(rs, (pr * 2 + (s & 1)) * 512 + 2 * tid);
            const bool ok = a.y == (unsigned)s + 1u && a.w == (unsigned)s + 1u;
            h = make_float2(__uint_as_float(a.x), __uint_as_float(a.z));
            if (!poll.retry(ok)) break;
        } while (true);
        if (poll.gave_up()) return;
        float v[C];
#pragma unroll
        for (int c = 0; c < C; ++c) v[c] = fmaf(w[c].y, h.y, w[c].x * h.x);
        pd_wave_sum_multi<C>(v, lane);
        if (lane < 4) {
#pragma unroll
            for (int i = 0; i < C / 4; ++i) red[rpar][wave][4 * i + lane] = v[i];
        }
        __syncthreads();
        if (tid < U) {
            float gsum[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) gsum[q] = ((red[rpar][0][4 * cu + q] + red[rpar][1][4 * cu + q]) + (red[rpar][2][4 * cu + q] + red[rpar][3][4 * cu + q])) + pre[q];
            cst = sigmoidf_(gsum[1]) * cst + sigmoidf_(gsum[0]) * tanhf(gsum[2]);
            hlast = sigmoidf_(gsum[3]) * tanhf(cst);
            publish((s + 1) & 1, (unsigned)s + 2u, hlast);
            p.rnn[((int64_t)b * T + t) * 1024 + d * 512 + unit] = hlast;
        }
        rpar ^= 1;
    }
    if (tid < U) {      // finals: forward -> decoder layer 0, backward -> layer 1 (decoder.py:398-399); cells side by side for E_C
        p.h_state[(int64_t)d * ((p.B + 15) & ~15) * 512 + frag16_index(b, unit, 512)] = hlast;
        p.cellcat[(int64_t)b * 1024 + d * 512 + unit] = cst;
    }
}

int64_t pbilstm_ws_bytes() { return (int64_t)pb_rstride(2) * 8 * PD_MAXREP + 256; }
bool pbilstm_supported(int B, int T) { return B >= 1 && B <= 2 && T >= 1 && T <= 300; }
// a launch whose workgroups gave up (no fresh granule for PD_GIVE_UP_TICKS: the chip was not theirs) must not hand back plausible numbers: every output
// of the launch is overwritten with NaN and the process-wide count of such launches (pinned host memory, readable without a synchronize:
// l2s_persist_timeouts) goes up by one - the next persistent launch of the process then fails with an error instead of queueing behind a wedged device
__global__ void pdecode_guard_kernel(const unsigned* status, float* a, float* b, float* c, int na, int nb, int nc, unsigned* timeouts) {
    if (__hip_atomic_load(status, PD_RLX) == 0u) return;
    const float nan = __uint_as_float(0x7fc00000u);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < na; i += gridDim.x * blockDim.x) a[i] = nan;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nb; i += gridDim.x * blockDim.x) b[i] = nan;
    if (c) for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nc; i += gridDim.x * blockDim.x) c[i] = nan;
    if (timeouts && blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_fetch_add(timeouts, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

int64_t pdecode_ws_bytes(int) { return (int64_t)pd_rstride(2) * 8 * PD_MAXREP + 256; }      // laid out for two clips per launch
bool pdecode_supported(int B, int T, int m) {
    return B >= 1 && B <= PD_MAXB && T >= 1 && T <= PD_MAXT && m >= 1 && m <= PD_MAXM && pd_lds_floats(2, T, m) * 4 <= PD_LDS_MAX;
}

// every persistent launch needs all its workgroups resident at once: two of them in flight on different streams could each hold half of the chip and
// wait for the other half for ever, so they are chained through one event per device (a launch waits for the previous persistent launch on its device)
struct PdDevice {
    bool init = false;
    int cus = 0;
    bool resident = false;        // every persistent kernel fits one workgroup per compute unit AND nothing in the environment takes compute units away
    hipEvent_t ev = nullptr;
    unsigned* timeouts = nullptr; // pinned, device-visible word of THIS device: its launches whose workgroups gave up (pdecode_guard_kernel)
    unsigned seen = 0;            // ... of which the gate has already reported (pdecode_gate)
    unsigned armed_at = 0;        // the count when the persistent forms were last (re-)armed on this device: they are off while *timeouts != armed_at
};
constexpr int PD_MAX_DEVICES = 64;
static std::mutex g_pd_mu;
static PdDevice g_pd_dev[PD_MAX_DEVICES];
static unsigned long long* g_pd_ts = nullptr;
static int g_pd_ts_step = 0;

#ifdef L2S_DIAG
void pdecode_set_timeline(unsigned long long* ts, int step) { g_pd_ts = ts; g_pd_ts_step = step; }
#endif

int pdecode_timeouts() {      // process-wide: the sum over the devices this process has used
    std::lock_guard<std::mutex> lock(g_pd_mu);
    unsigned n = 0;
    for (const PdDevice& d : g_pd_dev) if (d.timeouts) n += __atomic_load_n(d.timeouts, __ATOMIC_RELAXED);
    return (int)n;
}

template <typename K>
static bool pd_fits(K kernel, int lds) {
    int nb = 0;
    return hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kernel, PD_NT, (size_t)lds) == hipSuccess && nb >= 1;
}

// the current device's entry (g_pd_mu held); nullptr when the device cannot be queried
static PdDevice* pd_device_locked() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= PD_MAX_DEVICES) return nullptr;
    PdDevice& d = g_pd_dev[dev];
    if (!d.init) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return nullptr;
        if (hipEventCreateWithFlags(&d.ev, hipEventDisableTiming) != hipSuccess) return nullptr;
        if (hipHostMalloc(reinterpret_cast<void**>(&d.timeouts), 64, hipHostMallocMapped | hipHostMallocPortable) != hipSuccess) d.timeouts = nullptr;
        else *d.timeouts = 0u;
        d.cus = prop.multiProcessorCount;
        // Co-residency, checked up front: the workgroups spin on each other, so a launch is only made where all of them can be resident at once -
        // 256 compute units, one workgroup of every persistent kernel fits a compute unit at the largest LDS request (the occupancy query), and no
        // compute-unit mask in the environment (the device then still reports 256 but hands out fewer).  What cannot be seen from here - another
        // process on the device - is bounded by the kernels' own 2 s give-up and reported through the guard kernel.
        const bool masked = std::getenv("HSA_CU_MASK") || std::getenv("ROC_GLOBAL_CU_MASK");
        bool fits = d.cus >= PD_WG && !masked && d.timeouts;
        if (fits) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(pdecode_kernel<1, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, PD_LDS_MAX);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(pdecode_kernel<2, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, PD_LDS_MAX);
            fits = pd_fits(pdecode_kernel<1, 2>, PD_LDS_MAX) && pd_fits(pdecode_kernel<2, 2>, PD_LDS_MAX) && pd_fits(pbilstm_kernel<1>, 0) && pd_fits(pbilstm_kernel<2>, 0);
        }
        d.resident = fits;
        d.init = true;
    }
    return &d;
}

// the persistent forms need every workgroup resident at once (above): callers fall back to the launch path where that cannot be promised, and after a
// launch on this device has timed out (the device is evidently shared) - per device, until the caller re-arms it (pdecode_rearm)
static bool pd_armed(const PdDevice* d) { return d && d->resident && __atomic_load_n(d->timeouts, __ATOMIC_RELAXED) == d->armed_at; }
bool pdecode_device_ok() {
    std::lock_guard<std::mutex> lock(g_pd_mu);
    return pd_armed(pd_device_locked());
}

// The gate every persistent-eligible call goes through (decode_run / prologue_run): 1 = take the persistent form, 0 = take the launch path, -1 = a
// persistent launch on this device gave up since the last call through the gate - its outputs are NaN - and THIS call fails once with that error
// (l2s_last_error); the calls after it take the launch path until the device is re-armed.  The count sits in pinned host memory and is written by
// the guard kernel at the end of the timed-out launch, so a caller that has synchronized with that launch is certain to see it here.
int pdecode_gate() {
    std::lock_guard<std::mutex> lock(g_pd_mu);
    PdDevice* const d = pd_device_locked();
    if (!d || !d->resident) return 0;
    const unsigned n = __atomic_load_n(d->timeouts, __ATOMIC_RELAXED);
    if (n != d->seen) {
        d->seen = n;
        set_error("an earlier persistent launch on this device gave up after 2 s without progress (its workgroups were not all resident: shared or CU-masked "
                  "device); its outputs were overwritten with NaN; further calls take the launch-per-phase path (re-arm: set option persist_decode again)");
        return -1;
    }
    return n == d->armed_at ? 1 : 0;
}

// option "persist_decode" set to a positive value: the caller asks for the persistent forms (again) - the current device's time-outs so far are
// forgiven (a transient co-tenant need not switch the latency path off for the rest of the process); the process-wide count keeps counting
void pdecode_rearm() {
    std::lock_guard<std::mutex> lock(g_pd_mu);
    PdDevice* const d = pd_device_locked();
    if (d && d->timeouts) d->armed_at = __atomic_load_n(d->timeouts, __ATOMIC_RELAXED);      // `seen` stays: a time-out nobody has been told about is still reported by the gate
}

int launch_pdecode(const PDecP& p, void* ws, int64_t ws_bytes, hipStream_t s) {
    L2S_REQUIRE(pdecode_supported(p.B, p.T, p.m), "persistent decode: <= 4 clips of <= 32 frames whose values fit the LDS");
    L2S_REQUIRE(ws && ws_bytes >= pdecode_ws_bytes(2), "persistent decode: exchange buffer too small");
    std::lock_guard<std::mutex> lock(g_pd_mu);
    PdDevice* const dv = pd_device_locked();
    L2S_REQUIRE(pd_armed(dv), "persistent decode needs 256 compute units, one resident workgroup each (none masked), and no timed-out launch since the device was armed");
    const int lds = std::max(pd_lds_floats(2, p.T, p.m) * 4, PD_LDS_MIN);
    L2S_CHECK_HIP(hipStreamWaitEvent(s, dv->ev, 0));      // a never-recorded event is complete
    ProfScope ps("decode_persistent", s);
    // clips two at a time (three or four clips: two launches one after the other - still shorter than 300 x four launches)
    for (int b0 = 0; b0 < p.B; b0 += 2) {
        const int n = p.B - b0 >= 2 ? 2 : 1;
        PDecP q = p;
        q.b0 = b0;
        q.xch = reinterpret_cast<u64*>(ws);
        q.status = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(ws) + (int64_t)pd_rstride(2) * 8 * PD_MAXREP);
        q.ts = g_pd_ts; q.ts_step = g_pd_ts_step;
        L2S_CHECK_HIP(hipMemsetAsync(ws, 0, (size_t)pdecode_ws_bytes(2), s));      // tags and the status word start at zero EVERY launch
        // 128 workgroups per clip, each standing for two of the one-per-CU layout: one clip leaves half the chip idle and is FASTER for it (7.7 against
        // 9.7 us per step on 256 workgroups: an edge among 128 workgroups costs ~1.2 us, among 256 ~1.8; 64 workgroups of four: 12.2, the weights no
        // longer fit the registers)
        // test hook, diagnostic build only (libl2s_diag.so; tests/test_persist_timeout.py, its own process): one workgroup short, so that the others wait
        // for granules that never come and the give-up path runs - 2 s, NaN outputs, l2s_persist_timeouts() = 1
#ifdef L2S_DIAG
        static const int starve = std::getenv("L2S_TEST_PDECODE_STARVE") ? 1 : 0;
#else
        constexpr int starve = 0;
#endif
        if (n == 1) hipLaunchKernelGGL((pdecode_kernel<1, 2>), dim3(PD_WG / 2 - starve), dim3(PD_NT), lds, s, q);
        else hipLaunchKernelGGL((pdecode_kernel<2, 2>), dim3(PD_WG - starve), dim3(PD_NT), lds, s, q);
        hipLaunchKernelGGL(pdecode_guard_kernel, dim3(8), dim3(256), 0, s, q.status, q.mel, q.stop, q.attn, q.B * q.S * 80, q.B * q.S, q.attn ? q.B * q.S * q.T : 0, dv->timeouts);
        L2S_CHECK_HIP(hipGetLastError());
    }
    L2S_CHECK_HIP(hipEventRecord(dv->ev, s));
    return 0;
}

int launch_pbilstm(const PBiP& p, void* ws, int64_t ws_bytes, hipStream_t s) {
    L2S_REQUIRE(pbilstm_supported(p.B, p.T), "persistent BiLSTM: one or two clips");
    L2S_REQUIRE(ws && ws_bytes >= pbilstm_ws_bytes(), "persistent BiLSTM: exchange buffer too small");
    std::lock_guard<std::mutex> lock(g_pd_mu);
    PdDevice* const dv = pd_device_locked();
    L2S_REQUIRE(pd_armed(dv), "persistent BiLSTM needs 256 compute units, one resident workgroup each (none masked), and no timed-out launch since the device was armed");
    PBiP q = p;
    q.xch = reinterpret_cast<u64*>(ws);
    q.status = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(ws) + (int64_t)pb_rstride(2) * 8 * PD_MAXREP);
    L2S_CHECK_HIP(hipMemsetAsync(ws, 0, (size_t)pbilstm_ws_bytes(), s));
    // the frag16 state rows past B stay zero, as the launch path leaves them
    L2S_CHECK_HIP(hipMemsetAsync(q.h_state, 0, sizeof(float) * 2 * ((p.B + 15) & ~15) * 512, s));
    L2S_CHECK_HIP(hipStreamWaitEvent(s, dv->ev, 0));
    ProfScope ps("bilstm_persistent", s);
    if (p.B == 1) hipLaunchKernelGGL(pbilstm_kernel<1>, dim3(PD_WG), dim3(PD_NT), 0, s, q);
    else hipLaunchKernelGGL(pbilstm_kernel<2>, dim3(PD_WG), dim3(PD_NT), 0, s, q);
    hipLaunchKernelGGL(pdecode_guard_kernel, dim3(8), dim3(256), 0, s, q.status, q.rnn, q.cellcat, q.h_state, p.B * p.T * 1024, p.B * 1024, 2 * ((p.B + 15) & ~15) * 512, dv->timeouts);
    L2S_CHECK_HIP(hipGetLastError());
    L2S_CHECK_HIP(hipEventRecord(dv->ev, s));
    return 0;
}

}  // namespace l2s
