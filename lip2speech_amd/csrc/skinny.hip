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
#include "skinny_dev.h"

#include <algorithm>

namespace l2s {

// ---- block-stamp log (libl2s_diag.so only; include/l2s_diag.h l2s_op_stamp_log): thread 0 of every block of the decode step's kernels appends
// {entry, exit, tag} on the 100 MHz constant clock - the overlap proof of profiles/ that does not depend on the host's clock or on the profiler
#ifdef L2S_DIAG
__device__ unsigned long long* d_stamp_log = nullptr;
__device__ unsigned long long d_stamp_cap = 0;
struct BlockStamp {
    unsigned long long t0 = 0;
    unsigned long long* log = nullptr;
    __device__ __forceinline__ void begin() { log = d_stamp_log; if (log && threadIdx.x == 0) t0 = wall_clock64(); }
    __device__ __forceinline__ void end(unsigned kind, const void* key) {
        if (!log) return;
        __syncthreads();                                   // the block's last wave is through (its stores may still drain)
        if (threadIdx.x == 0) {
            const unsigned long long t1 = wall_clock64();
            const unsigned long long slot = atomicAdd(log, 1ull);
            if (slot < d_stamp_cap) {
                unsigned long long* r = log + 1 + 3 * slot;
                r[0] = t0; r[1] = t1; r[2] = ((unsigned long long)kind << 60) | ((unsigned long long)(uintptr_t)key & 0xFFFFFFFFFFFFull);
            }
        }
    }
};
int set_stamp_log(unsigned long long* log, long long cap) {
    const unsigned long long c = cap > 0 ? (unsigned long long)cap : 0ull;
    if (hipMemcpyToSymbol(HIP_SYMBOL(d_stamp_cap), &c, sizeof(c)) != hipSuccess) return 1;
    if (hipMemcpyToSymbol(HIP_SYMBOL(d_stamp_log), &log, sizeof(log)) != hipSuccess) return 1;
    return 0;
}
#define L2S_BLOCK_STAMP_BEGIN() BlockStamp bs_; bs_.begin()
#define L2S_BLOCK_STAMP_END(kind, key) bs_.end(kind, key)
#else
#define L2S_BLOCK_STAMP_BEGIN() do {} while (0)
#define L2S_BLOCK_STAMP_END(kind, key) do {} while (0)
#endif

// segment layouts with their own instance (chunks of 16 columns per segment): the decode step's and the BiLSTM's operand shapes
//   1: [32]            K = 512   fc_out+stop, prenet1∘fc_out, BiLSTM / speaker LSTM recurrences
//   2: [32 | 32]       K = 1024  LSTM1 on [h0' | h1], Q on [h0 | h1], content Q on [c0 | c1]
//   3: [16|16|32|32]   K = 1536  LSTM0 on [content | prenet | a.v | h0] (attention_proj folded in)
//   4: [16]            K = 256   prenet layers
//   5: [16|16|16|32]   K = 1280  LSTM0 on [content | prenet | a.V' | h0] (attention_proj hoisted into the prologue)
//   6: [16|16+16|32]   K = 1024  LSTM0 on [content | prenet + a.V' | h0]: segment 1 summed from two sources by the loader (SkinnyP::a_sum)
static int skinny_layout_of(const SkinnyP& p) {
    const int n0 = p.seg[0].nchunks, n1 = p.seg[1].nchunks, n2 = p.seg[2].nchunks, n3 = p.seg[3].nchunks;
    if (n0 == 32 && n1 == 0 && n2 == 0 && n3 == 0) return 1;
    if (n0 == 32 && n1 == 32 && n2 == 0 && n3 == 0) return 2;
    if (n0 == 16 && n1 == 16 && n2 == 32 && n3 == 32) return 3;
    if (n0 == 16 && n1 == 0 && n2 == 0 && n3 == 0) return 4;
    if (n0 == 16 && n1 == 16 && n2 == 16 && n3 == 32) return 5;
    if (n0 == 16 && n1 == 16 && n2 == 32 && n3 == 0 && p.a_sum) return 6;
    return 0;
}

// MAXC = chunks per wave the instance is unrolled for (K <= 128 * MAXC): the 24 operand registers per chunk pair are what sets the
// kernel's VGPR count, so launches whose longest K is 512 / 1024 get their own, smaller instances
template <int MAXC>
__global__ __launch_bounds__(512) void skinny_kernel(const SkinnyBatch batch) {
    __shared__ float red[SK_RED_FLOATS];
    const int g = blockIdx.z;
    const SkinnyP& p = batch.p[g];
    const int lay = p.layout;                       // block-uniform
    if (MAXC == 12 && lay == 3) skinny_block<false, 12, false, SegLay<16, 16, 32, 32>>(p, blockIdx.x, blockIdx.y, red, batch.ntiles[g]);
    else if (MAXC >= 8 && lay == 2) skinny_block<false, 8, false, SegLay<32, 32, 0, 0>>(p, blockIdx.x, blockIdx.y, red, batch.ntiles[g]);
    else if (MAXC >= 4 && lay == 1) skinny_block<false, 4, false, SegLay<32, 0, 0, 0>>(p, blockIdx.x, blockIdx.y, red, batch.ntiles[g]);
    else if (lay == 4) skinny_block<false, 2, false, SegLay<16, 0, 0, 0>>(p, blockIdx.x, blockIdx.y, red, batch.ntiles[g]);
    else skinny_block<false, MAXC>(p, blockIdx.x, blockIdx.y, red, batch.ntiles[g]);
}
// instances with their operand loads in NB batches (options "skinny_split" = NB for K <= 1536, "skinny_split8" = NB for K <= 1024)
template <int MAXC, int NB, int WPS>
__global__ __launch_bounds__(512, WPS) void skinny_kernel_split(const SkinnyBatch batch) {
    __shared__ float red[SK_RED_FLOATS];
    const int g = blockIdx.z;
    skinny_block<false, MAXC, false, SegRuntime, NB>(batch.p[g], blockIdx.x, blockIdx.y, red, batch.ntiles[g]);
}

// register-blocked instances for many batch rows (grouped decode, skinny_dev.h skinny_block_rc): RT x CT tiles of 16x16 per block
template <int RT, int CT, int MAXC, int JB, int WPS, int DEPTH = 2>
__global__ __launch_bounds__(512, WPS) void skinny_rc_kernel(const SkinnyBatch batch, int mts) {
    __shared__ float red[SkRc<RT, CT>::RED_FLOATS];
    const int g = blockIdx.z;
    skinny_block_rc<RT, CT, MAXC, JB, DEPTH>(batch.p[g], blockIdx.x, blockIdx.y, red, batch.ntiles[g], mts);
}
// the straight-line form (skinny_dev.h skinny_block_rcs) for launches whose groups all have one of the decode step's K layouts
template <int LAYID> struct SkLay;
template <> struct SkLay<1> { using T = SegLay<32, 0, 0, 0>; };
template <> struct SkLay<2> { using T = SegLay<32, 32, 0, 0>; };
template <> struct SkLay<3> { using T = SegLay<16, 16, 32, 32>; };
template <> struct SkLay<5> { using T = SegLay<16, 16, 16, 32>; };
template <> struct SkLay<6> { using T = SegLay<16, 16, 32, 0, true>; };
template <int RT, int CT, int LAYID, int DEPTH, int WPS, bool IS_LSTM>
__global__ __launch_bounds__(512, WPS) void skinny_rcs_kernel(const SkinnyBatch batch, int mts) {
    __shared__ float red[SkRc<RT, CT>::RED_FLOATS];
    const int g = blockIdx.z;
    skinny_block_rcs<RT, CT, typename SkLay<LAYID>::T, DEPTH, IS_LSTM>(batch.p[g], blockIdx.x, blockIdx.y, red, batch.ntiles[g], mts);
}
// four real waves per block (one per SIMD, each playing two K slices): skinny_block_rcs<..., NW = 4>
template <int RT, int CT, int LAYID, int DEPTH, bool IS_LSTM>
__global__ __launch_bounds__(256, 1) void skinny_rc4_kernel(const SkinnyBatch batch, int mts) {
    __shared__ float red[SkRc<RT, CT>::RED_FLOATS];
    const int g = blockIdx.z;
    skinny_block_rcs<RT, CT, typename SkLay<LAYID>::T, DEPTH, IS_LSTM, false, 4>(batch.p[g], blockIdx.x, blockIdx.y, red, batch.ntiles[g], mts);
}
// the same LSTM blocks on the bf16 matrix cores (skinny_block_rcs<..., X3>): every group of the launch carries pre-split weight planes (SkinnyP::W3)
template <int RT, int CT, int LAYID, int DEPTH>
__global__ __launch_bounds__(256, 1) void skinny_rc4x_kernel(const SkinnyBatch batch, int mts) {
    __shared__ float red[SkRc<RT, CT>::RED_FLOATS];
    const int g = blockIdx.z;
    skinny_block_rcs<RT, CT, typename SkLay<LAYID>::T, DEPTH, true, false, 4, true>(batch.p[g], blockIdx.x, blockIdx.y, red, batch.ntiles[g], mts);
}
// Half a compute unit per block (option "lstm_x3" = 3): four waves of at most 256 registers - each wave plays its two K slices one after the other
// on ONE set of accumulators (skinny_block_rcs<.., SEQ>: same partial sums, same bits as the other forms) - so that two blocks share a CU: of one
// launch (more than 256 blocks) or of two launch chains on different streams, whose kernels then OVERLAP instead of queueing (a block alone leaves
// its CU's load path and matrix pipe idle two thirds of its lifetime: first-operand latency, reduction, store drain, launch boundary)
template <int RT, int CT, int LAYID, int DEPTH>
__global__ __launch_bounds__(256, 2) void skinny_rc4h_kernel(const SkinnyBatch batch, int mts) {
    __shared__ float red[SkRc<RT, CT>::RED_FLOATS];
    const int g = blockIdx.z;
    L2S_BLOCK_STAMP_BEGIN();
    skinny_block_rcs<RT, CT, typename SkLay<LAYID>::T, DEPTH, true, false, 4, true, false, true>(batch.p[g], blockIdx.x, blockIdx.y, red, batch.ntiles[g], mts);
    L2S_BLOCK_STAMP_END(1u, batch.p[g].c_out);
}
template <int RT, int CT, int LAYID, int DEPTH>
__global__ __launch_bounds__(512, 1) void skinny_rc8x_kernel(const SkinnyBatch batch, int mts) {      // the same on eight waves: the two waves of a SIMD alternate split VALU and MFMAs
    __shared__ float red[SkRc<RT, CT>::RED_FLOATS];
    const int g = blockIdx.z;
    L2S_BLOCK_STAMP_BEGIN();
    skinny_block_rcs<RT, CT, typename SkLay<LAYID>::T, DEPTH, true, false, 8, true>(batch.p[g], blockIdx.x, blockIdx.y, red, batch.ntiles[g], mts);
    L2S_BLOCK_STAMP_END(1u, batch.p[g].c_out);
}
// weight planes of the split-bf16 LSTM blocks from the packed fp32 fragments ([tile][chunk][lane] float4): chunk pair ip of K slice s = chunks
// (s + 16 ip, s + 16 ip + 8); out[tile][s][ip][plane][lane] = 8 bf16 (the lane's quad of the first chunk, then of the second)
__global__ __launch_bounds__(256) void skx_planes_kernel(const float4* __restrict__ Wf, int ntiles, int NC, uint4* __restrict__ out) {
    const int NPI = NC / 16;
    const int64_t total = (int64_t)ntiles * 8 * NPI * 64;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int lane = (int)(i & 63);
        int64_t r = i >> 6;
        const int ip = (int)(r % NPI); r /= NPI;
        const int sl = (int)(r & 7);
        const int64_t tile = r >> 3;
        const float4 f0 = Wf[(tile * NC + sl + 16 * ip) * 64 + lane], f1 = Wf[(tile * NC + sl + 16 * ip + 8) * 64 + lane];
        uint2 h0, m0, l0, h1, m1, l1;
        skx_split4(f0, h0, m0, l0); skx_split4(f1, h1, m1, l1);
        uint4* o = out + (((tile * 8 + sl) * NPI + ip) * 3) * 64 + lane;
        o[0] = make_uint4(h0.x, h0.y, h1.x, h1.y); o[64] = make_uint4(m0.x, m0.y, m1.x, m1.y); o[128] = make_uint4(l0.x, l0.y, l1.x, l1.y);
    }
}
int launch_skx_planes(const float* Wf, int ntiles, int K, void* out, hipStream_t s) {
    L2S_REQUIRE(Wf && out && K % 256 == 0, "split-bf16 LSTM planes: K must be a multiple of 256 (whole chunk pairs per slice)");
    const int64_t total = (int64_t)ntiles * 8 * (K / 256) * 64;
    hipLaunchKernelGGL(skx_planes_kernel, dim3((unsigned)std::min<int64_t>((total + 255) / 256, 4096)), dim3(256), 0, s, reinterpret_cast<const float4*>(Wf), ntiles, K / 16,
                       reinterpret_cast<uint4*>(out));
    L2S_CHECK_HIP(hipGetLastError());
    return 0;
}

template <int RT, int CT, int DEPTH>
static bool launch_rc4(const SkinnyBatch& bl, int lay, int kind, int maxt, int mts, hipStream_t s, int x3 = 0) {
    const dim3 grid((maxt + CT - 1) / CT, (mts + RT - 1) / RT, bl.count), blk(256);
    constexpr int HD = 3;       // operand slots in flight of the half-CU 4x2 form (four: its 256 registers spill)
    if constexpr (RT * CT >= 8) {       // the half-CU form exists for the 4x2 block; the smaller shapes already leave room beside them
        if (kind == 2 && x3 == 3 && (lay == 1 || lay == 2 || lay == 6)) {
            if (lay == 1) hipLaunchKernelGGL((skinny_rc4h_kernel<RT, CT, 1, HD>), grid, blk, 0, s, bl, mts);
            else if (lay == 2) hipLaunchKernelGGL((skinny_rc4h_kernel<RT, CT, 2, HD>), grid, blk, 0, s, bl, mts);
            else hipLaunchKernelGGL((skinny_rc4h_kernel<RT, CT, 6, HD>), grid, blk, 0, s, bl, mts);
            return true;
        }
    }
    if (kind == 2 && x3 >= 2 && (lay == 1 || lay == 2 || lay == 6)) {
        if (lay == 1) hipLaunchKernelGGL((skinny_rc8x_kernel<RT, CT, 1, DEPTH>), grid, dim3(512), 0, s, bl, mts);      // the BiLSTM recurrence (K = 512)
        else if (lay == 2) hipLaunchKernelGGL((skinny_rc8x_kernel<RT, CT, 2, DEPTH>), grid, dim3(512), 0, s, bl, mts);
        else hipLaunchKernelGGL((skinny_rc8x_kernel<RT, CT, 6, DEPTH>), grid, dim3(512), 0, s, bl, mts);
        return true;
    }
    if (kind == 2 && x3 && (lay == 1 || lay == 2 || lay == 6)) {
        if (lay == 1) hipLaunchKernelGGL((skinny_rc4x_kernel<RT, CT, 1, DEPTH>), grid, blk, 0, s, bl, mts);
        else if (lay == 2) hipLaunchKernelGGL((skinny_rc4x_kernel<RT, CT, 2, DEPTH>), grid, blk, 0, s, bl, mts);
        else hipLaunchKernelGGL((skinny_rc4x_kernel<RT, CT, 6, DEPTH>), grid, blk, 0, s, bl, mts);
        return true;
    }
    if (kind == 2) {
        if (lay == 1) hipLaunchKernelGGL((skinny_rc4_kernel<RT, CT, 1, DEPTH, true>), grid, blk, 0, s, bl, mts);
        else if (lay == 2) hipLaunchKernelGGL((skinny_rc4_kernel<RT, CT, 2, DEPTH, true>), grid, blk, 0, s, bl, mts);
        else if (lay == 3) hipLaunchKernelGGL((skinny_rc4_kernel<RT, CT, 3, DEPTH, true>), grid, blk, 0, s, bl, mts);
        else if (lay == 5) hipLaunchKernelGGL((skinny_rc4_kernel<RT, CT, 5, DEPTH, true>), grid, blk, 0, s, bl, mts);
        else if (lay == 6) hipLaunchKernelGGL((skinny_rc4_kernel<RT, CT, 6, DEPTH, true>), grid, blk, 0, s, bl, mts);
        else return false;
    } else if (kind == 1) {
        if (lay == 1) hipLaunchKernelGGL((skinny_rc4_kernel<RT, CT, 1, DEPTH, false>), grid, blk, 0, s, bl, mts);
        else if (lay == 2) hipLaunchKernelGGL((skinny_rc4_kernel<RT, CT, 2, DEPTH, false>), grid, blk, 0, s, bl, mts);
        else return false;
    } else return false;
    return true;
}
#ifdef L2S_DIAG
// measurement build of the 4x2 LSTM form: thread 0 of every block stamps its phases (8 x 64-bit per block, tools/skinny_timeline.py ROWS=256)
template <int LAYID, bool X3 = false, int NW = 4>
__global__ __launch_bounds__(64 * NW, 1) void skinny_rcs_timed_kernel(const SkinnyBatch batch, int mts, unsigned long long* ts) {
    __shared__ float red[SkRc<4, 2>::RED_FLOATS];
    const int blk = blockIdx.y * gridDim.x + blockIdx.x;
    skinny_block_rcs<4, 2, typename SkLay<LAYID>::T, 4, true, true, NW, X3>(batch.p[0], blockIdx.x, blockIdx.y, red, batch.ntiles[0], mts, ts + (int64_t)blk * 8);
}
#endif
// lay: the K layout every group shares; kind: 2 = every group is an LSTM cell, 1 = none is
template <int RT, int CT, int DEPTH>
static bool launch_rcs(const SkinnyBatch& bl, int lay, int kind, int maxt, int mts, hipStream_t s) {
    const dim3 grid((maxt + CT - 1) / CT, (mts + RT - 1) / RT, bl.count), blk(512);
    if (kind == 2) {
        if (lay == 1) hipLaunchKernelGGL((skinny_rcs_kernel<RT, CT, 1, DEPTH, 2, true>), grid, blk, 0, s, bl, mts);
        else if (lay == 2) hipLaunchKernelGGL((skinny_rcs_kernel<RT, CT, 2, DEPTH, 2, true>), grid, blk, 0, s, bl, mts);
        else if (lay == 3) hipLaunchKernelGGL((skinny_rcs_kernel<RT, CT, 3, DEPTH, 2, true>), grid, blk, 0, s, bl, mts);
        else if (lay == 5) hipLaunchKernelGGL((skinny_rcs_kernel<RT, CT, 5, DEPTH, 2, true>), grid, blk, 0, s, bl, mts);
        else return false;
    } else if (kind == 1) {
        if (lay == 1) hipLaunchKernelGGL((skinny_rcs_kernel<RT, CT, 1, DEPTH, 2, false>), grid, blk, 0, s, bl, mts);
        else if (lay == 2) hipLaunchKernelGGL((skinny_rcs_kernel<RT, CT, 2, DEPTH, 2, false>), grid, blk, 0, s, bl, mts);
        else return false;
    } else return false;
    return true;
}

template <int RT, int CT>
static void launch_rc(const SkinnyBatch& bl, int cls, int maxt, int mts, hipStream_t s, int rc_jb, int lay, int kind, int x3 = 0) {
    const dim3 grid((maxt + CT - 1) / CT, (mts + RT - 1) / RT, bl.count), blk(512);
    // every group in one of the decode step's K layouts and no forced batching: the straight-line form (exact waits, DEPTH - 1 chunks in flight
    // under every chunk's MFMAs)
    // Default ("skinny_rc_jb" = 0) where every group has one of the decode step's K layouts: FOUR waves per block, one per SIMD, each playing two of
    // the eight K slices, straight-line code with exact waits and four chunks in flight (skinny_block_rcs<.., NW = 4>).  With eight waves the two
    // waves of a SIMD do not share the matrix pipe evenly: the older one runs ahead, the younger follows in its shadow and finishes ~4 us later,
    // alone (stamped build, tools/skinny_timeline.py ROWS=256: block lifetime 14.3 us, kernel span 19.4 us at K = 1024); with one wave per SIMD the
    // pipe is 90 % busy through the K loop and every block ends within 0.2 us of the others (span 12.4 us).  Same bits.
    if (rc_jb == 0 && lay && launch_rc4<RT, CT, 4>(bl, lay, kind, maxt, mts, s, x3)) return;
    if (rc_jb == 28 && lay && launch_rcs<RT, CT, (RT * CT >= 8 ? 4 : 6)>(bl, lay, kind, maxt, mts, s)) return;      // the same straight-line form on eight waves
    if constexpr (RT * CT >= 8) {
        // the 4x2 form holds 6 fragments per chunk and runs alone on its CU.  Default ("skinny_rc_jb" = 0): one-chunk operand batches, four in
        // flight - the registers of two two-chunk batches, 1.5x the latency tolerance: 20.2 -> 19.4 us per LSTM launch at 256 rows; five in flight
        // (15): 19.9; 2 / 4: the two-chunk batches, two in flight, of round 2.  (Two 2x2 blocks SHARING a CU - one-chunk batches, three in flight,
        // 118 VGPRs - are slower, 21.2 us: the same CU then pulls 786 KB instead of 590 KB; what bounds these launches is bytes per CU.)
        if (rc_jb == 2 || rc_jb == 4) {
            if (cls == 4) hipLaunchKernelGGL((skinny_rc_kernel<RT, CT, 4, 2, 2>), grid, blk, 0, s, bl, mts);
            else if (cls == 8) hipLaunchKernelGGL((skinny_rc_kernel<RT, CT, 8, 2, 2>), grid, blk, 0, s, bl, mts);
            else hipLaunchKernelGGL((skinny_rc_kernel<RT, CT, 12, 2, 2>), grid, blk, 0, s, bl, mts);
        } else if (rc_jb == 15) {
            if (cls == 4) hipLaunchKernelGGL((skinny_rc_kernel<RT, CT, 4, 1, 2, 4>), grid, blk, 0, s, bl, mts);
            else if (cls == 8) hipLaunchKernelGGL((skinny_rc_kernel<RT, CT, 8, 1, 2, 5>), grid, blk, 0, s, bl, mts);
            else hipLaunchKernelGGL((skinny_rc_kernel<RT, CT, 12, 1, 2, 5>), grid, blk, 0, s, bl, mts);
        } else {
            if (cls == 4) hipLaunchKernelGGL((skinny_rc_kernel<RT, CT, 4, 1, 2, 4>), grid, blk, 0, s, bl, mts);
            else if (cls == 8) hipLaunchKernelGGL((skinny_rc_kernel<RT, CT, 8, 1, 2, 4>), grid, blk, 0, s, bl, mts);
            else hipLaunchKernelGGL((skinny_rc_kernel<RT, CT, 12, 1, 2, 4>), grid, blk, 0, s, bl, mts);
        }
    } else if (rc_jb == 4) {
        if (cls == 4) hipLaunchKernelGGL((skinny_rc_kernel<RT, CT, 4, 4, 2>), grid, blk, 0, s, bl, mts);
        else if (cls == 8) hipLaunchKernelGGL((skinny_rc_kernel<RT, CT, 8, 4, 2>), grid, blk, 0, s, bl, mts);
        else hipLaunchKernelGGL((skinny_rc_kernel<RT, CT, 12, 4, 2>), grid, blk, 0, s, bl, mts);
    } else {
        if (cls == 4) hipLaunchKernelGGL((skinny_rc_kernel<RT, CT, 4, 2, 2>), grid, blk, 0, s, bl, mts);
        else if (cls == 8) hipLaunchKernelGGL((skinny_rc_kernel<RT, CT, 8, 2, 2>), grid, blk, 0, s, bl, mts);
        else hipLaunchKernelGGL((skinny_rc_kernel<RT, CT, 12, 2, 2>), grid, blk, 0, s, bl, mts);
    }
}

// ---- balanced flat launch for phases that carry several GEMM groups (the step's first phase: prenet1 o fc_out, Q, content Q, fc_out+stop).
// A (tiles x rows x groups) grid with ONE block shape leaves the chip unbalanced: at 256 rows the 2x2 grid has 280 live blocks for 256 CUs, so
// 24 CUs run two blocks - two K = 1024 blocks pull 524 KB through one CU's vector-memory path while others pull 131 KB - and the launch lasts as
// long as its slowest CU.  Here every group gets its OWN block shape, chosen so that all groups together have at most one block per CU and the
// largest block moves the fewest bytes (K = 1024 groups as 2x2 = 262 KB, K = 512 groups as 4x2 = 197 KB at 256 rows: 236 blocks, each alone on
// its CU); the blocks sit in one flat grid, longest first.  Per output element the arithmetic is unchanged (skinny_block_rc): same bits.
struct SkinnyFlat {
    int first[SKINNY_MAX_GROUP + 1];     // first[k] .. first[k+1]: the flat block range of the k-th longest group
    int gid[SKINNY_MAX_GROUP];           // which group (index into SkinnyBatch::p) that is
    int ncol[SKINNY_MAX_GROUP];          // by rank k: column blocks (ceil(tiles / CT))
    int wide[SKINNY_MAX_GROUP];          // by rank k: 1 = a K <= 1024 group (shape S8), 0 = a K <= 512 group (shape S4)
    int nrow[SKINNY_MAX_GROUP];          // by rank k: row blocks (ceil(mts / RT))
    int xcd[SKINNY_MAX_GROUP];           // by rank k: 1 = XCD-affine block -> tile map (below), 0 = row-major
};
// XCD-affine map ("flat_xcd"): workgroup b of a launch runs on XCD b mod 8, and every XCD has its own L2, which the kernel boundary invalidates - so
// what a launch pulls through the fabric is (weights) x (XCDs that see each weight tile) + (activations) x (XCDs that see each row).  Row-major order
// gives every XCD all rows and an eighth of the columns: weights once, activations EIGHT times (the first phase at 256 rows: 3.8 + 8 x 2.0 MB).  Here
// XCD x owns row half x & 1 and column quarter x >> 1 of the group: weights twice, activations four times (7.6 + 8.0 MB).  Needs the group's first
// block on a multiple of 8, an even number of row blocks and a multiple of four column blocks.
__device__ __forceinline__ void flat_xcd_map(int b, int local, int nc, int nr, int& tp, int& mg) {
    const int x = b & 7, j = local >> 3;
    const int ncq = nc >> 2, nrh = nr >> 1;
    mg = (x & 1) * nrh + j / ncq;
    tp = (x >> 1) * ncq + j % ncq;
}
// One instance per pair of shapes (S8 for the K <= 1024 groups, S4 for the K <= 512 groups; RT * 10 + CT): an instance that carries every
// shape is 60 KB of code and pays ~2.5 us of instruction fetch per launch.
template <int S8, int S4, int STATIC = 0, bool TIMED = false>
__global__ __launch_bounds__(STATIC == 2 ? 256 : 512, STATIC == 2 ? 1 : 2) void skinny_flat_kernel(const SkinnyBatch batch, const SkinnyFlat fl, int mts, unsigned long long* ts = nullptr) {
    constexpr int R8 = S8 / 10, C8 = S8 % 10, R4 = S4 / 10, C4 = S4 % 10;
    constexpr int RED = SkRc<R8, C8>::RED_FLOATS > SkRc<R4, C4>::RED_FLOATS ? SkRc<R8, C8>::RED_FLOATS : SkRc<R4, C4>::RED_FLOATS;
    __shared__ float red[RED];
    const int b = blockIdx.x;
    int k = 0;
#pragma unroll
    for (int i = 1; i < SKINNY_MAX_GROUP; ++i) k += (i < batch.count && b >= fl.first[i]) ? 1 : 0;
    const int g = fl.gid[k];
    const int local = b - fl.first[k], nc = fl.ncol[k];
    int tp = local % nc, mg = local / nc;
    if (fl.xcd[k]) flat_xcd_map(b, local, nc, fl.nrow[k], tp, mg);
    const SkinnyP& p = batch.p[g];
    L2S_BLOCK_STAMP_BEGIN();
    if constexpr (STATIC == 2) {  // every wide group is [h | h] (K = 1024), every narrow one [h] (K = 512): straight-line four-wave blocks (256 threads)
        if (fl.wide[k]) skinny_block_rcs<R8, C8, SegLay<32, 32, 0, 0>, 4, false, false, 4>(p, tp, mg, red, batch.ntiles[g], mts);
        else skinny_block_rcs<R4, C4, SegLay<32, 0, 0, 0>, 4, false, false, 4>(p, tp, mg, red, batch.ntiles[g], mts);
    } else if constexpr (STATIC == 1) {      // the same on eight waves
        unsigned long long* const tb = TIMED ? ts + (int64_t)b * 8 : nullptr;       // measurement build (l2s_op_flat_timeline): 8 stamps per block
        if (fl.wide[k]) skinny_block_rcs<R8, C8, SegLay<32, 32, 0, 0>, 8, false, TIMED>(p, tp, mg, red, batch.ntiles[g], mts, tb);
        else skinny_block_rcs<R4, C4, SegLay<32, 0, 0, 0>, 4, false, TIMED>(p, tp, mg, red, batch.ntiles[g], mts, tb);
    } else {
        if (fl.wide[k]) skinny_block_rc<R8, C8, 8, 2>(p, tp, mg, red, batch.ntiles[g], mts);
        else skinny_block_rc<R4, C4, 4, 2>(p, tp, mg, red, batch.ntiles[g], mts);
    }
    L2S_BLOCK_STAMP_END(2u, batch.p[0].out);
}
// choose the pair of shapes: at most `cap` blocks in total (one per CU) and the shortest longest block.  A block's time is modelled from the
// measurements of tools/time_step_phases.py as bytes / 36.5 GB/s (what one block streams through its CU's vector-memory path) plus its MFMA
// work at 350 GFLOP/s per CU (not overlapped: a 4x2 block of a K = 512 group has twice the matrix work of a 2x1 block of a K = 1024 group with
// the same bytes, and lasts longer).  Returns 0 when nothing fits (then the uniform grid runs), else S8 * 100 + S4.
static int plan_flat(const SkinnyBatch& bl, int mts, int cap, SkinnyFlat& fl, bool half = false, bool xcd = false) {
    static const int S8[2] = {21, 22}, S4[3] = {21, 22, 42};
    const int n = bl.count;
    auto shape_of = [&](int g, int s8, int s4) { return bl.p[g].K > 512 ? s8 : s4; };
    auto nblocks = [&](int g, int sh) { return ((bl.ntiles[g] + sh % 10 - 1) / (sh % 10)) * ((mts + sh / 10 - 1) / (sh / 10)); };
    auto nbytes = [&](int g, int sh) { return (int64_t)(16 * (sh / 10) + 16 * (sh % 10)) * bl.p[g].K * 4; };
    auto cost = [&](int g, int sh) { return (double)nbytes(g, sh) / 36.5e3 + 2.0 * 256 * (sh / 10) * (sh % 10) * bl.p[g].K / 350e3; };      // us
    int best8 = 0, best4 = 0;
    double best_max = -1;
    int64_t best_sum = 0;
    for (int s8 : S8)
        for (int s4 : S4) {
            if (half && s4 == 42) continue;      // the 4x2 four-wave block needs 291 registers: not a half-CU block
            int blocks = 0; double mx = 0; int64_t sum = 0;
            for (int g = 0; g < n; ++g) {
                const int sh = shape_of(g, s8, s4);
                blocks += nblocks(g, sh); mx = std::max(mx, cost(g, sh)); sum += nbytes(g, sh) * nblocks(g, sh);
            }
            if (blocks <= cap && (best_max < 0 || mx < best_max - 1e-9 || (mx < best_max + 1e-9 && sum < best_sum))) { best_max = mx; best_sum = sum; best8 = s8; best4 = s4; }
        }
    if (best_max < 0) return 0;
    // longest blocks first (the dispatcher hands blocks out in grid order)
    int order[SKINNY_MAX_GROUP] = {0, 1, 2, 3};
    std::stable_sort(order, order + n, [&](int a, int b) { return nbytes(a, shape_of(a, best8, best4)) > nbytes(b, shape_of(b, best8, best4)); });
    int pos = 0;
    for (int k = 0; k < n; ++k) {
        const int g = order[k], sh = shape_of(g, best8, best4);
        fl.first[k] = pos; fl.gid[k] = g;
        fl.ncol[k] = (bl.ntiles[g] + sh % 10 - 1) / (sh % 10);
        fl.wide[k] = bl.p[g].K > 512 ? 1 : 0;
        fl.nrow[k] = (mts + sh / 10 - 1) / (sh / 10);
        fl.xcd[k] = (xcd && pos % 8 == 0 && fl.nrow[k] % 2 == 0 && fl.ncol[k] % 4 == 0) ? 1 : 0;
        pos += nblocks(g, sh);
    }
    for (int k = n; k <= SKINNY_MAX_GROUP; ++k) fl.first[k] = pos;
    return best8 * 100 + best4;
}
#ifdef L2S_DIAG
static unsigned long long* g_flat_ts = nullptr;
void skinny_set_flat_timeline(unsigned long long* ts) { g_flat_ts = ts; }
#endif
template <int S8, int S4>
static void launch_flat(const SkinnyBatch& bl, const SkinnyFlat& fl, int mts, hipStream_t s, int stat) {
#ifdef L2S_DIAG
    if (g_flat_ts && stat == 1) {
        if constexpr (S8 == 22 && S4 == 42) { hipLaunchKernelGGL((skinny_flat_kernel<S8, S4, 1, true>), dim3(fl.first[SKINNY_MAX_GROUP]), dim3(512), 0, s, bl, fl, mts, g_flat_ts); return; }
    }
#endif
    if (stat == 2) hipLaunchKernelGGL((skinny_flat_kernel<S8, S4, 2>), dim3(fl.first[SKINNY_MAX_GROUP]), dim3(256), 0, s, bl, fl, mts, (unsigned long long*)nullptr);
    else if (stat == 1) hipLaunchKernelGGL((skinny_flat_kernel<S8, S4, 1>), dim3(fl.first[SKINNY_MAX_GROUP]), dim3(512), 0, s, bl, fl, mts, (unsigned long long*)nullptr);
    else hipLaunchKernelGGL((skinny_flat_kernel<S8, S4, 0>), dim3(fl.first[SKINNY_MAX_GROUP]), dim3(512), 0, s, bl, fl, mts, (unsigned long long*)nullptr);
}

#ifdef L2S_DIAG
// measurement build of the same kernel: every block's thread 0 stamps its phases (8 stamps per block, block index = (z*gridDim.y + y)*gridDim.x + x)
__global__ __launch_bounds__(512) void skinny_kernel_timed(const SkinnyBatch batch, unsigned long long* ts) {
    __shared__ float red[SK_RED_FLOATS];
    const int g = blockIdx.z;
    const int blk = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    const SkinnyP& p = batch.p[g];
    if (p.layout == 3) skinny_block<false, 12, true, SegLay<16, 16, 32, 32>>(p, blockIdx.x, blockIdx.y, red, batch.ntiles[g], nullptr, ts + (int64_t)blk * 8);
    else if (p.layout == 2) skinny_block<false, 8, true, SegLay<32, 32, 0, 0>>(p, blockIdx.x, blockIdx.y, red, batch.ntiles[g], nullptr, ts + (int64_t)blk * 8);
    else skinny_block<false, SK_MAXC, true>(p, blockIdx.x, blockIdx.y, red, batch.ntiles[g], nullptr, ts + (int64_t)blk * 8);
}
#endif
#ifdef L2S_DIAG
static unsigned long long* g_skinny_ts = nullptr;
#else
static constexpr unsigned long long* g_skinny_ts = nullptr;      // the product never launches a stamped build
#endif
// options (l2s_common.h Options): "skinny_static" = compile-time segment layouts: +2 % one batch at a time (23.6 -> 23.1 us/step), -3 % with four
// batches in flight (1.60 -> 1.56 M mel-frames/s), off by default; "skinny_sized" = instances sized for the launch's longest K;
// "skinny_split" / "skinny_split8" = operand loads of the K <= 1536 / K <= 1024 instance in this many batches
#ifdef L2S_DIAG
void skinny_set_timeline(unsigned long long* ts) { g_skinny_ts = ts; }
#endif

static bool skinny_shape_known(int shape) { return shape == 0 || shape == 11 || shape == 21 || shape == 22 || shape == 42; }
// a forced "skinny_rc" / "skinny_rc_multi" outside {11, 21, 22, 42} falls through to skinny_kernel<cls>, which knows nothing of SkinnyP::a_sum: the
// caller then keeps LSTM layer 0 on [content | prenet | o | h0] (K = 1280, hoist_vproj = 1)
bool skinny_sum_supported(const Options& o) { return o.rc_jb == 0 && g_skinny_ts == nullptr && skinny_shape_known(o.rc_shape) && skinny_shape_known(o.rc_shape_multi); }

int launch_skinny(const SkinnyBatch& b, hipStream_t s, const char* name, const Options& o) {
    L2S_REQUIRE(b.count >= 1 && b.count <= SKINNY_MAX_GROUP, "skinny group size");
    int maxt = 0, mts = 0;
    for (int i = 0; i < b.count; ++i) {
        const SkinnyP& p = b.p[i];
        int k = 0;
        for (int j = 0; j < 4; ++j) k += 16 * p.seg[j].nchunks;
        L2S_REQUIRE(k == p.K && p.K % 16 == 0 && p.K <= 16 * SK_WAVES * SK_MAXC, "skinny K segments");
        L2S_REQUIRE(p.B >= 1, "skinny B");
        maxt = b.ntiles[i] > maxt ? b.ntiles[i] : maxt;
        int m = (p.B + 15) / 16;
        L2S_REQUIRE(mts == 0 || mts == m, "skinny groups must share B");
        mts = m;
    }
    SkinnyBatch bl = b;
    for (int i = 0; i < bl.count; ++i) {
        for (int j = bl.p[i].nseg; j < 4; ++j) { bl.p[i].seg[j].nchunks = 0; bl.p[i].seg[j].a = bl.p[i].seg[0].a; }
        bl.p[i].layout = o.skinny_static ? skinny_layout_of(bl.p[i]) : 0;
    }
    ProfScope ps(name, s);
    int maxk = 0;
    for (int i = 0; i < bl.count; ++i) maxk = bl.p[i].K > maxk ? bl.p[i].K : maxk;
    const int cls = !o.skinny_sized ? 12 : maxk <= 512 ? 4 : maxk <= 1024 ? 8 : 12;
    // many batch rows: register-blocked blocks, the largest shape that still gives the chip one block per CU
    int shape = (bl.count > 1 && o.rc_shape_multi) ? o.rc_shape_multi : o.rc_shape;
    bool n_lstm_all = o.rc_jb == 0;       // every group an LSTM cell with pre-split planes: the launch can take the half-CU 4x2 form
    for (int i = 0; i < bl.count; ++i) n_lstm_all = n_lstm_all && bl.p[i].epi == SK_LSTM && bl.p[i].W3;
    if (shape == 0) {
        shape = 11;
        if (mts >= 4) {
            auto blocks = [&](int rt, int ct) { int n = 0; for (int i = 0; i < bl.count; ++i) n += (bl.ntiles[i] + ct - 1) / ct; return n * ((mts + rt - 1) / rt); };
            // (a 4x4 block - 33 % fewer operand bytes per tile - is MFMA-bound at 30.4 us per block: 512 rows in 31.9 us against 33.0 us with 4x2
            //  blocks in two rounds, and 256 accumulator + operand registers with spills; not kept)
            if (blocks(4, 2) >= 224 || (o.lstm_x3 == 3 && chains_hint() >= 2 && o.half_min_mts > 0 && mts >= o.half_min_mts && n_lstm_all)) shape = 42;
            else if (blocks(2, 2) >= 224) shape = 22;
            else if (blocks(2, 1) >= 224) shape = 21;
        }
    }
    // one of the decode step's K layouts shared by every group of the launch (1: [32], 2: [32|32], 3: [16|16|32|32] chunks), else 0
    int rc_lay = skinny_layout_of(bl.p[0]);
    for (int i = 1; i < bl.count; ++i) if (skinny_layout_of(bl.p[i]) != rc_lay) rc_lay = 0;
    if (rc_lay == 4) rc_lay = 0;      // the 256-wide prenet layers: two chunks per slice, general blocks
    int n_lstm = 0;
    for (int i = 0; i < bl.count; ++i) n_lstm += bl.p[i].epi == SK_LSTM ? 1 : 0;
    const int rc_kind = n_lstm == bl.count ? 2 : n_lstm == 0 ? 1 : 0;
    // Block forms by how many launch chains the caller keeps in flight (l2s_set_thread_chains): with two or more, blocks of half a compute unit, so
    // that kernels of different chains run side by side on the CUs; a chain that has the chip to itself keeps the eight-wave blocks (3-4 % faster
    // alone).  Same bits either way.
    const bool overlap = chains_hint() >= 2;
    int x3 = (o.rc_jb == 0 && rc_kind == 2 && (rc_lay == 1 || rc_lay == 2 || rc_lay == 6)) ? o.lstm_x3 : 0;      // LSTM launches on the bf16 matrix cores (1: four waves, 2: eight, 3: half-CU 4x2 blocks when chains overlap)
    if (x3 == 3 && !overlap) x3 = 2;
    for (int i = 0; i < bl.count; ++i) if (!bl.p[i].W3) x3 = 0;
    // several groups, >= 128 rows (below that the 1x1 / 2x1 uniform grids with two blocks per CU are faster, tools/time_step_phases.py): per-group
    // block shapes in one flat grid of at most one block per CU ("skinny_flat", default on)
    if (!g_skinny_ts && bl.count > 1 && mts >= 8 && o.skinny_flat && !o.rc_shape && !o.rc_shape_multi && maxk <= 1024 && rc_kind == 1) {      // a forced block shape wins; LSTM groups (the BiLSTM's two directions) keep the uniform grid of four-wave blocks
        SkinnyFlat fl{};
        const bool half = o.flat_half != 0 && (o.flat_half >= 2 || chains_hint() >= 2);
        const int plan = plan_flat(bl, mts, half ? 512 : 256, fl, half, o.flat_xcd != 0);
        // 0 = general blocks, 1 = straight-line eight-wave blocks (default: 12.4 us at 256 rows against 12.9 general), 2 = straight-line four-wave
        // blocks ("skinny_rc_jb" = 44: 14.9 us - these blocks move 262 KB for 3.4 us of matrix work; eight waves keep more loads in flight)
        int stat = half ? 2 : o.rc_jb == 0 || o.rc_jb == 28 ? 1 : o.rc_jb == 44 ? 2 : 0;
        for (int i = 0; i < bl.count; ++i) if (!(skinny_layout_of(bl.p[i]) == (bl.p[i].K > 512 ? 2 : 1) && bl.p[i].epi != SK_LSTM)) stat = 0;
        if (plan) {
            switch (plan) {
                case 2121: launch_flat<21, 21>(bl, fl, mts, s, stat); break;
                case 2122: launch_flat<21, 22>(bl, fl, mts, s, stat); break;
                case 2142: launch_flat<21, 42>(bl, fl, mts, s, stat); break;
                case 2221: launch_flat<22, 21>(bl, fl, mts, s, stat); break;
                case 2222: launch_flat<22, 22>(bl, fl, mts, s, stat); break;
                default: launch_flat<22, 42>(bl, fl, mts, s, stat); break;
            }
            L2S_CHECK_HIP(hipGetLastError());
            return 0;
        }
    }
    for (int i = 0; i < bl.count; ++i)
        L2S_REQUIRE(!bl.p[i].a_sum || (rc_lay == 6 && rc_kind == 2 && skinny_sum_supported(o) && skinny_shape_known(shape) && shape != 0),
                    "skinny: a summed segment needs the straight-line LSTM blocks (shape 11 / 21 / 22 / 42, default operand batching)");
    bool any_sum = false;
    for (int i = 0; i < bl.count; ++i) any_sum = any_sum || bl.p[i].a_sum;
#ifdef L2S_DIAG
    if (g_skinny_ts && shape == 42 && bl.count == 1 && rc_kind == 2 && (rc_lay == 2 || rc_lay == 3)) {
        const dim3 grid((maxt + 1) / 2, (mts + 3) / 4, 1);
        if (rc_lay == 3) hipLaunchKernelGGL(skinny_rcs_timed_kernel<3>, grid, dim3(256), 0, s, bl, mts, g_skinny_ts);
        else if (x3 == 2) hipLaunchKernelGGL((skinny_rcs_timed_kernel<2, true, 8>), grid, dim3(512), 0, s, bl, mts, g_skinny_ts);
        else if (x3) hipLaunchKernelGGL((skinny_rcs_timed_kernel<2, true>), grid, dim3(256), 0, s, bl, mts, g_skinny_ts);
        else hipLaunchKernelGGL(skinny_rcs_timed_kernel<2>, grid, dim3(256), 0, s, bl, mts, g_skinny_ts);
    }
    else if (g_skinny_ts) hipLaunchKernelGGL(skinny_kernel_timed, dim3(maxt, mts, b.count), dim3(512), 0, s, bl, g_skinny_ts);
    else
#endif
    if (shape == 11 && o.rc_jb == 0 && rc_lay && rc_kind && launch_rc4<1, 1, 4>(bl, rc_lay, rc_kind, maxt, mts, s, x3)) {}      // 16 x 16 tiles, four waves
    else if (shape == 42) launch_rc<4, 2>(bl, cls, maxt, mts, s, o.rc_jb, rc_lay, rc_kind, x3);
    else if (shape == 22) launch_rc<2, 2>(bl, cls, maxt, mts, s, o.rc_jb, rc_lay, rc_kind, x3);
    else if (shape == 21) launch_rc<2, 1>(bl, cls, maxt, mts, s, o.rc_jb, rc_lay, rc_kind, x3);
    else if (any_sum) L2S_REQUIRE(false, "skinny: no block form of this launch sums u = prenet + o in its loader (SkinnyP::a_sum would be ignored)");
    else if (cls == 4) hipLaunchKernelGGL(skinny_kernel<4>, dim3(maxt, mts, b.count), dim3(512), 0, s, bl);
    else if (cls == 8 && o.skinny_split8 == 2) hipLaunchKernelGGL((skinny_kernel_split<8, 2, 8>), dim3(maxt, mts, b.count), dim3(512), 0, s, bl);
    else if (cls == 8) hipLaunchKernelGGL(skinny_kernel<8>, dim3(maxt, mts, b.count), dim3(512), 0, s, bl);
    else if (cls == 12 && o.skinny_split == 2) hipLaunchKernelGGL((skinny_kernel_split<12, 2, 6>), dim3(maxt, mts, b.count), dim3(512), 0, s, bl);
    else if (cls == 12 && o.skinny_split == 3) hipLaunchKernelGGL((skinny_kernel_split<12, 3, 8>), dim3(maxt, mts, b.count), dim3(512), 0, s, bl);
    else hipLaunchKernelGGL(skinny_kernel<12>, dim3(maxt, mts, b.count), dim3(512), 0, s, bl);
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

constexpr int ATT_PRE2_MAXC = 2;      // prenet layer 2: K = 256 = 16 * 8 waves * 2 chunks
template <bool VL, bool SKIP0 = false>
__global__ __launch_bounds__(512) void step_attn_kernel(const StepB sb) {
    constexpr int SMF = ATT_SM_FLOATS + (VL && !SKIP0 ? ATT_VLDS_FLOATS : 0);
    __shared__ __attribute__((aligned(16))) float sm[SMF > SK_RED_FLOATS ? SMF : SK_RED_FLOATS];
    const int nb = sb.at.B, ptiles = sb.pre2_tiles;
    L2S_PIN_S("s"(nb), "s"(ptiles));
    const int bid = blockIdx.x;
    L2S_BLOCK_STAMP_BEGIN();
    if (bid < nb) {
        attention_block<false, false, VL, SKIP0>(sb.at, bid, sm);
    } else if (bid < 2 * nb) {
        content_block(sb.at, bid - nb, sm);
    } else {
        const int j = bid - 2 * nb;
        const int tile = j % ptiles, mt = j / ptiles;
        skinny_block<false, ATT_PRE2_MAXC>(sb.pre2, tile, mt, sm);       // K = 256: the 12-chunk instance would set this kernel's VGPR count
    }
    L2S_BLOCK_STAMP_END(3u, sb.at.q);
}

#ifdef L2S_DIAG
// measurement build (tools/attn_timeline.py): thread 0 of every attention block stamps the 100 MHz wall clock at seven points
__global__ __launch_bounds__(512) void step_attn_timed_kernel(const StepB sb, unsigned long long* ts) {
    __shared__ __attribute__((aligned(16))) float sm[(ATT_SM_FLOATS + ATT_VLDS_FLOATS) > SK_RED_FLOATS ? (ATT_SM_FLOATS + ATT_VLDS_FLOATS) : SK_RED_FLOATS];
    const int nb = sb.at.B, ptiles = sb.pre2_tiles;
    const int bid = blockIdx.x;
    if (bid < nb) {
        attention_block<false, true, true>(sb.at, bid, sm, nullptr, ts);
    } else if (bid < 2 * nb) {
        content_block(sb.at, bid - nb, sm);
    } else {
        const int j = bid - 2 * nb;
        skinny_block<false, ATT_PRE2_MAXC>(sb.pre2, j % ptiles, j / ptiles, sm);
    }
}
static unsigned long long* g_attn_ts = nullptr;
void attn_set_timeline(unsigned long long* ts) { g_attn_ts = ts; }

#endif
int launch_step_attn(const AttnP& at, const SkinnyP& pre2, int pre2_tiles, hipStream_t s, int lds_values, int skip0) {
    L2S_REQUIRE(at.T <= ATT_MAXT && at.m <= 16, "attention sizes");
    L2S_REQUIRE(pre2.K <= 16 * SK_WAVES * ATT_PRE2_MAXC, "prenet layer 2 is a 256-wide layer");
    StepB sb;
    sb.at = at;
    sb.pre2 = pre2;
    sb.pre2_tiles = pre2_tiles;
    sb.mts = (at.B + 15) / 16;
    ProfScope ps("step_attention_prenet2", s);
    // projected values of a short clip fetched as 16-byte rows through LDS (attention_block<.., VLDS>): faster alone and at up to 128 rows per launch
    // (32 rows: 8.64 against 8.85 us inside the step), not at 256 (11.2 against 10.6 us event-bracketed, the pass 0.06 ms longer) - unless the caller keeps several
    // chains in flight: three 74-register blocks per CU leave room for other chains' kernels (+1.4 % at three chains): option 1 = by rows and chains, 2 = always
    const bool vl = lds_values && (lds_values >= 2 || at.B <= 128 || chains_hint() >= 2) && at.vp != nullptr && at.T <= 32;
#ifdef L2S_DIAG
    if (g_attn_ts && vl) hipLaunchKernelGGL(step_attn_timed_kernel, dim3(2 * at.B + pre2_tiles * sb.mts), dim3(512), 0, s, sb, g_attn_ts);
    else
#endif
    if (vl && skip0 && (skip0 >= 2 || chains_hint() >= 2)) hipLaunchKernelGGL((step_attn_kernel<true, true>), dim3(2 * at.B + pre2_tiles * sb.mts), dim3(512), 0, s, sb);
    else if (vl) hipLaunchKernelGGL(step_attn_kernel<true>, dim3(2 * at.B + pre2_tiles * sb.mts), dim3(512), 0, s, sb);
    else hipLaunchKernelGGL(step_attn_kernel<false>, dim3(2 * at.B + pre2_tiles * sb.mts), dim3(512), 0, s, sb);
    L2S_CHECK_HIP(hipGetLastError());
    return 0;
}

#ifdef L2S_DIAG
// ---- launch-floor probes (tools/launch_floor.py): chains of dependent launches shaped like a decode phase
__global__ __launch_bounds__(512) void probe_empty_kernel(float* out) {
    if (threadIdx.x == 1023) out[0] = 0.f;
}
__global__ __launch_bounds__(512) void probe_touch_kernel(const float* __restrict__ in, float* __restrict__ out, int n_per_block) {
    // every block reads n_per_block KiB-sized wave loads (like a K slice) and writes one value per block
    const float4* src = reinterpret_cast<const float4*>(in) + (int64_t)blockIdx.x * n_per_block * 64 * 8;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int j = 0; j < n_per_block; ++j) {
        const float4 v = src[(int64_t)(j * 8 + (threadIdx.x >> 6)) * 64 + (threadIdx.x & 63)];
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[blockIdx.x] = acc.x;
    if (threadIdx.x == 0) out[blockIdx.x] = acc.x;
}
// every block spins for `ticks` periods of the 100 MHz constant clock: a kernel of known duration, to expose the GPU-side gap between
// dependent launches when the host enqueues faster than the GPU drains
__global__ __launch_bounds__(512) void probe_spin_kernel(float* out, int ticks) {
    const uint64_t t0 = wall_clock64();
    while ((int64_t)(wall_clock64() - t0) < ticks) __builtin_amdgcn_s_sleep(1);
    if (threadIdx.x == 1023) out[0] = 0.f;
}
int launch_probe(int kind, int blocks, int n_per_block, const float* in, float* out, hipStream_t s) {
    if (kind == 2) hipLaunchKernelGGL(probe_spin_kernel, dim3(blocks), dim3(512), 0, s, out, n_per_block);
    else if (kind == 0) hipLaunchKernelGGL(probe_empty_kernel, dim3(blocks), dim3(512), 0, s, out);
    else hipLaunchKernelGGL(probe_touch_kernel, dim3(blocks), dim3(512), 0, s, in, out, n_per_block);
    L2S_CHECK_HIP(hipGetLastError());
    return 0;
}

#endif

}  // namespace l2s
