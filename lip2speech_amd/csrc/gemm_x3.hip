// Tiled GEMM with fp32 operands on the bf16 matrix cores ("split-bf16", 3 x bf16 = 24 significand bits).
//
// gfx950 has no reduced-precision f32 MFMA (no xf32): v_mfma_f32_32x32x2_f32 runs at the fp32 VECTOR rate, 1/16 of the bf16 rate, so the
// dense kernels of this path top out at 157 TFLOP/s (gemm_nt.hip reaches 119).  Every fp32 value is EXACTLY the sum of three bf16 values,
//      x = hi + mid + lo,   hi = x with the low 16 significand bits cleared,  mid = (x - hi) likewise,  lo = x - hi - mid
// (each difference is exact in fp32, and the last remainder has <= 8 significant bits), and a product of two bf16 values is exact in the
// fp32 accumulator of v_mfma_f32_32x32x16_bf16.  Of the nine partial products of x*y the six with weight >= 2^-16 are kept,
//      x*y ~= hi*hi + hi*mid + mid*hi + mid*mid + hi*lo + lo*hi          (dropped: mid*lo + lo*mid + lo*lo <= 2^-22 |x*y|),
// six bf16 MFMAs for one K step of 16 where the f32 form needs eight MFMAs of K = 2 at 16x the cost each: 6/16 of the f32 matrix time.
// Accumulation is fp32 inside the MFMA as before; results differ from the f32-MFMA kernel by rounding-level amounts (tests/: measured
// against an fp64 product next to the f32 kernel's own error).
//
// Block = 512 threads = 4 MFMA waves (2x2, each 64x64 = 2x2 tiles of 32x32, 64 accumulator registers) + 4 staging waves, block tile
// 128(M) x 128(N) x 32(K).  Operands are fetched as fp32 float4 by range-checked buffer loads (the implicit-Conv1d addressing of
// gemm_nt.hip), split while they are staged into LDS - three bf16 planes per operand, rows of 80 bytes (64 + 16 pad: conflict-free
// ds_read_b128 for the 32x32x16 operand layout: lane l reads the 8 consecutive k of row l&31 at k offset 8*(l>>5)) - so a value is split once
// per block that uses it and every ds_read_b128 is one MFMA operand.  Two LDS stages (120 KB, one block per CU).
#include "l2s_common.h"
#include "gemm_dev.h"

namespace l2s {

constexpr int XM = 128, XN = 128, XK = 32;
constexpr int XLDB = 80;                         // bytes per LDS row (32 bf16 + pad)
constexpr int XPLANE = XM * XLDB;                // bytes per plane (128 rows)

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

struct X3Split { uint2 hi, mid, lo; };           // 4 consecutive k as bf16 pairs

__device__ __forceinline__ X3Split x3_split(const float4& v) {
    const float f[4] = {v.x, v.y, v.z, v.w};
    unsigned h[4], m[4], l[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const unsigned xb = __float_as_uint(f[e]);
        const unsigned hb = xb & 0xFFFF0000u;
        const float r1 = f[e] - __uint_as_float(hb);            // exact
        const unsigned mb = __float_as_uint(r1) & 0xFFFF0000u;
        const float r2 = r1 - __uint_as_float(mb);              // exact, <= 8 significant bits
        h[e] = xb; m[e] = __float_as_uint(r1); l[e] = __float_as_uint(r2);
    }
    X3Split s;
    // pack the upper halves of two words: {lo16 = even k, hi16 = odd k}
    s.hi = make_uint2(__builtin_amdgcn_perm(h[1], h[0], 0x07060302u), __builtin_amdgcn_perm(h[3], h[2], 0x07060302u));
    s.mid = make_uint2(__builtin_amdgcn_perm(m[1], m[0], 0x07060302u), __builtin_amdgcn_perm(m[3], m[2], 0x07060302u));
    s.lo = make_uint2(__builtin_amdgcn_perm(l[1], l[0], 0x07060302u), __builtin_amdgcn_perm(l[3], l[2], 0x07060302u));
    return s;
}

struct X3LoadP { int K, Cin, taps, Tin, lda, a_split, a_gap; };

// Wave-specialised: waves 0-3 (one per SIMD) only read operands from LDS and issue MFMAs, waves 4-7 only fetch, split and stage the next
// K tile into the other LDS stage - the split's VALU work and the global-load latency run beside the matrix pipe instead of in front of
// it, and there is ONE barrier per K tile (stage kt+1 written / stage kt read).
// TIMED: the measurement build (l2s_op_gemm_x3_timeline): lane 0 of every wave of ONE block stamps the shader clock at the points marked
// X3_STAMP - [wave][K tile][slot] - so that a K tile's time splits into producer work (fetch issue / loads landed / split + LDS writes),
// consumer work (MFMA groups) and the time either side waits at the barrier.  Same arithmetic; the product launches TIMED = false.
#define X3_STAMP(kt_, slot_) do { if constexpr (TIMED) { if (stamp_on && lane == 0 && (kt_) < 96) ts[((wave * 96) + (kt_)) * 8 + (slot_)] = clock64(); } } while (0)
template <bool TIMED>
__global__ __launch_bounds__(512, 2) void gemm_x3_kernel(const GemmBatch batch, unsigned long long* __restrict__ ts, int stamp_block) {
    const GemmP& p = batch.p[blockIdx.z];
    const bool stamp_on = TIMED && (int)(blockIdx.y * gridDim.x + blockIdx.x) == stamp_block && blockIdx.z == 0;
    // XCD-aware tile order (see gemm_nt.hip): each XCD gets a contiguous run of tiles, N index fastest
    int bx, by;
    {
        const int gx = gridDim.x, total = gx * gridDim.y;
        const int L = blockIdx.y * gx + blockIdx.x;
        const int xcd = L & 7, local = L >> 3;
        const int chunk = total >> 3, rem = total & 7;
        const int tile = xcd * chunk + (xcd < rem ? xcd : rem) + local;
        bx = tile % gx; by = tile / gx;
    }
    const int m0 = by * XM, n0 = bx * XN;
    if (m0 >= p.M || n0 >= p.N) return;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];      // 2 stages x {A planes, B planes}
    constexpr int STAGE = 6 * XPLANE;

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int nkt = (p.K + XK - 1) / XK;
    const int wm = (wave >> 1) & 1, wn = wave & 1;
    const int li = lane & 31, lg = lane >> 5;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    if (wave >= 4) {
        // ------------------------------------------------------------------------------------------------ producers
        const int pt = tid - 256;
        // staging role: 8 lanes per row (32 k); the two rows of a 16-lane ds_write group are 4 apart - with 80-byte rows their 16-dword
        // spans then fall on disjoint halves of the 32 banks a ds_write_b64 sees (consecutive rows overlap in 4 banks)
        const int oct = pt >> 3;
        const int lr = (oct & ~7) + ((oct & 1) << 2) + ((oct & 7) >> 1), kq = (pt & 7) * 4;
        bool avalid[4], wvalid[4];
        int atbase[4];
        unsigned arow_off[4];                                // float offset of the row's sequence inside A
        const X3LoadP lp{p.K, p.Cin, p.taps, p.Tin, p.lda, p.a_split, p.a_gap};
        asm volatile("" ::"s"(lp.K), "s"(lp.Cin), "s"(lp.taps), "s"(lp.Tin), "s"(lp.lda), "s"(lp.a_split), "s"(lp.a_gap));
        const int ldw = p.ldw ? p.ldw : lp.K;
        constexpr unsigned OOB = 0x80000000u;                // both extents are < 2 GiB (checked at launch)
        unsigned woff[4], aoff[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int m = m0 + lr + 32 * j;
            avalid[j] = m < p.M;
            const int mm = avalid[j] ? m : 0;
            const int b = mm / p.Tout, t = mm - b * p.Tout + p.win_off;
            atbase[j] = t * p.stride - p.pad;
            arow_off[j] = (unsigned)((int64_t)b * lp.Tin * lp.lda);
            const int n = n0 + lr + 32 * j;
            wvalid[j] = n < p.N;
            woff[j] = wvalid[j] ? (unsigned)((int64_t)n * ldw + kq) * 4u : OOB;
        }
        int tap = 0, ci = kq;
        if (lp.taps > 1) { tap = kq / lp.Cin; ci = kq - tap * lp.Cin; }
        const int nseq = (p.M + p.Tout - 1) / p.Tout;
        const __amdgpu_buffer_rsrc_t ra_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.A), 0, (int)((int64_t)nseq * lp.Tin * lp.lda * 4), 0x00020000);
        const __amdgpu_buffer_rsrc_t rw_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.W), 0, (int)((int64_t)p.N * ldw * 4), 0x00020000);
        auto set_tap = [&]() {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int tin = atbase[j] + tap;
                const bool ok = avalid[j] && tin >= 0 && tin < lp.Tin;
                aoff[j] = ok ? (arow_off[j] + (unsigned)(tin * lp.lda)) * 4u : OOB;
            }
        };
        set_tap();
        auto fetch = [&](int k, float4* ra, float4* rb) {
            const unsigned kbad = k < lp.K ? 0u : OOB;                                       // K is a multiple of 4: a quad is all in or all out
            const unsigned col = (unsigned)(ci + (ci >= lp.a_split ? lp.a_gap : 0)) * 4u;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                ra[j] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(ra_rs, (int)((aoff[j] | kbad) + (aoff[j] == OOB ? 0u : col)), 0, 0));
                rb[j] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rw_rs, (int)(woff[j] | kbad), 0, 0));
            }
        };
        auto advance = [&]() {
            ci += XK;
            if (lp.taps > 1 && ci >= lp.Cin) { ci -= lp.Cin; ++tap; set_tap(); }
#pragma unroll
            for (int j = 0; j < 4; ++j) woff[j] += wvalid[j] ? XK * 4u : 0u;
        };
        const int st_off = lr * XLDB + kq * 2;                              // byte offset of this thread's first staged row inside a plane
        auto stage = [&](const float4* ra, const float4* rb, int st, int kt_stamp) {
            unsigned char* base = smem + st * STAGE + st_off;
            if constexpr (TIMED) {                           // measurement build: all splits, stamp, all stores, stamp
                X3Split sa[4], sb[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) { sa[j] = x3_split(ra[j]); sb[j] = x3_split(rb[j]); }
                asm volatile("" :: "v"(sa[0].hi.x), "v"(sa[1].hi.x), "v"(sa[2].hi.x), "v"(sa[3].lo.y), "v"(sb[0].hi.x), "v"(sb[1].hi.x), "v"(sb[2].hi.x), "v"(sb[3].lo.y));
                __builtin_amdgcn_sched_barrier(0);
                X3_STAMP(kt_stamp, 4);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    unsigned char* ad = base + 32 * j * XLDB;
                    unsigned char* bd = ad + 3 * XPLANE;
                    *reinterpret_cast<uint2*>(ad) = sa[j].hi; *reinterpret_cast<uint2*>(ad + XPLANE) = sa[j].mid; *reinterpret_cast<uint2*>(ad + 2 * XPLANE) = sa[j].lo;
                    *reinterpret_cast<uint2*>(bd) = sb[j].hi; *reinterpret_cast<uint2*>(bd + XPLANE) = sb[j].mid; *reinterpret_cast<uint2*>(bd + 2 * XPLANE) = sb[j].lo;
                }
                return;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const X3Split sa = x3_split(ra[j]), sb = x3_split(rb[j]);
                unsigned char* ad = base + 32 * j * XLDB;
                unsigned char* bd = ad + 3 * XPLANE;
                *reinterpret_cast<uint2*>(ad) = sa.hi; *reinterpret_cast<uint2*>(ad + XPLANE) = sa.mid; *reinterpret_cast<uint2*>(ad + 2 * XPLANE) = sa.lo;
                *reinterpret_cast<uint2*>(bd) = sb.hi; *reinterpret_cast<uint2*>(bd + XPLANE) = sb.mid; *reinterpret_cast<uint2*>(bd + 2 * XPLANE) = sb.lo;
            }
        };
        // two register sets: tile kt+2 is requested before tile kt+1 is split, so a fetch has a whole K tile of MFMA time to land.  The loop body
        // is BRANCH-FREE on purpose: fetches past the last tile are range-checked to zero (kbad) and the stage they are written to is never
        // read.  With `if (kt + 2 < nkt)` guards around the fetches the compiler has to assume at the join that the newer register set may not
        // have been requested and waits for the OLDER set with vmcnt(7..0) - i.e. for the loads it has just issued as well: every K tile
        // then paid a full memory round trip under load (stamped build: 2 650 of the tile's 3 060 clk in the producers).
        float4 ra0[4], rb0[4], ra1[4], rb1[4];
        fetch(kq, ra0, rb0);
        advance(); fetch(XK + kq, ra1, rb1);
        stage(ra0, rb0, 0, 95);
        __syncthreads();                                     // stage 0 = tile 0
        for (int kt = 0; kt < nkt; kt += 2) {
            // iteration kt: consumers read stage 0; stage 1 <- tile kt+1 (registers set 1), request tile kt+2 into set 0
            X3_STAMP(kt, 0);
            advance(); fetch((kt + 2) * XK + kq, ra0, rb0);
            __builtin_amdgcn_sched_barrier(0);               // the requests go out BEFORE the older set is waited for and split
            if constexpr (TIMED) { asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); X3_STAMP(kt, 1); }      // the older register set has landed
            stage(ra1, rb1, 1, kt);
            if constexpr (TIMED) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
            X3_STAMP(kt, 2);
            __syncthreads();
            X3_STAMP(kt, 3);
            if (kt + 1 >= nkt) break;
            // iteration kt+1: consumers read stage 1; stage 0 <- tile kt+2 (set 0), request tile kt+3 into set 1
            X3_STAMP(kt + 1, 0);
            advance(); fetch((kt + 3) * XK + kq, ra1, rb1);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (TIMED) { asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); X3_STAMP(kt + 1, 1); }
            stage(ra0, rb0, 0, kt + 1);
            if constexpr (TIMED) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
            X3_STAMP(kt + 1, 2);
            __syncthreads();
            X3_STAMP(kt + 1, 3);
        }
    } else {
        // ------------------------------------------------------------------------------------------------ consumers
        // Operand fragments are double-buffered in registers: the ds_read_b128s of the next K step (or of the next tile's first step, right
        // after the barrier) are in flight while the 24 MFMAs of the current one issue, so the matrix pipe never waits for LDS.
        const unsigned char* a_rd = smem + (wm * 64 + li) * XLDB + lg * 16;               // this lane's operand rows: + i*32 rows, + s*32 bytes, + plane
        const unsigned char* b_rd = smem + 3 * XPLANE + (wn * 64 + li) * XLDB + lg * 16;
        struct Frags { bf16x8 ah[2], am[2], al[2], bh[2], bm[2], bl[2]; };
        auto read_frags = [&](Frags& f, int so, int st) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const unsigned char* ap = a_rd + so + i * 32 * XLDB + st * 32;
                const unsigned char* bp = b_rd + so + i * 32 * XLDB + st * 32;
                f.ah[i] = *reinterpret_cast<const bf16x8*>(ap); f.am[i] = *reinterpret_cast<const bf16x8*>(ap + XPLANE); f.al[i] = *reinterpret_cast<const bf16x8*>(ap + 2 * XPLANE);
                f.bh[i] = *reinterpret_cast<const bf16x8*>(bp); f.bm[i] = *reinterpret_cast<const bf16x8*>(bp + XPLANE); f.bl[i] = *reinterpret_cast<const bf16x8*>(bp + 2 * XPLANE);
            }
        };
        // smallest partial products first; the four accumulators interleave so that no MFMA waits on its predecessor
#define L2S_X3_TERM(A_, B_)                                                                           \
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A_[0], B_[0], acc[0][0], 0, 0, 0);     \
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A_[0], B_[1], acc[0][1], 0, 0, 0);     \
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A_[1], B_[0], acc[1][0], 0, 0, 0);     \
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A_[1], B_[1], acc[1][1], 0, 0, 0);
#define L2S_X3_MMA(F_) { L2S_X3_TERM(F_.al, F_.bh) L2S_X3_TERM(F_.ah, F_.bl) L2S_X3_TERM(F_.am, F_.bm) L2S_X3_TERM(F_.am, F_.bh) L2S_X3_TERM(F_.ah, F_.bm) L2S_X3_TERM(F_.ah, F_.bh) }
        Frags f0, f1;
        __syncthreads();                                     // stage 0 ready
        read_frags(f0, 0, 0);
        for (int kt = 0; kt < nkt; ++kt) {
            const int so = (kt & 1) * STAGE;
            X3_STAMP(kt, 0);
            read_frags(f1, so, 1);
            __builtin_amdgcn_sched_barrier(0);
            L2S_X3_MMA(f0)
            __builtin_amdgcn_sched_barrier(0);
            X3_STAMP(kt, 1);                                 // the 24 MFMAs of step 0 are issued (not necessarily retired)
            __syncthreads();                                 // this stage is read (f1 has landed: the barrier waits for it); the other one is written
            X3_STAMP(kt, 2);
            if (kt + 1 < nkt) read_frags(f0, STAGE - so, 0);
            __builtin_amdgcn_sched_barrier(0);
            L2S_X3_MMA(f1)
            __builtin_amdgcn_sched_barrier(0);
            X3_STAMP(kt, 3);
        }
#undef L2S_X3_MMA
#undef L2S_X3_TERM
    }

    // epilogue.  The accumulators leave through LDS, half a tile (the 32-row sub-tiles i of both wave rows = 64 rows x 128 columns, 32 KB)
    // at a time: every accumulator element is addressed with compile-time indices (a rolled loop over them would put all 64 in scratch
    // memory), the rolled store loop that follows - all eight waves - reads LDS, runs the fused epilogue once per element and writes rows
    // of 128 consecutive columns.  C/D layout of a 32x32 tile: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).
    constexpr int CTLD = XN + 1;                               // odd pitch: rows and columns of the tile are both conflict-free to walk
    float* const ct = reinterpret_cast<float*>(smem);          // [64][129]; the operand stages are dead (barrier at the loop end)
    const GemmP pl = p;                                        // epilogue parameters in SGPRs: through the kernarg reference the rolled loop
                                                               // below re-fetches them with scalar loads on every iteration
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        if (i) __syncthreads();
        if (wave < 4) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    ct[(wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lg) * CTLD + wn * 64 + j * 32 + li] = acc[i][j][r];
        }
        __syncthreads();
        if (pl.c_tr_T > 0) {
            // channel-first store (the post-net's last layer: out[(b, n, t)]): lanes run along the ROWS, so a wave writes 32 + 32 consecutive frames of
            // one channel (128-byte runs) instead of 64 channels 4 T bytes apart; the odd row pitch of the tile keeps the column reads conflict-free
            const int q = tid & 63, row = m0 + (q >> 5) * 64 + i * 32 + (q & 31);
            if (row < pl.M) {
                for (int cl = tid >> 6; cl < XN; cl += 8) {
                    const int col = n0 + cl;
                    if (col < pl.N) gemm_store(pl, row, col, ct[q * CTLD + cl], pl.scale ? pl.scale[col] : 1.0f, pl.shift ? pl.shift[col] : 0.0f);
                }
            }
            continue;
        }
        const int cl = tid & 127, col = n0 + cl;
        if (col < pl.N) {
            const float sc = pl.scale ? pl.scale[col] : 1.0f;
            const float sh = pl.shift ? pl.shift[col] : 0.0f;
#pragma unroll 4
            for (int q = tid >> 7; q < 64; q += 4) {              // local row q: wave-row q >> 5, row q & 31 of sub-tile i
                const int row = m0 + (q >> 5) * 64 + i * 32 + (q & 31);
                if (row < pl.M) gemm_store(pl, row, col, ct[q * CTLD + cl], sc, sh);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ wide tile: 128(M) x 256(N) x 16(K)
// What the stamped build of the kernel above showed (tools/gemm_x3_timeline.py, profiles/r03_gemm_x3_timeline.txt): per K tile of 32 the four MFMA
// waves are busy ~1 950 clk (1 536 of matrix time) and the four staging waves ~2 430 - 540 to issue the next fetch, 240 of split VALU and **1 510 clk
// for their 24 ds_write_b64**: the LDS STORE path (VGPR -> LDS, 2 clk per source dword per instruction, MI355X_MICROARCH.md section LDS) moves the
// tile's 48 KB of planes at 32 B/clk and that, not the operand fetch, sets the pace.  A 128x128 tile needs 32 B of LDS fill per clk of MFMA time;
// this one needs 24.  Block = 768 threads: EIGHT MFMA waves (2 x 4, two per SIMD, each 64x64 exactly as above: the pipe of a SIMD stays fed while
// one of its two waves waits) + the same four staging waves; a stage is ONE K step (rows of 48 bytes = 16 bf16 + pad: conflict-free for the
// ds_read_b128 lane groups), 2 x 54 KB of LDS.  A step runs as two column halves so that every LDS read has MFMAs to hide behind: B half 1 is read
// while half 0's 12 MFMAs issue, the barrier sits between the halves, and the next step's A (into its second register set) and B half 0 are read
// while half 1's MFMAs issue - 136 registers of accumulators and fragments, three waves per SIMD.  Per output element the accumulation order is the
// narrow kernel's (K steps ascending, the six terms in the same order): bit-identical results.
constexpr int WM = 128, WN = 256, WK = 16;
constexpr int WLDB = 48;                         // bytes per LDS row (16 bf16 + pad)
constexpr int WPA = WM * WLDB, WPB = WN * WLDB;  // bytes per plane
constexpr int WSTAGE = 3 * WPA + 3 * WPB;        // 55 296
constexpr int WTHREADS = 768;
// with the weight operand by LDS-DMA from pre-split planes (GemmP::W3): B rows of 32 bytes, no pad - the DMA's per-lane source address swaps the two
// 16-byte halves of rows 8-15 (mod 16), which makes the consumers' ds_read_b128 lane groups conflict-free without it
constexpr int WLDBD = 32, WPBD = WN * WLDBD;     // 8 192 bytes per plane
constexpr int WSTAGED = 3 * WPA + 3 * WPBD;      // 43 008

template <bool TIMED, bool DMAW>
__global__ __launch_bounds__(WTHREADS) void gemm_x3w_kernel(const GemmBatch batch, unsigned long long* __restrict__ ts, int stamp_block) {
    const GemmP& p = batch.p[blockIdx.z];
    const bool stamp_on = TIMED && (int)(blockIdx.y * gridDim.x + blockIdx.x) == stamp_block && blockIdx.z == 0;
    int bx, by;
    {
        const int gx = gridDim.x, total = gx * gridDim.y;
        const int L = blockIdx.y * gx + blockIdx.x;
        const int xcd = L & 7, local = L >> 3;
        const int chunk = total >> 3, rem = total & 7;
        const int tile = xcd * chunk + (xcd < rem ? xcd : rem) + local;
        bx = tile % gx; by = tile / gx;
    }
    const int m0 = by * WM, n0 = bx * WN;
    if (m0 >= p.M || n0 >= p.N) return;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];      // 2 stages x {A planes, B planes}
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int nks = (p.K + WK - 1) / WK;
    const int wm = (wave >> 2) & 1, wn = wave & 3;
    const int li = lane & 31, lg = lane >> 5;
    constexpr int STG = DMAW ? WSTAGED : WSTAGE;     // bytes per stage

    // the accumulators leave through LDS, one 32-row sub-tile of both wave rows (64 rows x 256 columns, 64 KB) at a time; all twelve waves run
    // the fused epilogue over it (rows of 256 consecutive columns).  The accumulators are declared in the consumer branch only: live across the
    // producer branch as well they cost it 64 registers and the allocator spills
    float* const ct = reinterpret_cast<float*>(smem);
    const GemmP pl = p;
    auto store_rows = [&](int i) {
        const int cl = tid & 255, col = n0 + cl;
        if (col < pl.N) {
            const float sc = pl.scale ? pl.scale[col] : 1.0f;
            const float sh = pl.shift ? pl.shift[col] : 0.0f;
#pragma unroll 2
            for (int q = tid >> 8; q < 64; q += 3) {
                const int row = m0 + (q >> 5) * 64 + i * 32 + (q & 31);
                if (row < pl.M) gemm_store(pl, row, col, ct[q * WN + cl], sc, sh);
            }
        }
    };

    if (wave >= 8) {
        // ------------------------------------------------------------------------------------------------ producers
        const int pt = tid - 512;
        // 4 lanes per row (16 k); the four rows of a 16-lane ds_write group are 2 apart: with 48-byte rows their 8-dword spans tile the 32 banks
        const int slot = pt >> 2;
        const int lr = (slot & ~7) + ((slot & 3) << 1) + ((slot >> 2) & 1), kq = (pt & 3) * 4;
        bool avalid[2];
        int atbase[2];
        unsigned arow_off[2];
        const X3LoadP lp{p.K, p.Cin, p.taps, p.Tin, p.lda, p.a_split, p.a_gap};
        asm volatile("" ::"s"(lp.K), "s"(lp.Cin), "s"(lp.taps), "s"(lp.Tin), "s"(lp.lda), "s"(lp.a_split), "s"(lp.a_gap));
        const int ldw = p.ldw ? p.ldw : lp.K;
        constexpr unsigned OOB = 0x80000000u;
        // Addressing with the K position kept UNIFORM (scalar registers): a step's 16 k are one tap and one side of the A split for every thread
        // (launch checks: Cin and a_split multiples of 16), so the per-step advance is scalar arithmetic, the column offset travels in the buffer
        // instruction's scalar offset, and the per-thread part - row base + kq - only changes when the tap does (a uniform branch).  The first form
        // (per-thread ci / tap, offsets rebuilt per load) cost the staging waves 580 clk of address VALU per step beside two MFMA waves per SIMD.
        unsigned wfix[DMAW ? 1 : 4], afix[2];                           // per-thread byte offsets (OOB: row outside the matrix / frame outside the sequence)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int m = m0 + lr + 64 * j;
            avalid[j] = m < p.M;
            const int mm = avalid[j] ? m : 0;
            const int b = mm / p.Tout, t = mm - b * p.Tout + p.win_off;
            atbase[j] = t * p.stride - p.pad;
            arow_off[j] = (unsigned)((int64_t)b * lp.Tin * lp.lda);
        }
        if constexpr (!DMAW) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int n = n0 + lr + 64 * j;
                wfix[j] = n < p.N ? (unsigned)((int64_t)n * ldw + kq) * 4u : OOB;
            }
        }
        int kb = 0, tap = 0, cib = 0;                        // uniform: first k of the step being requested, its tap, its first channel
        const int nseq = (p.M + p.Tout - 1) / p.Tout;
        const __amdgpu_buffer_rsrc_t ra_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.A), 0, (int)((int64_t)nseq * lp.Tin * lp.lda * 4), 0x00020000);
        const __amdgpu_buffer_rsrc_t rw_rs = DMAW ? __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.W3), 0, (int)((int64_t)p.N * lp.K * 6), 0x00020000)
                                                  : __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.W), 0, (int)((int64_t)p.N * ldw * 4), 0x00020000);
        // LDS-DMA of the weight planes of K step `ks` into stage `st`: 3 planes x 256 rows x 32 bytes = 24 contiguous 1-KB pieces (a piece = 32 rows of
        // one plane), six per staging wave; lane l fills 16-byte chunk l of its piece = (row l >> 1, half l & 1), reading the half the swizzle puts there
        const unsigned dma_lane = (unsigned)((lane >> 1) * 32 + (((lane & 1) ^ ((lane >> 4) & 1)) * 16));
        auto dma_b = [&](int ks, int st) {
            if (ks < nks) {                                                                  // uniform
#pragma unroll
                for (int e = 0; e < 6; ++e) {
                    const int q = (wave - 8) * 6 + e, pl = q >> 3, sub = q & 7;
                    const int soff = __builtin_amdgcn_readfirstlane((int)((((int64_t)ks * 3 + pl) * p.N + n0 + sub * 32) * 32));
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rw_rs, (__attribute__((address_space(3))) void*)(smem + st * STG + 3 * WPA + pl * WPBD + sub * 1024), 16,
                                                             (int)dma_lane, soff, 0, 0);
                }
            }
        };
        auto set_tap = [&]() {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int tin = atbase[j] + tap;
                const bool ok = avalid[j] && tin >= 0 && tin < lp.Tin;
                afix[j] = ok ? (arow_off[j] + (unsigned)(tin * lp.lda) + (unsigned)kq) * 4u : OOB;
            }
        };
        set_tap();
        // K is a multiple of 16 here (launch check), so no quad of a step is past K and the per-thread offsets are loop-invariant registers: no
        // VALU at all between the loads of one step and the next.  The requests past the last step (the loop runs two sets ahead) re-read the
        // last step instead of running off the matrix: the uniform position simply stops advancing.
        auto fetch = [&](float4* ra, float4* rb) {
            const int acol = __builtin_amdgcn_readfirstlane((cib + (cib >= lp.a_split ? lp.a_gap : 0)) * 4);
            const int wcol = __builtin_amdgcn_readfirstlane(kb * 4);
#pragma unroll
            for (int j = 0; j < 2; ++j)
                ra[j] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(ra_rs, (int)afix[j], acol, 0));
            if constexpr (!DMAW) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    rb[j] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rw_rs, (int)wfix[j], wcol, 0));
            }
        };
        auto advance = [&]() {
            if (kb + WK < lp.K) {                                                            // uniform
                kb += WK; cib += WK;
                if (lp.taps > 1 && cib >= lp.Cin) { cib -= lp.Cin; ++tap; set_tap(); }
            }
        };
        const int st_off = lr * WLDB + kq * 2;
        auto stage = [&](const float4* ra, const float4* rb, int st) {
            unsigned char* base = smem + st * STG + st_off;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const X3Split sa = x3_split(ra[j]);
                unsigned char* ad = base + 64 * j * WLDB;
                *reinterpret_cast<uint2*>(ad) = sa.hi; *reinterpret_cast<uint2*>(ad + WPA) = sa.mid; *reinterpret_cast<uint2*>(ad + 2 * WPA) = sa.lo;
            }
            if constexpr (!DMAW) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const X3Split sb = x3_split(rb[j]);
                    unsigned char* bd = base + 3 * WPA + 64 * j * WLDB;
                    *reinterpret_cast<uint2*>(bd) = sb.hi; *reinterpret_cast<uint2*>(bd + WPB) = sb.mid; *reinterpret_cast<uint2*>(bd + 2 * WPB) = sb.lo;
                }
            }
        };
        // branch-free body, requests before the older sets are waited for (see the narrow kernel).  THREE register sets: the data staged in
        // iteration ks (step ks+1) were requested in iteration ks-2 - two K steps (~3 500 clk) to land; with two sets (one step) the staging
        // waves still waited ~400 clk per step for rows of A that come from HBM
        float4 ra0[2], rb0[DMAW ? 1 : 4], ra1[2], rb1[DMAW ? 1 : 4], ra2[2], rb2[DMAW ? 1 : 4];
        if constexpr (DMAW) dma_b(0, 0);
        fetch(ra0, rb0);
        advance(); fetch(ra1, rb1);
        advance(); fetch(ra2, rb2);
        stage(ra0, rb0, 0);
        if constexpr (DMAW) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // stage 0's weight planes have landed before the barrier publishes it
        __syncthreads();                                     // stage 0 = step 0
        // every request of the prologue has landed before the loop is entered: with loads pending on the entry edge the wait-count pass merges
        // them with the back edge's at the loop header and drains ALL outstanding loads (vmcnt(0)) once per trip
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#define L2S_X3W_PRODUCE(KS_, RA_NEW, RB_NEW, RA_OLD, RB_OLD)                                                   \
        X3_STAMP(KS_, 0);                                                                                      \
        if constexpr (DMAW) dma_b((KS_) + 1, ((KS_) + 1) & 1);   /* that stage was last read in step KS_ - 1 */  \
        advance(); fetch(RA_NEW, RB_NEW);                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
        X3_STAMP(KS_, 4);                                                                                      \
        if constexpr (TIMED) { if constexpr (DMAW) asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); X3_STAMP(KS_, 1); } \
        stage(RA_OLD, RB_OLD, ((KS_) + 1) & 1);                                                                \
        if constexpr (TIMED) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }                            \
        X3_STAMP(KS_, 2);                                                                                      \
        /* all but the two newest requests (this step's activation rows) are back: the DMA pieces of step KS_ + 1 have landed */ \
        if constexpr (DMAW) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");                                   \
        __syncthreads();                                                                                       \
        X3_STAMP(KS_, 3);
        for (int ks = 0; ks < nks; ks += 3) {
            L2S_X3W_PRODUCE(ks, ra0, rb0, ra1, rb1)
            if (ks + 1 >= nks) break;
            L2S_X3W_PRODUCE(ks + 1, ra1, rb1, ra2, rb2)
            if (ks + 2 >= nks) break;
            L2S_X3W_PRODUCE(ks + 2, ra2, rb2, ra0, rb0)
        }
#undef L2S_X3W_PRODUCE
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if (i) __syncthreads();
            __syncthreads();
            store_rows(i);
        }
    } else {
        // ------------------------------------------------------------------------------------------------ consumers
        f32x16 acc[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        const unsigned char* a_rd = smem + (wm * 64 + li) * WLDB + lg * 16;
        const unsigned char* b_rd = DMAW ? smem + 3 * WPA + (wn * 64 + li) * WLDBD + ((lg ^ ((li >> 3) & 1)) * 16)
                                         : smem + 3 * WPA + (wn * 64 + li) * WLDB + lg * 16;
        constexpr int BROW = DMAW ? WLDBD : WLDB, BPL = DMAW ? WPBD : WPB;
        struct FA { bf16x8 h[2], m[2], l[2]; };
        struct FB { bf16x8 h, m, l; };
        auto read_a = [&](FA& f, int so) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const unsigned char* ap = a_rd + so + i * 32 * WLDB;
                f.h[i] = *reinterpret_cast<const bf16x8*>(ap); f.m[i] = *reinterpret_cast<const bf16x8*>(ap + WPA); f.l[i] = *reinterpret_cast<const bf16x8*>(ap + 2 * WPA);
            }
        };
        auto read_b = [&](FB& f, int so, int j) {
            const unsigned char* bp = b_rd + so + j * 32 * BROW;
            f.h = *reinterpret_cast<const bf16x8*>(bp); f.m = *reinterpret_cast<const bf16x8*>(bp + BPL); f.l = *reinterpret_cast<const bf16x8*>(bp + 2 * BPL);
        };
        // smallest partial products first; the two accumulators of a half alternate so that no MFMA waits on its predecessor
#define L2S_X3W_TERM(A_, B_, J_)                                                                              \
        acc[0][J_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A_[0], B_, acc[0][J_], 0, 0, 0);                \
        acc[1][J_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A_[1], B_, acc[1][J_], 0, 0, 0);
#define L2S_X3W_MMA_HEAD(FA_, FB_, J_) { L2S_X3W_TERM(FA_.l, FB_.h, J_) }
#define L2S_X3W_MMA_TAIL(FA_, FB_, J_) { L2S_X3W_TERM(FA_.h, FB_.l, J_) L2S_X3W_TERM(FA_.m, FB_.m, J_) L2S_X3W_TERM(FA_.m, FB_.h, J_) L2S_X3W_TERM(FA_.h, FB_.m, J_) L2S_X3W_TERM(FA_.h, FB_.h, J_) }
        // the LDS reads of a half are issued AFTER its first two MFMAs: right behind the barrier every MFMA wave of the block starts from an empty
        // pipe, and nine ds_read_b128 in front of the first MFMA were ~150 clk of that bubble per step
#define L2S_X3W_STEP(KS_, FA_CUR, FA_NXT, SO_)                                                                 \
        X3_STAMP(KS_, 0);                                                                                      \
        L2S_X3W_MMA_HEAD(FA_CUR, fb0, 0)                                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
        read_b(fb1, SO_, 1);                                                                                   \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
        L2S_X3W_MMA_TAIL(FA_CUR, fb0, 0)                                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
        X3_STAMP(KS_, 1);                                                                                      \
        __syncthreads();              /* this stage is read (fb1 has landed); the other one is written */      \
        X3_STAMP(KS_, 2);                                                                                      \
        L2S_X3W_MMA_HEAD(FA_CUR, fb1, 1)                                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
        read_a(FA_NXT, STG - (SO_)); read_b(fb0, STG - (SO_), 0);         /* past the last step: unused */     \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
        L2S_X3W_MMA_TAIL(FA_CUR, fb1, 1)                                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
        X3_STAMP(KS_, 3);
        FA fa0, fa1;
        FB fb0, fb1;
        __syncthreads();                                     // stage 0 ready
        read_a(fa0, 0); read_b(fb0, 0, 0);
        int ks = 0;
        for (; ks + 1 < nks; ks += 2) {                      // pairs of steps: no exit from the middle of the body (the accumulators would be copied at it)
            L2S_X3W_STEP(ks, fa0, fa1, 0)
            L2S_X3W_STEP(ks + 1, fa1, fa0, STG)
        }
        if (ks < nks) { L2S_X3W_STEP(ks, fa0, fa1, 0) }
#undef L2S_X3W_STEP
#undef L2S_X3W_MMA_HEAD
#undef L2S_X3W_MMA_TAIL
#undef L2S_X3W_TERM
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if (i) __syncthreads();
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    ct[(wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lg) * WN + wn * 64 + j * 32 + li] = acc[i][j][r];
            __syncthreads();
            store_rows(i);
        }
    }
}

static bool x3_aligned16(const void* ptr) { return (reinterpret_cast<uintptr_t>(ptr) & 15u) == 0; }

// A group runs on the split-bf16 kernel when every member can (below) AND the launch is large enough to fill the chip: at least one full round of tiles
// of ONE batch's rows.  Measured on MI355X (tools/time_gemm_x3.py, time_conv_x3.py, time_gemm_shapes.py; f32 kernel -> this one, round 3): 76800x512x2560
// 1 640 -> 990 us (204 TFLOP/s; the f32 matrix peak is 157), 9600x512x2560 233 -> 142 us, 7424x512x5632 383 -> 223 us; 928x512x2560 (32 tiles) loses.
// one member of a launch: operands the kernel can address (float4, no batch-statistics pass) and a shape it wins on - judged on the rows of ONE
// batch of a grouped launch, so a batch meets the same kernel alone and in a group
bool gemm_x3_member_ok(const GemmP& p) {
    const bool ok4 = p.vec == 4 && (p.K % 4 == 0) && (p.Cin % 4 == 0) && (p.lda % 4 == 0) && x3_aligned16(p.A) && x3_aligned16(p.W) &&
                     (p.a_split % 4 == 0) && (p.a_gap % 4 == 0) && (p.taps == 1 || p.Cin >= XK) &&
                     (int64_t)((p.M + p.Tout - 1) / p.Tout) * p.Tin * p.lda * 4 < (1ll << 31) && (int64_t)p.N * (p.ldw ? p.ldw : p.K) * 4 < (1ll << 31);
    if (!ok4 || p.stats) return false;
    if (p.x3 & 2) return true;                                 // forced (operator tests run every addressable shape)
    const int m_unit = p.M / (p.x3_group > 1 ? p.x3_group : 1);
    const int64_t t_unit = (int64_t)((m_unit + XM - 1) / XM) * ((p.N + XN - 1) / XN);
    // short K (the post-net's first layer: K = 400; conv_last: K = 464) only pays with many rows per batch: 9600 x 512 x 400 runs 42 us against 52 on
    // the f32 kernel, 928 x 512 x 512 36 against 21; narrow outputs (the post-net's last layer: N = 80, five eighths of a 128-wide tile) likewise:
    // 76800 x 80 x 2560 runs 320 us against 490 (tools/time_gemm_shapes.py)
    return p.N >= 64 && (p.N >= 96 || t_unit >= 64) && p.K >= 384 && (p.K >= 1024 || t_unit >= 128);
}

bool gemm_x3_eligible(const GemmBatch& b) {
    int64_t tiles = 0;
    for (int i = 0; i < b.count; ++i) {
        const GemmP& p = b.p[i];
        if (!gemm_x3_member_ok(p)) return false;
        const int m_unit = p.M / (p.x3_group > 1 ? p.x3_group : 1);                    // rows of ONE batch of a grouped launch
        tiles += (int64_t)((m_unit + XM - 1) / XM) * ((p.N + XN - 1) / XN);
    }
    return tiles >= (b.count > 1 ? 150 : 100) || (b.p[0].x3 & 2);
}

#ifdef L2S_DIAG      // the stamped measurement builds exist in libl2s_diag.so only (include/l2s_diag.h l2s_op_gemm_x3_timeline)
static unsigned long long* g_x3_ts = nullptr;
static int g_x3_stamp_block = 0;
void gemm_x3_set_timeline(unsigned long long* ts, int block) { g_x3_ts = ts; g_x3_stamp_block = block; }
#endif

// Which tile: both give the same bits, so the choice is free per launch.  The wide tile needs N, K, Cin and the A split to fit its uniform K steps, and
// it pays when the launch runs in fewer "rounds" of 256 blocks x tile time: a wide tile takes ~1.7x a narrow one for twice the work (2 250 against
// 2 x 1 320 clk per 16 k), so 9600 x 512 (150 wide tiles, one round, against 300 narrow = two) wins and 7424 x 512 (116 against 232: one round each) loses;
// grouped launches whose members differ in K (the MultiHop convs) have a longer tail with the longer tile and need a clearer margin.
static bool x3_wide(const GemmBatch& b) {
    if (b.p[0].x3 & 4) return false;                          // mode bit 4 (option "gemm_x3" = 5, operator flag 4): the 128x128x32 tile everywhere
    int64_t wide = 0, narrow = 0;
    for (int i = 0; i < b.count; ++i) {
        const GemmP& p = b.p[i];
        if (p.N % WN != 0 || p.K % WK != 0 || p.a_split % WK != 0 || (p.taps > 1 && p.Cin % WK != 0)) return false;
        wide += (int64_t)((p.M + WM - 1) / WM) * (p.N / WN);
        narrow += (int64_t)((p.M + XM - 1) / XM) * ((p.N + XN - 1) / XN);
    }
    if (b.p[0].x3 & 2) return true;                           // forced (operator tests): the wide tile wherever it fits
    bool dma = true;
    for (int i = 0; i < b.count; ++i) dma = dma && b.p[i].W3 && b.p[i].ldw == 0;
    // grouped launches whose members all bring weight planes (the MultiHop convs): the DMA form's K step is 2 013 clk against 2 x 1 320
    const double cost_w = (double)((wide + 255) / 256) * (b.count > 1 ? (dma ? 1.6 : 2.0) : 1.7), cost_n = (double)((narrow + 255) / 256);
    return cost_w < cost_n;
}

// W [N][K] (K contiguous) -> its split-bf16 planes [K / 16][3 planes][N][16 k], the layout gemm_x3w_kernel<., true> fetches by LDS-DMA (a K step of one
// plane = N rows of 32 bytes); the same truncation split the staging waves apply, so both forms of the kernel multiply the same operand bits
__global__ __launch_bounds__(256) void gemm_planes_kernel(const float* __restrict__ W, int N, int K, unsigned char* __restrict__ planes) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int kq4 = K / 4;
    if (idx >= (int64_t)N * kq4) return;
    const int n = (int)(idx / kq4), k = 4 * (int)(idx - (int64_t)n * kq4);
    const X3Split sp = x3_split(*reinterpret_cast<const float4*>(W + (int64_t)n * K + k));
    const int64_t step = (int64_t)(k >> 4) * 3 * N;
    unsigned char* d = planes + ((step + n) * 16 + (k & 15)) * 2;
    *reinterpret_cast<uint2*>(d) = sp.hi;
    *reinterpret_cast<uint2*>(d + (int64_t)N * 32) = sp.mid;
    *reinterpret_cast<uint2*>(d + (int64_t)N * 64) = sp.lo;
}
int launch_gemm_planes(const float* W, int N, int K, void* planes, hipStream_t s) {
    L2S_REQUIRE(W && planes && N % WN == 0 && K % WK == 0, "gemm planes: N a multiple of 256, K of 16");
    const int64_t n = (int64_t)N * (K / 4);
    hipLaunchKernelGGL(gemm_planes_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, W, N, K, reinterpret_cast<unsigned char*>(planes));
    L2S_CHECK_HIP(hipGetLastError());
    return 0;
}

int launch_gemm_x3(const GemmBatch& b, hipStream_t s, const char* name) {
    L2S_REQUIRE(b.count >= 1 && b.count <= GEMM_MAX_GROUP && gemm_x3_eligible(b), "split-bf16 gemm: group not eligible");
    int maxM = 0, maxN = 0;
    for (int i = 0; i < b.count; ++i) {
        L2S_REQUIRE(b.p[i].taps * b.p[i].Cin == b.p[i].K, "gemm K = taps*Cin");
        maxM = b.p[i].M > maxM ? b.p[i].M : maxM;
        maxN = b.p[i].N > maxN ? b.p[i].N : maxN;
    }
    ProfScope ps(name, s);
    if (x3_wide(b)) {
        dim3 grid((maxN + WN - 1) / WN, (maxM + WM - 1) / WM, b.count);
        constexpr int LDS_BYTES = 2 * WSTAGE;                 // 110 592
        constexpr int LDS_BYTES_D = 2 * WSTAGED;              // 86 016 (the epilogue's 64-row sub-tile needs 65 536)
        bool dma = true;
        for (int i = 0; i < b.count; ++i) dma = dma && b.p[i].W3 && b.p[i].ldw == 0;      // every member brings pre-split weight planes
        static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_x3w_kernel<false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        L2S_CHECK_HIP(attr);
        static const hipError_t attr_d = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_x3w_kernel<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES_D);
        L2S_CHECK_HIP(attr_d);
#ifdef L2S_DIAG
        if (g_x3_ts) {
            static const hipError_t attr_t = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_x3w_kernel<true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
            L2S_CHECK_HIP(attr_t);
            static const hipError_t attr_td = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_x3w_kernel<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES_D);
            L2S_CHECK_HIP(attr_td);
            if (dma) hipLaunchKernelGGL((gemm_x3w_kernel<true, true>), grid, dim3(WTHREADS), LDS_BYTES_D, s, b, g_x3_ts, g_x3_stamp_block);
            else hipLaunchKernelGGL((gemm_x3w_kernel<true, false>), grid, dim3(WTHREADS), LDS_BYTES, s, b, g_x3_ts, g_x3_stamp_block);
        } else
#endif
        if (dma) {
            hipLaunchKernelGGL((gemm_x3w_kernel<false, true>), grid, dim3(WTHREADS), LDS_BYTES_D, s, b, (unsigned long long*)nullptr, 0);
        } else {
            hipLaunchKernelGGL((gemm_x3w_kernel<false, false>), grid, dim3(WTHREADS), LDS_BYTES, s, b, (unsigned long long*)nullptr, 0);
        }
        L2S_CHECK_HIP(hipGetLastError());
        return 0;
    }
    dim3 grid((maxN + XN - 1) / XN, (maxM + XM - 1) / XM, b.count);
    constexpr int LDS_BYTES = 2 * 6 * XPLANE;              // 122 880: two operand stages (one block per CU)
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_x3_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    L2S_CHECK_HIP(attr);
#ifdef L2S_DIAG
    if (g_x3_ts) {
        static const hipError_t attr_t = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_x3_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        L2S_CHECK_HIP(attr_t);
        hipLaunchKernelGGL(gemm_x3_kernel<true>, grid, dim3(512), LDS_BYTES, s, b, g_x3_ts, g_x3_stamp_block);
    } else
#endif
    {
        hipLaunchKernelGGL(gemm_x3_kernel<false>, grid, dim3(512), LDS_BYTES, s, b, (unsigned long long*)nullptr, 0);
    }
    L2S_CHECK_HIP(hipGetLastError());
    return 0;
}

}  // namespace l2s
