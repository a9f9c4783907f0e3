// Tiled fp32 GEMM on the CDNA4 matrix cores (v_mfma_f32_32x32x2_f32: exact f32, a k-ordered fmaf chain).
//
// Serves every "dense over channel-last rows" op of the hot path: the ShuffleNet pointwise convs, the
// decoder's Linear layers, and all Conv1d stacks (MultiHopConv K/V, Content.agg, Postnet) as implicit
// GEMMs over (B,T,C) sequences - reference/model/modules/decoder.py:107-271, shufflenetv2.py:42-104.
//
// Block = 256 threads = 4 waves (2x2), block tile 64(M) x 64(N) x 32(K); each wave owns a 32x32 tile
// (one f32x16 accumulator).  A and W tiles are staged through LDS row-major with K contiguous and a
// +4 float row pad (144-B rows: ds_read_b128 conflict-free).  Operand fetch uses a K permutation so one
// ds_read_b128 per operand feeds four MFMAs: within an 8-deep chunk, lane group g = lane>>5 holds
// k = 8c + 4g + e (e = 0..3) and MFMA e consumes element e of both operands.
// Global loads of tile t+1 are issued before the MFMAs of tile t (register prefetch).
#include "l2s_common.h"
#include "gemm_dev.h"

namespace l2s {

constexpr int BM = 64, BN = 64, BK = 32, LDS_LD = BK + 4;

// the fields the operand loaders need, copied once into SGPRs (read through the kernarg reference they are re-fetched by scalar loads
// inside the K loop, each behind its own s_waitcnt)
struct LoadP { int K, Cin, taps, Tin, lda, a_split, a_gap; };

template <int VEC>
__device__ __forceinline__ float4 load_a(const LoadP& p, const float* rowbase, bool rowvalid, int tbase, int k, int tap4, int ci4) {
    float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
    if (!rowvalid) return r;
    if (VEC == 4) {
        if (k >= p.K) return r;
        const int tap = tap4, ci = ci4;          // (tap, ci) of k, tracked incrementally by the caller: no division in the K loop
        int tin = tbase + tap;
        if (tin < 0 || tin >= p.Tin) return r;
        int col = ci + (ci >= p.a_split ? p.a_gap : 0);
        return *reinterpret_cast<const float4*>(rowbase + (int64_t)tin * p.lda + col);
    } else {
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            int kk = k + e;
            v[e] = 0.f;
            if (kk < p.K) {
                int tap = 0, ci = kk;
                if (p.taps > 1) { tap = kk / p.Cin; ci = kk - tap * p.Cin; }
                int tin = tbase + tap;
                if (tin >= 0 && tin < p.Tin) {
                    int col = ci + (ci >= p.a_split ? p.a_gap : 0);
                    v[e] = rowbase[(int64_t)tin * p.lda + col];
                }
            }
        }
        return make_float4(v[0], v[1], v[2], v[3]);
    }
}

template <int VEC>
__device__ __forceinline__ float4 load_w(const LoadP& p, const float* wrow, bool valid, int k) {
    float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
    if (!valid) return r;
    if (VEC == 4) {
        if (k >= p.K) return r;
        return *reinterpret_cast<const float4*>(wrow + k);
    } else {
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = (k + e < p.K) ? wrow[k + e] : 0.f;
        return make_float4(v[0], v[1], v[2], v[3]);
    }
}

template <int VEC, bool BF16 = false>
__global__ __launch_bounds__(256) void gemm_nt_kernel(const GemmBatch batch) {
    const GemmP& p = batch.p[blockIdx.z];
    // XCD-aware tile order: the hardware deals consecutive workgroups round-robin to the 8 XCDs (private L2s), so give each XCD a
    // contiguous run of tiles with the N index fastest - the column tiles that share an A row panel then meet in ONE L2 instead of
    // fetching it from the Infinity Cache eight times (post-net layer: 309 -> 295 us)
    int bx, by;
    {
        const int gx = gridDim.x, total = gx * gridDim.y;
        const int L = blockIdx.y * gx + blockIdx.x;
        const int xcd = L & 7, local = L >> 3;
        const int chunk = total >> 3, rem = total & 7;
        const int tile = xcd * chunk + (xcd < rem ? xcd : rem) + local;
        bx = tile % gx; by = tile / gx;
    }
    const int m0 = by * BM, n0 = bx * BN;
    if (m0 >= p.M || n0 >= p.N) return;

    __shared__ __attribute__((aligned(16))) float As[BM * LDS_LD];
    __shared__ __attribute__((aligned(16))) float Bs[BN * LDS_LD];

    const int tid = threadIdx.x;
    const int lr = tid >> 3, kq = (tid & 7) * 4;

    // per-thread row descriptors for the two A rows and two W rows this thread stages
    const float* arow[2];
    bool avalid[2];
    int atbase[2];
    const float* wrow[2];
    bool wvalid[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        int m = m0 + lr + 32 * j;
        avalid[j] = m < p.M;
        int mm = avalid[j] ? m : 0;
        int b = mm / p.Tout, t = mm - b * p.Tout + p.win_off;
        atbase[j] = t * p.stride - p.pad;
        arow[j] = p.A + (int64_t)b * p.Tin * p.lda;
        int n = n0 + lr + 32 * j;
        wvalid[j] = n < p.N;
        wrow[j] = p.W + (int64_t)(wvalid[j] ? n : 0) * (p.ldw ? p.ldw : p.K);
    }

    const int wave = tid >> 6, lane = tid & 63;
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 31, lg = lane >> 5;

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;

    const LoadP lp{p.K, p.Cin, p.taps, p.Tin, p.lda, p.a_split, p.a_gap};
    asm volatile("" ::"s"(lp.K), "s"(lp.Cin), "s"(lp.taps), "s"(lp.Tin), "s"(lp.lda), "s"(lp.a_split), "s"(lp.a_gap));
    const int nkt = (lp.K + BK - 1) / BK;
    // Operand addressing of the float4 path: buffer loads through two descriptors (A panel, weight matrix) whose hardware range check
    // returns zeros - no predication, no zero-fill, no 64-bit address arithmetic in the K loop.  This thread's k = kt*BK + kq maps to
    // (tap, ci), advanced by BK per tile (Cin >= BK or taps == 1, checked at launch); per staged row the byte offset of input frame
    // tbase + tap (or an out-of-range offset when that frame is padding), refreshed only when the tap changes.
    constexpr unsigned OOB = 0x80000000u;               // both extents are < 2 GiB (checked at launch)
    int tap = 0, ci = kq;
    if (lp.taps > 1) { tap = kq / lp.Cin; ci = kq - tap * lp.Cin; }
    const int nseq = (p.M + p.Tout - 1) / p.Tout;
    const __amdgpu_buffer_rsrc_t ra_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.A), 0, (int)((int64_t)nseq * lp.Tin * lp.lda * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rw_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.W), 0, (int)((int64_t)p.N * (p.ldw ? p.ldw : lp.K) * 4), 0x00020000);
    unsigned aoff[2], woff[2];
    auto set_tap = [&]() {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int tin = atbase[j] + tap;
            const bool ok = avalid[j] && tin >= 0 && tin < lp.Tin;
            aoff[j] = ok ? (unsigned)((arow[j] - p.A) + (int64_t)tin * lp.lda) * 4u : OOB;
        }
    };
    set_tap();
#pragma unroll
    for (int j = 0; j < 2; ++j) woff[j] = wvalid[j] ? (unsigned)((wrow[j] - p.W) + kq) * 4u : OOB;
    auto fetch4 = [&](int k, float4* ra, float4* rb) {
        const unsigned kbad = k < lp.K ? 0u : OOB;                                       // K is a multiple of 4: a quad is all in or all out
        const unsigned col = (unsigned)(ci + (ci >= lp.a_split ? lp.a_gap : 0)) * 4u;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            ra[j] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(ra_rs, (int)((aoff[j] | kbad) + (aoff[j] == OOB ? 0u : col)), 0, 0));
            rb[j] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rw_rs, (int)(woff[j] | kbad), 0, 0));
        }
    };
    float4 ra[2], rb[2];
    if (VEC == 4) fetch4(kq, ra, rb);
    else {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            ra[j] = load_a<VEC>(lp, arow[j], avalid[j], atbase[j], kq, 0, 0);
            rb[j] = load_w<VEC>(lp, wrow[j], wvalid[j], kq);
        }
    }

    for (int kt = 0; kt < nkt; ++kt) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            if (BF16) {
                *reinterpret_cast<uint2*>(reinterpret_cast<unsigned char*>(As) + (lr + 32 * j) * GD_BROW + kq * 2) = gd_pack4(ra[j]);
                *reinterpret_cast<uint2*>(reinterpret_cast<unsigned char*>(Bs) + (lr + 32 * j) * GD_BROW + kq * 2) = gd_pack4(rb[j]);
            } else {
                *reinterpret_cast<float4*>(&As[(lr + 32 * j) * LDS_LD + kq]) = ra[j];
                *reinterpret_cast<float4*>(&Bs[(lr + 32 * j) * LDS_LD + kq]) = rb[j];
            }
        }
        __syncthreads();
        if (kt + 1 < nkt) {
            const int k = (kt + 1) * BK + kq;
            if (VEC == 4) {
                ci += BK;
                if (lp.taps > 1 && ci >= lp.Cin) { ci -= lp.Cin; ++tap; set_tap(); }
#pragma unroll
                for (int j = 0; j < 2; ++j) woff[j] += wvalid[j] ? BK * 4u : 0u;
                fetch4(k, ra, rb);
            } else {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    ra[j] = load_a<VEC>(lp, arow[j], avalid[j], atbase[j], k, 0, 0);
                    rb[j] = load_w<VEC>(lp, wrow[j], wvalid[j], k);
                }
            }
        }
        if (BF16) {
            acc = gd_mma_tile_bf16(reinterpret_cast<const unsigned char*>(As), reinterpret_cast<const unsigned char*>(Bs), wm, wn, li, lg, acc);
        } else {
            const float* ap = &As[(wm * 32 + li) * LDS_LD + 4 * lg];
            const float* bp = &Bs[(wn * 32 + li) * LDS_LD + 4 * lg];
#pragma unroll
            for (int c = 0; c < BK / 8; ++c) {
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

    if (p.stats) {      // batch-statistics pass: per-column sum and sum of squares of this tile's rows; the raw product is parked on request
        if (p.stats_raw) {
            const int colr = n0 + wn * 32 + li;
            if (colr < p.N) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lg;
                    if (row < p.M) p.Zout[(int64_t)row * p.ldc + (int64_t)colr * p.c_cstride] = acc[r];
                }
            }
        }
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lg;
            if (row < p.M) { s1 += acc[r]; s2 += acc[r] * acc[r]; }
        }
        float* red = As;                                   // [2][wm 2][lg 2][64 cols]
        red[((0 * 2 + wm) * 2 + lg) * 64 + wn * 32 + li] = s1;
        red[((1 * 2 + wm) * 2 + lg) * 64 + wn * 32 + li] = s2;
        __syncthreads();
        if (tid < 128) {
            const int k = tid >> 6, c = tid & 63;
            if (n0 + c < p.N) {
                const float* q = red + k * 256 + c;
                p.stats[((int64_t)by * 2 + k) * p.N + n0 + c] = (q[0] + q[64]) + (q[128] + q[192]);
            }
        }
        return;
    }
    // epilogue: C/D layout of the 32x32 tile: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    const int col = n0 + wn * 32 + li;
    if (col >= p.N) return;
    const float sc = p.scale ? p.scale[col] : 1.0f;
    const float sh = p.shift ? p.shift[col] : 0.0f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lg;
        if (row < p.M) gemm_store(p, row, col, acc[r], sc, sh);
    }
}

// split-K finish: add the K slices in order, then the epilogue of the unsplit GEMM
__global__ __launch_bounds__(256) void gemm_splitk_finish_kernel(const GemmP p, const float* __restrict__ part, int ksplit) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)p.M * p.N) return;
    const int row = (int)(idx / p.N), col = (int)(idx - (int64_t)row * p.N);
    float v = 0.f;
    for (int i = 0; i < ksplit; ++i) v += part[(int64_t)i * p.M * p.N + idx];
    gemm_store(p, row, col, v, p.scale ? p.scale[col] : 1.0f, p.shift ? p.shift[col] : 0.0f);
}

// epilogue over parked raw products (launch_gemm_finish): thread = one output element
__global__ __launch_bounds__(256) void gemm_finish_kernel(const GemmBatch b) {
    const GemmP& p = b.p[blockIdx.y];
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)p.M * p.N) return;
    const int row = (int)(idx / p.N), col = (int)(idx - (int64_t)row * p.N);
    const float raw = p.Zout[(int64_t)row * p.ldc + (int64_t)col * p.c_cstride];
    gemm_store(p, row, col, raw, p.scale ? p.scale[col] : 1.0f, p.shift ? p.shift[col] : 0.0f);
}
int launch_gemm_finish(const GemmBatch& b, hipStream_t s, const char* name) {
    L2S_REQUIRE(b.count >= 1 && b.count <= GEMM_MAX_GROUP, "gemm group size");
    int64_t maxe = 0;
    for (int i = 0; i < b.count; ++i) {
        L2S_REQUIRE(b.p[i].Zout && b.p[i].win_T == 0 && !b.p[i].stats, "gemm finish: needs the parked raw product (Zout), plain rows");
        maxe = std::max<int64_t>(maxe, (int64_t)b.p[i].M * b.p[i].N);
    }
    ProfScope ps(name, s);
    hipLaunchKernelGGL(gemm_finish_kernel, dim3((unsigned)((maxe + 255) / 256), b.count), dim3(256), 0, s, b);
    L2S_CHECK_HIP(hipGetLastError());
    return 0;
}

GemmP gemm_plain(const float* A, int lda, const float* W, float* C, int ldc, int M, int N, int K) {
    GemmP p{};
    p.A = A; p.W = W; p.C = C;
    p.M = M; p.N = N; p.K = K;
    p.lda = lda; p.Tout = M > 0 ? M : 1; p.Tin = p.Tout; p.taps = 1; p.stride = 1; p.pad = 0; p.Cin = K;
    p.a_split = 1 << 30; p.a_gap = 0;
    p.act = ACT_NONE;
    p.ldc = ldc; p.c_cstride = 1; p.c_tr_T = 0;
    p.vec = 4;
    p.x3 = gemm_x3_mode();
    p.x3_group = gemm_x3_group();
    return p;
}

int& gemm_x3_mode() { static thread_local int mode = 0; return mode; }
int& gemm_x3_group() { static thread_local int group = 1; return group; }
bool& grouped_entry() { static thread_local bool grouped = false; return grouped; }
int& chains_hint() { static thread_local int chains = 1; return chains; }
int& gemm_bf16_mode() { static thread_local int mode = 0; return mode; }

static bool aligned16(const void* ptr) { return (reinterpret_cast<uintptr_t>(ptr) & 15u) == 0; }

int launch_gemm(const GemmBatch& b, hipStream_t s, const char* name) {
    L2S_REQUIRE(b.count >= 1 && b.count <= GEMM_MAX_GROUP, "gemm group size");
    int maxM = 0, maxN = 0;
    bool vec4 = true;
    for (int i = 0; i < b.count; ++i) {
        const GemmP& p = b.p[i];
        L2S_REQUIRE(p.M > 0 && p.N > 0 && p.K > 0, "gemm dims");
        L2S_REQUIRE(p.taps * p.Cin == p.K, "gemm K = taps*Cin");
        maxM = p.M > maxM ? p.M : maxM;
        maxN = p.N > maxN ? p.N : maxN;
        bool ok4 = p.vec == 4 && (p.K % 4 == 0) && (p.Cin % 4 == 0) && (p.lda % 4 == 0) && aligned16(p.A) &&
                   aligned16(p.W) && (p.a_split % 4 == 0) && (p.a_gap % 4 == 0) && (p.taps == 1 || p.Cin >= BK) &&
                   (int64_t)((p.M + p.Tout - 1) / p.Tout) * p.Tin * p.lda * 4 < (1ll << 31) && (int64_t)p.N * (p.ldw ? p.ldw : p.K) * 4 < (1ll << 31);
        vec4 = vec4 && ok4;
    }
    bool x3 = vec4;
    for (int i = 0; i < b.count; ++i) x3 = x3 && b.p[i].x3 != 0;
    if (x3 && gemm_x3_eligible(b)) return launch_gemm_x3(b, s, name);
    if (x3 && b.count > 1) {
        // a group with members that cannot run on the split-bf16 kernel (the k = 1 MultiHop convs: K = 512) used to keep ALL of it on the f32
        // kernel; the members that can (per member, on one batch's rows: the same choice alone and in a group) now go there as their own launch
        GemmBatch yes{}, no{};
        for (int i = 0; i < b.count; ++i) {
            if (gemm_x3_member_ok(b.p[i])) yes.p[yes.count++] = b.p[i];
            else no.p[no.count++] = b.p[i];
        }
        if (yes.count >= 1 && no.count >= 1 && gemm_x3_eligible(yes)) {
            if (launch_gemm_x3(yes, s, name)) return 1;
            return launch_gemm(no, s, name);
        }
    }
    dim3 grid((maxN + BN - 1) / BN, (maxM + BM - 1) / BM, b.count);
    ProfScope ps(name, s);
    const bool bf16 = gemm_bf16_mode() != 0;                 // training, option "train_bf16": bf16 operands, fp32 accumulation
    if (vec4 && bf16) hipLaunchKernelGGL((gemm_nt_kernel<4, true>), grid, dim3(256), 0, s, b);
    else if (vec4) hipLaunchKernelGGL((gemm_nt_kernel<4, false>), grid, dim3(256), 0, s, b);
    else if (bf16) hipLaunchKernelGGL((gemm_nt_kernel<1, true>), grid, dim3(256), 0, s, b);
    else hipLaunchKernelGGL((gemm_nt_kernel<1, false>), grid, dim3(256), 0, s, b);
    L2S_CHECK_HIP(hipGetLastError());
    return 0;
}

// K slice i of ksplit of a plain (1-tap) GEMM, raw product to `out` [M][N]; honours the two-segment A rows (a_split / a_gap)
static GemmP gemm_kslice(const GemmP& p, int i, int ksplit, float* out) {
    const int kc = p.K / ksplit, k0 = i * kc;
    const bool past = k0 >= p.a_split;
    GemmP q = gemm_plain(p.A + k0 + (past ? p.a_gap : 0), p.lda, p.W + k0, out, p.N, p.M, p.N, kc);
    if (!past && p.a_split < k0 + kc) { q.a_split = p.a_split - k0; q.a_gap = p.a_gap; }
    q.ldw = p.ldw ? p.ldw : p.K;
    q.vec = p.vec; q.x3 = p.x3; q.x3_group = p.x3_group;
    return q;
}

int launch_gemm_splitk_group(const GemmBatch& g, int ksplit, float* part, hipStream_t s, const char* name) {
    L2S_REQUIRE(ksplit >= 2 && g.count >= 1 && g.count * ksplit <= GEMM_MAX_GROUP, "split-K: at most 8 slices per launch");
    GemmBatch b{};
    float* pj = part;
    for (int j = 0; j < g.count; ++j) {
        const GemmP& p = g.p[j];
        L2S_REQUIRE(p.taps == 1 && !p.stats && p.win_T == 0 && p.K % (4 * ksplit) == 0 && p.a_split % 4 == 0 && p.a_gap % 4 == 0,
                    "split-K: plain GEMMs only, K divisible by 4*ksplit");
        for (int i = 0; i < ksplit; ++i) b.p[b.count++] = gemm_kslice(p, i, ksplit, pj + (int64_t)i * p.M * p.N);
        pj += (int64_t)ksplit * p.M * p.N;
    }
    if (launch_gemm(b, s, name)) return 1;
    pj = part;
    for (int j = 0; j < g.count; ++j) {
        const GemmP& p = g.p[j];
        const int64_t total = (int64_t)p.M * p.N;
        hipLaunchKernelGGL(gemm_splitk_finish_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, p, pj, ksplit);
        pj += (int64_t)ksplit * total;
    }
    L2S_CHECK_HIP(hipGetLastError());
    return 0;
}

int launch_gemm_splitk(const GemmP& p, int ksplit, float* part, hipStream_t s, const char* name) {
    GemmBatch g{};
    g.p[0] = p; g.count = 1;
    return launch_gemm_splitk_group(g, ksplit, part, s, name);
}

int launch_gemm_tapsplit(const GemmBatch& convs, float* part, hipStream_t s, const char* name) {
    GemmBatch b{};
    auto flush = [&]() -> int {
        if (b.count == 0) return 0;
        if (launch_gemm(b, s, name)) return 1;
        b = GemmBatch{};
        return 0;
    };
    float* pj = part;
    const float* parts[GEMM_MAX_GROUP] = {};
    for (int j = 0; j < convs.count; ++j) {
        const GemmP& p = convs.p[j];
        if (p.taps == 1) {                       // nothing to split: runs with its own epilogue next to the slices
            b.p[b.count++] = p;
            if (b.count == GEMM_MAX_GROUP && flush()) return 1;
            continue;
        }
        L2S_REQUIRE(p.a_split >= p.K && !p.stats && p.win_T == 0 && p.Cin % 4 == 0, "tap split: plain Conv1d layers only");
        parts[j] = pj;
        for (int i = 0; i < p.taps; ++i) {
            GemmP q = gemm_plain(p.A, p.lda, p.W + (int64_t)i * p.Cin, pj + (int64_t)i * p.M * p.N, p.N, p.M, p.N, p.Cin);
            q.ldw = p.K; q.Tout = p.Tout; q.Tin = p.Tin; q.stride = p.stride; q.pad = p.pad - i; q.vec = p.vec; q.x3 = p.x3; q.x3_group = p.x3_group;
            b.p[b.count++] = q;
            if (b.count == GEMM_MAX_GROUP && flush()) return 1;
        }
        pj += (int64_t)p.taps * p.M * p.N;
    }
    if (flush()) return 1;
    for (int j = 0; j < convs.count; ++j) {
        if (!parts[j]) continue;
        const GemmP& p = convs.p[j];
        const int64_t total = (int64_t)p.M * p.N;
        hipLaunchKernelGGL(gemm_splitk_finish_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, p, parts[j], p.taps);
    }
    L2S_CHECK_HIP(hipGetLastError());
    return 0;
}

int launch_gemm1(const GemmP& p, hipStream_t s, const char* name) {
    GemmBatch b{};
    b.p[0] = p;
    b.count = 1;
    return launch_gemm(b, s, name);
}

}  // namespace l2s
