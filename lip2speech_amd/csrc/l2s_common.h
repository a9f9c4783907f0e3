// Internal declarations shared by the HIP translation units of libl2s_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>

#define L2S_WAVE 64

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace l2s {

// ---------------------------------------------------------------- error handling
void set_error(const std::string& msg);
#define L2S_CHECK_HIP(expr)                                                                   \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess) {                                                               \
            l2s::set_error(std::string(#expr) + ": " + hipGetErrorString(_e));                \
            return 1;                                                                         \
        }                                                                                     \
    } while (0)
#define L2S_REQUIRE(cond, msg)                                                                \
    do {                                                                                      \
        if (!(cond)) {                                                                        \
            l2s::set_error(std::string("l2s: ") + (msg) + " [" #cond "]");                    \
            return 1;                                                                         \
        }                                                                                     \
    } while (0)

// ---------------------------------------------------------------- optional per-kernel event timing
void prof_begin(const char* name, hipStream_t s);
void prof_end(hipStream_t s);
struct ProfScope {
    hipStream_t s;
    ProfScope(const char* name, hipStream_t st) : s(st) { prof_begin(name, st); }
    ~ProfScope() { prof_end(s); }
};

// ---------------------------------------------------------------- run-time options (include/l2s.h "run-time options")
// Every model carries its own copy (l2s_model::opt, copied from the process defaults at l2s_model_create), so host threads that drive
// different models - or one model - never read a switch another thread is flipping.
struct Options {
    int fold = 1;               // "fold_step_weights": phase-merged 4-launch step vs the literal 6-phase step
    int graph = 0;              // "use_graph": replay the decode loop from a captured hipGraph
    int overlap_postnet = 0;    // "overlap_postnet": windowed post-net on a second stream under the decode loop
    int fuse_trunk = 1;         // "fuse_trunk": stride-1 ShuffleNet units as one fused kernel each
    int fuse_s2 = 1;            // "fuse_s2": stride-2 ShuffleNet units as one fused kernel each (needs fuse_trunk)
    int trunk_x3 = 1;           // "trunk_x3": the fused units' pointwise convs on the bf16 matrix cores through the exact three-way split (activations split once where
                                //             they are written to LDS, weights as pre-split operand planes); 0 = the f32-MFMA units
    int refresh_map = 0;        // "refresh_map": l2s_model_finalize also builds the device-side refresh map (training)
    int skinny_static = 0;      // "skinny_static": compile-time K-segment layouts in the batch-row kernels
    int skinny_sized = 1;       // "skinny_sized": batch-row instances sized for the launch's longest K
    int skinny_split = 2;       // "skinny_split": operand-load batches of the K <= 1536 instance
    int skinny_split8 = 1;      // "skinny_split8": the same for the K <= 1024 instance
    int rc_shape = 0;           // "skinny_rc": register-blocked batch-row blocks for >= 64 rows: 0 = by tile count, 11 = never, 21 / 22 / 42 = force RT x CT
    int rc_jb = 0;              // "skinny_rc_jb": operand batching of the register-blocked blocks: 0 = 4x2 blocks one chunk per batch and four batches in flight, smaller shapes two chunks per batch and two in flight; 2 / 4 = that many chunks per batch, two in flight, every shape; 15 = 4x2 with five in flight
    int gemm_x3_dma = 1;        // "gemm_x3_dma": constant Conv1d / Linear weights of the split-bf16 GEMMs as pre-split planes fetched by LDS-DMA (ConvW::W3)
    int lstm_x3 = 3;            // "lstm_x3": the decode step's LSTM launches on the bf16 matrix cores (exact three-way split, pre-split weight planes): 2 = eight-wave
                                //   blocks, 1 = four-wave blocks, 3 (default) = as 2, but the 4x2 blocks (>= 192 rows) as four-wave blocks of at most 256 registers - half
                                //   a compute unit, so that kernels of other launch chains run beside them (all the same bits), 0 = the f32 MFMA form
    int half_min_mts = 12;      // "half_min_mts": with "lstm_x3" = 3 and chains overlapping, an LSTM launch takes the half-CU 4x2 form from this many 16-row tiles
                                //   on (13: where 4x2 blocks give >= 224 blocks anyway; 12: also the 192-row launches of a six-batch group, whose 384 2x2
                                //   eight-wave blocks otherwise fill every CU for two rounds - the driver's 20-step command cuts into 7 + 7 + 6 batches:
                                //   3.14-3.18 -> 3.21-3.29 M in three A/B pairs; 8: no better there, worse at 128 / 160 rows); same bits
    int flat_half = 1;          // "flat_half": the flat first phase of the step on four-wave 2x1 / 2x2 blocks (at most 153 registers, 39 KB of LDS: two or
                                //   three per CU, up to 512 per launch) instead of eight-wave blocks that sit alone on their CU - so that other chains'
                                //   kernels run beside them; same bits
    int attn_lds = 1;           // "attn_lds": the step's attention blocks fetch keys / projected values by buffer loads, the values as 16-byte rows through LDS:
                                //   1 = at up to 128 rows per launch, 2 = always, 0 = never
    int hoist_vproj = 2;        // "hoist_vproj": the phase-merged step reads o = a @ V' with V' = V W_ap^T + b_ap computed once in the prologue: 2 = LSTM0 on
                                //   [content | prenet + o | h0] (K = 1024, the sum formed by the operand loader: the reference's own u = prenet + o), 1 = on
                                //   [content | prenet | o | h0] through a second copy of W_ih's u columns (K = 1280; every block form), 0 = a @ v through the
                                //   pre-multiplied W_ih W_ap (K = 1536)
    int skinny_flat = 1;        // "skinny_flat": launches with several GEMM groups at >= 64 rows run per-group block shapes in one flat grid of at most one
                                //   block per CU (the step's first phase), instead of one block shape for every group
    int rc_shape_multi = 0;     // "skinny_rc_multi": the same choice for launches that carry several GEMM groups (the step's first phase); 0 = as "skinny_rc"
    int gemm_x3 = 1;            // "gemm_x3": inference GEMMs / Conv1d stacks on the split-bf16 kernel (gemm_x3.hip) where eligible; 5 = its 128x128x32 tile only
    int frontend_x3 = 3;        // "frontend_x3": the inference front-end conv on the split-bf16 matrix path (frontend3d_x3_kernel); 2: its two-output-frames-per-block
                                //   form (frontend3d_x3p_kernel); 3 (default): the same with the next slab's staging interleaved between the MFMA groups of the
                                //   current one (frontend3d_x3q_kernel: double-buffered input planes, one barrier per slab; same bits: 0.593 -> 0.562 ms at 32 clips,
                                //   1.907 -> 1.832 at 128, 3.61 -> 3.59 at 256 - at 256 clips the kernel sits at the chip's clock under MFMA load)
    int train_bf16 = 0;         // "train_bf16": the training step's GEMMs / Conv1d stacks (forward and backward) with bf16 operands on the bf16 matrix cores
    int persist = 4;            // "persist_decode": the free-running decode loop of a single-batch call with at most this many clips (pdecode.hip: up to 4 clips
                                //   of <= 32 frames, two per launch) as persistent weight-stationary launches instead of four launches per step; 0 = never.  A
                                //   latency form: one call owns the chip, such launches are chained one after the other; grouped calls (l2s_*_multi) never take it
    int frontend_solo = 0;      // "frontend_solo" (diagnostic A/B): when chains overlap, the front-end conv takes ONE block per CU (an LDS pad) so that step kernels of other
                                //   chains (half-CU blocks) run beside it instead of waiting for its 530-us blocks to retire
    int trunk_chain = 1;        // "trunk_chain" (diagnostic A/B): the consecutive stride-1 units of a ShuffleNet stage as ONE launch (the map stays on chip between the
                                //   units; stages 2 and 3 - at 3x3 the per-unit launches are 5 % faster): 3 + 7 launches become 2, the trunk 3.82 -> 3.52 ms per 256 clips,
                                //   same bits; 0 = one launch per unit, 2 = the 3x3 stage as well
    int flat_xcd = 1;           // "flat_xcd" (diagnostic A/B): the flat first phase's block -> tile map XCD-affine (an XCD = one row half x one column quarter of a group:
                                //   weights through two L2s, activations through four, instead of one and eight): 22.3 -> 18.4 MB through the fabric per launch; same bits
    int attn_skip0 = 1;         // "attn_skip0" (diagnostic A/B): when chains overlap, the attention blocks fetch a frame's projected values only if its soft-max weight is
                                //   not exactly zero (28.4 -> 23.5 MB per launch; same bits); 2 = also for a chain alone (one more round trip in the block: slower there), 0 = never
    int infer_bf16 = 0;         // "infer_bf16": the bf16 leg of inference / evaluate: front-end conv, GEMMs and Conv1d stacks of encoder, prologue, post-net and
                                //   voice tower with bf16 operands (fp32 accumulation); the recurrent loops, the fused ShuffleNet units and all statistics stay fp32
};
int set_option_field(Options& o, const char* name, int value);    // 0 = ok, 1 = unknown name

enum Act { ACT_NONE = 0, ACT_RELU = 1, ACT_SILU = 2, ACT_PSINE = 3, ACT_PRELU = 4 /* backward only */ };

// ---------------------------------------------------------------- tiled fp32 MFMA GEMM (gemm_nt.hip)
// C[m, n] = epilogue( sum_k A(m,k) * W[n*K + k] ), A addressed as an implicit Conv1d over channel-last
// sequences: row m = (b, t), k = (tap, ci), element X[(b*Tin + t*stride + tap - pad) * lda + col(ci)].
struct GemmP {
    const float* A;
    const float* W;       // [N][K], K contiguous, k = tap*Cin + ci
    const float* scale;   // [N] or null (1)
    const float* shift;   // [N] or null (0)     v = acc*scale + shift
    const float* actw;    // [N] for ACT_PSINE
    const float* R1;      // optional addend after the activation: R1[(r1_mod ? m % r1_mod : m)*ldr1 + n]
    const float* R2;
    float* C;
    int M, N, K;
    int lda, Tout, Tin, taps, stride, pad, Cin;
    int a_split, a_gap;   // channel ci >= a_split reads column ci + a_gap (two-segment A rows)
    int act;
    int ldr1, r1_mod, ldr2, r2_mod;
    int r2_div;           // > 0: R2 row index = m / r2_div (a per-sequence row broadcast over its time steps)
    int ldc, c_cstride;   // C[m*ldc + n*c_cstride]
    int c_tr_T;           // > 0: C[((m / T)*N + n)*T + m % T]  (channel-first store per sequence)
    float* Zout;          // training: pre-activation value (after scale/shift), same addressing as C (row-major form only)
    int win_T, win_off;   // win_T > 0: rows cover the time window [win_off, win_off + Tout) of sequences of length win_T:
                          //   input frame = (win_off + t)*stride + tap - pad, output/residual row = b*win_T + win_off + t
    int vec;              // 4: float4 operand loads (K, Cin, lda, offsets multiples of 4); 1: scalar loads
    const float* mask;    // training: dropout multiplier mask[m*ldmask + n], applied after activation and addends
    int ldmask, mask_pre; //   (mask_pre: after the activation, before the addends)
    int ldw;              // row stride of W in floats (0: K) - a K slice of a wider matrix (split-K)
    const void* W3;       // W as pre-split bf16 planes [K / 16][3 planes][N][16 k] (launch_gemm_planes; N % 256 == 0, K % 16 == 0, ldw == 0) or null: the
                          //   128x256x16 split-bf16 tile then brings its weight operand in by LDS-DMA instead of load + split + ds_write
    int x3;               // bit 1: run on the split-bf16 kernel (gemm_x3.hip) when the whole launch group is eligible; bit 2: whatever its size
                          //   (operator tests); bit 4: the 128x128x32 tile only; set from gemm_x3_mode()
    int x3_group;         // batches sharing this launch (grouped inference): the size thresholds of the kernel choice look at M / x3_group,
                          //   so a batch meets the same kernels alone and in a group (results stay bit-identical)
    float* stats;         // training, batch-statistics BatchNorm: STATS PASS - nothing is stored; per-column sums of the raw product
                          //   (no scale/shift) go to stats[(blockIdx.y*2 + {0: sum, 1: sum of squares})*N + n]
    int stats_raw;        // STATS PASS with stats_raw: the raw product is also parked at the Zout addresses, and launch_gemm_finish runs the
                          //   epilogue over it once this batch's scale/shift exist - ONE pass over the product instead of two
};
inline int64_t gemm_stats_floats(int M, int N) { return (int64_t)((M + 63) / 64) * 2 * N; }
constexpr int GEMM_MAX_GROUP = 8;
struct GemmBatch {
    GemmP p[GEMM_MAX_GROUP];
    int count;
};
GemmP gemm_plain(const float* A, int lda, const float* W, float* C, int ldc, int M, int N, int K);
// split-bf16 GEMM (gemm_x3.hip): fp32 operands split into 3 bf16 planes on their way into LDS, six bf16 MFMAs per K step instead of the
// f32 MFMA chain.  gemm_plain() stamps new descriptors with the calling thread's current mode; the inference launch sequences open an
// X3Scope with their model's "gemm_x3" option, everything else (training, operator tests) stays on the f32 kernel unless it asks.
int& gemm_x3_mode();                     // thread-local
int& gemm_x3_group();                    // thread-local: batches per launch chain (1 outside l2s_inference_multi)
struct X3Scope { int prev; explicit X3Scope(int on) : prev(gemm_x3_mode()) { gemm_x3_mode() = on; } ~X3Scope() { gemm_x3_mode() = prev; } };
int& chains_hint();                     // thread-local: launch chains the CALLER keeps in flight on the device (l2s_set_thread_chains; 1 = this thread's calls have the chip to themselves)
bool& grouped_entry();                  // thread-local: inside an l2s_*_multi call, whatever G (such calls never take the persistent latency forms)
struct X3Group {
    int prev; bool prev_grouped;
    explicit X3Group(int g) : prev(gemm_x3_group()), prev_grouped(grouped_entry()) { gemm_x3_group() = g; grouped_entry() = true; }
    ~X3Group() { gemm_x3_group() = prev; grouped_entry() = prev_grouped; }
};
// bf16-operand mode of the TRAINING GEMMs (forward gemm_nt.hip and backward gemm_bwd.hip; option "train_bf16"): operands rounded to bf16
// (RNE) while they are staged into LDS, v_mfma_f32_32x32x16_bf16, fp32 accumulation, fp32 results.  Thread-local, opened by the l2s_train_*
// entry points; the recurrent loop, the front-end conv, BatchNorm statistics and every elementwise kernel stay fp32.
int& gemm_bf16_mode();
struct Bf16Scope { int prev; explicit Bf16Scope(int on) : prev(gemm_bf16_mode()) { gemm_bf16_mode() = on; } ~Bf16Scope() { gemm_bf16_mode() = prev; } };
bool gemm_x3_eligible(const GemmBatch& b);
bool gemm_x3_member_ok(const GemmP& p);               // one member's operands and shape (no launch-size threshold)
int launch_gemm_planes(const float* W, int N, int K, void* planes, hipStream_t s);      // planes: N * K * 6 bytes
int launch_gemm_x3(const GemmBatch& b, hipStream_t s, const char* name);
void gemm_x3_set_timeline(unsigned long long* ts, int block);      // non-null: launch the stamped measurement build (tools/gemm_x3_timeline.py)
// split-K for plain GEMMs whose 64x64 tiles are too few to fill the chip (M <= 128 rows in the content path, the B-row Linears):
// <= 8 K slices as ONE grouped launch writing raw partial products to part[slice][M][N], then one kernel that adds the slices in
// order and runs the usual epilogue of `p`. Deterministic. part: ksplit*M*N floats.
int launch_gemm_splitk(const GemmP& p, int ksplit, float* part, hipStream_t s, const char* name);
int launch_gemm_splitk_group(const GemmBatch& g, int ksplit, float* part, hipStream_t s, const char* name);   // count*ksplit <= 8; part: count*ksplit*M*N
// the same for a group of Conv1d layers whose tile counts are small and whose K = taps*Cin is long (Content.agg: 2-15 row tiles,
// K up to 3584): one K slice per TAP (a 1-tap conv with a shifted pad), all slices of all layers in grouped launches, then one finish
// kernel per multi-tap layer. part: sum over the multi-tap layers of taps*M*N floats.
int launch_gemm_tapsplit(const GemmBatch& convs, float* part, hipStream_t s, const char* name);
int launch_gemm(const GemmBatch& b, hipStream_t s, const char* name);
// train() mode, convs in front of a batch-statistics BatchNorm: after a stats pass with stats_raw = 1 (and bn_stats_finalize), the fused epilogue
// of `b` (scale/shift, Zout, activation, masks, addends, any store form) element by element over the parked raw products - the same gemm_store
// on the same accumulator values as a second GEMM pass, bit for bit, for an M x N elementwise pass instead of the product
int launch_gemm_finish(const GemmBatch& b, hipStream_t s, const char* name);
int launch_gemm1(const GemmP& p, hipStream_t s, const char* name);

// ---------------------------------------------------------------- backward GEMMs (gemm_bwd.hip)
enum BwdMode { BWD_DX = 0, BWD_DW = 1 };
struct BwdGemmP {
    int mode;
    const float* A; int lda;     // dZ
    const float* B; int ldb;     // DX: forward-layout weight [Nout][taps*Cin];  DW: forward input X (B,Tx,Cin)
    float* C; int ldc;           // DX: dX;  DW: dWp [Nout][taps*Cin]
    int M, N, K;                 // C is M x N, reduction length K
    int Tx, Tz, taps, stride, pad, padp, Nout, Cin;
    int c_T; int64_t c_seq_stride;   // c_T > 0: C row = (m / c_T) * c_seq_stride + (m % c_T) * ldc
    float alpha; int accumulate;
    int vec;                     // set by launch_gemm_bwd: operands allow 16-byte loads
    int ksplit; int64_t c_split_stride;   // DW: reduction split over gridDim.z into partial matrices (launch_gemm_bwd_splitk)
};
BwdGemmP bwd_dx(const float* dZ, int ldz, const float* Wp, float* dX, int ldx, int B, int Tz, int Tx, int Nout, int Cin, int taps, int pad, bool accumulate);
BwdGemmP bwd_dw(const float* dZ, int ldz, const float* X, int ldx, float* dWp, int B, int Tz, int Tx, int Nout, int Cin, int taps, int stride, int pad, bool accumulate);
int launch_gemm_bwd(const BwdGemmP& p, hipStream_t s, const char* name);
int64_t gemm_bwd_splitk_floats(const BwdGemmP& p, int splits);
int launch_gemm_bwd_splitk(const BwdGemmP& p, int splits, float* partials, hipStream_t s, const char* name);
// weight gradient (split-K as above) and input gradient of one 1x1 layer in ONE launch (they share dZ and are independent)
int launch_gemm_bwd_dw_dx(const BwdGemmP& pdw, int splits, float* partials, const BwdGemmP& pdx, hipStream_t s, const char* name);

// Backward of a fused GEMM epilogue  y = act(z) [+ residual],  z = conv * s + shift  (s, shift = eval-mode BatchNorm and/or bias):
//   dpre = dy * act'(z);  dconv = dpre * s;  per-column sums  r0 = sum dpre,  r1 = sum dpre * (z - beta)/gamma,
//   r2 = sum dy * sin(z) (PSine) or sum dy * min(z, 0) (PReLU).  Two-stage column reduction, deterministic (train_decoder.hip).
constexpr int AB_RS = 1024;   // row splits: narrow layers (24-116 channels) have one column block, the splits are what fills the chip (four blocks per CU:
                              // a block keeps only 8 KB of loads in flight, and at one block per CU the pass ran at a quarter of the HBM rate)
struct ActBwdP {
    const float* dy; const float* z; float* dconv;     // [rows][ld]
    int ld_dy, ld_z, ld_dconv;                          // 0 = C
    int64_t rows; int C;
    int act;                                            // ACT_NONE / ACT_SILU / ACT_PSINE / ACT_RELU / ACT_PRELU
    const float* actw;                                  // PSine w / PReLU slope
    const float* scale;                                 // BN scale s (null = 1)
    const float* gamma; const float* beta;              // BN affine (null = no BN)
    float* partials;                                    // [AB_RS][3][C]
    int cs_dy, co_dy, cs_z, co_z;                       // column stride (0 = 1) and offset of dy / z (shuffled channel positions)
};
int act_bwd(const ActBwdP& p, float* d_shift, float* d_gamma, float* d_actw, float* d_convbias, bool accumulate, hipStream_t s, float* totals = nullptr);

// ---- batch-statistics BatchNorm (train mode), train_kernels.hip.  The forward kernels keep their fused "acc*scale + shift" epilogues:
// a STATS PASS of the same kernel first reduces the raw conv output per channel, bn_stats_finalize turns the sums into this batch's
// scale/shift (and updates the running statistics like nn.BatchNorm, momentum 0.1, unbiased running variance), then the real pass runs
// with them.  Backward: act_bwd as for eval statistics (dconv = dpre*scale, totals r0 = sum dpre, r1 = sum dpre*xhat), then
// bn_train_fix subtracts scale*(r0/n + xhat*r1/n).
struct BnLayer {
    const float* gamma; const float* beta; float* rmean; float* rvar; const float* conv_bias;   // canonical tensors (bias may be null)
    float* scale; float* shift;                                                                  // outputs [C] (kept on the tape)
    int C;
};
int bn_stats_finalize(const float* partials, int nblk, int blk_stride, int64_t count, const BnLayer& L, float momentum, hipStream_t s);
int bn_train_fix(float* dconv, int ld_dconv, const float* z, int ld_z, int cs_z, int co_z, const float* gamma, const float* beta, const float* scale,
                 const float* totals, int64_t rows, int C, hipStream_t s);
// depthwise 3x3 stats pass: partials[(rs*2 + k)*C + c], DWS_RS row splits
constexpr int DWS_RS = 1024;
int launch_dwconv_stats(const float* in, int N, int Hi, int Wi, int ldi, int ci_off, int C, int stride, const float* w9, float* partials, hipStream_t s,
                        float* raw_out = nullptr, int ldo = 0, int co_off = 0);     // raw_out: also park the raw conv output (then launch_bn_apply instead of a second conv pass)
int launch_bn_apply(float* x, int64_t rows, int C, int ld, int co_off, const float* scale, const float* shift, hipStream_t s, bool fused = false);

// ---------------------------------------------------------------- encoder kernels (encoder_kernels.hip)
struct FrontendW {          // device pointers into the weight blob
    const float* w;         // [15 slabs (ci*5+kt)][50 (kh*7+kw, padded)][32 (co, padded)]
    const float* scale;     // [24] BN scale
    const float* shift;     // [24]
    const float* slope;     // [24] PReLU
    const float* w3;        // split-bf16 operand planes of w for frontend3d_x3_kernel: [15 slabs][4 steps][3 planes][32 co][16 taps, 48-byte rows];
                            //   channel rows 24-31 are zeros (frontend3d_x3p_kernel takes the zero rows of an absent frame from row 24)
                            //   (null: f32 MFMA kernel)
    int pair = 0;           // with w3: two output frames per block (frontend3d_x3p_kernel) - set by the callers from option "frontend_x3" >= 2
    int solo = 0;           // with pipe: one block per CU (diagnostic option "frontend_solo" with chains overlapping)
    int pipe = 0;           // with pair: the next slab's staging interleaved with the current slab's MFMAs (frontend3d_x3q_kernel) - option "frontend_x3" == 3
    const float* w1;        // the same as ONE plane rounded to nearest even: [15 slabs][4 steps][32 co][16 taps, 48-byte rows] (set by the callers
                            //   of launch_frontend only for a model with "infer_bf16"; takes precedence over w3)
};
// where the clips of a launch live: clip b is clip (b % per) of the (per,3,T,H,W) tensor p[b / per] - the G batches of a grouped pass
// (l2s_inference_multi) stay where their caller put them
constexpr int MAX_GROUP = 8;       // = L2S_MAX_GROUP (include/l2s.h)
struct FrameSrc { const float* p[MAX_GROUP]; int per; };
int launch_frontend(const FrontendW& w, const FrameSrc& video, int B, int T, int H, int W, float* out, hipStream_t s, float* zout = nullptr);
inline int launch_frontend(const FrontendW& w, const float* video, int B, int T, int H, int W, float* out, hipStream_t s, float* zout = nullptr) {
    FrameSrc f{}; f.p[0] = video; f.per = B;
    return launch_frontend(w, f, B, T, H, W, out, s, zout);
}
// batch-statistics pass of the front-end conv (training): partials[(block*2 + k)*24 + ch], *nblocks blocks
int launch_frontend_stats(const FrontendW& w, const float* video, int B, int T, int H, int W, float* partials /*[blocks][2][24]*/, int* nblocks, hipStream_t s, float* raw_out = nullptr);
// raw_out (NF,H/2,W/2,24): the statistics pass also parks the raw conv map; launch_bn_apply on it + launch_frontend_pool then replace the second conv pass
int launch_frontend_pool(const float* z, const float* slope, int NF, int Hc, int Wc, float* out, hipStream_t s);

// data boundary: packed uint8 RGB clips (clip i = frames[i] x H x W x 3 bytes at packed + offsets[i]) -> video (B,3,T,H,W) fp32,
// /255 then ImageNet mean/std, clips shorter than T zero-padded; offsets / frames are HOST arrays
constexpr int MAX_COLLATE_CLIPS = 64;        // per launch (the clip table travels in the kernel arguments)
int launch_normalise_pad(const uint8_t* packed, const int64_t* offsets, const int* frames, int B, int T, int H, int W, float* video, hipStream_t s);
// depthwise 3x3, pad 1, channel-last: in (N,Hi,Wi,ldi) channels [ci_off, ci_off+C) -> out (N,Ho,Wo,ldo) at co_off
int launch_dwconv(const float* in, int N, int Hi, int Wi, int ldi, int ci_off, int C, int stride,
                  const float* w9 /*[9][C]*/, const float* scale, const float* shift,
                  float* out, int ldo, int co_off, hipStream_t s);
// fused stride-1 ShuffleNet unit (encoder_kernels.hip)
struct ShuffleS1P {
    const float* x; float* out;                              // (NF, h, h, 2*half) channel-last
    const float* w1f; const float* s1; const float* b1;      // pw1: frag16 [pad16(half)][Kpad], BN scale/shift
    const float* wd; const float* sd; const float* bd;       // dw: [9][half], BN scale/shift
    const float* w2f; const float* s2; const float* b2;      // pw2
    int NF, h, half, Kpad, F;
    const void* w1p; const void* w2p;                        // pw1 / pw2 as pre-split bf16 operand planes (launch_su_planes) or null: both set -> the split-bf16 unit
};
int64_t su_planes_bytes(int N, int K);
int launch_su_planes(const float* W, int N, int K, void* out, hipStream_t s);      // [N][K] fp32 -> [ceil(N/16)][pad32(K)/32][3 planes][64 lanes] 16 bytes
int launch_shuffle_s1(const ShuffleS1P& p, hipStream_t s);
constexpr int S1_CHAIN_MAX = 7;
int launch_shuffle_s1_chain(const ShuffleS1P* units, int n, hipStream_t s);      // n consecutive stride-1 units of one stage in ONE launch (the map stays on chip between them); same bits
void shuffle_set_timeline(unsigned long long* ts, int h);     // non-null: launch the stamped measurement build (tools/fused_unit_timeline.py)
// fused stride-2 ShuffleNet unit (encoder_kernels.hip): banch1 (dw s2 -> pw) and banch2 (pw -> dw s2 -> pw) of one strip of Ro
// output rows per block; the full-resolution pw1 map (124 MB at B=32 in stage 2) never leaves the CU
struct ShuffleS2P {
    const float* x; float* out;                              // (NF, h, h, cin) -> (NF, ho, ho, 2*half), channel-last
    const float* wd1; const float* sd1; const float* bd1;    // banch1 dw: [9][cin], BN scale/shift
    const float* wb1f; const float* sb1; const float* bb1;   // banch1 pw: frag16 [pad16(half)][Kin]
    const float* w1f; const float* s1; const float* b1;      // banch2 pw1: frag16 [pad16(half)][Kin]
    const float* wd; const float* sd; const float* bd;       // banch2 dw: [9][half]
    const float* w2f; const float* s2; const float* b2;      // banch2 pw2: frag16 [pad16(half)][Kh]
    int NF, h, ho, cin, half, Kin, Kh, Ro;
    const void* wb1p; const void* w1p; const void* w2p;      // the three pointwise weights as pre-split bf16 operand planes (launch_su_planes) or null
};
int launch_shuffle_s2(const ShuffleS2P& p, hipStream_t s);
// out[r*ldo + off_o + c*cs_o] = in[r*ldi + off_i + c]
int launch_copy_cols(const float* in, int ldi, int off_i, float* out, int ldo, int off_o, int cs_o,
                     int64_t rows, int cols, hipStream_t s);
// x (NF, P, C) -> mean over P -> L2 normalise over C -> vis[f*ldv + c]; emb (B,E) tiled into vis[f*ldv + C + e]
int launch_pool_norm_cat(const float* x, int NF, int P, int C, const float* emb, int E, int T,
                         float* vis, int ldv, float* feat /*optional (NF,C)*/, hipStream_t s);

// ---------------------------------------------------------------- skinny (batch-row) MFMA kernels (skinny.hip)
// "frag16" layout of a row-major X[R][K] (R padded to 16, K multiple of 16):
//   F[(rt*(K/16) + c)*256 + l*4 + e] = X[16*rt + (l&15)][16*c + 4*(l>>4) + e]      (l = lane 0..63)
// One float4 per lane per 16-deep K chunk = exactly the A (or B) operands of four v_mfma_f32_16x16x4_f32.
__host__ __device__ inline int64_t frag16_index(int row, int k, int K) {
    int rt = row >> 4, i = row & 15, c = k >> 4, g = (k >> 2) & 3, e = k & 3;
    return ((int64_t)(rt * (K >> 4) + c) * 64 + (g * 16 + i)) * 4 + e;
}

enum SkinnyEpi {
    SK_PLAIN = 0,      // out[row*ldo + n] = act(v)                       (+ optional pos/add rows)
    SK_FRAG = 1,       // out frag16 (K = N) = act(v) (+add)
    SK_LSTM = 2,       // LSTM cell: rows of W permuted to (unit, gate); updates h (frag), c (frag)
    SK_MEL = 3,        // fc_out(+stop row): writes mel[b][step][n], y frag16, stop[b][step]
};
struct SkinnySeg { const float* a; int nchunks; };   // one K segment of A in frag16 layout (K = 16*nchunks)
struct SkinnyP {
    SkinnySeg seg[4];
    int nseg;
    const float* a_sum;    // optional second frag16 source of segment 1 (same shape): the segment's operand is seg[1].a + a_sum, added by the loader
                           //   (the decode step's u = prenet + attention_proj(a @ v), decoder.py:421; straight-line four-wave blocks only)
    int layout;            // set by launch_skinny: index of a compile-time segment layout the kernel has an instance for (0 = general path)
    const float* W;        // packed frag16 of the [Npad][K] weight, K = sum of segments
    const void* W3;        // optional: the same weight as three bf16 planes for the split-bf16 LSTM blocks ([tile][K slice 8][chunk pair][plane][lane] 16 B;
                           //   skx_planes_kernel); with it and option "lstm_x3" the launch runs on the bf16 matrix cores
    const float* bias;     // [Npad] (permuted order for SK_LSTM)
    const float* actw;     // [N] psine weights
    const float* add;      // optional plain [B][ld_add] added after activation (SK_PLAIN/SK_FRAG)
    int ld_add;
    const float* addrow;   // optional [N] row added after activation (positional encoding of this step)
    float* out;            // SK_PLAIN: [B][ldo]; SK_FRAG: frag16 with K = N
    int ldo;
    int B, N, K;           // N = valid output columns
    int act;
    int epi;
    // SK_LSTM
    const float* pre;      // optional precomputed input gates: pre[b*ld_pre + gate*H + unit]
    int64_t ld_pre;
    const float* c_in;     // frag16 (K = H) previous cell
    float* c_out;          // frag16
    float* h_out;          // frag16 (K = h_out_K) at column offset h_out_off
    int h_out_K, h_out_off;
    float* h_seq;          // optional plain: h_seq[b*ld_hseq + unit]
    int64_t ld_hseq;
    float* h_plain;        // optional second plain copy
    int ld_hplain;
    int H;
    // SK_MEL
    float* mel; int64_t ld_mel_b;      // mel[b*ld_mel_b + n] (already offset to this step)
    float* stop; int64_t ld_stop_b;    // stop[b*ld_stop_b]   (already offset to this step)
    const float* stop_const;           // [B]
    float* yfrag;                      // frag16, K = 80
};
constexpr int SKINNY_MAX_GROUP = 4;
struct SkinnyBatch { SkinnyP p[SKINNY_MAX_GROUP]; int ntiles[SKINNY_MAX_GROUP]; int count; };
int launch_skinny(const SkinnyBatch& b, hipStream_t s, const char* name, const Options& o = Options());
int launch_skx_planes(const float* Wf, int ntiles, int K, void* out, hipStream_t s);      // bf16 planes of a packed LSTM weight (SkinnyP::W3): ntiles * K * 16 * 6 bytes
bool skinny_sum_supported(const Options& o);           // launches with SkinnyP::a_sum need the straight-line four-wave blocks: default operand batching, no stamped build
void attn_set_timeline(unsigned long long* ts);        // non-null: the stamped attention kernel (tools/attn_timeline.py)
void skinny_set_flat_timeline(unsigned long long* ts);   // non-null: the stamped build of the step's flat first phase (tools/flat_timeline.py)
void skinny_set_timeline(unsigned long long* ts);      // non-null: launch the stamped measurement build (tools/skinny_timeline.py)

int launch_probe(int kind, int blocks, int n_per_block, const float* in, float* out, hipStream_t s);      // libl2s_diag.so only, like every *_set_timeline above
void skinny_set_flat_timeline(unsigned long long* ts);
int set_stamp_log(unsigned long long* log, long long cap);      // the block-stamp log of the step kernels (skinny.hip; libl2s_diag.so only)

// ---------------------------------------------------------------- decoder helper kernels (decoder_kernels.hip)
struct AttnP {
    const float* q; int ldq;          // [B][512]
    const float* k;                   // [B][T][512]
    const float* v;                   // [B][T][512]
    const float* vp;                  // optional [B][T][256] = V W_ap^T + b_ap: then av_frag receives o = a @ V' (frag16, K = 256) instead of a @ v
    const float* tau;                 // device scalar
    float* av_frag;                   // frag16, K = 512
    float* attn_out; int64_t ld_attn_b; int attn_logits;   // optional [b*ld + t]
    const float* qc; int ldqc;        // [B][256]
    const float* ckey;                // [B][m][256]
    const float* cval;                // [B][m][256]
    const float* tau_c;
    float* cc_frag;                   // frag16, K = 256
    int B, T, m;
};
// attention role + second prenet layer in one grid (skinny.hip)
int launch_step_attn(const AttnP& at, const SkinnyP& pre2, int pre2_tiles, hipStream_t s, int lds_values = 1, int skip0 = 0);
// adaptive average pooling of up to 5 channel-last maps into a concatenated (B, m, nmaps*C) buffer
struct PoolCatP { const float* x[5]; int L[5]; int ld[5]; int nmaps; int B, m, C; float* out; };
int launch_pool_cat(const PoolCatP& p, hipStream_t s);
// rows softmax: z = softmax((l+g)/tau) into zpad [rows][ldz] (zero padded), dis = softmax(l) (optional)
int launch_gumbel_softmax(const float* logits, const float* gumbel, int rows, int n, float tau,
                          float* z, int ldz, float* dis, hipStream_t s);
// plain [B][K] -> frag16 (rows padded with zeros); K multiple of 16; optional broadcast of a single row
int launch_to_frag(const float* x, int ldx, int B, int K, float* frag, int Kfrag, int koff, int broadcast_row, hipStream_t s);
int launch_fill(float* p, int64_t n, float v, hipStream_t s);
// frag16 -> plain: out[b*ldo + ooff + k] = frag[b][k]
int launch_from_frag(const float* frag, int Kfrag, int B, int K, float* out, int ldo, int ooff, hipStream_t s);
// dst[(b*T + t)*ldd + c] = src[b*lds + c]
int launch_tile_rows(const float* src, int lds, float* dst, int ldd, int B, int T, int C, hipStream_t s);
// (B,S,C) -> (B,C,S)
int launch_transpose_bsc(const float* in, int B, int S, int C, float* out, hipStream_t s);
int launch_output_lengths(const float* stop, int B, int S, int64_t* lengths, hipStream_t s);
int launch_frame_window(const float* audio, int B, int N, int L, int n_fft, int hop, const float* window, float* frames, hipStream_t s);
int launch_power(const float* spec, int lds, int64_t rows, int nf, float* power, int ldp, hipStream_t s);
// stop_const[b] = dot(ecell[b], w[512:1024]) + bias
int launch_stop_const(const float* ecell, const float* w_tail, const float* bias, int B, float* out, hipStream_t s);

}  // namespace l2s
