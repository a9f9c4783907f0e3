/*
 * l2s_diag.h - the DIAGNOSTIC surface of the MI355X-native Lip2Speech hot path (libl2s_diag.so).
 *
 * libl2s_diag.so is built from the sources of libl2s_hip.so with -DL2S_DIAG (lip2speech_amd/csrc/Makefile).  It exports the whole product
 * ABI of l2s.h (same kernels, same launch code) PLUS what only measurements and operator tests need:
 *   - operator-level entry points (one GEMM / Conv1d / front-end conv) for the operator parity tests,
 *   - chain microbenches and launch-floor probes,
 *   - stamped ("timeline") builds of the kernels and the block-stamp log behind the overlap proof of profiles/,
 *   - the run-time switches that only choose between block forms of the same arithmetic (measured-and-rejected forms kept for A/B timing),
 *   - the test hook of the persistent loop's give-up path (environment variable L2S_TEST_PDECODE_STARVE).
 * None of this is in libl2s_hip.so, and the package (lip2speech_amd/native.py lib(), the callers, bench.py's timed region) never loads this
 * library: tools/ set L2S_LIB to it, tests bind it through native.diag().  An l2s_model created by one library may be handed to the other
 * (same struct, same process heap, same device context).
 */
#ifndef L2S_DIAG_H
#define L2S_DIAG_H

#include "l2s.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- diagnostic run-time options (l2s_set_option / l2s_model_set_option of libl2s_diag.so accept these on top of l2s.h's; the product library
 * answers "unknown option").  Defaults are the forms the product runs; every other value selects a form that measured slower or equal:
 *   "fuse_trunk"        (1)  stride-1 ShuffleNet units as one fused kernel each; 0 = pw/dw/pw/copy launches
 *   "fuse_s2"           (1)  stride-2 ShuffleNet units as one fused kernel each (needs fuse_trunk); 0 = dw/pw + pw/dw/pw launches
 *   "overlap_postnet"   (0)  l2s_inference: windowed post-net on a second stream under the decode loop
 *   "gemm_x3_dma"       (1)  constant weights of the split-bf16 GEMMs as pre-split bf16 planes by LDS-DMA; 0 = load + split + LDS store in the staging waves
 *   "hoist_vproj"       (2)  attention_proj applied to the VALUES once per clip in the prologue: 2 = LSTM0 on [content | prenet + o | h0] (K = 1024), 1 = on
 *                            [content | prenet | o | h0] (K = 1280), 0 = a @ v through the pre-multiplied W_ih W_ap (K = 1536)
 *   "attn_lds"          (1)  attention blocks with the projected values staged through LDS: 1 = at up to 128 rows and whenever chains overlap, 2 = always, 0 = never
 *   "flat_half"         (1)  the step's first launch on four-wave half-CU blocks when chains overlap; 2 = always, 0 = never
 *   "half_min_mts"      (12) with "lstm_x3" = 3 and chains overlapping: an all-LSTM launch takes the half-CU 4x2 form from this many 16-row tiles on
 *   "trunk_chain"       (1)  the stride-1 units of ShuffleNet stages 2 and 3 as ONE launch per stage (the map stays on chip between the units); 2 = stage 4 too, 0 = one launch per unit
 *   "frontend_solo"     (0)  when chains overlap, the front-end conv one block per CU (an LDS pad) so that other chains' step kernels run beside it: measured -8 %, off
 *   "flat_xcd"          (1)  the step's first launch with an XCD-affine block -> tile map (an XCD = one row half x one column quarter of a group); 0 = row-major
 *   "attn_skip0"        (1)  when chains overlap, attention blocks fetch only the projected-value rows whose soft-max weight is not exactly zero; 2 = always, 0 = never
 *   "skinny_flat"       (1)  multi-group batch-row launches at >= 128 rows as one flat grid of per-group block shapes; 0 = one shape for all groups
 *   "skinny_rc"         (0)  batch-row kernels at >= 64 rows: 0 = by tile count, 11 = 1x1 blocks only, 21 / 22 / 42 = force RT x CT tiles;
 *   "skinny_rc_multi"   (0)  the same for launches that carry several GEMM groups;  "skinny_rc_jb" (0): operand batching of those blocks (2 / 4 / 15 / 28 / 44)
 *   "skinny_static"     (0)  compile-time K-segment layouts in the 16x16 batch-row kernels;  "skinny_sized" (1): instances sized for the launch's longest K;
 *   "skinny_split"      (2) / "skinny_split8" (1): operand loads of the K <= 1536 / K <= 1024 instance in this many batches */

/* ---- operator-level entry points (the operator parity tests of tests/test_gpu_parity.py) -------- */
/* C[M,N] = act((A[M,K] @ Wt[N,K]^T) * scale[N] + shift[N]);  act: 0 none, 1 relu, 2 silu, 3 sin(x)*actw[n] */
int l2s_op_gemm(const float* A, const float* Wt, const float* scale, const float* shift, const float* actw,
                float* C, int M, int N, int K, int act, void* stream);
/* Conv1d over channel-last sequences as an implicit GEMM: X (B,Tin,Cin), Wp (Cout, taps*Cin) with
 * k = tap*Cin + ci, out (B,Tout,Cout), Tout = (Tin + 2*pad - taps)/stride + 1 */
int l2s_op_conv1d(const float* X, const float* Wp, const float* scale, const float* shift, const float* actw,
                  float* out, int B, int Tin, int Cin, int Cout, int taps, int stride, int pad, int act,
                  void* stream);
/* the same two operators with flags: bit 0 = run on the split-bf16 kernel (fp32 operands split into 3 bf16 planes, six bf16 MFMAs per
 * K step; eligible shapes only - otherwise the f32 kernel runs); bit 1 = bf16 operands (round to nearest even on the way into LDS, one
 * bf16 MFMA per K step, fp32 accumulation); bit 2 (with bit 0) = the split-bf16 kernel's 128x128x32 tile instead of its default 128x256x16 one
 * (same bits out; kept for A/B timing); bit 3 (with bit 0) = the weight operand as pre-split bf16 planes fetched by LDS-DMA - what a model with
 * "gemm_x3_dma" runs in its post-net - derived per call into scratch memory the library owns (N % 256 == 0, K % 16 == 0, else ignored; same bits out) */
int l2s_op_gemm_ex(const float* A, const float* Wt, const float* scale, const float* shift, const float* actw, float* C, int M, int N,
                   int K, int act, int flags, void* stream);
int l2s_op_conv1d_ex(const float* X, const float* Wp, const float* scale, const float* shift, const float* actw, float* out, int B,
                     int Tin, int Cin, int Cout, int taps, int stride, int pad, int act, int flags, void* stream);
/* backward of l2s_op_conv1d (no scale/shift/activation): dZ (B,Tout,Cout), X (B,Tin,Cin), Wp (Cout, taps*Cin) ->
 * dX (B,Tin,Cin) (stride 1 only; may be NULL) and dWp (Cout, taps*Cin) (may be NULL) */
int l2s_op_conv1d_bwd(const float* dZ, const float* X, const float* Wp, float* dX, float* dWp, int B, int Tin, int Cin, int Cout, int taps,
                      int stride, int pad, void* stream);
/* fused Conv3d(3->24,5x7x7,s(1,2,2),p(2,3,3)) + BN + PReLU + MaxPool(1,3,3)/s(1,2,2)/p(0,1,1) of the model:
 * video dev (B,3,T,H,W) -> out dev (B*T, H/4, W/4, 24) channel-last */
int l2s_op_frontend(l2s_model* m, const float* video, int B, int T, int H, int W, float* out, void* stream);
/* average duration (us) of the decoder LSTM-cell kernel over a chain of n_pairs x {layer 0, layer 1} launches bracketed by ONE pair
 * of HIP events on `stream` (bench.py's roofline figure; synchronises) */
int l2s_op_lstm_cell_chain(l2s_model* m, int B, int n_pairs, void* ws, int64_t ws_bytes, void* stream, double* avg_us);
/* launch-floor probe: n dependent launches of an empty kernel (kind 0) or of a kernel in which each of `blocks` 512-thread
 * blocks streams n_per_block x 8 KiB from `in` (kind 1) - the cost model of a latency-bound decode phase (tools/launch_floor.py) */
int l2s_op_launch_chain(int kind, int n_launches, int blocks, int n_per_block, const float* in, float* out, void* stream);
/* measurement: with ts_dev != NULL every batch-row ("skinny") launch runs a stamped build of the same kernel - thread 0 of each block
 * writes 8 x 64-bit 100 MHz wall-clock stamps (entry, parameters in SGPRs, loads issued, first operands landed, MFMAs done, after the
 * reduction barrier, after the gate barrier, stores drained) to ts_dev[block*8 ..]; NULL restores the production kernel */
int l2s_op_skinny_timeline(void* ts_dev);
/* the same for the attention blocks of the step's second launch: 8 x uint64 per block (entry, requests issued, q visible, logits, after the barrier,
   weights visible, stored) of the 100 MHz wall clock.  tools/attn_timeline.py */
int l2s_op_attn_timeline(void* ts_dev);
/* and for the step's first launch (the flat grid of per-group block shapes, at >= 128 rows): 8 stamps per block as for l2s_op_skinny_timeline; the last
   such launch leaves its stamps.  tools/flat_timeline.py */
int l2s_op_flat_timeline(void* ts_dev);
/* persistent decode loop (option "persist_decode"): ts_dev = [256 workgroups][16] uint64 stamps of step `step` (100 MHz clock), or NULL to stop */
int l2s_op_pdecode_timeline(void* ts_dev, int step);
/* measurement build of the split-bf16 GEMM: lane 0 of each of the eight waves of block `block` stamps the shader clock per K tile
   ([12 waves][96 K tiles][8 slots] uint64); NULL switches it off again.  tools/gemm_x3_timeline.py */
int l2s_op_gemm_x3_timeline(void* ts_dev, int block);
/* measurement: the same for the fused stride-1 ShuffleNet units of spatial size h (12, 6 or 3; the last such launch leaves its stamps) - 10 x 64-bit words per block to ts_dev[block*10 ..]: 8 stamps (entry,
 * input in LDS, after the barrier, pw1 done, depthwise taps done, depthwise written, pw2 done, stores drained), HW_ID, XCC_ID.  h = -24 / -12 / -6: the
 * stride-2 unit whose INPUT map is that size (9 stamps: entry, input issued, after the barrier, banch1 depthwise, banch2 pw1, banch2 depthwise, banch1 pw,
 * banch2 pw2, stores drained; block = frame * strips + strip).  ([wave][96 K tiles][8 slots] for the GEMM hook above.) */
int l2s_op_fused_unit_timeline(void* ts_dev, int h);
/* measurement: n back-to-back launches of the decode step's attention kernel alone on the state of l2s_decoder_prologue (zero queries): whether a
 * clip's K / V survive in its XCD's L2 between launches when nothing else runs in between (tools/attn_l2_probe.py; workspace: l2s_workspace_bytes) */
int l2s_op_step_attn_chain(l2s_model* m, float* state, int B, int T, int n_launches, void* ws, int64_t ws_bytes, void* stream);
/* the same chain issued alternately on two streams (two independent dependency chains): does a second chain hide the launch floor? */
int l2s_op_launch_chain2(int kind, int n_launches, int blocks, int n_per_block, const float* in, float* out, void* stream_a, void* stream_b);
/* Block-stamp log (the overlap proof of profiles/rNN_overlap_stamps.txt): with log_dev != NULL every block of the decode step's kernels - LSTM launches,
 * first phase, attention - appends one record {entry stamp, exit stamp, tag} of 3 x uint64 (100 MHz constant clock, s_memrealtime; tag = kernel kind
 * 1 LSTM / 2 first phase / 3 attention in bits 60-63, the low 48 bits of a per-chain operand pointer below) to log_dev[1 + 3 * slot ..], slot taken
 * from the atomic counter at log_dev[0]; capacity in records; records past it are dropped (the counter keeps counting).  NULL switches it off. */
int l2s_op_stamp_log(void* log_dev, int64_t capacity);

#ifdef __cplusplus
}
#endif
#endif /* L2S_DIAG_H */
