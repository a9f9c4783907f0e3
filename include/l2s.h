/*
 * l2s.h - C-ABI of the MI355X-native Lip2Speech hot path (libl2s_hip.so).
 *
 * This is the drop-in boundary.  The reference has no native code:
 * its hot path is the Python call chain
 *
 *     Lip2Speech.inference            /root/reference/model/model.py:43-59
 *       VideoExtractor.forward        /root/reference/model/modules/video.py:76-87
 *       Decoder.inference             /root/reference/model/modules/decoder.py:382-444
 *     Lip2Speech.forward (eval)       /root/reference/model/model.py:23-40
 *       Decoder.forward               /root/reference/model/modules/decoder.py:320-379
 *
 * and a binding for this library replaces the bodies of exactly those methods
 * (INTEGRATION.md shows the ctypes stub a maintainer of the reference would add).
 *
 * Conventions
 *   - plain pointers and sizes only; no framework types.  `stream` is a hipStream_t
 *     passed as void* (NULL = the default stream).  Every entry point only enqueues
 *     work on `stream`; nothing synchronises the device.
 *   - all tensors are fp32, contiguous; "dev" = device memory, "host" = host memory.
 *   - the caller owns every buffer, including the workspace; the library allocates
 *     nothing persistent except the packed weight blob owned by an l2s_model, freed by
 *     l2s_model_destroy.
 *   - every function returns 0 on success, non-zero on error; l2s_last_error() gives
 *     the message for the calling thread.
 *   - lengths are ignored exactly as in the reference (no key masking, zero-padded
 *     frames are encoded and attended like real ones; decoder.py:320-379).
 */
#ifndef L2S_H
#define L2S_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct l2s_model l2s_model;

/* fixed geometry of the path (reference/hparams.py, decoder.py:285-318, video.py:55-72) */
#define L2S_N_MELS      80
#define L2S_D_MODEL     512
#define L2S_D_FEAT      768
#define L2S_D_EMB       256
#define L2S_D_VIS       1024
#define L2S_VOCAB       501
#define L2S_MAX_STEPS   300

int         l2s_abi_version(void);
const char* l2s_last_error(void);

/* ---- weights: replaces nn.Module.load_state_dict for encoder.* / decoder.* keys -------------------
 * A model may hold the encoder.* keys, the decoder.* keys, or both (net.encoder / net.decoder are usable on
 * their own in the reference); a stage whose half is absent returns an error.
 * (the on-disk format is the reference's state_dict; demo.py:30-38, evaluate.py:73-77).
 * set_tensor copies `numel` floats from host memory under the checkpoint key (e.g.
 * "decoder.K.0.conv.3.0.weight"); finalize validates that every key of the path is present with the
 * right element count, derives BatchNorm scale/shift, re-lays the weights out for the kernels
 * and uploads one device blob. */
int l2s_model_create(l2s_model** out);
int l2s_model_set_tensor(l2s_model* m, const char* key, const float* host_data, int64_t numel);
int l2s_model_finalize(l2s_model* m, void* stream);
int l2s_model_destroy(l2s_model* m);

/* ---- sizes ---------------------------------------------------------------------------------------- */
/* number of content slots min_T of Content.encode (decoder.py:239-246): min over the strided branches */
int     l2s_min_T(int T);
/* bytes of caller-provided device workspace for one call at these sizes (S = decode steps) */
int64_t l2s_workspace_bytes(int B, int T, int H, int W, int S);
/* floats in the decoder state buffer produced by l2s_decoder_prologue */
int64_t l2s_state_floats(int B, int T);
/* float offset of a field inside the state buffer; layouts:
 *   L2S_ST_K      (B,T,512)   keys,   k[b][t][c]  (the reference holds (B,512,T))
 *   L2S_ST_V      (B,T,512)   values
 *   L2S_ST_CKEY   (B,m,256)   content keys   (reference: self.key (B,256,m))
 *   L2S_ST_CVAL   (B,m,256)   content values (reference: self.value)
 *   L2S_ST_ECELL  (B,512)     encoder_cell
 *   L2S_ST_H / L2S_ST_C   decoder LSTM state, 2 layers, fragment layout
 *   L2S_ST_ENC    (B,T,512)   encoder_outputs after encoder_proj + site + residual
 *   L2S_ST_VP     (B,T,256)   V' = values W_ap^T + b_ap: attention_proj (decoder.py:420) applied to the values once, so that the step's
 *                             a @ V' IS attention_proj(a @ v) (the attention weights sum to one) */
enum { L2S_ST_K = 0, L2S_ST_V = 1, L2S_ST_CKEY = 2, L2S_ST_CVAL = 3, L2S_ST_ECELL = 4,
       L2S_ST_H = 5, L2S_ST_C = 6, L2S_ST_ENC = 7, L2S_ST_STOPC = 8, L2S_ST_VP = 9 };
int64_t l2s_state_offset(int B, int T, int field);

/* ---- stages ---------------------------------------------------------------------------------------- */
/* Data boundary on the device (datasets/lrw/dataset.py:83-86,123-146 + datasets/__init__.py:7-46, the model-facing half): the decoded
 * uint8 RGB frames of B clips, packed back to back in ONE device buffer (clip i = frames[i] x H x W x 3 bytes at packed_u8 + offsets[i],
 * offsets multiples of 4; both tables are HOST arrays of B entries), become the model's input video dev (B,3,T,H,W) fp32:
 * x/255, then (x - mean_c)/std_c with the ImageNet constants, clips shorter than T zero-padded.  Same fp32 operations in the same order
 * as the reference's host transforms: bit-identical, at a quarter of the PCIe bytes. */
int l2s_normalise_pad_frames(const uint8_t* packed_u8, const int64_t* offsets, const int32_t* frames, int B, int T, int H, int W,
                             float* video, void* stream);

/* VideoExtractor.forward (video.py:76-87): video dev (B,3,T,H,W) -> feat dev (B,T,768), L2-normalised.
 * H = W in {88, 96}. */
int l2s_encoder_fwd(l2s_model* m, const float* video, int B, int T, int H, int W,
                    float* feat, void* ws, int64_t ws_bytes, void* stream);

/* model.py:52-55: vis[b][t] = cat(feat[b][t] (768), emb[b] (256)) -> dev (B,T,1024) */
int l2s_build_visual(const float* feat, const float* emb, int B, int T, float* vis, void* stream);

/* Decoder prologue (decoder.py:383-410 / 321-351): vis dev (B,T,1024) = the `encoder_outputs` argument
 * of Decoder.forward/inference, emb dev (B,256) = its `face_features[:, 0]`, gumbel dev (B*min_T,501) =
 * the noise F.gumbel_softmax would draw (decoder.py:257).  Writes the state buffer and content_dis dev
 * (B*min_T,501) (softmax of the content logits, the 6th output of Decoder.forward; may be NULL). */
int l2s_decoder_prologue(l2s_model* m, const float* vis, const float* emb, const float* gumbel,
                         int B, int T, float* state, float* content_dis,
                         void* ws, int64_t ws_bytes, void* stream);

/* The autoregressive loop (decoder.py:412-429 / 353-375), S steps from the state buffer.
 *   teacher      dev (B,S,80) or NULL: frame fed at step i when teacher_mask[i] != 0
 *                (= cat(BOS, mels)[:, i], decoder.py:349,357)
 *   teacher_mask host (S) bytes or NULL: the scheduled-sampling decisions, made by the caller
 *   mel          dev (B,S,80)  pre-postnet frames, channel-last
 *   stop         dev (B,S)     stop-token logits
 *   attn         dev (B,S,T) or NULL; post-softmax weights, or tau*q.k logits if attn_logits != 0
 *                (inference() returns the former, forward() the latter) */
int l2s_decode_steps(l2s_model* m, float* state, int B, int T, int S,
                     const float* teacher, const uint8_t* teacher_mask,
                     float* mel, float* stop, float* attn, int attn_logits,
                     void* ws, int64_t ws_bytes, void* stream);

/* Postnet + residual (decoder.py:143-156, 438-439): mel dev (B,S,80) -> mel_post dev (B,80,S) in the
 * reference's layout; mel_cf dev (B,80,S) or NULL additionally receives the transposed pre-postnet mel. */
int l2s_postnet(l2s_model* m, const float* mel, int B, int S, float* mel_post, float* mel_cf,
                void* ws, int64_t ws_bytes, void* stream);

/* SpeakerEncoder.inference (model/modules/audio.py:131-150; called at demo.py:84): audio dev (B, n_samples) 16 kHz ->
 * 40-band mel power spectrogram (n_fft 400, hop 160, hann, centre/reflect, HTK, no log) -> 3 x LSTM(256) ->
 * Linear(last hidden) -> ReLU -> L2 normalise -> emb dev (B,256).  Needs the speaker_encoder.* keys in the model. */
int64_t l2s_speaker_workspace_bytes(int B, int n_samples);
int l2s_speaker_encoder_fwd(l2s_model* m, const float* audio, int B, int n_samples, float* emb,
                            void* ws, int64_t ws_bytes, void* stream);

/* ---- vocoder + metric of evaluate.py -------------------------------------------------------------
 * The reference vocodes the predicted mels with torchaudio 0.9.0 (datasets/spectograms.py:76-95: exp -> InverseMelScale ->
 * GriffinLim, 256 iterations each) and scores them with pystoi 0.3.3 (evaluate.py:41-45: stoi(gt, pred, fs, extended=True)).
 * Both packages are absent from the build image: these entry points run the published algorithms as restated in
 * lip2speech_amd/datasets/spectrograms.py and lip2speech_amd/metrics.py - PARITY UNPINNED against the packages themselves.
 * The random start iterates of both vocoder stages are INPUTS (the reference draws them with torch.rand).
 *
 * l2s_inverse_mel: torchaudio.transforms.InverseMelScale.forward.  mel dev (N, n_mels, L): power mel, or log-mel when log_input
 *   (spectral_de_normalize = exp is then applied on the fly); fb dev (n_freqs, n_mels) with fb_nnz non-zero entries (exact count,
 *   the caller knows its filterbank; at most 2048); init dev (N*L, n_freqs), row n*L + l = the start spectrum of frame l of clip n.
 *   The N clips are N / rows_per_call independent calls (the SGD's 1/(rows*L) gradient scale, the loss mean and the two stopping
 *   rules - loss < 1e-5, |loss - previous loss| < 1e-8, the update of the stopping iteration still applied - are per call).
 *   spec dev (N, n_freqs, L) >= 0.  loss_per_iter dev (calls, iters) or NULL; iters_run dev int (calls) or NULL.
 * l2s_griffin_lim: torchaudio.functional.griffinlim (power 2, rand_init replaced by init_angles dev (N, n_freqs, L, 2) used as
 *   given, momentum 0.99): power_spec dev (N, 513, L) -> wave dev (N, hop*(L-1)).  n_fft = win_length = 1024, hop = 256, L <= 121.
 * l2s_estoi: pystoi.stoi(clean, pred, fs, extended=True) per clip: clean / pred dev (N, n_samples); fir dev = the polyphase
 *   resampling filter of scipy.signal.resample_poly(x, up, down) INCLUDING its leading zero pad (n_fir taps, already scaled by up),
 *   n_pre_remove / n_resampled as resample_poly computes them, or fir = NULL when fs is already 10 kHz; band_lo_hi_host = HOST
 *   array of 2 x 15 ints, first / last+1 bin of each one-third octave band; score dev (N). */
int64_t l2s_inverse_mel_workspace_bytes(int N, int L, int n_mels, int n_freqs, int rows_per_call, int iters);
int l2s_inverse_mel(const float* mel, int log_input, const float* fb, int fb_nnz, const float* init, int N, int L, int n_mels, int n_freqs,
                    int rows_per_call, int iters, float* spec, float* loss_per_iter, int* iters_run, void* ws, int64_t ws_bytes, void* stream);
int64_t l2s_griffin_lim_workspace_bytes(int N, int L);
int l2s_griffin_lim(const float* power_spec, const float* init_angles, int N, int L, int n_fft, int hop, int iters, float momentum,
                    float* wave, void* ws, int64_t ws_bytes, void* stream);
int64_t l2s_estoi_workspace_bytes(int N);
int l2s_estoi(const float* clean, const float* pred, int N, int n_samples, const float* fir, int n_fir, int up, int down, int n_pre_remove,
              int n_resampled, const int* band_lo_hi_host, float* score, void* ws, int64_t ws_bytes, void* stream);

/* decoder.py:429-435: lengths[b] = first i+1 with stop logit > 0, else S.  lengths dev (B) int64. */
int l2s_output_lengths(const float* stop, int B, int S, int64_t* lengths, void* stream);

/* Lip2Speech.inference with a supplied speaker embedding (model.py:43-59), all stages on `stream`:
 * mel_post dev (B,80,S), lengths dev (B) int64, attn dev (B,S,T) or NULL. */
int l2s_inference(l2s_model* m, const float* video, const float* emb, const float* gumbel,
                  int B, int T, int H, int W, int S,
                  float* mel_post, int64_t* lengths, float* attn,
                  void* ws, int64_t ws_bytes, void* stream);

/* Grouped inference ("advance G independent batches per launch"): G batches of B clips each - separate tensors, wherever the caller's
 * loader put them - run as rows g*B .. g*B+B-1 of ONE launch chain on ONE weight blob: the 300 x 4 step launches, the BiLSTM recurrence and
 * every GEMM are issued once per G batches instead of once per batch, and from M = 64 rows on the batch-row kernels use register-blocked
 * 2x1 / 2x2 / 4x2 tiles (half the operand traffic per row; at M = 128 the LSTM launches are MFMA-bound).  Replaces G passes through the
 * reference's loop (model/modules/decoder.py:412-435) over G DataLoader batches (demo.py:60-90 / evaluate.py:22-51 iterate them one by one).
 * Every kernel of the path is row-independent, so each batch's mel / lengths / attention are bit-identical to l2s_inference on that batch.
 * video[g] dev (B,3,T,H,W), emb[g] dev (B,256), gumbel[g] dev (B*l2s_min_T(T),501) - host arrays of G device pointers;
 * mel_post dev (G*B,80,S), lengths dev (G*B) int64, attn dev (G*B,S,T) or NULL - batch g is the g-th slice of B rows. */
#define L2S_MAX_GROUP 8
int64_t l2s_workspace_bytes_multi(int G, int B, int T, int H, int W, int S);
int l2s_inference_multi(l2s_model* m, int G, const float* const* video, const float* const* emb, const float* const* gumbel,
                        int B, int T, int H, int W, int S,
                        float* mel_post, int64_t* lengths, float* attn,
                        void* ws, int64_t ws_bytes, void* stream);

/* Lip2Speech.forward(..., tf_ratio) in eval() mode (model/model.py:23-40 + model/modules/decoder.py:320-379) - what evaluate.py:32-38 runs on
 * every batch at tf_ratio = 1 - as ONE launch chain: encoder, prologue, S = mels.shape[2] steps, post-net.
 *   teacher      dev (B,S,80) or NULL: cat(BOS, mels)[:, i] (decoder.py:349), fed at the steps whose teacher_mask byte is set
 *   teacher_mask host (S) bytes or NULL: the caller's scheduled-sampling draws (decoder.py:355-357); both or neither
 *   mel_cf       dev (B,80,S) or NULL  pre-postnet mel, the reference's layout (outputs[0])
 *   mel_post     dev (B,80,S)          (outputs[1])
 *   stop         dev (B,S)             stop-token logits (outputs[2] without its trailing 1)
 *   attn_logits  dev (B,S,T) or NULL   tau * q.k, PRE-softmax (outputs[4]; train.py:244 applies the softmax itself)
 *   content_dis  dev (B*min_T,501) or NULL  (outputs[5])
 * Workspace: l2s_workspace_bytes(B,T,H,W,S).  Same kernels as the staged calls l2s_encoder_fwd .. l2s_postnet: bit-identical to them. */
int l2s_forward_eval(l2s_model* m, const float* video, const float* emb, const float* gumbel, int B, int T, int H, int W, int S,
                     const float* teacher, const uint8_t* teacher_mask, float* mel_cf, float* mel_post, float* stop, float* attn_logits,
                     float* content_dis, void* ws, int64_t ws_bytes, void* stream);
/* The grouped form (see l2s_inference_multi): G batches of evaluate.py's loop (evaluate.py:32-38 iterates them one by one) as rows
 * g*B .. g*B+B-1 of ONE launch chain; per batch bit-identical to l2s_forward_eval.  video / emb / gumbel / teacher: host arrays of G device
 * pointers (teacher NULL = free-running).  The batches of a group share S and teacher_mask - at tf_ratio = 1 the mask is empty, so any G
 * batches of one shape group.  Outputs are (G*B, ...) tensors, batch g = the g-th slice of B rows (content_dis: of B*min_T rows).
 * Workspace: l2s_workspace_bytes_multi(G,B,T,H,W,S). */
int l2s_forward_eval_multi(l2s_model* m, int G, const float* const* video, const float* const* emb, const float* const* gumbel,
                           const float* const* teacher, const uint8_t* teacher_mask, int B, int T, int H, int W, int S, float* mel_cf,
                           float* mel_post, float* stop, float* attn_logits, float* content_dis, void* ws, int64_t ws_bytes, void* stream);

/* ---- training-side primitives of the data-parallel step (train.py:102-104,172-193; train_utils/losses.py:69-77) ----------
 * scratch: l2s_train_scratch_bytes() of device memory.  Reductions are two-stage fp64 (deterministic). */
int64_t l2s_train_scratch_bytes(void);
/* losses5 dev = {mel MSE, 10 * post-net mel MSE, gate BCE-with-logits, KLD(content_dis || uniform), sum}; mel/mel_post/mel_target are
 * (B,80,S) channel-first as Decoder.forward returns them, stop/gate (B,S), content_dis (R,501).  d* (may be NULL) receive the
 * gradients of the summed loss. */
int l2s_loss(const float* mel, const float* mel_post, const float* mel_target, const float* stop, const float* gate_target,
             const float* content_dis, int B, int S, int R, float* losses5, float* dmel, float* dmel_post, float* dstop, float* ddis,
             void* scratch, void* stream);
/* norm_out dev [1] = ||grads||_2 over a flat gradient range (torch.nn.utils.clip_grad_norm_'s total norm) */
int l2s_grad_norm(const float* grads, int64_t n, void* scratch, float* norm_out, void* stream);
/* torch.optim.AdamW(amsgrad=True) on a flat range, fused with gradient averaging and clipping:
 * g_eff = grads * grad_mul * min(1, max_norm / (grad_norm[0] * grad_mul + 1e-6)); grad_norm dev [1] or NULL (no clipping).
 * grad_mul = 1/world_size after a sum all-reduce.  step counts from 1. */
int l2s_adamw_amsgrad_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, float* max_exp_avg_sq, int64_t n,
                           float lr, float beta1, float beta2, float eps, float weight_decay, int step, const float* grad_norm,
                           float grad_mul, float max_norm, void* stream);

/* ---- training path of the model (forward with tapes + backward of encoder, prologue, loop, post-net) ---------------------------
 * l2s_train_bind: device pointers of a parameter in its canonical (checkpoint) layout and of its gradient slot (may be NULL), by key.
 * The backward entry points write parameter gradients into the bound slots (overwrite, not accumulate). */
int l2s_train_bind(l2s_model* m, const char* key, float* param_dev, float* grad_dev);
/* BatchNorm behaviour of the l2s_train_* entry points: batch_stats = 1 -> nn.Module.train() semantics (normalise with the statistics of
 * this batch, update the BOUND running_mean / running_var in place with `momentum`, unbiased running variance; per process - no cross-rank
 * synchronisation, like the single-device reference); 0 (default) -> running statistics (eval()). */
int l2s_train_set_bn(l2s_model* m, int batch_stats, float momentum);
/* After an optimizer step: rebuild the packed blob ON THE DEVICE from the bound tensors (l2s_train_bind; bind the BatchNorm running
 * statistics and the other buffers as well, grad_dev = NULL).  Needs l2s_set_option("refresh_map", 1) before l2s_model_finalize: finalize
 * then records, for every blob float that copies a checkpoint element verbatim, where it comes from; BatchNorm folds and bias sums are
 * recomputed by small kernels; the phase-merged step weights (fp64 products) are marked stale and l2s_decode_steps / l2s_inference use
 * the literal step until the next l2s_model_finalize.  Replaces the host re-pack (~190 ms) by ~1 ms of device work. */
int l2s_train_refresh_weights(l2s_model* m, void* stream);

/* Post-net forward with a tape (decoder.py:143-156, eval-mode BatchNorm statistics) and its backward:
 * mel dev (B,S,80) -> mel_post dev (B,80,S);  dmel_post dev (B,80,S) -> dmel dev (B,S,80) is ACCUMULATED into.
 * drop: NULL (no dropout) or the five dropout multipliers (0 or 1/(1-p), p = 0.5; decoder.py:152,154) channel-last and back to back:
 * layers 0..3 (B*S,512) each at offset l*B*S*512, layer 4 (B*S,80) at offset 4*B*S*512 - explicit inputs, like the Gumbel noise. */
int64_t l2s_train_postnet_tape_floats(int B, int S);
int64_t l2s_train_postnet_ws_bytes(int B, int S);
int l2s_train_postnet_fwd(l2s_model* m, const float* mel, int B, int S, float* tape, float* mel_post, const float* drop, void* stream);
int l2s_train_postnet_bwd(l2s_model* m, const float* mel, const float* dmel_post, int B, int S, float* tape, float* dmel, const float* drop,
                          void* ws, int64_t ws_bytes, void* stream);

/* Stage 2: the autoregressive loop with a tape and its back-propagation through time (decoder.py:353-375; the literal 6-phase
 * step, eval-mode statistics, no dropout).  `state` is the buffer of l2s_decoder_prologue.  Forward: mel dev (B,S,80), stop dev (B,S),
 * attn_logits dev (B,S,T) (also read by the backward).  Backward: dmel dev (B,S,80) = total gradient of the pre-postnet frames,
 * dstop dev (B,S); wbuf = l2s_train_steps_weights_floats() floats filled by l2s_train_steps_pack_weights (transposed step weights packed on
 * the device from the bound canonical parameters - repack after every optimizer step).  Outputs: parameter gradients into the bound
 * slots, dk / dv dev (B,T,512), dckey / dcval dev (B,min_T,256), dh_init dev (2,B,512), de_c dev (B,512).
 * Train-mode dropout sites of the loop as explicit multiplier inputs (0 or 1/(1-p); NULL = site off), the same tensors for fwd and bwd:
 * drop_prenet dev (S,B,256) after the first prenet PSine (p 0.2, decoder.py:308); drop_attn dev (S,B,T) on the attention logits (p 0.1,
 * :363 - the returned attn_logits are the dropped ones, as in the reference); drop_rnn dev (S,B,512) on h0 as input of LSTM layer 1
 * (p 0.1, nn.LSTM(dropout=0.1), :312). */
int64_t l2s_train_steps_tape_floats(int B, int S);
int64_t l2s_train_steps_weights_floats(void);
int64_t l2s_train_steps_ws_bytes(int B, int S);
int l2s_train_steps_pack_weights(l2s_model* m, float* wbuf, void* stream);
int l2s_train_steps_fwd(l2s_model* m, float* state, int B, int T, int S, const float* teacher, const uint8_t* teacher_mask,
                        const uint8_t* teacher_mask_dev, float* tape, float* mel, float* stop, float* attn_logits, const float* drop_prenet,
                        const float* drop_attn, const float* drop_rnn, void* ws, int64_t ws_bytes, void* stream);
int l2s_train_steps_bwd(l2s_model* m, float* state, int B, int T, int S, const uint8_t* teacher_mask, float* tape, const float* attn_logits,
                        const float* dmel, const float* dstop, float* wbuf, float* dk, float* dv, float* dckey, float* dcval, float* dh_init,
                        float* de_c, const float* drop_prenet, const float* drop_attn, const float* drop_rnn, void* ws, int64_t ws_bytes,
                        void* stream);

/* Stage 3: the decoder prologue with a tape (decoder.py:321-351 / 383-410) and its backward down to the visual features.
 * Forward = l2s_decoder_prologue (same `state` buffer, same arguments) plus the tape.  Backward: the gradients of the state the loop
 * consumed (the outputs of l2s_train_steps_bwd; dcontent_dis dev (B*min_T,501) = gradient of the content distribution, may be NULL)
 * -> every prologue parameter gradient into the bound slots and dvis dev (B,T,1024). */
int64_t l2s_train_prologue_tape_floats(int B, int T);
int64_t l2s_train_prologue_ws_bytes(int B, int T);
int l2s_train_prologue_fwd(l2s_model* m, const float* vis, const float* emb, const float* gumbel, int B, int T, float* state, float* content_dis,
                           float* tape, void* ws, int64_t ws_bytes, void* stream);
int l2s_train_prologue_bwd(l2s_model* m, const float* vis, const float* emb, int B, int T, float* state, float* tape, float* wbuf, const float* dk,
                           const float* dv, const float* dckey, const float* dcval, const float* dh_init, const float* de_c, const float* dcontent_dis,
                           float* dvis, void* ws, int64_t ws_bytes, void* stream);

/* Visual encoder with a tape (video.py:76-87, shufflenetv2.py:42-152) and its backward.  Forward = l2s_encoder_fwd (same outputs:
 * vis dev (B,T,1024) and/or feat dev (B,T,768)) plus the tape.  Backward: dfeat = gradient wrt the normalised features, row stride
 * ld_dfeat floats (pass the dvis of l2s_train_prologue_bwd with ld_dfeat = 1024) -> every encoder parameter gradient into the bound
 * slots (the reference computes no gradient wrt the video). */
int64_t l2s_train_encoder_tape_floats(int B, int T, int H);
int64_t l2s_train_encoder_ws_bytes(int B, int T, int H);
int l2s_train_encoder_fwd(l2s_model* m, const float* video, int B, int T, int H, int W, const float* emb, float* vis, float* feat, float* tape, void* stream);
int l2s_train_encoder_bwd(l2s_model* m, const float* video, int B, int T, int H, int W, const float* dfeat, int ld_dfeat, float* tape, void* ws, int64_t ws_bytes,
                          void* stream);

/* ---- run-time options ------------------------------------------------------------------------------------------------------------
 * l2s_set_option changes the PROCESS DEFAULTS: what l2s_model_create copies into a new model.  l2s_model_set_option changes one model.
 * Launch sequences only ever read their own model's copy, so a thread that flips a switch cannot disturb batches other threads have in
 * flight on other models.  Several host threads may also drive ONE model at once (lip2speech_amd.parallel.InflightPool: chains in flight on
 * one weight blob): the blob is read-only, every workspace is its caller's, the per-model side stream / events / graph cache behind
 * "use_graph" are serialised by a per-model mutex, and the l2s_profile_* accumulators are guarded - only changing an option of a model WHILE
 * other threads run batches on it is the caller's race.  An unknown name is an error.  The options of the product library choose WHAT is
 * computed (precision leg, semantics) or a documented mode; the switches that only choose between block forms of the same arithmetic
 * (measured and rejected forms, kept for A/B timing) exist in the diagnostic build alone: include/l2s_diag.h.
 *   "persist_decode"    (4)  the free-running decode loop (decoder.py:412-435) of a single-batch call with at most this many clips - up to 4, clips of
 *                            <= 32 frames (demo.py runs one clip, BASELINE config 1 two) - as ONE persistent launch of 128 resident workgroups per clip
 *                            that keep the step weights in registers and exchange h / c / q / prenet as tagged 8-byte granules (pdecode.hip): 7.7 instead
 *                            of 20.4 us per step at one clip, 8.2 at two; three or four clips run as two such launches one after the other.  Another
 *                            order of the same fp32 sums (within 5e-4 of the launch path, < 1e-3 of the reference).  The form is only taken where
 *                            l2s_persist_available() says every workgroup can be resident; a launch that makes no progress for 2 s gives up,
 *                            overwrites mel / stop / attention with NaN and counts in l2s_persist_timeouts(); the NEXT persistent-eligible call on that
 *                            device fails once with that error, later ones take the launch path until the option is set to a positive value again
 *                            (which re-arms the device).  0 = always four launches per step; l2s_*_multi never uses it
 *   "use_graph"         (0)  replay the decode loop from a captured hipGraph (BASELINE config 4's streaming decoder; slower than plain launches
 *                            on this runtime at every size measured, so off by default)
 *   "fold_step_weights" (1)  4-launch step with pre-multiplied prenet1*fc_out and attention_proj hoisted onto the values; 0 = the literal 6-phase
 *                            step of decoder.py:412-429 (what a model runs after l2s_train_refresh_weights until its merged weights are rebuilt)
 *   "refresh_map"       (0)  l2s_model_finalize also builds the map l2s_train_refresh_weights needs (training)
 *   "infer_bf16"        (0)  the bf16 leg of the INFERENCE / evaluate entry points: the front-end conv on one bf16 plane, GEMMs / Conv1d stacks of
 *                            encoder, prologue, post-net and voice tower with bf16 operands and fp32 accumulation; the decode loop, the BiLSTM, the
 *                            fused ShuffleNet units and every activation in HBM stay fp32.  Outside the 1e-3 fp32 gate by construction
 *   "train_bf16"        (0)  TRAINING entry points (encoder, prologue, post-net; forward and backward): GEMMs / Conv1d stacks round their operands to
 *                            bf16 (RNE) on the way into LDS, fp32 accumulation and results; the recurrent loop, the Conv3d front-end, BatchNorm
 *                            statistics, master weights and optimizer stay fp32
 *   "gemm_x3"           (1)  inference GEMMs / Conv1d stacks on the bf16 matrix cores through the EXACT three-way split (x = hi + mid + lo, six bf16
 *                            MFMAs per K step of 16 instead of eight f32 MFMAs of K = 2) where the shapes are eligible; 0 = the f32 MFMA kernel
 *   "frontend_x3"       (3)  the inference front-end conv (Conv3d 5x7x7 + BN + PReLU + MaxPool) on the same split-bf16 path: 3 = two consecutive
 *                            output frames per block with the next slab's staging interleaved between the MFMAs, 2 = the same with a staging phase of its
 *                            own (same bits), 1 = one frame per block, 0 = the f32 MFMA kernel
 *   "trunk_x3"          (1)  the fused ShuffleNet units' pointwise convs on the split-bf16 path; 0 = f32 MFMA
 *   "lstm_x3"           (3)  the decode step's two LSTM launches on the split-bf16 path (1 / 2 / 3: four-wave / eight-wave / half-CU block forms of the
 *                            same arithmetic, same bits - 3 picks by l2s_set_thread_chains); 0 = f32 MFMAs (other bits, rounding-level) */
int l2s_set_option(const char* name, int value);
int l2s_model_set_option(l2s_model* m, const char* name, int value);
/* The persistent forms (option "persist_decode") spin on other workgroups and need all of them resident at once.  l2s_persist_available: 1 where that
 * holds on the current device - 256 compute units, no compute-unit mask in the environment (HSA_CU_MASK / ROC_GLOBAL_CU_MASK), one workgroup of every
 * persistent kernel fits a compute unit (occupancy query), and no persistent launch on this device has timed out since it was last armed - else 0: calls
 * inside the envelope then take the launch-per-phase path.  l2s_persist_timeouts: how many persistent launches of this process gave up (2 s without
 * progress) and had their outputs overwritten with NaN; read from pinned host memory, no synchronize - meaningful after the host has synchronized with
 * the stream.  The library itself reports a time-out through the next persistent-eligible call on that device (it fails once; l2s_last_error). */
int l2s_persist_available(void);
int l2s_persist_timeouts(void);
/* How many launch chains the CALLER keeps in flight on the device, for the calling host thread (default 1): a scheduling hint, never arithmetic.  With
 * n >= 2 the step kernels of this thread's calls take blocks of half a compute unit, so that kernels of the other chains run beside them on the same
 * CUs - 33.6 -> 27-28 us per decode step for the chip at 256 rows with three chains; with n = 1 they keep the blocks that fill a CU, which are 3-4 %
 * faster when the chain has the chip to itself.  lip2speech_amd.parallel.InflightPool sets it in its worker threads. */
int l2s_set_thread_chains(int n);
/* per-kernel timing: when enabled every launch is bracketed by HIP events on its stream; read back with
 * l2s_profile_get (which synchronises the events it reads).  Off by default. */
int l2s_profile_enable(int on);
int l2s_profile_reset(void);
int l2s_profile_count(void);
int l2s_profile_get(int idx, const char** name, int64_t* launches, double* total_ms);

#ifdef __cplusplus
}
#endif
#endif /* L2S_H */
