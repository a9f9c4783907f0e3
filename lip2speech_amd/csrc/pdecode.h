// Persistent (single-launch, weight-stationary) decode loop for a few clips: parameters and launcher (pdecode.hip).
#pragma once
#include "l2s_common.h"

namespace l2s {

typedef unsigned long long u64;

struct PDecP {
    // step weights: frag16 fragments of the blob, read once into registers
    const float *Wq, *bq, *aq;            // Q: [512][1024], bias, PSine w
    const float *Wcq, *bcq;               // content Q: [256][1024]
    const float *Wp1f, *bp1f, *ap1;       // prenet1 o fc_out over h1: [256][512]
    const float *Wp1, *bp1;               // literal prenet1 over a frame (step 0: BOS): [256][80]
    const float *Wp2, *bp2, *ap2;         // prenet2: [256][256]
    const float *Wl0, *bl0, *Wl1, *bl1;   // LSTM layers: [2048][1024], rows (unit, gate); b_ih + b_hh
    const float *Wfc, *bfc;               // fc_out + stop row: [96][512]
    const float *pos, *tau, *tau_c, *bos;
    // per-call state (decoder prologue outputs)
    const float *k, *vp, *ckey, *cval;    // [B][T][512], [B][T][256], [B][m][256] x 2
    const float *h_init;                  // frag16 [2][pad16(B)][512]: h0 then h1
    const float *stop_const;              // [B]
    float *mel, *stop, *attn;             // [B][S][80], [B][S], [B][S][T] or null
    u64* xch;                             // exchange granules (zeroed before the launch)
    unsigned* status;                     // [0]: set to 1 by a workgroup whose poll timed out (every workgroup then leaves)
    int attn_logits, B, T, m, S;
    int b0;                               // set by launch_pdecode: first clip of THIS launch (clips go two at a time)
    unsigned long long* ts; int ts_step;  // measurement (tools/pdecode_timeline.py): [256 workgroups][16] stamps of step ts_step, or null
};

// the prologue's BiLSTM recurrence in the same form (pbilstm_kernel)
struct PBiP {
    const float *Whh0, *Whh1;             // recurrent weights of the two directions, frag16 [2048][512], rows (unit, gate)
    const float *gin;                     // [B][T][4096] input gates W_ih x + b_ih + b_hh: [direction][gate][unit]
    const float *s_e;                     // [B][512] h0 = c0
    float *rnn;                           // [B][T][1024] out: [forward | backward]
    float *h_state;                       // frag16 [2][pad16(B)][512] out: final h (forward, backward)
    float *cellcat;                       // [B][1024] out: final c (forward | backward)
    u64* xch; unsigned* status;           // set by launch_pbilstm
    int B, T;
};
int64_t pbilstm_ws_bytes();
bool pbilstm_supported(int B, int T);                   // one or two clips
int launch_pbilstm(const PBiP& p, void* ws, int64_t ws_bytes, hipStream_t s);
bool pdecode_device_ok();                               // every workgroup of a persistent launch can be resident: 256 compute units, none masked, kernels fit, no timed-out launch so far
int pdecode_gate();                                     // per call of a persistent-eligible entry: 1 = persistent form, 0 = launch path, -1 = an earlier launch timed out: fail this call once (error set)
void pdecode_rearm();                                   // option "persist_decode" > 0 was set: forgive the current device's time-outs so far
int pdecode_timeouts();                                 // persistent launches of this process whose workgroups gave up (outputs NaN)
int64_t pdecode_ws_bytes(int B);                       // exchange granules + status word
bool pdecode_supported(int B, int T, int m);            // <= 4 clips of <= 32 frames
void pdecode_set_timeline(unsigned long long* ts, int step);      // non-null: thread 0 of every workgroup stamps the phases of that step
int launch_pdecode(const PDecP& p, void* ws, int64_t ws_bytes, hipStream_t s);      // xch / status are carved from ws

}  // namespace l2s
