// Vocoder + metric of evaluate.py on the device:
//   MelSpec2Audio (datasets/spectograms.py:76-95) = exp -> torchaudio 0.9.0 InverseMelScale (SGD on spec @ fb = mel) -> GriffinLim
//   ESTOI         (evaluate.py:41-45: pystoi.stoi(clean, pred, fs, extended=True))
// torchaudio and pystoi are absent from the build image: what these kernels implement is the published algorithm as restated in
// lip2speech_amd/datasets/spectrograms.py and lip2speech_amd/metrics.py - PARITY UNPINNED against the third-party packages; the kernels
// are tested against those restatements with the random initial iterates passed in explicitly.
#include "fft_dev.h"
#include "l2s_common.h"
#include "../../include/l2s.h"

#include <math.h>

namespace l2s {

// =====================================================================================================================================
// InverseMelScale: minimise mean_{rows} |mel - spec @ fb|^2 over spec >= 0 by SGD (lr 0.1, momentum 0.9), `iters` iterations, from a
// given start.  A row (one mel frame of one clip) meets the others only through the loss that drives the two stopping rules, and the
// filterbank is banded (a frequency bin feeds at most a few mel bands): ONE WAVE PER ROW keeps spec and velocity in registers (bin
// f = lane + 64 j), exchanges spec / diff through 2.4 KB of LDS and walks only the non-zero band of each filter - no GEMM: the dense
// 513 x 80 product would be 97 % zeros.  Per-iteration per-row losses go to `loss_rows` [iter][row]; inverse_mel_stop_kernel then
// evaluates the reference's rules per call (new_loss < 1e-5, |loss - new_loss| < 1e-8; the update of the stopping iteration is still
// applied) and a second pass re-runs only the calls that stop early, with their iteration count.
// =====================================================================================================================================
constexpr int IM_MAXF = 576;      // 9 bins per lane
constexpr int IM_MAXM = 128;      // 2 mel bands per lane
constexpr int IM_ROWS = 8;        // rows (waves) per block: they share the compact filterbank tables in LDS
constexpr int IM_NNZ = 2048;      // capacity of each compact table (the 513 x 80 HTK filterbank has ~1 100 non-zeros)

struct InvMelP {
    const float* mel;        // (N, n_mels, L) power mel, or log-mel when log_input
    const float* init;       // (N*L, n_freqs) start iterate, row = n*L + l
    const int* tab;          // ws: [n_mels][3] first bin / last+1 / offset into fwd; [n_freqs][3] first band / last+1 / offset into bwd; then nnz_fwd, nnz_bwd
    const float* fwd;        // ws: the non-zeros of fb, band-major (band m: fb[flo..fhi) [m])
    const float* bwd;        // ws: the same non-zeros, bin-major (bin f: fb[f][mlo..mhi))
    float* spec;             // (N, n_freqs, L) out
    float* loss_rows;        // [iters][N*L] or null
    const int* iters_call;   // per call: iterations to run in THIS pass (null: `iters` for all); a call whose entry equals `iters` is skipped when `second`
    int N, L, n_mels, n_freqs, rows_per_call, iters, log_input, second;
};

// one block: band structure of the filterbank + its non-zeros in the two orders the SGD walks them
__global__ __launch_bounds__(256) void inverse_mel_bands_kernel(const float* fb, int n_freqs, int n_mels, int* tab, float* fwd, float* bwd) {
    __shared__ int cnt[IM_MAXM + IM_MAXF];
    const int tid = threadIdx.x;
    for (int i = tid; i < n_mels + n_freqs; i += 256) {
        int lo = 0, hi = 0;
        if (i < n_mels) {
            lo = n_freqs;
            for (int f = 0; f < n_freqs; ++f) if (fb[(int64_t)f * n_mels + i] != 0.f) { lo = f < lo ? f : lo; hi = f + 1; }
        } else {
            const int f = i - n_mels;
            lo = n_mels;
            for (int m = 0; m < n_mels; ++m) if (fb[(int64_t)f * n_mels + m] != 0.f) { lo = m < lo ? m : lo; hi = m + 1; }
        }
        if (hi == 0) lo = 0;
        tab[3 * i] = lo; tab[3 * i + 1] = hi;
        cnt[i] = hi - lo;
    }
    __syncthreads();
    if (tid == 0) {
        int o = 0;
        for (int m = 0; m < n_mels; ++m) { tab[3 * m + 2] = o; o += cnt[m]; }
        tab[3 * (n_mels + n_freqs)] = o;
        o = 0;
        for (int f = 0; f < n_freqs; ++f) { tab[3 * (n_mels + f) + 2] = o; o += cnt[n_mels + f]; }
        tab[3 * (n_mels + n_freqs) + 1] = o;
    }
    __syncthreads();
    for (int i = tid; i < n_mels + n_freqs; i += 256) {
        const int lo = tab[3 * i], hi = tab[3 * i + 1], o = tab[3 * i + 2];
        if (i < n_mels) { for (int f = lo; f < hi; ++f) if (o + f - lo < IM_NNZ) fwd[o + f - lo] = fb[(int64_t)f * n_mels + i]; }
        else { const int f = i - n_mels; for (int m = lo; m < hi; ++m) if (o + m - lo < IM_NNZ) bwd[o + m - lo] = fb[(int64_t)f * n_mels + m]; }
    }
}

__global__ __launch_bounds__(IM_ROWS * 64) void inverse_mel_kernel(const InvMelP p) {
    __shared__ float s_fwd[IM_NNZ], s_bwd[IM_NNZ];
    __shared__ float s_specs[IM_ROWS][IM_MAXF];
    __shared__ float s_diffs[IM_ROWS][IM_MAXM];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int L = p.L, M = p.n_mels, F = p.n_freqs;
    {
        const int nf = p.tab[3 * (M + F)], nb = p.tab[3 * (M + F) + 1];
        for (int i = tid; i < nf && i < IM_NNZ; i += IM_ROWS * 64) s_fwd[i] = p.fwd[i];
        for (int i = tid; i < nb && i < IM_NNZ; i += IM_ROWS * 64) s_bwd[i] = p.bwd[i];
    }
    __syncthreads();                                   // the only block barrier: from here on every wave runs its own row
    const int64_t row = (int64_t)blockIdx.x * IM_ROWS + wave;
    if (row >= (int64_t)p.N * L) return;
    float* s_spec = s_specs[wave];
    float* s_diff = s_diffs[wave];
    const int n = (int)(row / L), l = (int)(row - (int64_t)n * L);
    const int call = (int)(row / ((int64_t)p.rows_per_call * L));
    int iters = p.iters;
    if (p.iters_call) {
        iters = p.iters_call[call];
        if (p.second && iters >= p.iters) return;             // this call ran to the end in the first pass
    }
    const float gscale = -2.0f / (float)(p.rows_per_call * L);
    float spec[9], vel[9];
    int mlo[9], mcnt[9], boff[9];
#pragma unroll
    for (int j = 0; j < 9; ++j) {
        const int f = lane + 64 * j;
        spec[j] = f < F ? p.init[row * F + f] : 0.f;
        vel[j] = 0.f;
        mlo[j] = f < F ? p.tab[3 * (M + f)] : 0;
        mcnt[j] = f < F ? p.tab[3 * (M + f) + 1] - mlo[j] : 0;
        boff[j] = f < F ? p.tab[3 * (M + f) + 2] : 0;
    }
    float target[2];
    int flo[2], fcnt[2], foff[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int m = lane + 64 * i;
        float t = m < M ? p.mel[((int64_t)n * M + m) * L + l] : 0.f;
        if (p.log_input && m < M) t = expf(t);
        target[i] = t;
        flo[i] = m < M ? p.tab[3 * m] : 0;
        fcnt[i] = m < M ? p.tab[3 * m + 1] - flo[i] : 0;
        foff[i] = m < M ? p.tab[3 * m + 2] : 0;
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 9; ++j) s_spec[lane + 64 * j] = spec[j];
        wave_lds_sync();
        float lsum = 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int m = lane + 64 * i;
            float acc = 0.f;
            for (int c = 0; c < fcnt[i]; ++c) acc = fmaf(s_spec[flo[i] + c], s_fwd[foff[i] + c], acc);      // bins ascending
            const float d = target[i] - acc;
            if (m < M) { s_diff[m] = d; lsum = fmaf(d, d, lsum); }
        }
        wave_lds_sync();
        if (p.loss_rows) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) lsum += __shfl_xor(lsum, o);
            if (lane == 0) p.loss_rows[(int64_t)it * ((int64_t)p.N * L) + row] = lsum;
        }
#pragma unroll
        for (int j = 0; j < 9; ++j) {
            float g = 0.f;
            for (int c = 0; c < mcnt[j]; ++c) g = fmaf(s_diff[mlo[j] + c], s_bwd[boff[j] + c], g);          // bands ascending
            const float nv = 0.9f * vel[j] + gscale * g;                 // torch.optim.SGD(lr 0.1, momentum 0.9): buf = 0.9 buf + grad
            vel[j] = nv;
            spec[j] = fmaxf(spec[j] - 0.1f * nv, 0.f);                   // p -= lr buf; clamp_(min=0)
        }
        wave_lds_sync();
    }
#pragma unroll
    for (int j = 0; j < 9; ++j) {
        const int f = lane + 64 * j;
        if (f < F) p.spec[((int64_t)n * F + f) * L + l] = spec[j];
    }
}

// per call: the mean loss of every iteration (rows summed in index order, fp64) -> the iteration count the reference's stopping rules leave
__global__ __launch_bounds__(256) void inverse_mel_stop_kernel(const float* loss_rows, int rows_total, int rows_call, int iters, int* iters_call, float* loss_out) {
    __shared__ double red[256];
    __shared__ int stop_at;
    const int call = blockIdx.x, tid = threadIdx.x;
    if (tid == 0) stop_at = iters;
    float prev = INFINITY;
    __syncthreads();
    for (int it = 0; it < iters; ++it) {
        double s = 0.0;
        const float* lr = loss_rows + (int64_t)it * rows_total + (int64_t)call * rows_call;
        for (int r = tid; r < rows_call; r += 256) s += (double)lr[r];
        red[tid] = s;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) { if (tid < o) red[tid] += red[tid + o]; __syncthreads(); }
        const float nl = (float)(red[0] / (double)rows_call);
        __syncthreads();
        if (tid == 0 && loss_out) loss_out[(int64_t)call * iters + it] = nl;
        const bool stop = nl < 1e-5f || fabsf(prev - nl) < 1e-8f;        // prev = inf at it = 0: |inf - x| = inf
        prev = nl;
        if (stop) { if (tid == 0) stop_at = it + 1; break; }             // block-uniform: every thread computed the same nl
    }
    __syncthreads();
    if (tid == 0) iters_call[call] = stop_at;
}

// =====================================================================================================================================
// Griffin-Lim (torchaudio 0.9.0 functional.griffinlim: power 2, momentum 0.99, given start angles), n_fft = win = 1024, hop = 256:
//   repeat n_iter times:  rebuilt = stft(istft(mag * ang));  ang = rebuilt - prev * (0.99 / 1.99);  ang /= |ang| + 1e-16;  prev = rebuilt
//   wave = istft(mag * ang)
// ONE BLOCK PER CLIP, NW waves, one wave per STFT frame at a time.  The clip's waveform (hop (L-1) samples) lives in LDS for the whole
// loop: the inverse transforms overlap-add into it in four barrier-separated phases (frames t = p mod 4 do not overlap each other, so
// the order of the additions is fixed: deterministic), the forward transforms read it back with the reflect padding of
// torch.stft(center=True) folded into the index.  The half spectra (L x 513 complex: 316 KB per clip) do not fit next to it: the last
// two `rebuilt` spectra live in the caller's workspace (ping-pong; ang is recomputed from them where it is consumed and never stored),
// frame-major so that a wave's 64 lanes touch consecutive bins.  FFTs: fft_dev.h (three radix-8 stages per 1024-point real transform).
// =====================================================================================================================================
constexpr int GL_NFFT = 1024, GL_HOP = 256, GL_NBIN = 513, GL_LDK = 520;      // bins per frame padded to 520 in the workspace

struct GriffinP {
    const float* power;      // (N, 513, L) power spectrogram (>= 0; clamped)
    const float* init;       // (N, 513, L, 2) start angles (complex, used as given)
    float* mag;              // ws: (N, L, 520) sqrt(power), frame-major
    float2* reb;             // ws: (N, 2, L, 520) the last two rebuilt spectra (slot 0 starts as the transposed init)
    float* wave;             // (N, hop (L-1)) out
    int N, L, iters;
    float momentum;          // 0.99 / 1.99
};

template <int NW, int YCAP>
__global__ __launch_bounds__(NW * 64) void griffin_lim_kernel(const GriffinP p) {
    __shared__ __attribute__((aligned(16))) float y[YCAP];
    __shared__ __attribute__((aligned(16))) float2 scratch[NW][GL_LDK];
    __shared__ float w2[GL_NFFT];                                  // window^2 for the overlap-add envelope
    const int clip = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int L = p.L, ylen = GL_HOP * (L - 1);
    float2* sc = scratch[wave];
    float* mag = p.mag + (int64_t)clip * L * GL_LDK;
    float2* reb0 = p.reb + (int64_t)clip * 2 * L * GL_LDK;
    float2* reb1 = reb0 + (int64_t)L * GL_LDK;
    // ---- prologue: frame-major copies of sqrt(power) and of the start angles; window constants
    {
        const float* pw = p.power + (int64_t)clip * GL_NBIN * L;
        const float2* in = reinterpret_cast<const float2*>(p.init) + (int64_t)clip * GL_NBIN * L;
        for (int i = tid; i < GL_NBIN * L; i += NW * 64) {
            const int k = i / L, t = i - k * L;
            mag[(int64_t)t * GL_LDK + k] = sqrtf(fmaxf(pw[i], 0.f));
            reb0[(int64_t)t * GL_LDK + k] = in[i];
        }
        for (int i = tid; i < GL_NFFT; i += NW * 64) { const float w = 0.5f - 0.5f * cospif((float)i / 512.0f); w2[i] = w * w; }
    }
    Fft512Tw tw;
    tw.init(lane);
    float wn[8][2];                                                  // hann(periodic) at samples 2 n, 2 n + 1 of a frame, n = lane + 64 r
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const int m = 2 * (lane + 64 * r);
        wn[r][0] = 0.5f - 0.5f * cospif((float)m / 512.0f);
        wn[r][1] = 0.5f - 0.5f * cospif((float)(m + 1) / 512.0f);
    }
    __threadfence_block();
    __syncthreads();

    for (int it = 0; it <= p.iters; ++it) {
        const float2* cur = (it & 1) ? reb1 : reb0;                  // written by phase B of iteration it - 1 (it = 0: the start angles)
        const float2* prv = (it & 1) ? reb0 : reb1;                  // written at it - 2; not read for it < 2
        // ---------------- phase A: y = istft(mag * ang)
        for (int i = tid; i < ylen; i += NW * 64) y[i] = 0.f;
        __syncthreads();
        for (int ph = 0; ph < 4; ++ph) {
            for (int t = ph + 4 * wave; t < L; t += 4 * NW) {
                const float* mg = mag + (int64_t)t * GL_LDK;
                const float2* c = cur + (int64_t)t * GL_LDK;
                const float2* q = prv + (int64_t)t * GL_LDK;
                float2 v[8];
                auto spec_at = [&](int k) -> float2 {
                    float2 a = c[k];
                    if (it >= 1) {
                        if (it >= 2) { const float2 b = q[k]; a.x -= p.momentum * b.x; a.y -= p.momentum * b.y; }
                        const float inv = 1.0f / (sqrtf(a.x * a.x + a.y * a.y) + 1e-16f);
                        a.x *= inv; a.y *= inv;
                    }
                    const float g = mg[k];
                    return make_float2(g * a.x, g * a.y);
                };
#pragma unroll
                for (int r = 0; r < 8; ++r) v[r] = spec_at(lane + 64 * r);
                float2 s512 = make_float2(0.f, 0.f);
                if (lane == 0) s512 = spec_at(512);
                irfft1024_pre(v, s512, sc, lane, tw);
                fft512<+1>(v, sc, lane, tw);
                const int base = t * GL_HOP - GL_NFFT / 2;           // output index of the frame's sample 0 (istft trims n_fft / 2)
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    const int idx = base + 2 * (lane + 64 * r);
                    if (idx >= 0 && idx < ylen) {                    // idx is even and ylen is even: both samples in or out together
                        float2 o = *reinterpret_cast<float2*>(y + idx);
                        o.x += v[r].x * (1.0f / 1024.0f) * wn[r][0];
                        o.y += v[r].y * (1.0f / 1024.0f) * wn[r][1];
                        *reinterpret_cast<float2*>(y + idx) = o;
                    }
                }
            }
            __syncthreads();
        }
        // overlap-add envelope: sum of window^2 over the frames that cover the sample
        for (int i = tid; i < ylen; i += NW * 64) {
            const int pp = i + GL_NFFT / 2;
            int t0 = (pp - (GL_NFFT - 1) + GL_HOP - 1) / GL_HOP; t0 = t0 < 0 ? 0 : t0;
            int t1 = pp / GL_HOP; t1 = t1 > L - 1 ? L - 1 : t1;
            float env = 0.f;
            for (int t = t0; t <= t1; ++t) env += w2[pp - t * GL_HOP];
            y[i] = y[i] / env;
        }
        __syncthreads();
        if (it == p.iters) break;
        // ---------------- phase B: rebuilt = stft(y) (centre, reflect), into the slot phase A of the next iteration reads as `cur`
        float2* dst = (it & 1) ? reb0 : reb1;
        for (int t = wave; t < L; t += NW) {
            float2 v[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                int s0 = t * GL_HOP + 2 * (lane + 64 * r) - GL_NFFT / 2, s1 = s0 + 1;
                s0 = s0 < 0 ? -s0 : s0; s0 = s0 >= ylen ? 2 * (ylen - 1) - s0 : s0;
                s1 = s1 < 0 ? -s1 : s1; s1 = s1 >= ylen ? 2 * (ylen - 1) - s1 : s1;
                v[r] = make_float2(y[s0] * wn[r][0], y[s1] * wn[r][1]);
            }
            fft512<-1>(v, sc, lane, tw);
            const float nyq = rfft1024_post(v, sc, lane, tw);
            float2* d = dst + (int64_t)t * GL_LDK;
#pragma unroll
            for (int r = 0; r < 8; ++r) d[lane + 64 * r] = v[r];
            if (lane == 0) d[512] = make_float2(nyq, 0.f);
        }
        __threadfence_block();                                       // the spectra go through global memory: this block's own stores, re-read by other waves
        __syncthreads();
    }
    float* out = p.wave + (int64_t)clip * ylen;
    for (int i = tid; i < ylen; i += NW * 64) out[i] = y[i];
}

// =====================================================================================================================================
// ESTOI (pystoi 0.3.3 stoi(x, y, fs, extended=True) as restated in lip2speech_amd/metrics.py), one block per clip:
//   resample_poly(., 10000 / g, fs / g) with the caller's polyphase FIR (pystoi's resample_oct window) -> drop the frames of the clean signal more than 40 dB below its
//   loudest frame (256-sample hann frames, hop 128) from both signals and overlap-add the rest -> 512-point spectra of 256-sample frames ->
//   15 one-third octave bands -> every run of 30 frames: rows and columns normalised to zero mean / unit norm -> mean correlation.
// The work per clip is small (two signals x ~93 frames x 257 bins): the spectra are direct sums against a 512-entry twiddle table.
// =====================================================================================================================================
constexpr int ES_FRAME = 256, ES_HOP = 128, ES_NFFT = 512, ES_BANDS = 15, ES_SEG = 30, ES_MAXFRAMES = 128, ES_MAXLEN = ES_HOP * (ES_MAXFRAMES + 1);
constexpr int ES_NBIN = ES_NFFT / 2 + 1;

struct EstoiP {
    const float* clean; const float* pred;    // (N, n_samples)
    const float* fir;                          // resampling filter h (already scaled by `up`, pre-padded), n_fir taps; null when fs == 10000
    const int* band_lo; const int* band_hi;    // third-octave band edges (bins), 15 each
    float* rs;                                 // ws: (N, 2, ES_MAXLEN) resampled signals
    float* pw;                                 // ws: (N, 2, ES_MAXFRAMES, ES_NBIN) power spectra of the frames
    float* score;                              // (N)
    int N, n_samples, n_fir, up, down, n_pre_remove, n_res;
};

__global__ __launch_bounds__(1024) void estoi_kernel(const EstoiP p) {
    __shared__ float sig[2][ES_MAXLEN];                               // the silent-frame-free signals (x = clean, y = pred); later: row statistics + contributions
    __shared__ float tob[2][ES_BANDS][ES_MAXFRAMES];
    __shared__ float2 twd[ES_NFFT];
    __shared__ float win[ES_FRAME];
    __shared__ float energy[ES_MAXFRAMES];
    __shared__ int keep_pos[ES_MAXFRAMES];
    __shared__ float red[1024];
    __shared__ int n_keep_s;
    const int clip = blockIdx.x, tid = threadIdx.x;
    float* rs = p.rs + (int64_t)clip * 2 * ES_MAXLEN;
    float* pw = p.pw + (int64_t)clip * 2 * ES_MAXFRAMES * ES_NBIN;
    const int nr = p.n_res;
    constexpr float EPS = 2.220446049250313e-16f;
    // ---- tables
    for (int i = tid; i < ES_NFFT; i += 1024) { float s, c; sincospif((float)i / 256.0f, &s, &c); twd[i] = make_float2(c, -s); }      // exp(-2 pi i n / 512)
    for (int i = tid; i < ES_FRAME; i += 1024) win[i] = 0.5f - 0.5f * cospif(2.0f * (float)(i + 1) / (float)(ES_FRAME + 1));          // np.hanning(258)[1:-1]
    // ---- resample (upfirdn, zero-padded ends): out[n] = sum_i x[i] h[(n + n_pre_remove) down - i up]
    for (int s = 0; s < 2; ++s) {
        const float* x = (s ? p.pred : p.clean) + (int64_t)clip * p.n_samples;
        for (int n = tid; n < nr; n += 1024) {
            float acc;
            if (p.fir) {
                const int64_t c = (int64_t)(n + p.n_pre_remove) * p.down;
                int64_t ihi = c / p.up; if (ihi > p.n_samples - 1) ihi = p.n_samples - 1;
                int64_t ilo = c - (p.n_fir - 1) <= 0 ? 0 : (c - (p.n_fir - 1) + p.up - 1) / p.up;
                double a = 0.0;
                for (int64_t i = ilo; i <= ihi; ++i) a += (double)x[i] * (double)p.fir[c - i * p.up];
                acc = (float)a;
            } else {
                acc = x[n];
            }
            rs[s * ES_MAXLEN + n] = acc;
        }
    }
    __syncthreads();
    // ---- silent-frame removal: energies of the clean signal's windowed frames
    // pystoi 0.3.3 frames with range(0, len - framelen, hop): ceil((len - framelen) / hop) frames - the one ending on the last sample is not taken
    const int nf = nr > ES_FRAME ? (nr - ES_FRAME + ES_HOP - 1) / ES_HOP : 0;
    {
        const int wv = tid >> 6, ln = tid & 63;
        for (int f = wv; f < nf; f += 16) {
            float e = 0.f;
            for (int i = ln; i < ES_FRAME; i += 64) { const float v = rs[f * ES_HOP + i] * win[i]; e = fmaf(v, v, e); }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) e += __shfl_xor(e, o);
            if (ln == 0) energy[f] = 20.0f * log10f(sqrtf(e) + EPS);
        }
    }
    __syncthreads();
    if (tid == 0) {
        float mx = -INFINITY;
        for (int f = 0; f < nf; ++f) mx = fmaxf(mx, energy[f]);
        int k = 0;
        for (int f = 0; f < nf; ++f) if (mx - 40.0f - energy[f] < 0.f) keep_pos[k++] = f;
        n_keep_s = k;
    }
    __syncthreads();
    const int nk = n_keep_s;
    const int len2 = nk > 0 ? (nk - 1) * ES_HOP + ES_FRAME : 0;
    // overlap-add of the kept frames: output sample i is covered by kept frames j = i / hop - 1 and i / hop
    for (int s = 0; s < 2; ++s)
        for (int i = tid; i < len2; i += 1024) {
            const int j1 = i / ES_HOP, j0 = j1 - 1;
            float acc = 0.f;
            if (j0 >= 0 && j0 < nk) { const int o = i - j0 * ES_HOP; acc += rs[s * ES_MAXLEN + keep_pos[j0] * ES_HOP + o] * win[o]; }
            if (j1 < nk) { const int o = i - j1 * ES_HOP; acc += rs[s * ES_MAXLEN + keep_pos[j1] * ES_HOP + o] * win[o]; }
            sig[s][i] = acc;
        }
    __syncthreads();
    // ---- power spectra of the bins the bands use (direct sums against the twiddle table), then the band magnitudes
    const int nf2 = len2 > ES_FRAME ? (len2 - ES_FRAME + ES_HOP - 1) / ES_HOP : 0;      // = nk - 1: the same rule on the overlap-added signal
    if (nf2 < ES_SEG) { if (tid == 0) p.score[clip] = 1e-5f; return; }      // pystoi: not enough frames -> 1e-5 (with a warning)
    const int k_lo = p.band_lo[0], k_hi = p.band_hi[ES_BANDS - 1], nkb = k_hi - k_lo;
    for (int item = tid; item < 2 * nf2 * nkb; item += 1024) {
        const int s = item / (nf2 * nkb), rem = item - s * nf2 * nkb, f = rem / nkb, k = k_lo + rem - f * nkb;
        const float* fr = sig[s] + f * ES_HOP;
        float re = 0.f, im = 0.f;
        for (int i = 0; i < ES_FRAME; ++i) {
            const float v = fr[i] * win[i];
            const float2 w = twd[(k * i) & (ES_NFFT - 1)];
            re = fmaf(v, w.x, re); im = fmaf(v, w.y, im);
        }
        pw[(s * ES_MAXFRAMES + f) * ES_NBIN + k] = re * re + im * im;
    }
    __syncthreads();
    for (int item = tid; item < 2 * nf2 * ES_BANDS; item += 1024) {
        const int s = item / (nf2 * ES_BANDS), rem = item - s * nf2 * ES_BANDS, f = rem / ES_BANDS, b = rem - f * ES_BANDS;
        float a = 0.f;
        for (int k = p.band_lo[b]; k < p.band_hi[b]; ++k) a += pw[(s * ES_MAXFRAMES + f) * ES_NBIN + k];
        tob[s][b][f] = sqrtf(a);
    }
    __syncthreads();
    // ---- segments of 30 frames.  Row (band) statistics per (signal, segment, band): mean and 1 / (norm + eps) over the 30 frames
    const int nseg = nf2 - ES_SEG + 1;
    float* rstat = &sig[0][0];                                          // [2][nseg][15][2]   (the signals are dead)
    float* contrib = rstat + 2 * ES_MAXFRAMES * ES_BANDS * 2;           // [nseg][30]
    for (int item = tid; item < 2 * nseg * ES_BANDS; item += 1024) {
        const int s = item / (nseg * ES_BANDS), rem = item - s * nseg * ES_BANDS, seg = rem / ES_BANDS, b = rem - seg * ES_BANDS;
        float mean = 0.f;
        for (int i = 0; i < ES_SEG; ++i) mean += tob[s][b][seg + i];
        mean /= (float)ES_SEG;
        float nrm = 0.f;
        for (int i = 0; i < ES_SEG; ++i) { const float v = tob[s][b][seg + i] - mean; nrm = fmaf(v, v, nrm); }
        rstat[item * 2] = mean; rstat[item * 2 + 1] = 1.0f / (sqrtf(nrm) + EPS);
    }
    __syncthreads();
    // column (frame) normalisation over the 15 bands of the row-normalised values, and the frame's share of the correlation
    for (int item = tid; item < nseg * ES_SEG; item += 1024) {
        const int seg = item / ES_SEG, i = item - seg * ES_SEG;
        float v[2][ES_BANDS];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            float mean = 0.f;
#pragma unroll
            for (int b = 0; b < ES_BANDS; ++b) {
                const float* st = rstat + ((s * nseg + seg) * ES_BANDS + b) * 2;
                v[s][b] = (tob[s][b][seg + i] - st[0]) * st[1];
                mean += v[s][b];
            }
            mean /= (float)ES_BANDS;
            float nrm = 0.f;
#pragma unroll
            for (int b = 0; b < ES_BANDS; ++b) { v[s][b] -= mean; nrm = fmaf(v[s][b], v[s][b], nrm); }
            const float inv = 1.0f / (sqrtf(nrm) + EPS);
#pragma unroll
            for (int b = 0; b < ES_BANDS; ++b) v[s][b] *= inv;
        }
        float c = 0.f;
#pragma unroll
        for (int b = 0; b < ES_BANDS; ++b) c = fmaf(v[0][b], v[1][b], c);
        contrib[item] = c / (float)ES_SEG;
    }
    __syncthreads();
    float part = 0.f;
    for (int item = tid; item < nseg * ES_SEG; item += 1024) part += contrib[item];
    red[tid] = part;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) { if (tid < o) red[tid] += red[tid + o]; __syncthreads(); }
    if (tid == 0) p.score[clip] = red[0] / (float)nseg;
}

}  // namespace l2s

using namespace l2s;

extern "C" {

static int64_t im_align(int64_t x) { return (x + 255) / 256 * 256; }
int64_t l2s_inverse_mel_workspace_bytes(int N, int L, int n_mels, int n_freqs, int rows_per_call, int iters) {
    const int64_t rows = (int64_t)N * L;
    const int calls = rows_per_call > 0 ? N / rows_per_call : 1;
    return 256 + im_align((3 * (int64_t)(n_mels + n_freqs) + 2) * 4) + 2 * im_align(IM_NNZ * 4) + im_align(rows * (iters > 0 ? iters : 1) * 4) + im_align((int64_t)calls * 4);
}

int l2s_inverse_mel(const float* mel, int log_input, const float* fb, int fb_nnz, const float* init, int N, int L, int n_mels, int n_freqs, int rows_per_call,
                    int iters, float* spec, float* loss_per_iter, int* iters_run, void* ws, int64_t ws_bytes, void* stream) {
    L2S_REQUIRE(mel && fb && init && spec && ws, "inverse_mel: null argument");
    L2S_REQUIRE(N > 0 && L > 0 && iters >= 0, "inverse_mel: sizes");
    L2S_REQUIRE(n_mels > 0 && n_mels <= IM_MAXM && n_freqs > 0 && n_freqs <= IM_MAXF, "inverse_mel: at most 128 mel bands and 576 frequency bins");
    L2S_REQUIRE(fb_nnz > 0 && fb_nnz <= IM_NNZ, "inverse_mel: the filterbank's non-zero count (host-side, exact) must be given and at most 2048");
    if (rows_per_call <= 0) rows_per_call = N;
    L2S_REQUIRE(N % rows_per_call == 0, "inverse_mel: rows_per_call must divide N");
    L2S_REQUIRE(ws_bytes >= l2s_inverse_mel_workspace_bytes(N, L, n_mels, n_freqs, rows_per_call, iters), "inverse_mel: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    const int calls = N / rows_per_call;
    const int64_t rows = (int64_t)N * L;
    char* w = (char*)(((uintptr_t)ws + 255) / 256 * 256);
    int* tab = (int*)w; w += im_align((3 * (int64_t)(n_mels + n_freqs) + 2) * 4);
    float* fwd = (float*)w; w += im_align(IM_NNZ * 4);
    float* bwd = (float*)w; w += im_align(IM_NNZ * 4);
    float* loss_rows = (float*)w; w += im_align(rows * (iters > 0 ? iters : 1) * 4);
    int* iters_call = iters_run ? iters_run : (int*)w;
    ProfScope ps("vocoder_inverse_mel", s);
    hipLaunchKernelGGL(inverse_mel_bands_kernel, dim3(1), dim3(256), 0, s, fb, n_freqs, n_mels, tab, fwd, bwd);
    InvMelP p{mel, init, tab, fwd, bwd, spec, iters > 0 ? loss_rows : nullptr, nullptr, N, L, n_mels, n_freqs, rows_per_call, iters, log_input, 0};
    const unsigned blocks = (unsigned)((rows + IM_ROWS - 1) / IM_ROWS);
    hipLaunchKernelGGL(inverse_mel_kernel, dim3(blocks), dim3(IM_ROWS * 64), 0, s, p);
    if (iters > 0) {
        hipLaunchKernelGGL(inverse_mel_stop_kernel, dim3(calls), dim3(256), 0, s, loss_rows, (int)rows, rows_per_call * L, iters, iters_call, loss_per_iter);
        p.loss_rows = nullptr; p.iters_call = iters_call; p.second = 1;
        hipLaunchKernelGGL(inverse_mel_kernel, dim3(blocks), dim3(IM_ROWS * 64), 0, s, p);      // only the calls a stopping rule ended early re-run
    }
    L2S_CHECK_HIP(hipGetLastError());
    return 0;
}

int64_t l2s_griffin_lim_workspace_bytes(int N, int L) {
    return (int64_t)N * L * GL_LDK * 4 + (int64_t)N * 2 * L * GL_LDK * 8 + 512;
}

int l2s_griffin_lim(const float* power_spec, const float* init_angles, int N, int L, int n_fft, int hop, int iters, float momentum,
                    float* wave, void* ws, int64_t ws_bytes, void* stream) {
    L2S_REQUIRE(power_spec && init_angles && wave && ws, "griffin_lim: null argument");
    L2S_REQUIRE(n_fft == GL_NFFT && hop == GL_HOP, "griffin_lim: built for n_fft = win_length = 1024, hop 256 (hparams.py)");
    L2S_REQUIRE(N > 0 && L >= 5 && iters >= 0, "griffin_lim: sizes");
    L2S_REQUIRE(GL_HOP * (L - 1) <= 30720, "griffin_lim: at most 121 frames per clip (the waveform stays in LDS)");
    L2S_REQUIRE(ws_bytes >= l2s_griffin_lim_workspace_bytes(N, L), "griffin_lim: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    char* w = (char*)(((uintptr_t)ws + 255) / 256 * 256);
    GriffinP p{power_spec, init_angles, (float*)w, (float2*)(w + (int64_t)N * L * GL_LDK * 4), wave, N, L, iters, momentum / (1.0f + momentum)};
    ProfScope ps("vocoder_griffin_lim", s);
    if (GL_HOP * (L - 1) <= 19456) hipLaunchKernelGGL((griffin_lim_kernel<12, 19456>), dim3(N), dim3(768), 0, s, p);
    else hipLaunchKernelGGL((griffin_lim_kernel<8, 30720>), dim3(N), dim3(512), 0, s, p);
    L2S_CHECK_HIP(hipGetLastError());
    return 0;
}

int64_t l2s_estoi_workspace_bytes(int N) { return 512 + 256 + (int64_t)N * 2 * ES_MAXLEN * 4 + (int64_t)N * 2 * ES_MAXFRAMES * ES_NBIN * 4; }

int l2s_estoi(const float* clean, const float* pred, int N, int n_samples, const float* fir, int n_fir, int up, int down, int n_pre_remove, int n_resampled,
              const int* band_lo_hi_host, float* score, void* ws, int64_t ws_bytes, void* stream) {
    L2S_REQUIRE(clean && pred && score && ws && band_lo_hi_host, "estoi: null argument");
    L2S_REQUIRE(N > 0 && n_samples > 0, "estoi: sizes");
    L2S_REQUIRE(n_resampled > 0 && n_resampled <= ES_MAXLEN, "estoi: at most 16 512 samples at 10 kHz per clip (the signals stay in LDS)");
    L2S_REQUIRE(fir ? (n_fir > 0 && up > 0 && down > 0 && n_pre_remove >= 0) : n_resampled == n_samples, "estoi: resampler arguments");
    L2S_REQUIRE(ws_bytes >= l2s_estoi_workspace_bytes(N), "estoi: workspace too small");
    for (int b = 0; b < ES_BANDS; ++b)
        L2S_REQUIRE(band_lo_hi_host[b] >= 0 && band_lo_hi_host[b] <= band_lo_hi_host[ES_BANDS + b] && band_lo_hi_host[ES_BANDS + b] <= ES_NBIN &&
                    (b == 0 || band_lo_hi_host[b] >= band_lo_hi_host[b - 1]), "estoi: band edges");
    hipStream_t s = (hipStream_t)stream;
    char* w = (char*)(((uintptr_t)ws + 255) / 256 * 256);
    int* bands = (int*)w;
    float* rs = (float*)(w + 256);
    float* pw = rs + (int64_t)N * 2 * ES_MAXLEN;
    L2S_CHECK_HIP(hipMemcpyAsync(bands, band_lo_hi_host, 2 * ES_BANDS * sizeof(int), hipMemcpyHostToDevice, s));
    EstoiP p{clean, pred, fir, bands, bands + ES_BANDS, rs, pw, score, N, n_samples, n_fir, up, down, n_pre_remove, n_resampled};
    ProfScope ps("metric_estoi", s);
    hipLaunchKernelGGL(estoi_kernel, dim3(N), dim3(1024), 0, s, p);
    L2S_CHECK_HIP(hipGetLastError());
    return 0;
}

}  // extern "C"
