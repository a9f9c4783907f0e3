// Small decoder-side kernels: Content.encode's adaptive pooling + Gumbel softmax (decoder.py:239-260),
// layout changes into the frag16 operand layout, the stop-token bookkeeping (decoder.py:429-435) and
// the channel-first transposes of the boundary.
#include "l2s_common.h"

namespace l2s {

// adaptive_avg_pool1d of each map (B, L_j, C) to m bins, bin i = [floor(i*L/m), ceil((i+1)*L/m)), written
// side by side: out[b][i][j*C + c]
__global__ __launch_bounds__(256) void pool_cat_kernel(const PoolCatP p) {
    const int64_t total = (int64_t)p.B * p.m * p.nmaps * p.C;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int c = idx % p.C;
        int64_t r = idx / p.C;
        const int j = r % p.nmaps;
        r /= p.nmaps;
        const int i = r % p.m;
        const int b = r / p.m;
        const int L = p.L[j];
        const int s = (i * L) / p.m;
        const int e = ((i + 1) * L + p.m - 1) / p.m;
        const float* x = p.x[j] + (int64_t)b * L * p.ld[j] + c;
        float acc = 0.f;
        for (int t = s; t < e; ++t) acc += x[(int64_t)t * p.ld[j]];
        p.out[((int64_t)b * p.m + i) * (p.nmaps * p.C) + j * p.C + c] = acc / (float)(e - s);
    }
}

int launch_pool_cat(const PoolCatP& p, hipStream_t s) {
    const int64_t total = (int64_t)p.B * p.m * p.nmaps * p.C;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    ProfScope ps("content_adaptive_pool_cat", s);
    hipLaunchKernelGGL(pool_cat_kernel, dim3(blocks), dim3(256), 0, s, p);
    L2S_CHECK_HIP(hipGetLastError());
    return 0;
}

// one block per row: z = softmax((l + g) / tau) -> z[row*ldz + j] (columns n..ldz-1 zeroed),
// dis = softmax(l) (optional)
__device__ __forceinline__ float blk_reduce(float x, float* scratch, bool is_max) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float y = __shfl_xor(x, o);
        x = is_max ? fmaxf(x, y) : x + y;
    }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = x;
    __syncthreads();
    return is_max ? fmaxf(fmaxf(scratch[0], scratch[1]), fmaxf(scratch[2], scratch[3]))
                  : (scratch[0] + scratch[1]) + (scratch[2] + scratch[3]);
}

__global__ __launch_bounds__(256) void gumbel_softmax_kernel(const float* __restrict__ logits, const float* __restrict__ gumbel,
                                                             int n, float tau, float* __restrict__ z, int ldz,
                                                             float* __restrict__ dis) {
    __shared__ float scratch[4];
    const int row = blockIdx.x, tid = threadIdx.x;
    const float* l = logits + (int64_t)row * n;
    const float* g = gumbel + (int64_t)row * n;
    float y[2], x[2];
    float my = -INFINITY, mxv = -INFINITY;
    int cnt = 0;
    for (int j = tid; j < n; j += 256, ++cnt) {
        x[cnt] = l[j];
        y[cnt] = (l[j] + g[j]) / tau;
        my = fmaxf(my, y[cnt]);
        mxv = fmaxf(mxv, x[cnt]);
    }
    my = blk_reduce(my, scratch, true);
    mxv = blk_reduce(mxv, scratch, true);
    float sy = 0.f, sx = 0.f;
    cnt = 0;
    for (int j = tid; j < n; j += 256, ++cnt) {
        y[cnt] = expf(y[cnt] - my);
        x[cnt] = expf(x[cnt] - mxv);
        sy += y[cnt];
        sx += x[cnt];
    }
    sy = blk_reduce(sy, scratch, false);
    sx = blk_reduce(sx, scratch, false);
    cnt = 0;
    for (int j = tid; j < n; j += 256, ++cnt) {
        z[(int64_t)row * ldz + j] = y[cnt] / sy;
        if (dis) dis[(int64_t)row * n + j] = x[cnt] / sx;
    }
    for (int j = n + tid; j < ldz; j += 256) z[(int64_t)row * ldz + j] = 0.f;
}

int launch_gumbel_softmax(const float* logits, const float* gumbel, int rows, int n, float tau, float* z, int ldz,
                          float* dis, hipStream_t s) {
    L2S_REQUIRE(n <= 512, "gumbel softmax width");
    ProfScope ps("content_gumbel_softmax", s);
    hipLaunchKernelGGL(gumbel_softmax_kernel, dim3(rows), dim3(256), 0, s, logits, gumbel, n, tau, z, ldz, dis);
    L2S_CHECK_HIP(hipGetLastError());
    return 0;
}

// plain x[b][k] (or one broadcast row) -> frag16 buffer with Kfrag columns at column offset koff; rows padded to 16
// are written as zeros so padded batch rows stay finite.
__global__ __launch_bounds__(256) void to_frag_kernel(const float* __restrict__ x, int ldx, int B, int K,
                                                      float* __restrict__ frag, int Kfrag, int koff, int broadcast_row) {
    const int Bp = (B + 15) & ~15;
    const int64_t total = (int64_t)Bp * K;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int k = idx % K;
        const int b = idx / K;
        float v = 0.f;
        if (b < B) v = broadcast_row ? x[k] : x[(int64_t)b * ldx + k];
        frag[frag16_index(b, koff + k, Kfrag)] = v;
    }
}

int launch_to_frag(const float* x, int ldx, int B, int K, float* frag, int Kfrag, int koff, int broadcast_row, hipStream_t s) {
    const int64_t total = (int64_t)((B + 15) & ~15) * K;
    int blocks = (int)((total + 255) / 256);
    ProfScope ps("to_frag16", s);
    hipLaunchKernelGGL(to_frag_kernel, dim3(blocks), dim3(256), 0, s, x, ldx, B, K, frag, Kfrag, koff, broadcast_row);
    L2S_CHECK_HIP(hipGetLastError());
    return 0;
}

__global__ __launch_bounds__(256) void from_frag_kernel(const float* __restrict__ frag, int Kfrag, int B, int K,
                                                        float* __restrict__ out, int ldo, int ooff) {
    const int64_t total = (int64_t)B * K;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int k = idx % K;
        const int b = idx / K;
        out[(int64_t)b * ldo + ooff + k] = frag[frag16_index(b, k, Kfrag)];
    }
}
int launch_from_frag(const float* frag, int Kfrag, int B, int K, float* out, int ldo, int ooff, hipStream_t s) {
    const int64_t total = (int64_t)B * K;
    ProfScope ps("from_frag16", s);
    hipLaunchKernelGGL(from_frag_kernel, dim3((int)((total + 255) / 256)), dim3(256), 0, s, frag, Kfrag, B, K, out, ldo, ooff);
    L2S_CHECK_HIP(hipGetLastError());
    return 0;
}

__global__ __launch_bounds__(256) void tile_rows_kernel(const float* __restrict__ src, int lds, float* __restrict__ dst, int ldd,
                                                        int B, int T, int C) {
    const int64_t total = (int64_t)B * T * C;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int c = idx % C;
        const int64_t r = idx / C;
        const int b = r / T;
        dst[r * ldd + c] = src[(int64_t)b * lds + c];
    }
}
int launch_tile_rows(const float* src, int lds, float* dst, int ldd, int B, int T, int C, hipStream_t s) {
    const int64_t total = (int64_t)B * T * C;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    ProfScope ps("tile_rows", s);
    hipLaunchKernelGGL(tile_rows_kernel, dim3(blocks), dim3(256), 0, s, src, lds, dst, ldd, B, T, C);
    L2S_CHECK_HIP(hipGetLastError());
    return 0;
}

__global__ __launch_bounds__(256) void fill_kernel(float* p, int64_t n, float v) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) p[i] = v;
}
int launch_fill(float* p, int64_t n, float v, hipStream_t s) {
    if (n <= 0) return 0;
    int blocks = (int)((n + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    ProfScope ps("fill", s);
    hipLaunchKernelGGL(fill_kernel, dim3(blocks), dim3(256), 0, s, p, n, v);
    L2S_CHECK_HIP(hipGetLastError());
    return 0;
}

// (B,S,C) -> (B,C,S) through a 32x33 LDS tile
__global__ __launch_bounds__(256) void transpose_bsc_kernel(const float* __restrict__ in, int S, int C, float* __restrict__ out) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z, s0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8) {
        const int s = s0 + r, c = c0 + tx;
        tile[r][tx] = (s < S && c < C) ? in[((int64_t)b * S + s) * C + c] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int c = c0 + r, s = s0 + tx;
        if (c < C && s < S) out[((int64_t)b * C + c) * S + s] = tile[tx][r];
    }
}
int launch_transpose_bsc(const float* in, int B, int S, int C, float* out, hipStream_t s) {
    ProfScope ps("transpose_bsc", s);
    hipLaunchKernelGGL(transpose_bsc_kernel, dim3((S + 31) / 32, (C + 31) / 32, B), dim3(256), 0, s, in, S, C, out);
    L2S_CHECK_HIP(hipGetLastError());
    return 0;
}

// lengths[b] = first i+1 with stop[b][i] > 0 (sigmoid > 0.5), else S
__global__ __launch_bounds__(64) void output_lengths_kernel(const float* __restrict__ stop, int S, int64_t* __restrict__ lengths) {
    const int b = blockIdx.x, lane = threadIdx.x;
    int first = S;
    for (int i = lane; i < S; i += 64)
        if (stop[(int64_t)b * S + i] > 0.f) { first = i + 1; break; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) first = min(first, __shfl_xor(first, o));
    if (lane == 0) lengths[b] = first;
}
int launch_output_lengths(const float* stop, int B, int S, int64_t* lengths, hipStream_t s) {
    ProfScope ps("output_lengths", s);
    hipLaunchKernelGGL(output_lengths_kernel, dim3(B), dim3(64), 0, s, stop, S, lengths);
    L2S_CHECK_HIP(hipGetLastError());
    return 0;
}

// stop_const[b] = dot(ecell[b][0:512], w_tail[0:512]) + bias   (the encoder_cell half of stop_token_layer,
// constant over the decode steps; decoder.py:429)
__global__ __launch_bounds__(64) void stop_const_kernel(const float* __restrict__ ecell, const float* __restrict__ w_tail,
                                                        const float* __restrict__ bias, float* __restrict__ out) {
    const int b = blockIdx.x, lane = threadIdx.x;
    float acc = 0.f;
    for (int j = lane; j < 512; j += 64) acc = fmaf(ecell[(int64_t)b * 512 + j], w_tail[j], acc);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if (lane == 0) out[b] = acc + bias[0];
}
int launch_stop_const(const float* ecell, const float* w_tail, const float* bias, int B, float* out, hipStream_t s) {
    ProfScope ps("stop_const", s);
    hipLaunchKernelGGL(stop_const_kernel, dim3(B), dim3(64), 0, s, ecell, w_tail, bias, out);
    L2S_CHECK_HIP(hipGetLastError());
    return 0;
}

// SpeakerEncoder front-end (audio.py:121,131: torchaudio MelSpectrogram n_fft 400 / hop 160, centre + reflect padding):
// frames[(b*L + l)*400 + j] = reflect_pad(audio[b])[l*160 + j] * hann[j]
__global__ __launch_bounds__(256) void frame_window_kernel(const float* __restrict__ audio, int B, int N, int L, int n_fft, int hop,
                                                           const float* __restrict__ window, float* __restrict__ frames) {
    const int64_t total = (int64_t)B * L * n_fft;
    const int half = n_fft / 2;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int j = idx % n_fft;
        const int64_t r = idx / n_fft;
        const int l = r % L;
        const int b = r / L;
        int i = l * hop + j - half;
        if (i < 0) i = -i;
        if (i >= N) i = 2 * (N - 1) - i;
        frames[idx] = audio[(int64_t)b * N + i] * window[j];
    }
}
int launch_frame_window(const float* audio, int B, int N, int L, int n_fft, int hop, const float* window, float* frames, hipStream_t s) {
    const int64_t total = (int64_t)B * L * n_fft;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 8192) blocks = 8192;
    ProfScope ps("spk_frame_window", s);
    hipLaunchKernelGGL(frame_window_kernel, dim3(blocks), dim3(256), 0, s, audio, B, N, L, n_fft, hop, window, frames);
    L2S_CHECK_HIP(hipGetLastError());
    return 0;
}

// power[r][k] = re[r][k]^2 + im[r][k]^2 from the DFT GEMM output spec[r] = [re(0..nf-1) | im(0..nf-1)] (ld = lds); columns nf..ldp-1 zero
__global__ __launch_bounds__(256) void power_kernel(const float* __restrict__ spec, int lds, int64_t rows, int nf, float* __restrict__ power, int ldp) {
    const int64_t total = rows * ldp;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int k = idx % ldp;
        const int64_t r = idx / ldp;
        float v = 0.f;
        if (k < nf) {
            const float re = spec[r * lds + k], im = spec[r * lds + nf + k];
            v = re * re + im * im;
        }
        power[idx] = v;
    }
}
int launch_power(const float* spec, int lds, int64_t rows, int nf, float* power, int ldp, hipStream_t s) {
    const int64_t total = rows * ldp;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 8192) blocks = 8192;
    ProfScope ps("spk_power_spectrum", s);
    hipLaunchKernelGGL(power_kernel, dim3(blocks), dim3(256), 0, s, spec, lds, rows, nf, power, ldp);
    L2S_CHECK_HIP(hipGetLastError());
    return 0;
}

}  // namespace l2s
