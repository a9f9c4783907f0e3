// How many bytes per second can ONE CU pull when all 256 stream at once?  (the batch-row kernels of the decode step sit at
// ~37 GB/s per CU.)  256 blocks x 512 threads, one per CU; every lane streams float4s with DEPTH independent loads in flight.
//   mode 0  private:   block b reads its own `bytes` (L2 misses; served by the Infinity Cache after the first launch when the total fits)
//   mode 1  xcd-shared: the 32 blocks of an XCD (b % 8) read the SAME `bytes` (first touch misses, the rest hit the XCD's L2)
//   mode 2  lstm-like: `bytes`/3 shared by the 4 blocks {b, b+8.. same column group}, 2*`bytes`/3 shared by 8 blocks - the operand reuse of the
//           4x2 LSTM blocks (weights shared by 4 row groups, activations by 8 column groups of the XCD)
// build: hipcc --offload-arch=gfx950 -O3 -o membw membw.hip ; run: ./membw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int DEPTH>
__global__ __launch_bounds__(512) void stream_kernel(const float4* __restrict__ base, int mode, int64_t f4_per_block, float* __restrict__ sink) {
    const int b = blockIdx.x, tid = threadIdx.x;
    const float4* p0; const float4* p1 = nullptr; int64_t n0 = f4_per_block, n1 = 0;
    if (mode == 0) p0 = base + (int64_t)b * f4_per_block;
    else if (mode == 1) p0 = base + (int64_t)(b % 8) * f4_per_block;
    else {
        const int x = b % 64, y = b / 64;                     // 64 column groups x 4 row groups; XCD = x % 8
        n0 = f4_per_block / 3; n1 = f4_per_block - n0;
        p0 = base + (int64_t)x * n0;                          // "weights" of column group x: read by the 4 row groups
        p1 = base + (int64_t)64 * n0 + (int64_t)y * n1;       // "activations" of row group y: read by all 64 column groups (8 per XCD)
    }
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    auto sweep = [&](const float4* p, int64_t n) {
        for (int64_t i = tid; i + (int64_t)512 * (DEPTH - 1) < n; i += (int64_t)512 * DEPTH) {
            float4 v[DEPTH];
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) v[d] = p[i + (int64_t)512 * d];
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) { acc.x += v[d].x; acc.y += v[d].y; acc.z += v[d].z; acc.w += v[d].w; }
        }
    };
    sweep(p0, n0);
    if (p1) sweep(p1, n1);
    if (acc.x + acc.y + acc.z + acc.w == 1234.5f) sink[b] = acc.x;
}

int main() {
    const int64_t cap = (int64_t)256 * 1024 * 1024;          // 256 MiB buffer
    float4* buf; float* sink;
    hipMalloc(&buf, cap); hipMalloc(&sink, 4096);
    hipMemset(buf, 0, cap);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const char* names[3] = {"private", "xcd-shared", "lstm-like"};
    printf("256 blocks x 512 threads (one per CU), every lane streaming float4; GB/s per CU and aggregate, median of launches 3..12 of 12\n");
    for (int mode = 0; mode < 3; ++mode)
        for (int64_t kb : {128, 256, 590, 1024})
            for (int depth : {4, 8, 16}) {
                const int64_t f4 = kb * 1024 / 16;
                std::vector<float> ts;
                for (int it = 0; it < 12; ++it) {
                    hipEventRecord(e0);
                    if (depth == 4) hipLaunchKernelGGL(stream_kernel<4>, dim3(256), dim3(512), 0, 0, buf, mode, f4, sink);
                    else if (depth == 8) hipLaunchKernelGGL(stream_kernel<8>, dim3(256), dim3(512), 0, 0, buf, mode, f4, sink);
                    else hipLaunchKernelGGL(stream_kernel<16>, dim3(256), dim3(512), 0, 0, buf, mode, f4, sink);
                    hipEventRecord(e1); hipEventSynchronize(e1);
                    float ms; hipEventElapsedTime(&ms, e0, e1);
                    if (it >= 2) ts.push_back(ms);
                }
                std::sort(ts.begin(), ts.end());
                const double us = ts[ts.size() / 2] * 1e3;
                printf("%-10s %5lld KB per block, %2d loads in flight per lane: %7.2f us per launch  (event pair; ~2 us of it is launch)  %6.1f GB/s per CU  %6.2f TB/s aggregate\n",
                       names[mode], (long long)kb, depth, us, kb * 1024 / us / 1e3, 256.0 * kb * 1024 / us / 1e6);
            }
    return 0;
}
