// What does ONE all-to-all edge of a persistent (single-launch) decode step cost on this chip?  VERDICT r3 item 4 / SURVEY section 7 item 7:
// a weight-stationary step for <= 32 rows would keep the 21 MB of step weights in LDS (82 KB per CU) and exchange h / c / q / u between its
// four phases inside the launch.  This skeleton runs that exchange pattern with nothing else: 256 workgroups (one per CU, forced by their LDS
// size), `steps` x 4 edges; per edge every workgroup publishes its 1/256 share of an n-float vector as 8-byte {epoch, value} granules (one sc1
// store each: the data is the flag, cdna_hip_programming.md Guideline 16 R2), then gathers ALL n granules (relaxed agent-scope 8-byte loads,
// re-swept until every tag carries the epoch), stages them in LDS and does a token amount of arithmetic on them.  Spins are bounded.
// Prints us per edge (host events around the launch / (steps * 4), and the device clock of workgroup 0) for n = 256 .. 16384 floats.
//   build: hipcc --offload-arch=gfx950 -O3 -o persist_probe persist_probe.hip ; run: ./persist_probe      -> profiles/r04_persist_edge_probe.txt
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

typedef unsigned long long u64;
constexpr int NWG = 256, NT = 256, MAXN = 16384;

__global__ __launch_bounds__(NT) void edge_kernel(u64* gran /*[4][n]*/, int n, int steps, int work, unsigned* tmo, float* out, u64* clk) {
    __shared__ float vec[MAXN];
    __shared__ float pad[20000];                  // > 80 KB with vec: one workgroup per CU
    __shared__ int bad;
    const int wg = blockIdx.x, tid = threadIdx.x;
    const int share = n / NWG;                    // floats this workgroup publishes per edge
    if (tid == 0) bad = 0;
    pad[tid] = (float)tid;
    float mine = 1.0f + 0.001f * wg;
    __syncthreads();
    u64 t0 = 0;
    for (int s = 0; s < steps; ++s) {
        if (wg == 0 && tid == 0 && s == 8) t0 = wall_clock64();
        for (int e = 0; e < 4; ++e) {
            const unsigned epoch = (unsigned)(s * 4 + e + 1);
            u64* g = gran + (size_t)e * n;
            // publish
            if (tid < share) __hip_atomic_store(g + wg * share + tid, ((u64)epoch << 32) | (u64)__float_as_uint(mine + tid), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // gather everything
            for (int base = 0; base < n; base += NT * 4) {
                unsigned spins = 0;
                for (;;) {
                    bool ok = true;
                    float v[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int i = base + k * NT + tid;
                        u64 x = i < n ? __hip_atomic_load(g + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : ((u64)epoch << 32);
                        ok &= (unsigned)(x >> 32) == epoch;
                        v[k] = __uint_as_float((unsigned)x);
                    }
                    if (__all(ok)) {
#pragma unroll
                        for (int k = 0; k < 4; ++k) { const int i = base + k * NT + tid; if (i < n) vec[i] = v[k]; }
                        break;
                    }
                    if (++spins > 200000u || __hip_atomic_load(tmo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { __hip_atomic_store(tmo, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); bad = 1; break; }
                    __builtin_amdgcn_s_sleep(1);
                }
                if (bad) break;
            }
            __syncthreads();
            if (bad) return;
            // token phase work on the gathered vector (a GEMV slice would sit here): `work` passes over it from LDS
            float acc = 0.f;
            for (int w = 0; w < work; ++w)
                for (int i = tid; i < n; i += NT) acc = fmaf(vec[i], pad[(i + w) & 16383 % 20000], acc);
            mine = 1.0f + 1e-9f * acc;
            __syncthreads();
        }
    }
    if (wg == 0 && tid == 0) { clk[0] = wall_clock64() - t0; }
    if (tid == 0) out[wg] = mine;
}

int main() {
    u64* gran; unsigned* tmo; float* out; u64* clk;
    hipMalloc(&gran, sizeof(u64) * 4 * MAXN); hipMalloc(&tmo, 4); hipMalloc(&out, 4 * NWG); hipMalloc(&clk, 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    int nb = 0;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, edge_kernel, NT, 0);
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    printf("device %s, %d CUs; occupancy query: %d workgroup(s) of the probe per CU\n", prop.name, prop.multiProcessorCount, nb);
    if (prop.multiProcessorCount < NWG || nb < 1) { printf("cannot keep %d workgroups resident\n", NWG); return 1; }
    printf("%8s %6s %6s | %12s %12s | %s\n", "n floats", "KB gr", "work", "us/edge host", "us/edge dev", "status");
    const int steps = 300;
    for (int work : {0, 4})
        for (int n : {256, 512, 1024, 2048, 4096, 8192, 16384}) {
            std::vector<float> ts; double dev = 0; unsigned bad = 0;
            for (int rep = 0; rep < 5; ++rep) {
                hipMemset(gran, 0, sizeof(u64) * 4 * MAXN); hipMemset(tmo, 0, 4);
                hipDeviceSynchronize();
                hipEventRecord(e0);
                hipLaunchKernelGGL(edge_kernel, dim3(NWG), dim3(NT), 0, 0, gran, n, steps, work, tmo, out, clk);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                u64 c; hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost); hipMemcpy(&bad, tmo, 4, hipMemcpyDeviceToHost);
                if (rep) { ts.push_back(ms); dev = (double)c / 100.0 / ((steps - 8) * 4); }      // 100 MHz wall clock
                if (bad) break;
            }
            std::sort(ts.begin(), ts.end());
            printf("%8d %6.1f %6d | %12.2f %12.2f | %s\n", n, n * 8 / 1024.0, work, ts.empty() ? -1.0 : ts[ts.size() / 2] * 1e3 / (steps * 4), dev, bad ? "TIMEOUT" : "ok");
            fflush(stdout);
            if (bad) break;
        }
    return 0;
}
