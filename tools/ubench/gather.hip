// Microbenchmark: cost of 8-byte gathers from an L2-resident table on gfx950, as a function
// of active lanes and address pattern.  hipcc --offload-arch=gfx950 -O3 -o gather gather.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdlib>
typedef long long i64;
__global__ void k_gather(const i64* __restrict__ table, const int* __restrict__ idx, i64* out, int iters, int active, int stride_mode, unsigned mask) {
    const int lane = threadIdx.x & 63;
    const unsigned wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    i64 acc = 0;
    if (lane < active) {
        unsigned base = wave * 977u;
        for (int i = 0; i < iters; i += 4) {
            unsigned a[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                unsigned j = base + (unsigned)(i + u) * 64u;
                a[u] = stride_mode == 0 ? (unsigned)idx[(j + lane) & mask] : ((j + (unsigned)lane) & mask);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) acc += table[a[u] & mask];
        }
    }
    if (acc == 0x1234567) out[0] = acc;
}
int main() {
    const unsigned tsz = 1u << 17;  // 128k entries * 8 B = 1 MB
    std::vector<i64> h(tsz); std::vector<int> hi(tsz);
    for (unsigned i = 0; i < tsz; ++i) { h[i] = i; hi[i] = rand() & (tsz - 1); }
    i64 *d, *o; int* di;
    hipMalloc(&d, tsz * 8); hipMalloc(&o, 8); hipMalloc(&di, tsz * 4);
    hipMemcpy(d, h.data(), tsz * 8, hipMemcpyHostToDevice); hipMemcpy(di, hi.data(), tsz * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 256;
    for (int mode = 0; mode < 2; ++mode)
        for (int active : {64, 16, 4, 1}) {
            for (int blocks : {256, 1024}) {
                hipLaunchKernelGGL(k_gather, dim3(blocks), dim3(256), 0, 0, d, di, o, iters, active, mode, tsz - 1);
                hipEventRecord(e0);
                hipLaunchKernelGGL(k_gather, dim3(blocks), dim3(256), 0, 0, d, di, o, iters, active, mode, tsz - 1);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                double winstr = (double)blocks * 4 * iters;  // wave gather instructions
                printf("%s active %2d blocks %4d: %.1f us, %.2f ns per wave-gather-instr per CU-slot (%.1f cycles@2.1GHz per instr per CU)\n",
                       mode == 0 ? "random " : "contig ", active, blocks, ms * 1e3, ms * 1e6 / (winstr / 256.0), ms * 1e6 / (winstr / 256.0) * 2.1);
            }
        }
    return 0;
}
