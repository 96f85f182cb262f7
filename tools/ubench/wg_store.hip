// Microbenchmark: how fast can ONE workgroup (4 waves) write a contiguous run of 16-byte items?
// Each wave writes runs of `per` items side by side (64 lanes, one int4 each), `nruns` runs.
// Reports cycles from first issue to last issue, and to all stores acknowledged.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256) void k_store(int4 *dst, unsigned per, unsigned nruns, unsigned long long *out) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (unsigned j = w; j < nruns; j += 4) {
        for (unsigned i = lane; i < per; i += 64) dst[(size_t)j * per + i] = make_int4((int)j, (int)i, (int)(i + 1), 7);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_s_waitcnt(0);
    const unsigned long long t2 = __builtin_amdgcn_s_memtime();
    if (lane == 0) { out[(blockIdx.x * 4 + w) * 2] = t1 - t0; out[(blockIdx.x * 4 + w) * 2 + 1] = t2 - t0; }
}
int main() {
    int4 *dst; unsigned long long *out, h[64];
    hipMalloc(&dst, 64 << 20); hipMalloc(&out, sizeof(h));
    for (unsigned blocks : {1u, 8u}) for (unsigned per : {71u, 64u, 256u}) for (unsigned nruns : {4u, 16u, 41u, 64u}) {
        for (int rep = 0; rep < 3; ++rep) { hipLaunchKernelGGL(k_store, dim3(blocks), dim3(256), 0, 0, dst, per, nruns, out); hipDeviceSynchronize(); }
        hipMemcpy(h, out, sizeof(unsigned long long) * 8, hipMemcpyDeviceToHost);
        printf("blocks %u runs %2u of %3u items (%6.1f KB per block): wave 0 issue %5llu cycles, acknowledged %5llu cycles (%.2f us at 2.3 GHz)\n", blocks, nruns, per, nruns * per * 16 / 1024.0, h[0], h[1], h[1] / 2300.0);
    }
    return 0;
}
