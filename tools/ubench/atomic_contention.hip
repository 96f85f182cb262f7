// Microbenchmark: cost of agent-scope atomics that many workgroups aim at few addresses, the pattern
// of the solver's counters and hub flushes.  One lane per wave (or per workgroup) issues the atomic.
//   hipcc --offload-arch=gfx950 -O3 -o atomic_contention atomic_contention.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

__global__ void k_atomics(unsigned long long *x, int n_addr, int per_wave, int returning, unsigned long long *sink, int stride_words = 32) {
    const int lane = threadIdx.x & 63, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const bool mine = per_wave ? lane == 0 : threadIdx.x == 0;
    if (!mine) return;
    const int who = per_wave ? wave : (int)blockIdx.x;
    unsigned long long *p = x + (size_t)(who % n_addr) * stride_words;  // 256 bytes apart by default
    if (returning) {
        const unsigned long long old = __hip_atomic_fetch_add(p, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old == ~0ull) *sink = old;
    } else {
        __hip_atomic_fetch_add(p, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
__global__ void k_empty(unsigned long long *x) { if (x == nullptr) printf("x"); }

int main() {
    unsigned long long *x, *sink;
    hipMalloc(&x, 4096 * 256); hipMalloc(&sink, 8);
    hipMemset(x, 0, 4096 * 256);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto timeit = [&](auto launch) {
        std::vector<float> t;
        for (int r = 0; r < 30; ++r) {
            hipEventRecord(e0); for (int i = 0; i < 20; ++i) launch(); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); t.push_back(ms * 1000.f / 20.f);
        }
        std::sort(t.begin(), t.end()); return t[t.size() / 2];
    };
    for (int grid : {64, 256, 1024}) {
        const float base = timeit([&] { hipLaunchKernelGGL(k_empty, dim3(grid), dim3(256), 0, 0, x); });
        printf("grid %4d: empty kernel %.2f us per launch (back to back)\n", grid, base);
        for (int per_wave : {0, 1})
            for (int returning : {0, 1})
                for (int n_addr : {1, 4, 64, 4096}) {
                    const float us = timeit([&] { hipLaunchKernelGGL(k_atomics, dim3(grid), dim3(256), 0, 0, x, n_addr, per_wave, returning, sink); });
                    printf("  %s, %s, %4d addresses: %.2f us  (%d atomics, +%.2f us over empty)\n", per_wave ? "one per wave " : "one per group", returning ? "returning" : "no return",
                           n_addr, us, grid * (per_wave ? 4 : 1), us - base);
                }
    }
    // several counters inside one cache line: do they serialise like one address?
    for (int n_addr : {1, 4, 8})
        for (int stride : {1, 32}) {
            const float us = timeit([&] { hipLaunchKernelGGL(k_atomics, dim3(1024), dim3(256), 0, 0, x, n_addr, 1, 0, sink, stride); });
            printf("grid 1024, one per wave, %d addresses %d bytes apart: %.2f us\n", n_addr, stride * 8, us);
        }
    return 0;
}
