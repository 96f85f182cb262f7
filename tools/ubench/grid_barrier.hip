// Microbenchmark: cost of a grid-wide barrier built from device-scope atomics on MI355X, against
// the cost of a kernel boundary inside a hipGraph, for 256 / 512 / 1024 resident workgroups.
// flat: every workgroup arrives on one counter; tree: one counter per XCD (blockIdx % 8), the last
// arrival of each XCD arrives on a top counter.  The last arrival bumps a generation word that the
// others poll.  Every spin is bounded: a barrier that does not complete sets a flag and all leave.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <chrono>
struct Bar { unsigned cnt[9 * 32]; unsigned gen; unsigned fail; };
__device__ __forceinline__ unsigned ld(unsigned *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ bool barrier(Bar *b, unsigned nwg, unsigned &my_gen, int tree) {
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        bool last;
        if (tree) {
            const unsigned x = blockIdx.x & 7, per = (nwg - x + 7) / 8;
            last = false;
            if (__hip_atomic_fetch_add(&b->cnt[x * 32], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == per - 1) {
                __hip_atomic_store(&b->cnt[x * 32], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                last = __hip_atomic_fetch_add(&b->cnt[8 * 32], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (nwg < 8 ? nwg : 8) - 1;
                if (last) __hip_atomic_store(&b->cnt[8 * 32], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        } else {
            last = __hip_atomic_fetch_add(&b->cnt[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nwg - 1;
            if (last) __hip_atomic_store(&b->cnt[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (last) {
            __hip_atomic_fetch_add(&b->gen, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            unsigned spins = 0;
            while (ld(&b->gen) == my_gen) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > (1u << 22) || ld(&b->fail)) { __hip_atomic_store(&b->fail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); ok = false; break; }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    ++my_gen;
    __syncthreads();
    return ok;
}
__global__ __launch_bounds__(256) void k_bar(Bar *b, unsigned nwg, int iters, int tree, unsigned *data) {
    unsigned my_gen = ld(&b->gen);   // all workgroups read it before anyone can bump it? no: read at launch, host resets between runs
    my_gen = 0;
    for (int i = 0; i < iters; ++i) {
        if (threadIdx.x == 0) data[blockIdx.x * 32] = (unsigned)i;   // some work visible across the barrier
        if (!barrier(b, nwg, my_gen, tree)) return;
    }
}
__global__ void k_empty(unsigned *data) { if (threadIdx.x == 0 && data[0] == 0xFFFFFFFFu) data[1] = 1; }
int main() {
    Bar *b; unsigned *data;
    hipMalloc(&b, sizeof(Bar)); hipMalloc(&data, 1024 * 32 * 4);
    hipMemset(data, 0, 1024 * 32 * 4);
    hipStream_t s; hipStreamCreate(&s);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000;
    for (int tree : {0, 1}) for (unsigned nwg : {64u, 256u, 512u, 1024u}) {
        float best = 1e9f; unsigned fail = 0;
        for (int rep = 0; rep < 3; ++rep) {
            hipMemsetAsync(b, 0, sizeof(Bar), s);
            hipEventRecord(e0, s);
            hipLaunchKernelGGL(k_bar, dim3(nwg), dim3(256), 0, s, b, nwg, iters, tree, data);
            hipEventRecord(e1, s);
            hipStreamSynchronize(s);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            best = ms < best ? ms : best;
            Bar h; hipMemcpy(&h, b, sizeof(Bar), hipMemcpyDeviceToHost); fail |= h.fail;
        }
        printf("%s barrier, %4u workgroups: %.2f us per barrier%s\n", tree ? "tree" : "flat", nwg, best * 1e3 / iters, fail ? "  (FAILED: a spin ran out)" : "");
        fflush(stdout);
    }
    // kernel boundary inside a graph: chain of empty kernels with the same grids
    for (unsigned nwg : {1u, 256u, 1024u}) {
        hipGraph_t g; hipGraphExec_t ge;
        hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
        for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(k_empty, dim3(nwg), dim3(256), 0, s, data);
        hipStreamEndCapture(s, &g);
        hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
        float best = 1e9f;
        for (int rep = 0; rep < 5; ++rep) {
            hipEventRecord(e0, s); hipGraphLaunch(ge, s); hipEventRecord(e1, s); hipStreamSynchronize(s);
            float ms; hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best;
        }
        printf("graph of 200 empty kernels, %4u workgroups each: %.2f us per kernel\n", nwg, best * 1e3 / 200);
        hipGraphExecDestroy(ge); hipGraphDestroy(g);
    }
    return 0;
}
