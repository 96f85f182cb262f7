// Microbenchmark: stream 2 KB rows of four 43 MB arrays (10k rows used, one wave per row,
// 8+1 ints per lane) warm vs after the caches were flushed by a 1 GB sweep.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int int4u __attribute__((ext_vector_type(4), aligned(4)));
__global__ __launch_bounds__(256) void k_rows(const int *a, const int *b, const int *c, const int *d, int narr, unsigned first, unsigned nrows, unsigned stride, long long *sink) {
    const int lane = threadIdx.x & 63;
    const unsigned wave = (blockIdx.x * 256 + threadIdx.x) >> 6, nw = (gridDim.x * 256) >> 6;
    long long acc = 0;
    for (unsigned r = wave; r < nrows; r += nw) {
        const unsigned a0 = first + r * stride + lane * 8;
        const int *arr[4] = {a, b, c, d};
        for (int k = 0; k < narr; ++k) {
            int4u lo = *reinterpret_cast<const int4u *>(arr[k] + a0), hi = *reinterpret_cast<const int4u *>(arr[k] + a0 + 4);
            acc += lo.x + lo.y + lo.z + lo.w + hi.x + hi.y + hi.z + hi.w;
        }
    }
    if (acc == 0x12345) sink[0] = acc;
}
__global__ void k_flush(const int4 *p, size_t n, long long *sink) {
    long long acc = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += p[i].x;
    if (acc == 0x12345) sink[0] = acc;
}
int main() {
    const size_t m2 = 10664000;
    int *arr[4]; long long *sink; int4 *big;
    for (int k = 0; k < 4; ++k) { hipMalloc(&arr[k], m2 * 4); hipMemset(arr[k], 1, m2 * 4); }
    hipMalloc(&sink, 8); hipMalloc(&big, 1ull << 30); hipMemset(big, 0, 1ull << 30);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int narr : {1, 3, 4}) {
        for (int cold = 0; cold < 2; ++cold) {
            float tot = 0;
            for (int rep = 0; rep < 5; ++rep) {
                if (cold) hipLaunchKernelGGL(k_flush, dim3(2048), dim3(256), 0, 0, big, (1ull << 30) / 16, sink);
                else hipLaunchKernelGGL(k_rows, dim3(1024), dim3(256), 0, 0, arr[0], arr[1], arr[2], arr[3], narr, 12012u, 10000u, 513u, sink);
                hipEventRecord(e0);
                hipLaunchKernelGGL(k_rows, dim3(1024), dim3(256), 0, 0, arr[0], arr[1], arr[2], arr[3], narr, 12012u, 10000u, 513u, sink);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1); tot += ms;
            }
            printf("%d arrays x 10000 rows of 2 KB (%.0f MB) %s: %.1f us\n", narr, narr * 10000 * 2052 / 1e6, cold ? "cold" : "warm", tot / 5 * 1e3);
        }
    }
    return 0;
}
