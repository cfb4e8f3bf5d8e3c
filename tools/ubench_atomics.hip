// What do a launch's closing atomics cost?  G workgroups each issue K non-returning 32-bit atomic adds from K lanes of one wave at
// the END of an otherwise empty kernel; the launch's duration (HIP events, mean of 50) against where the adds go:
//   same word | K words of one 128-byte line | K words of K lines (the same for every workgroup) | words of the workgroup's own
// hipcc --offload-arch=gfx950 -O3 -o /tmp/ubench_atomics tools/ubench_atomics.hip && /tmp/ubench_atomics
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void k(uint32_t *buf, int K, int mode, int ret, uint32_t *sink) {
    const int t = threadIdx.x;
    if (t < K) {
        uint32_t *p = mode == 0 ? buf : mode == 1 ? buf + (t & 31) : mode == 2 ? buf + t * 64 : buf + 4096 + (blockIdx.x * 64 + t) * 64;
        if (ret) { uint32_t v = atomicAdd(p, 1u); if (v == 0xFFFFFFFFu) *sink = v; }
        else __hip_atomic_fetch_add(p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
int main() {
    uint32_t *buf, *sink;
    hipMalloc(&buf, 256u << 20); hipMemset(buf, 0, 256u << 20); hipMalloc(&sink, 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const char *names[4] = {"same word", "one line", "K lines shared", "own lines"};
    for (int ret = 0; ret < 2; ret++)
    for (int G : {206, 256, 4096})
        for (int K : {0, 1, 8, 32})
            for (int mode = 0; mode < 4; mode++) {
                if (K == 0 && mode) continue;
                for (int w = 0; w < 5; w++) hipLaunchKernelGGL(k, dim3(G), dim3(256), 0, 0, buf, K, mode, ret, sink);
                hipDeviceSynchronize();
                float tot = 0;
                for (int r = 0; r < 50; r++) {
                    hipEventRecord(a); hipLaunchKernelGGL(k, dim3(G), dim3(256), 0, 0, buf, K, mode, ret, sink); hipEventRecord(b);
                    hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b); tot += ms;
                }
                printf("%s G %4d K %2d %-15s %7.2f us per launch, %6.1f ns per atomic\n", ret ? "returning" : "no return", G, K, K ? names[mode] : "(no atomics)", tot / 50 * 1000, K ? tot / 50 * 1e6 / (G * K) : 0.0);
            }
    return 0;
}
