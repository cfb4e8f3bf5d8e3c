// Microbenchmark: latency of ONE round of scattered 4-byte loads (one per lane, 3 workgroups of 1024 -- the shape of
// k_open's level-0 tracing at 1e8 agents) from a buffer of a given size, cold and then repeated over OTHER random
// addresses of the same buffer, with a 400 MB stream in between or not.  Question: are the 14 us that the first
// dependent hop of the tracing loop takes at 1e8 agents (2.6 us for the same hop one level later) address translation?
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/ubench_scatter tools/ubench_scatter.hip && /tmp/ubench_scatter
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__global__ __launch_bounds__(1024) void k_scatter(const uint32_t *buf, uint64_t words, uint32_t seed, int rounds, uint64_t *t, uint32_t *sink) {
    const uint32_t lane = blockIdx.x * 1024 + threadIdx.x;
    uint32_t acc = 0;
    for (int r = 0; r < rounds; r++) {
        uint64_t x = (uint64_t)(lane + 1) * 0x9E3779B97F4A7C15ull + (uint64_t)(seed + r) * 0xD1B54A32D192ED03ull;
        x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32;
        const uint64_t at = x % words;
        const uint64_t t0 = wall_clock64();
        const uint32_t v = __builtin_nontemporal_load(buf + at);
        acc += v;
        // the wave's time until its slowest lane has its word
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const uint64_t t1 = wall_clock64();
        if ((threadIdx.x & 63) == 0) atomicMax((unsigned long long *)&t[r], (unsigned long long)(t1 - t0));
        __syncthreads();
    }
    if (acc == 0x12345u) sink[0] = acc;
}

__global__ __launch_bounds__(1024) void k_stream(const uint4 *buf, uint64_t n16, uint32_t *sink) {
    uint32_t acc = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * 1024 + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * 1024) {
        const uint4 v = buf[i];
        acc += v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345u) sink[0] = acc;
}

// what the day's last launch leaves behind: scattered 16-byte stores and integer atomics all over the buffer
__global__ __launch_bounds__(1024) void k_dirty(uint32_t *buf, uint64_t words, uint32_t seed, int per_lane) {
    const uint32_t lane = blockIdx.x * 1024 + threadIdx.x;
    for (int r = 0; r < per_lane; r++) {
        uint64_t x = (uint64_t)(lane + 1) * 0x9E3779B97F4A7C15ull + (uint64_t)(seed + r) * 0xD1B54A32D192ED03ull;
        x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32;
        const uint64_t at = (x % words) & ~3ull;
        *(uint4 *)(buf + at) = make_uint4(1, 1, 1, 1);
        atomicAdd(buf + ((x >> 7) % words), 1u);
    }
}

int main() {
    const uint64_t sizes[] = {64ull << 20, 400ull << 20, 3200ull << 20, 6400ull << 20, 12800ull << 20};
    uint64_t *t; uint32_t *sink;
    hipMalloc(&t, 64); hipMalloc(&sink, 4);
    uint32_t *stream; hipMalloc(&stream, 400ull << 20); hipMemset(stream, 1, 400ull << 20);
    for (uint64_t sz : sizes) {
        uint32_t *buf;
        if (hipMalloc(&buf, sz) != hipSuccess) { printf("no %llu MB\n", (unsigned long long)(sz >> 20)); continue; }
        hipMemset(buf, 1, sz);
        for (int with_stream = 0; with_stream < 4; with_stream++) {
            double sum[4] = {0, 0, 0, 0};
            const int reps = 20;
            for (int rep = 0; rep < reps; rep++) {
                hipMemset(t, 0, 64);
                if (with_stream == 1) k_stream<<<256, 1024>>>((const uint4 *)stream, (400ull << 20) / 16, sink);
                if (with_stream >= 2) k_dirty<<<256, 1024>>>(buf, sz / 4, 77 + rep, with_stream == 2 ? 1 : 8);
                k_scatter<<<3, 1024>>>(buf, sz / 4, 1000 + rep * 7, 4, t, sink);
                uint64_t h[4];
                hipMemcpy(h, t, 32, hipMemcpyDeviceToHost);
                for (int r = 0; r < 4; r++) sum[r] += (double)h[r] / 100.0;   // 100 MHz wall clock -> us
            }
            printf("%6llu MB  %s  slowest wave, us: round 0 %6.2f   1 %6.2f   2 %6.2f   3 %6.2f\n", (unsigned long long)(sz >> 20),
                   with_stream == 0 ? "back to back           " : with_stream == 1 ? "after a 400 MB stream  " : with_stream == 2 ? "after 262 K scattered writes" : "after 2.1 M scattered writes", sum[0] / reps, sum[1] / reps, sum[2] / reps, sum[3] / reps);
        }
        hipFree(buf);
    }
    return 0;
}
