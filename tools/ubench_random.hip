// What is the chip's RANDOM-ACCESS rate?  k_hosp_install on a peak day is a few million scattered accesses to per-agent records spread
// over 10 GB (a target's record and word, its source's count, two bit-plane words per infection; an onset's word and record; ...):
// its bound is not HBM bandwidth (the bytes are a few MB) but how many independent 32-byte sectors per second the memory system
// serves at random -- this measures that roofline.  Every lane makes R accesses at hashed addresses of a buffer of `footprint`
// bytes, `ilp` of them independent (issued back to back), the rest dependent on the previous value; kinds:
//   load4   4-byte loads        load32  32 bytes (2 x dwordx4) of one sector      store4  4-byte stores
//   or      non-returning 32-bit atomic OR      add_ret returning atomic add      cas     compare-and-swap (returning)
// over 16 or 32 waves per CU.  Output: G accesses/s and ns per wave step.
// hipcc --offload-arch=gfx950 -O3 -o /tmp/ubench_random tools/ubench_random.hip && /tmp/ubench_random
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <initializer_list>

__device__ __forceinline__ uint64_t mix(uint64_t x) {
    x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32; x *= 0x94D049BB133111EBull; x ^= x >> 29;
    return x;
}

template <int KIND, int ILP>
__global__ __launch_bounds__(1024) void k_rand(uint32_t *buf, uint64_t words, int rounds, uint32_t seed, uint32_t *sink) {
    const uint64_t lane = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t acc = seed;
    for (int r = 0; r < rounds; r++) {
        uint32_t got[ILP];
#pragma unroll
        for (int k = 0; k < ILP; k++) {
            // (the address depends on the previous round's value: rounds are dependent, the ILP accesses of a round are not)
            const uint64_t at = mix((lane + 1) * 0x9E3779B97F4A7C15ull + (uint64_t)(r * ILP + k) * 0xD1B54A32D192ED03ull + (acc & 1u)) % words;
            uint32_t *p = buf + (at & ~7ull);   // sector-aligned
            if (KIND == 0) got[k] = __builtin_nontemporal_load(p);
            else if (KIND == 1) { const uint4 a = *reinterpret_cast<const uint4 *>(p), b = *reinterpret_cast<const uint4 *>(p + 4); got[k] = a.x ^ b.w; }
            else if (KIND == 2) { *p = acc + k; got[k] = 0; }
            else if (KIND == 3) { __hip_atomic_fetch_or(p, 1u << (r & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); got[k] = 0; }
            else if (KIND == 4) got[k] = __hip_atomic_fetch_add(p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else { uint32_t e = 0; __hip_atomic_compare_exchange_strong(p, &e, 7u, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); got[k] = e; }
        }
#pragma unroll
        for (int k = 0; k < ILP; k++) acc += got[k];
    }
    if (acc == 0x12345u) sink[0] = acc;
}

template <int KIND, int ILP>
static void run(const char *name, uint32_t *buf, uint64_t words, int wgs, uint32_t *sink) {
    const int rounds = 16 / ILP > 0 ? 16 / ILP : 1;
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    for (int w = 0; w < 2; w++) hipLaunchKernelGGL((k_rand<KIND, ILP>), dim3(wgs), dim3(1024), 0, 0, buf, words, rounds, 17u + w, sink);
    (void)hipDeviceSynchronize();
    float tot = 0;
    const int reps = 5;
    for (int r = 0; r < reps; r++) {
        (void)hipEventRecord(a);
        hipLaunchKernelGGL((k_rand<KIND, ILP>), dim3(wgs), dim3(1024), 0, 0, buf, words, rounds, 100u + r, sink);
        (void)hipEventRecord(b);
        (void)hipEventSynchronize(b);
        float ms; (void)hipEventElapsedTime(&ms, a, b); tot += ms;
    }
    const double acc = (double)wgs * 1024 * rounds * ILP, s = tot / reps * 1e-3;
    printf("%-8s ilp %d  %3d WGs x 1024 (%2d waves/CU)  footprint %6.0f MB: %7.2f G accesses/s  (%6.1f us per launch of %.1f M accesses; %5.0f ns per wave step)\n",
           name, ILP, wgs, wgs * 16 / 256, words * 4 / 1e6, acc / s / 1e9, s * 1e6, acc / 1e6, s * 1e9 / (rounds));
}

int main() {
    uint32_t *buf, *sink;
    const uint64_t big = 2400ull << 20;   // words: 9.6 GB -- the per-agent state of 10^8 agents
    if (hipMalloc(&buf, big * 4) != hipSuccess) { printf("hipMalloc failed\n"); return 1; }
    (void)hipMemset(buf, 0, big * 4);
    (void)hipMalloc(&sink, 4);
    for (uint64_t words : {big, (uint64_t)(48ull << 20)}) {   // 9.6 GB | 192 MB (inside the 256 MB Infinity Cache)
        for (int wgs : {256, 512}) {
            run<0, 1>("load4", buf, words, wgs, sink); run<0, 2>("load4", buf, words, wgs, sink); run<0, 4>("load4", buf, words, wgs, sink);
            run<1, 1>("load32", buf, words, wgs, sink); run<1, 4>("load32", buf, words, wgs, sink);
            run<2, 1>("store4", buf, words, wgs, sink); run<2, 4>("store4", buf, words, wgs, sink);
            run<3, 1>("or", buf, words, wgs, sink); run<3, 4>("or", buf, words, wgs, sink);
            run<4, 1>("add_ret", buf, words, wgs, sink); run<4, 4>("add_ret", buf, words, wgs, sink);
            run<5, 1>("cas", buf, words, wgs, sink); run<5, 4>("cas", buf, words, wgs, sink);
        }
    }
    return 0;
}
