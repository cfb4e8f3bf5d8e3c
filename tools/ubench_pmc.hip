// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on the access shapes of the day's kernels (round-5 verdict item 5;
// /opt/skills/guides/MI355X_MICROARCH.md, HBM: "Other access widths are uncalibrated: calibrate on a known byte count in your own
// access pattern before trusting an absolute").  Every kernel below makes a KNOWN number of accesses of ONE shape over a footprint
// far beyond the 256 MB Infinity Cache; run it once per counter under the profiler, program directly behind `--`:
//     rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out_fetch -- /tmp/ubench_pmc
//     rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d out_write -- /tmp/ubench_pmc
// and tools/pmc_calibration.py divides the counter by the accesses: reported bytes per access, per shape.  The shapes:
//     stream16   coalesced 16 B / lane reads (k_day's bit plane, a dense day's hot words)     -- the guide: reports 1/2
//     stream4    coalesced 4 B / lane reads
//     wstream16  coalesced 16 B / lane stores
//     load4      scattered 4-byte loads, one 32-byte sector each (k_day's word fetches, the infected-plane lookups, a record's claim)
//     load32     scattered 32-byte loads (2 x dwordx4 of one sector: the inline infectee block, a cold record)
//     store4     scattered 4-byte stores (hot words, infector links)
//     store16    scattered 16-byte stores (half a cold record)
//     or         scattered non-returning atomic OR (the bit planes)        add_ret  returning atomic add (a source's count)
//     min64      scattered returning 64-bit atomic min (the claims)
//     pair32 / pair64   two 4-byte loads of one 128-byte line, 32 / 64 bytes apart, counted as ONE access: the request granularity
// The program prints every kernel's name, launches and accesses per launch (stdout -> the json beside the counters).
// hipcc --offload-arch=gfx950 -O3 -o /tmp/ubench_pmc tools/ubench_pmc.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

__device__ __forceinline__ uint64_t mix(uint64_t x) {
    x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32; x *= 0x94D049BB133111EBull; x ^= x >> 29;
    return x;
}

enum { K_STREAM16 = 0, K_STREAM4, K_WSTREAM16, K_LOAD4, K_LOAD32, K_STORE4, K_STORE16, K_OR, K_ADD_RET, K_MIN64, K_PAIR32, K_PAIR64, K_NR };
static const char *NAMES[K_NR] = {"stream16", "stream4", "wstream16", "load4", "load32", "store4", "store16", "or", "add_ret", "min64", "pair32", "pair64"};

// streams: `n` elements, grid-stride; scattered: every lane makes `rounds` accesses at hashed sector addresses of `words` words
template <int KIND>
__global__ __launch_bounds__(1024) void k_pmc(uint32_t *buf, uint64_t words, uint64_t n, int rounds, uint32_t seed, uint32_t *sink) {
    const uint64_t lane = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, lanes = (uint64_t)gridDim.x * blockDim.x;
    uint32_t acc = seed;
    if (KIND == K_STREAM16) {
        typedef uint32_t v4u __attribute__((ext_vector_type(4)));
        const v4u *p = reinterpret_cast<const v4u *>(buf);
        for (uint64_t i = lane; i < n; i += lanes) { const v4u v = __builtin_nontemporal_load(p + i); acc += v.x ^ v.w; }
    } else if (KIND == K_STREAM4) {
        for (uint64_t i = lane; i < n; i += lanes) acc += __builtin_nontemporal_load(buf + i);
    } else if (KIND == K_WSTREAM16) {
        uint4 *p = reinterpret_cast<uint4 *>(buf);
        for (uint64_t i = lane; i < n; i += lanes) p[i] = make_uint4(acc, (uint32_t)i, 0u, 0u);
    } else {
        for (int r = 0; r < rounds; r++) {
            const uint64_t at = mix((lane + 1) * 0x9E3779B97F4A7C15ull + (uint64_t)r * 0xD1B54A32D192ED03ull + seed) % words;
            uint32_t *p = buf + (at & ~7ull);   // sector-aligned
            if (KIND == K_PAIR32 || KIND == K_PAIR64) {
                // two 4-byte loads of ONE 128-byte line, 32 or 64 bytes apart (counted as one access): one request or two?
                uint32_t *q = buf + (at & ~31ull);
                acc += __builtin_nontemporal_load(q) + __builtin_nontemporal_load(q + (KIND == K_PAIR32 ? 8 : 16));
            } else if (KIND == K_LOAD4) acc += __builtin_nontemporal_load(p);
            else if (KIND == K_LOAD32) { const uint4 a = *reinterpret_cast<const uint4 *>(p), b = *reinterpret_cast<const uint4 *>(p + 4); acc += a.x ^ b.w; }
            else if (KIND == K_STORE4) *p = acc + (uint32_t)r;
            else if (KIND == K_STORE16) *reinterpret_cast<uint4 *>(p) = make_uint4(acc, (uint32_t)r, 0u, 0u);
            else if (KIND == K_OR) __hip_atomic_fetch_or(p, 1u << (r & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else if (KIND == K_ADD_RET) acc += __hip_atomic_fetch_add(p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else acc += (uint32_t)__hip_atomic_fetch_min(reinterpret_cast<unsigned long long *>(p), (unsigned long long)lane << 8 | (unsigned)r,
                                                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (acc == 0x12345u) sink[0] = acc;
}

template <int KIND>
static void run(uint32_t *buf, uint64_t words, uint32_t *sink) {
    const bool stream = KIND <= K_WSTREAM16;
    const int wgs = 512, rounds = 16, launches = 3;
    // streams: 4 GiB of the buffer per launch; scattered: 512 x 1024 lanes x 16 accesses = 8.4 M per launch over all of it
    const uint64_t n = stream ? ((KIND == K_STREAM4) ? (1ull << 30) : (1ull << 28)) : 0ull;
    const double accesses = stream ? (double)n : (double)wgs * 1024 * rounds;
    const double bytes_each = KIND == K_STREAM4 ? 4 : (KIND == K_STREAM16 || KIND == K_WSTREAM16) ? 16 : KIND == K_LOAD32 ? 32 : KIND == K_STORE16 ? 16 : KIND == K_MIN64 ? 8 : 4;
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    (void)hipEventRecord(a);
    for (int l = 0; l < launches; l++) hipLaunchKernelGGL((k_pmc<KIND>), dim3(wgs), dim3(1024), 0, 0, buf, words, n, rounds, 17u + l, sink);
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, a, b);
    printf("{\"shape\": \"%s\", \"kernel\": \"k_pmc<%d>\", \"launches\": %d, \"accesses_per_launch\": %.0f, \"bytes_per_access\": %.0f, \"us_per_launch\": %.1f}\n",
           NAMES[KIND], KIND, launches, accesses, bytes_each, ms * 1000.0 / launches);
    fflush(stdout);
}

int main() {
    uint32_t *buf, *sink;
    const uint64_t words = 2400ull << 20;   // 9.6 GB: the per-agent state of 10^8 agents
    if (hipMalloc(&buf, words * 4) != hipSuccess) { printf("hipMalloc failed\n"); return 1; }
    (void)hipMemset(buf, 0xFF, words * 4);   // (0xFF..: the 64-bit minima really change what they find)
    (void)hipMalloc(&sink, 4);
    (void)hipDeviceSynchronize();
    run<K_STREAM16>(buf, words, sink); run<K_STREAM4>(buf, words, sink); run<K_WSTREAM16>(buf, words, sink);
    run<K_LOAD4>(buf, words, sink); run<K_LOAD32>(buf, words, sink); run<K_STORE4>(buf, words, sink); run<K_STORE16>(buf, words, sink);
    run<K_OR>(buf, words, sink); run<K_ADD_RET>(buf, words, sink); run<K_MIN64>(buf, words, sink);
    run<K_PAIR32>(buf, words, sink); run<K_PAIR64>(buf, words, sink);
    (void)hipDeviceSynchronize();
    return 0;
}
