// Microbenchmark: cost per call (SIMD cycles per wave) of the numeric primitives of reina_prims.h and of candidate
// replacements, at the occupancy k_day runs at (4 waves per SIMD).  hipcc --offload-arch=gfx950 -O3 -ffp-contract=off
// -o /tmp/ubench tools/ubench_prims.hip && /tmp/ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include "../reina_model_amd/csrc/reina_prims.h"

// Threefry2x32 (Salmon et al. SC'11), R rounds
template <int R>
__device__ __forceinline__ rp_u2 threefry2x32(uint32_t k0, uint32_t k1, uint32_t c0, uint32_t c1) {
    const uint32_t ks2 = 0x1BD11BDAu ^ k0 ^ k1;
    const int rot[8] = {13, 15, 26, 6, 17, 29, 16, 24};
    uint32_t x0 = c0 + k0, x1 = c1 + k1;
#pragma unroll
    for (int r = 0; r < R; r++) {
        x0 += x1;
        x1 = (x1 << rot[r & 7]) | (x1 >> (32 - rot[r & 7]));
        x1 ^= x0;
        if ((r & 3) == 3) {
            const int s = r / 4 + 1;
            const uint32_t ka = s % 3 == 0 ? k0 : s % 3 == 1 ? k1 : ks2, kb = (s + 1) % 3 == 0 ? k0 : (s + 1) % 3 == 1 ? k1 : ks2;
            x0 += ka;
            x1 += kb + (uint32_t)s;
        }
    }
    rp_u2 o; o.v[0] = x0; o.v[1] = x1; return o;
}

template <int WHAT>
__global__ __launch_bounds__(1024) void k_bench(uint32_t *out, int iters, uint32_t seed) {
    uint32_t a = threadIdx.x * 2654435761u + blockIdx.x + seed, b = a ^ 0x9E3779B9u, acc = 0;
    for (int i = 0; i < iters; i++) {
        if (WHAT == 0) { rp_u4 r = rp_philox(seed, 7, a, b, 3, i); a += r.v[0]; b ^= r.v[1]; acc += r.v[2] ^ r.v[3]; }
        if (WHAT == 1) { rp_u2 r = rp_philox2(seed, a, b + i); a += r.v[0]; b ^= r.v[1]; }
        if (WHAT == 2) { float z = rp_normal_from_u32(a); a = a * 1664525u + 1013904223u + rp_f2u(z); }
        if (WHAT == 3) { float z = rp_expf(rp_uniform24(a) * 4.0f - 2.0f); a = a * 1664525u + 1013904223u + rp_f2u(z); }
        if (WHAT == 4) { float z = rp_logf(rp_uniform24(a) + 0.01f); a = a * 1664525u + 1013904223u + rp_f2u(z); }
        if (WHAT == 5) { float z = rp_gamma_mu_cv(5.1f, 0.86f, seed, 7, a, i, 3, 1); a += rp_f2u(z); }
        if (WHAT == 6) { rp_u2 r = threefry2x32<13>(seed, 7, a, b + i); a += r.v[0]; b ^= r.v[1]; }
        if (WHAT == 7) { rp_u2 r = threefry2x32<20>(seed, 7, a, b + i); a += r.v[0]; b ^= r.v[1]; }
        if (WHAT == 8) { a = a * 1664525u + 1013904223u; b ^= a; }   // baseline: the loop itself
        if (WHAT == 9) { uint64_t p = (uint64_t)a * 0xD256D193u; a = (uint32_t)(p >> 32) ^ b; b = (uint32_t)p + i; }   // one 32x32->64 multiply
    }
    if (acc + a + b == 0x12345u) out[0] = acc;
}

template <int WHAT>
static void run(const char *name, int iters) {
    uint32_t *d; hipMalloc(&d, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k_bench<WHAT><<<256, 1024>>>(d, 16, 1);   // warm-up
    hipEventRecord(e0);
    k_bench<WHAT><<<256, 1024>>>(d, iters, 2);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // 256 workgroups x 16 waves = 4 waves per SIMD on 256 CUs; cycles per call per wave at 2.4 GHz, SIMD shared by 4 waves
    const double cyc = ms * 1e-3 * 2.4e9 / iters / 4.0;
    printf("%-28s %8.3f ms  %7.1f SIMD-cycles per wave-call\n", name, ms, cyc);
    hipFree(d);
}

int main() {
    run<8>("loop baseline (lcg)", 4000);
    run<9>("one 32x32->64 multiply", 4000);
    run<1>("philox2x32-10", 2000);
    run<0>("philox4x32-10", 2000);
    run<6>("threefry2x32-13", 2000);
    run<7>("threefry2x32-20", 2000);
    run<2>("normal_from_u32", 1000);
    run<3>("expf", 2000);
    run<4>("logf", 2000);
    run<5>("gamma(5.1, 0.86)", 200);
    return 0;
}
