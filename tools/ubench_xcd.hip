// Round-4 verdict item 6: what does a barrier between the phases of a day cost when the whole engine sits on ONE XCD?
// A HUS-sized population (1.7 M agents: 6.7 MB of hot words, 211 KB per bit plane) is a chain of three launches at the
// dispatch floor; every cross-workgroup hand-off inside a launch pays agent-scope coherence across the 8 XCDs (L2 write-back +
// L1 invalidate).  On one XCD all 32 CUs share ONE L2: stores land there, loads that bypass the per-CU L1 (sc1) read them back,
// no L2 write-back is needed.  Measured here:
//   (1) which XCDs a CU-masked stream's workgroups land on (HW_REG_XCC_ID), for two ways of choosing the 32 mask bits;
//   (2) P phases separated by a counter barrier among W workgroups, every phase handing 1 KB per workgroup to its neighbour
//       (checked: a stale read fails the run), in three forms:
//         xcd-local   masked stream, relaxed atomics + sc1 loads, no fences            (valid only inside one XCD)
//         agent       the same barrier with release / acquire fences at agent scope    (what the engine's launches use today)
//         launches    one launch per phase                                             (the boundary the barrier replaces)
// hipcc --offload-arch=gfx950 -O3 -o /tmp/ubench_xcd tools/ubench_xcd.hip && /tmp/ubench_xcd
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define GAS __attribute__((address_space(1)))

__device__ __forceinline__ uint32_t xcc_id() {
    uint32_t v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 15u;
}

__global__ void k_where(uint32_t *xcc_of_block) {
    if (threadIdx.x == 0) xcc_of_block[blockIdx.x] = xcc_id();
}

// mode 0: xcd-local (no fences, sc1 loads of the payload); mode 1: agent-scope release / acquire around the counter
__device__ __forceinline__ void barrier(uint32_t *counter, uint32_t target, int mode) {
    __syncthreads();
    if (threadIdx.x == 0) {
        if (mode == 1) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        uint32_t polls = 0;   // (bounded: a launch that is not resident as a whole must not hang the box)
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target && ++polls < (1u << 22)) __builtin_amdgcn_s_sleep(1);
        if (mode == 1) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
}

// P phases; in phase p workgroup b writes 256 words (p, b-tagged) to its slot, after the barrier reads its neighbour's slot
__global__ __launch_bounds__(256) void k_phases(uint32_t *slots, uint32_t *counter, uint32_t *bad, int P, int mode, uint32_t base) {
    const uint32_t W = gridDim.x, b = blockIdx.x, t = threadIdx.x;
    for (int p = 0; p < P; p++) {
        // (every store drained by the barrier's s_waitcnt / release)
        slots[(size_t)b * 256 + t] = base + (uint32_t)p * 1000003u + b * 257u + t;
        barrier(counter, (uint32_t)(p + 1) * W, mode);
        const uint32_t nb = (b + 1) % W;
        uint32_t got;
        if (mode == 0) got = __hip_atomic_load(&slots[(size_t)nb * 256 + t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // sc1: past this CU's L1
        else got = slots[(size_t)nb * 256 + t];
        if (got != base + (uint32_t)p * 1000003u + nb * 257u + t) atomicAdd(bad, 1u);
        // (the neighbour must not overwrite its slot before everybody has read: a second barrier, counted on the same counter)
        barrier(counter + 64, (uint32_t)(p + 1) * W, mode);
    }
}
__global__ __launch_bounds__(256) void k_one_phase(uint32_t *slots, uint32_t *bad, int p, uint32_t base, int write) {
    const uint32_t W = gridDim.x, b = blockIdx.x, t = threadIdx.x;
    if (write) slots[(size_t)b * 256 + t] = base + (uint32_t)p * 1000003u + b * 257u + t;
    else {
        const uint32_t nb = (b + 1) % W;
        if (slots[(size_t)nb * 256 + t] != base + (uint32_t)p * 1000003u + nb * 257u + t) atomicAdd(bad, 1u);
    }
}

static void histogram(const char *name, hipStream_t s, int W, uint32_t *d_x) {
    std::vector<uint32_t> h(W);
    hipLaunchKernelGGL(k_where, dim3(W), dim3(64), 0, s, d_x);
    hipStreamSynchronize(s);
    hipMemcpy(h.data(), d_x, W * 4, hipMemcpyDeviceToHost);
    int cnt[16] = {0};
    for (int i = 0; i < W; i++) cnt[h[i] & 15]++;
    printf("%-44s %3d workgroups on XCC:", name, W);
    for (int x = 0; x < 8; x++) printf(" %d:%d", x, cnt[x]);
    printf("\n");
}

int main() {
    uint32_t *slots, *counter, *bad, *d_x;
    hipMalloc(&slots, 256 * 256 * 4); hipMalloc(&counter, 1024); hipMalloc(&bad, 4); hipMalloc(&d_x, 4096);
    hipMemset(bad, 0, 4);
    hipStream_t plain, first32, every8;
    hipStreamCreate(&plain);
    // 256 CUs = 8 mask words.  (a) the first 32 bits; (b) every 8th bit -- which of them is ONE XCD is what (1) answers
    uint32_t mA[8] = {0xFFFFFFFFu, 0, 0, 0, 0, 0, 0, 0}, mB[8];
    for (int w = 0; w < 8; w++) mB[w] = 0x01010101u;
    if (hipExtStreamCreateWithCUMask(&first32, 8, mA) != hipSuccess || hipExtStreamCreateWithCUMask(&every8, 8, mB) != hipSuccess) {
        printf("hipExtStreamCreateWithCUMask failed\n");
        return 1;
    }
    printf("== (1) placement\n");
    histogram("unmasked stream", plain, 256, d_x);
    histogram("unmasked stream", plain, 32, d_x);
    histogram("CU mask: first 32 bits", first32, 32, d_x);
    histogram("CU mask: every 8th bit", every8, 32, d_x);
    histogram("CU mask: first 32 bits (64 workgroups)", first32, 64, d_x);
    histogram("CU mask: every 8th bit (64 workgroups)", every8, 64, d_x);

    printf("== (2) P phases, two barriers each, 1 KB handed to the neighbour per phase (us per launch, us per barrier; stale reads)\n");
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    struct Cfg { const char *name; hipStream_t s; int W; int mode; };
    Cfg cfgs[] = {{"xcd-local barrier, mask first 32", first32, 32, 0}, {"xcd-local barrier, mask every 8th", every8, 32, 0},
                  {"agent-scope barrier, mask first 32", first32, 32, 1}, {"agent-scope barrier, mask every 8th", every8, 32, 1},
                  {"agent-scope barrier, unmasked 32 WGs", plain, 32, 1}, {"agent-scope barrier, unmasked 256 WGs", plain, 256, 1},
                  {"NO-fence barrier, unmasked 32 WGs (stale?)", plain, 32, 0}};
    uint32_t base = 1;
    for (auto &c : cfgs) {
        for (int P : {0, 1, 3, 9}) {
            float tot = 0; uint32_t hb = 0;
            const int reps = 30;
            for (int r = -3; r < reps; r++) {
                hipMemsetAsync(counter, 0, 1024, c.s);
                hipMemsetAsync(bad, 0, 4, c.s);
                hipStreamSynchronize(c.s);
                hipExtLaunchKernelGGL(k_phases, dim3(c.W), dim3(256), 0, c.s, a, b, 0, slots, counter, bad, P, c.mode, base);
                base += 7919;
                hipEventSynchronize(b);
                float ms; hipEventElapsedTime(&ms, a, b);
                uint32_t x; hipMemcpy(&x, bad, 4, hipMemcpyDeviceToHost); hb += x;
                if (r >= 0) tot += ms;
            }
            printf("%-44s P %d: %7.2f us per launch%s  stale reads %u\n", c.name, P, tot / reps * 1000, "", hb);
        }
    }
    // launches: 3 phases = 3 x (write launch + read launch) on the plain stream, timed end to end by events
    for (int W : {32, 256}) {
        float tot = 0;
        const int reps = 30;
        for (int r = -3; r < reps; r++) {
            hipMemsetAsync(bad, 0, 4, plain);
            hipStreamSynchronize(plain);
            hipEventRecord(a, plain);
            for (int p = 0; p < 3; p++) {
                hipLaunchKernelGGL(k_one_phase, dim3(W), dim3(256), 0, plain, slots, bad, p, base, 1);
                hipLaunchKernelGGL(k_one_phase, dim3(W), dim3(256), 0, plain, slots, bad, p, base, 0);
            }
            hipEventRecord(b, plain);
            hipEventSynchronize(b);
            base += 7919;
            float ms; hipEventElapsedTime(&ms, a, b);
            if (r >= 0) tot += ms;
        }
        printf("launches: 3 phases as 6 dependent launches of %3d WGs: %7.2f us in all, %5.2f us per boundary\n", W, tot / reps * 1000, tot / reps * 1000 / 6);
    }
    return 0;
}
