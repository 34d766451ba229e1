// How many workgroups of a given shape does a CU of gfx950 hold at once?  (Round 4: a 320-thread workgroup — four shading waves +
// one loader wave — at 96 VGPRs was expected to fit 4 x per CU (20 waves = 5 per SIMD) and did not.)
// Every workgroup spins for a fixed time; 1024 workgroups on 256 CUs take (1024 / 256 / resident) x that time.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/residency_probe.out tools/experiments/residency_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int THREADS, int WAVES, int VGPR>
__global__ __launch_bounds__(THREADS) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES))) void spin(uint32_t* out, uint64_t ticks) {
    extern __shared__ uint8_t smem[];
    if (VGPR > 64) asm volatile("v_mov_b32 v65, 0" ::: "v65");
    if (VGPR > 72) asm volatile("v_mov_b32 v73, 0" ::: "v73");
    if (VGPR > 80) asm volatile("v_mov_b32 v81, 0" ::: "v81");
    if (VGPR > 88) asm volatile("v_mov_b32 v89, 0" ::: "v89");
    if (VGPR > 96) asm volatile("v_mov_b32 v97, 0" ::: "v97");
    const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    if (threadIdx.x == 0) { smem[0] = 1; out[blockIdx.x] = smem[0]; }
}
template <int THREADS, int WAVES, int VGPR> void run(uint32_t* out, size_t lds) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const uint64_t ticks = 5000;  // 50 us at 100 MHz
    hipFuncSetAttribute((const void*)spin<THREADS, WAVES, VGPR>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    spin<THREADS, WAVES, VGPR><<<1024, THREADS, lds>>>(out, ticks);
    hipEventRecord(e0);
    spin<THREADS, WAVES, VGPR><<<1024, THREADS, lds>>>(out, ticks);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    int n = 0;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, spin<THREADS, WAVES, VGPR>, THREADS, lds);
    printf("threads %3d  waves/SIMD attr %d  VGPR >= %3d  LDS %6zu : %.0f us -> %.2f generations of 50 us, i.e. %.1f workgroups per CU resident (occupancy API says %d)\n", THREADS, WAVES, VGPR, lds, ms * 1e3,
           ms * 1e3 / 50.0, 4.0 / (ms * 1e3 / 50.0), n);
}
int main() {
    uint32_t* out;
    hipMalloc(&out, 4096 * 4);
    for (size_t lds : {size_t(1024), size_t(34416), size_t(40016)}) {
        run<256, 4, 96>(out, lds);
        run<256, 5, 96>(out, lds);
        run<320, 5, 96>(out, lds);
        run<320, 5, 88>(out, lds);
        run<320, 6, 80>(out, lds);
        run<320, 7, 72>(out, lds);
        run<320, 8, 64>(out, lds);
        run<384, 6, 80>(out, lds);
        run<512, 8, 64>(out, lds);
        run<512, 4, 96>(out, lds);
    }
    return 0;
}
