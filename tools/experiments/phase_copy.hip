// Does HBM move a read + write mix faster when the whole chip alternates between a READ window and a WRITE window (DRAM bus
// turnarounds amortised) than when reads and writes of different workgroups interleave freely?  torch on the same box: 512 MiB
// copy 4.7 TB/s, 1 GiB copy 5.15 TB/s, pure fill 6.9 TB/s, pure read 5.5 TB/s.  The s_memrealtime clock (100 MHz) is one clock
// for the whole device, so workgroups can phase-lock without talking to each other (bounded waits only).
//   mode 0: free running (load a piece, store it)
//   mode 1: phase locked: loads only while (t mod P) < Pr, stores only while (t mod P) >= Pr
//   mode 2: the same windows, but every second workgroup shifted by Pr (its reads meet the others' writes): the control
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/phase_copy.out tools/experiments/phase_copy.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
template <int PIECES>
__global__ __launch_bounds__(256) void k(const uint8_t* src, uint8_t* dst, uint32_t bytes_per_wg, int mode, uint32_t P, uint32_t Pr) {
    const uint32_t q = gridDim.x / 8, xcd = blockIdx.x % 8, i = blockIdx.x / 8, work = xcd * q + i;
    const uint8_t* s = src + uint64_t(work) * bytes_per_wg + threadIdx.x * 16;
    uint8_t* d = dst + uint64_t(work) * bytes_per_wg + threadIdx.x * 16;
    const uint32_t shift = (mode == 2 && (work & 1u)) ? Pr : 0u;
    auto phase = [&]() -> uint32_t { return uint32_t((__builtin_amdgcn_s_memrealtime() + shift) % P); };
    for (uint32_t off = 0; off < bytes_per_wg; off += PIECES * 4096u) {
        if (mode) while (phase() >= Pr) __builtin_amdgcn_s_sleep(2);
        u32x4 v[PIECES];
#pragma unroll
        for (int j = 0; j < PIECES; j++) v[j] = *(const u32x4*)(s + off + j * 4096u);
        if (mode) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            while (phase() < Pr) __builtin_amdgcn_s_sleep(2);
        }
#pragma unroll
        for (int j = 0; j < PIECES; j++) *(u32x4*)(d + off + j * 4096u) = v[j];
    }
}
template <typename F>
static float timeit(F f) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 20; i++) f();
    hipEventRecord(e0);
    for (int i = 0; i < 50; i++) f();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms * 20;  // us per launch
}
int main() {
    const uint32_t wgs = 1024, per = 524288;
    uint8_t *src, *dst;
    hipMalloc(&src, uint64_t(wgs) * per); hipMalloc(&dst, uint64_t(wgs) * per);
    hipMemset(src, 1, uint64_t(wgs) * per);
    const double gb = 2.0 * wgs * per / 1e9;
    for (int i = 0; i < 200; i++) k<4><<<wgs, 256>>>(src, dst, per, 0, 400, 200);
    auto report = [&](const char* name, float us) { printf("%-60s %7.1f us  %.2f TB/s\n", name, us, gb / us * 1e3); };
    report("free running, 16 KB per workgroup and step", timeit([&] { k<4><<<wgs, 256>>>(src, dst, per, 0, 400, 200); }));
    report("free running, 32 KB per workgroup and step", timeit([&] { k<8><<<wgs, 256>>>(src, dst, per, 0, 400, 200); }));
    for (uint32_t P : {300u, 400u, 500u, 600u, 800u, 1200u}) {  // ticks of 10 ns
        for (uint32_t pr_pct : {40u, 50u}) {
            const uint32_t Pr = P * pr_pct / 100;
            char name[128];
            snprintf(name, sizeof name, "P = %.1f us, read window %.1f us, 16 KB: locked", P / 100.0, Pr / 100.0);
            report(name, timeit([&] { k<4><<<wgs, 256>>>(src, dst, per, 1, P, Pr); }));
            snprintf(name, sizeof name, "P = %.1f us, read window %.1f us, 16 KB: every 2nd shifted", P / 100.0, Pr / 100.0);
            report(name, timeit([&] { k<4><<<wgs, 256>>>(src, dst, per, 2, P, Pr); }));
            snprintf(name, sizeof name, "P = %.1f us, read window %.1f us, 32 KB: locked", P / 100.0, Pr / 100.0);
            report(name, timeit([&] { k<8><<<wgs, 256>>>(src, dst, per, 1, P, Pr); }));
            snprintf(name, sizeof name, "P = %.1f us, read window %.1f us, 32 KB: every 2nd shifted", P / 100.0, Pr / 100.0);
            report(name, timeit([&] { k<8><<<wgs, 256>>>(src, dst, per, 2, P, Pr); }));
        }
    }
    return 0;
}
