// Round 3: settle the copy floor.  /opt/skills/guides/MI355X_MICROARCH.md quotes 6.29 TB/s for a float4 copy; rounds 1-2
// measured 5.1-5.5 TB/s on their leases with one kernel shape.  This sweep covers the shape space the guide's number could
// come from, on THIS lease, with the clocks recorded next to it (tools/experiments/copy_floor3.sh dumps rocm-smi before and after):
//   bytes   : 0.25 .. 8 GiB read (+ the same written)          -- 128 MiB and below sit in the Infinity Cache
//   U       : 1 / 2 / 4 / 8 16-byte loads in flight per lane
//   wg/CU   : 1 .. 16 resident 256-thread workgroups per CU (grid = 256 * k, grid-stride), and one-chunk-per-workgroup grids
//   nt      : loads and / or stores non-temporal
//   layout  : interleaved (consecutive workgroups touch consecutive 4 KiB pieces) or blocked (a contiguous range each)
//   mix     : copy (1 : 1), read only, write only, and the 16k job's mix (3 reads : 4 writes)
// Output: one line per configuration, TB/s counting bytes read + bytes written.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/copy_floor3.out tools/experiments/copy_floor3.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// mode: 0 copy, 1 read only (xor-reduced, stored once per lane at the end), 2 write only, 3 job mix (every third vector is
// also written to dst2: 3 reads, 4 writes)
template <int U, bool NTL, bool NTS, int MODE, bool BLOCKED>
__global__ __launch_bounds__(256) void copyk(const u32x4* __restrict__ src, u32x4* __restrict__ dst, u32x4* __restrict__ dst2, uint64_t nvec) {
    const uint64_t step = 256ull * U;                    // vectors per workgroup iteration
    const uint64_t iters = nvec / step;                  // total iterations over all workgroups
    uint64_t it0, it1, stride;
    if (BLOCKED) {
        const uint64_t per = (iters + gridDim.x - 1) / gridDim.x;
        it0 = per * blockIdx.x; it1 = std::min(it0 + per, iters); stride = 1;
    } else {
        it0 = blockIdx.x; it1 = iters; stride = gridDim.x;
    }
    u32x4 acc = {0, 0, 0, 0};
    for (uint64_t it = it0; it < it1; it += stride) {
        const uint64_t base = it * step + threadIdx.x;
        u32x4 v[U];
        if (MODE != 2) {
#pragma unroll
            for (int j = 0; j < U; j++) v[j] = NTL ? __builtin_nontemporal_load(src + base + 256ull * j) : src[base + 256ull * j];
        } else {
#pragma unroll
            for (int j = 0; j < U; j++) v[j] = u32x4{(uint32_t)base, (uint32_t)j, 0u, 1u};
        }
        if (MODE == 1) {
#pragma unroll
            for (int j = 0; j < U; j++) acc ^= v[j];
        } else {
#pragma unroll
            for (int j = 0; j < U; j++) {
                if (NTS) __builtin_nontemporal_store(v[j], dst + base + 256ull * j); else dst[base + 256ull * j] = v[j];
            }
            if (MODE == 3) {
                // a quarter-rate second write stream: the vectors of every third iteration-slot go to dst2 as well
#pragma unroll
                for (int j = 0; j < U; j++)
                    if (((it * U + j) % 3) == 0) {
                        const uint64_t o = (it * U + j) / 3 * 256ull + threadIdx.x;
                        if (NTS) __builtin_nontemporal_store(v[j], dst2 + o); else dst2[o] = v[j];
                    }
            }
        }
    }
    if (MODE == 1 && (acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) dst[threadIdx.x] = acc;
}

static hipEvent_t e0, e1;
template <typename F>
static double time_us(F f, int reps) {
    f(); f();
    hipEventRecord(e0);
    for (int i = 0; i < reps; i++) f();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms * 1000.0 / reps;
}

template <int U, bool NTL, bool NTS, int MODE, bool BLOCKED>
static void run(const char* tag, const u32x4* s, u32x4* d, u32x4* d2, uint64_t bytes, int grid) {
    const uint64_t nvec = bytes / 16;
    const int reps = bytes >= (2ull << 30) ? 8 : 30;
    const double us = time_us([&] { copyk<U, NTL, NTS, MODE, BLOCKED><<<grid, 256>>>(s, d, d2, nvec); }, reps);
    const double moved = MODE == 0 ? 2.0 * bytes : MODE == 3 ? bytes * (1.0 + 4.0 / 3.0) : 1.0 * bytes;
    printf("%-6s bytes %6.2f GiB U %d grid %6d (%5.1f wg/CU) ntl %d nts %d %s : %9.1f us  %6.3f TB/s\n", tag, bytes / 1073741824.0, U, grid,
           grid / 256.0, (int)NTL, (int)NTS, BLOCKED ? "blocked    " : "interleaved", us, moved / us * 1e-6);
    fflush(stdout);
}

template <int U, int MODE, bool BLOCKED>
static void nt_variants(const char* tag, const u32x4* s, u32x4* d, u32x4* d2, uint64_t bytes, int grid, bool all_nt) {
    run<U, false, false, MODE, BLOCKED>(tag, s, d, d2, bytes, grid);
    if (!all_nt) return;
    run<U, true, false, MODE, BLOCKED>(tag, s, d, d2, bytes, grid);
    run<U, false, true, MODE, BLOCKED>(tag, s, d, d2, bytes, grid);
    run<U, true, true, MODE, BLOCKED>(tag, s, d, d2, bytes, grid);
}

template <int MODE>
static void sweep(const char* tag, const u32x4* s, u32x4* d, u32x4* d2, uint64_t bytes, bool full) {
    const int grids_full[] = {256, 512, 1024, 2048, 4096};
    const int grids_short[] = {1024, 2048};
    const int* grids = full ? grids_full : grids_short;
    const int ng = full ? 5 : 2;
    for (int gi = 0; gi < ng; gi++) {
        const int g = grids[gi];
        nt_variants<1, MODE, false>(tag, s, d, d2, bytes, g, full);
        nt_variants<2, MODE, false>(tag, s, d, d2, bytes, g, false);
        nt_variants<4, MODE, false>(tag, s, d, d2, bytes, g, full);
        nt_variants<8, MODE, false>(tag, s, d, d2, bytes, g, false);
        nt_variants<4, MODE, true>(tag, s, d, d2, bytes, g, false);
    }
    // one iteration per workgroup (the classic "one thread = one float4 x U" launch)
    nt_variants<1, MODE, false>(tag, s, d, d2, bytes, (int)std::min<uint64_t>(bytes / 16 / 256, 1u << 22), false);
    nt_variants<4, MODE, false>(tag, s, d, d2, bytes, (int)std::min<uint64_t>(bytes / 16 / 1024, 1u << 22), false);
}

int main(int argc, char** argv) {
    const bool quick = argc > 1 && !strcmp(argv[1], "quick");
    hipEventCreate(&e0); hipEventCreate(&e1);
    const uint64_t max_bytes = 8ull << 30;
    u32x4 *s, *d, *d2;
    if (hipMalloc(&s, max_bytes) != hipSuccess || hipMalloc(&d, max_bytes) != hipSuccess || hipMalloc(&d2, max_bytes / 2) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(s, 1, max_bytes); hipMemset(d, 0, max_bytes); hipMemset(d2, 0, max_bytes / 2);
    // spin-up: clocks ramp over the first tens of milliseconds
    for (int i = 0; i < 50; i++) copyk<4, false, false, 0, false><<<2048, 256>>>(s, d, d2, (1ull << 30) / 16);
    hipDeviceSynchronize();
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    printf("device %s, %d CUs, clockRate %d kHz, memoryClockRate %d kHz, bus %d bits\n", p.name, p.multiProcessorCount, p.clockRate, p.memoryClockRate, p.memoryBusWidth);
    // hipMemcpyAsync D2D as the library's own answer
    for (uint64_t b : {512ull << 20, 1ull << 30, 4ull << 30}) {
        const double us = time_us([&] { hipMemcpyAsync(d, s, b, hipMemcpyDeviceToDevice, 0); }, 10);
        printf("hipMemcpyAsync D2D %6.2f GiB: %9.1f us  %6.3f TB/s\n", b / 1073741824.0, us, 2.0 * b / us * 1e-6);
    }
    const uint64_t sizes[] = {128ull << 20, 256ull << 20, 512ull << 20, 537ull * 1000 * 1000 / 4096 * 4096, 1ull << 30, 2ull << 30, 4ull << 30, 8ull << 30};
    for (uint64_t b : sizes) {
        const bool full = !quick && (b == (512ull << 20) || b == (1ull << 30) || b == (4ull << 30));
        sweep<0>("copy", s, d, d2, b, full);
    }
    for (uint64_t b : {512ull << 20, 1ull << 30, 4ull << 30}) {
        sweep<1>("read", s, d, d2, b, false);
        sweep<2>("write", s, d, d2, b, false);
        sweep<3>("mix34", s, d, d2, b, b == (512ull << 20));
    }
    return 0;
}
