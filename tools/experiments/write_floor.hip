// Write-only floor of the box (round 2): how fast can HBM take stores at all, and does the parent-LOD shape matter?
//   fill_lin   : grid-stride 16-byte stores over a linear buffer
//   fill_tiles : 1024 workgroups, each fills "its" 512 KB tile row by row (1 KB rows, 4 waves = 4 rows per step), lockstep
//   fill_par   : the same workgroups write only a quarter-size parent tile (128 rows of 1 KB), one row per wave and step,
//                or (shape 1) every wave writes a 256-byte quarter of each row — the shape fused_main uses
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/write_floor.out tools/experiments/write_floor.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void fill_lin(u32x4* dst, uint64_t n) {
    const u32x4 v = {1, 2, 3, threadIdx.x};
    for (uint64_t i = uint64_t(blockIdx.x) * 256 + threadIdx.x; i < n; i += uint64_t(gridDim.x) * 256) dst[i] = v;
}
__global__ __launch_bounds__(256) void fill_tiles(uint8_t* tiles, uint32_t rows, uint64_t stride) {
    const uint32_t q = gridDim.x / 8, work = (blockIdx.x % 8) * q + blockIdx.x / 8, ty = work / 32, tx = work % 32;
    uint8_t* d = tiles + uint64_t(tx * 32 + ty) * stride + (threadIdx.x & 63u) * 16;
    const u32x4 v = {1, 2, 3, threadIdx.x};
    for (uint32_t r = threadIdx.x >> 6; r < rows; r += 4) *(u32x4*)(d + uint64_t(r) * 1024) = v;
}
__global__ __launch_bounds__(256) void fill_par(uint8_t* parents, int shape) {
    const uint32_t q = gridDim.x / 8, work = (blockIdx.x % 8) * q + blockIdx.x / 8, ty = work / 32, tx = work % 32;
    // parent tile (tx/2, ty/2), quadrant (tx&1, ty&1): 254-ish rows x 508 bytes -> modelled as 128 rows x 512 bytes
    uint8_t* d = parents + uint64_t((tx / 2) * 16 + ty / 2) * 524288 + uint64_t(ty & 1u) * 262144 + (tx & 1u) * 512;
    for (uint32_t r = 0; r < 256; r++) {
        if (shape == 0) *(uint32_t*)(d + uint64_t(r) * 1024 + (threadIdx.x & 127u) * 4) = r;            // 2 waves x 256 B (dword per lane), waves 2, 3 duplicate
        else if ((threadIdx.x & 1u) == 0) *(uint32_t*)(d + uint64_t(r) * 1024 + (threadIdx.x >> 1) * 4) = r;  // fused_main: even lanes, 128 B per wave
    }
}
template <typename F>
static float timeit(F f) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 50; i++) f();
    hipEventRecord(e0);
    for (int i = 0; i < 100; i++) f();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms * 10;
}
int main() {
    uint8_t* buf; hipMalloc(&buf, 1400ull << 19);
    for (int i = 0; i < 300; i++) fill_lin<<<1024, 256>>>((u32x4*)buf, (1024ull << 19) / 16);
    for (int blocks : {1024, 4096, 16384}) {
        const float us = timeit([&] { fill_lin<<<blocks, 256>>>((u32x4*)buf, (1024ull << 19) / 16); });
        printf("linear fill 537 MB, %d workgroups: %.1f us  %.2f TB/s\n", blocks, us, 536.870912 / us);
    }
    { const float us = timeit([&] { fill_lin<<<1024, 256>>>((u32x4*)buf, (256ull << 19) / 16); }); printf("linear fill 134 MB: %.1f us  %.2f TB/s\n", us, 134.217728 / us); }
    { const float us = timeit([&] { fill_tiles<<<1024, 256>>>(buf, 512, 524288); }); printf("tile fill 1024 x 512 KB lockstep: %.1f us  %.2f TB/s\n", us, 536.870912 / us); }
    { const float us = timeit([&] { fill_tiles<<<1024, 256>>>(buf, 128, 131072); }); printf("quarter tiles (1024 x 128 KB, whole 1 KB rows): %.1f us  %.2f TB/s\n", us, 134.217728 / us); }
    for (int shape : {0, 1}) { const float us = timeit([&] { fill_par<<<1024, 256>>>(buf, shape); }); printf("parent quadrants (256 rows x 512 B per workgroup), shape %d: %.1f us  %.2f TB/s\n", shape, us, 134.217728 / us); }
    return 0;
}
