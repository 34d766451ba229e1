// Follow-up to copy_floor.hip (round 2): WHY does the parent-LOD write stream cost twice its bytes?
// The 16k job's geometry again (1024 workgroups in tile-row order, 1 KB raster-row windows in, 512 KB tiles out, a
// quarter-size parent write), with the knobs that separate the hypotheses:
//   pad_t / pad_p : extra bytes between consecutive tiles / parent tiles (power-of-two strides put the 128 workgroups of
//                   an XCD on the same L2 sets and DRAM banks at the same time)
//   nt            : parent rows as non-temporal stores
//   lin3          : the same three streams, all linear (is a second write stream expensive as such?)
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/copy_floor2.out tools/copy_floor2.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k(const uint8_t* src, uint8_t* tiles, uint8_t* parents, uint64_t stride_t, uint64_t stride_p, int mode,
                                         int rows_per_step, int map = 0) {
    const uint32_t total = gridDim.x, q = total / 8, xcd = blockIdx.x % 8, i = blockIdx.x / 8;
    // workgroup -> tile: 0 XCD-contiguous tile-row order (fused_main), 1 tile-row order without the XCD remap, 2 XCD-contiguous
    // x-major (atlas) order, 3 x-major without remap, 4 an 8 x 16 block of tiles per XCD, 5 a 16 x 8 block per XCD
    uint32_t work = (map == 1 || map == 3) ? blockIdx.x : xcd * q + i, ty = work / 32, tx = work % 32;
    if (map == 2 || map == 3) { tx = work / 32; ty = work % 32; }
    if (map == 4) { ty = (xcd / 2) * 8 + i / 16; tx = (xcd % 2) * 16 + i % 16; }
    if (map == 5) { ty = (xcd / 4) * 16 + i / 8; tx = (xcd % 4) * 8 + i % 8; }
    const uint32_t lane16 = threadIdx.x & 63u, rsub = threadIdx.x >> 6;
    const uint8_t* s = src + (uint64_t(ty) * 512) * 32768 + uint64_t(tx) * 1024 + lane16 * 16;
    uint8_t* d = tiles + uint64_t(tx * 32 + ty) * stride_t + lane16 * 16;
    uint8_t* p = parents + uint64_t(tx * 32 + ty) * stride_p + lane16 * 16;
    for (uint32_t r0 = 0; r0 < 512; r0 += 4 * rows_per_step) {
        u32x4 v[4];
#pragma unroll
        for (int j = 0; j < 4; j++) if (j < rows_per_step) v[j] = *(const u32x4*)(s + uint64_t(r0 + 4 * j + rsub) * 32768);
#pragma unroll
        for (int j = 0; j < 4; j++) if (j < rows_per_step) {
            const uint32_t r = r0 + 4 * j + rsub;
            if (mode & 1) *(u32x4*)(d + uint64_t(r) * 1024) = v[j];
            if ((mode & 2) && (r & 3u) == 0) *(u32x4*)(p + uint64_t(r >> 2) * 1024) = v[j];
            if ((mode & 4) && (r & 3u) == 0) __builtin_nontemporal_store(v[j], (u32x4*)(p + uint64_t(r >> 2) * 1024));
            // parent rows in the SAME allocation as the tiles (the atlas: parents are just more layers), written by wave 0 only
            if ((mode & 8) && (r & 3u) == 0) *(u32x4*)(tiles + uint64_t(1024 + (tx * 32 + ty) / 4) * stride_t + uint64_t((tx * 32 + ty) & 3u) * 131072 + uint64_t(r >> 2) * 1024 + lane16 * 16) = v[j];
        }
    }
}
// fused_main's store shape: one dword per lane, a wave = 256 contiguous bytes of a 1 KB tile row; `shift` = 4 starts the
// row's centre at byte 4 like the tiles do (b = 2 texels of 2 bytes): every wave store then straddles a 128-byte line
__global__ __launch_bounds__(256) void kshape(const uint8_t* src, uint8_t* tiles, int shift, int do_read) {
    const uint32_t total = gridDim.x, q = total / 8, xcd = blockIdx.x % 8, i = blockIdx.x / 8;
    const uint32_t work = xcd * q + i, ty = work / 32, tx = work % 32;
    const uint32_t t = threadIdx.x;
    const uint8_t* s = src + (uint64_t(ty) * 512) * 32768 + uint64_t(tx) * 1024;
    uint8_t* d = tiles + uint64_t(tx * 32 + ty) * 524288;
    const uint32_t dword = shift ? (t + 1u) & 255u : t;  // shift: thread t owns dword t + 1 (the last thread wraps to dword 0)
    for (uint32_t r0 = 0; r0 < 512; r0 += 8) {
        u32x4 v[2];
        if (do_read) {  // 8 rows x 1 KB in, 16 bytes per lane and load like fused_main's staging
            v[0] = *(const u32x4*)(s + uint64_t(r0 + (t >> 6)) * 32768 + (t & 63u) * 16);
            v[1] = *(const u32x4*)(s + uint64_t(r0 + 4 + (t >> 6)) * 32768 + (t & 63u) * 16);
        } else {
            v[0] = v[1] = u32x4{t, t, t, t};
        }
#pragma unroll
        for (uint32_t r = 0; r < 8; r++) *(uint32_t*)(d + uint64_t(r0 + r) * 1024 + dword * 4) = v[r >> 2][r & 3];
    }
}
__global__ __launch_bounds__(256) void lin3(const uint8_t* src, uint8_t* dst, uint8_t* dst2, int third, int window = 0) {
    const uint32_t total = gridDim.x, q = total / 8, xcd = blockIdx.x % 8, i = blockIdx.x / 8, work = xcd * q + i;
    const uint32_t lane16 = threadIdx.x & 63u, rsub = threadIdx.x >> 6;
    for (uint32_t r = rsub; r < 512; r += 4) {
        const uint64_t off = uint64_t(work) * 524288 + uint64_t(r) * 1024 + lane16 * 16;
        // window = 1: the reads follow the raster windows of tile (tx, ty) = (work % 32, work / 32); the writes stay linear
        const uint64_t soff = window ? (uint64_t(work / 32) * 512 + r) * 32768 + uint64_t(work % 32) * 1024 + lane16 * 16 : off;
        const u32x4 v = *(const u32x4*)(src + soff);
        *(u32x4*)(dst + off) = v;
        if (third && (r & 3u) == 0) *(u32x4*)(dst2 + uint64_t(work) * 131072 + uint64_t(r >> 2) * 1024 + lane16 * 16) = v;
    }
}
template <typename F>
static float timeit(F f) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 100; i++) f();
    hipEventRecord(e0);
    for (int i = 0; i < 100; i++) f();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms * 10;  // us per launch
}
#include <cstdlib>
#include <cstring>
int main(int argc, char** argv) {
    uint8_t *src, *tiles, *parents;
    hipMalloc(&src, 16384ull * 16384 * 2); hipMalloc(&tiles, 1400ull * (524288 + 65536) + (4ull << 20)); hipMalloc(&parents, 1024ull * (131072 + 65536) + (4ull << 20));
    hipMemset(src, 1, 16384ull * 16384 * 2);
    if (argc > 1) {  // one configuration, a few launches (for rocprofv3 --pmc passes): lin3w | k0 | k2
        for (int i = 0; i < 10; i++) {
            if (!strcmp(argv[1], "lin3w")) lin3<<<1024, 256>>>(src, tiles, parents, 1, 1);
            if (!strcmp(argv[1], "k0")) k<<<1024, 256>>>(src, tiles, parents, 524288, 131072, 3, 2, 0);
            if (!strcmp(argv[1], "k2")) k<<<1024, 256>>>(src, tiles, parents, 524288, 131072, 3, 2, 2);
        }
        hipDeviceSynchronize();
        return 0;
    }
    for (int i = 0; i < 300; i++) lin3<<<1024, 256>>>(src, tiles, parents, 0);  // spin-up
    for (int rd : {0, 1})
        for (int shift : {0, 4})
            printf("dword-per-lane tile rows, %s, row start shifted by %d bytes: %.1f us\n", rd ? "reads + writes" : "writes only", shift,
                   timeit([&] { kshape<<<1024, 256>>>(src, tiles, shift, rd); }));
    printf("linear r + w: %.1f us;  + third linear stream of 1/4: %.1f us\n", timeit([&] { lin3<<<1024, 256>>>(src, tiles, parents, 0); }),
           timeit([&] { lin3<<<1024, 256>>>(src, tiles, parents, 1); }));
    printf("raster-window reads, linear writes: r + w %.1f us;  + third linear stream: %.1f us\n", timeit([&] { lin3<<<1024, 256>>>(src, tiles, parents, 0, 1); }),
           timeit([&] { lin3<<<1024, 256>>>(src, tiles, parents, 1, 1); }));
    for (int rows : {1, 2, 4})
        for (int mode : {1, 3, 5, 9})
            printf("rows in flight %d, mode %d: %.1f us\n", rows, mode, timeit([&] { k<<<1024, 256>>>(src, tiles, parents, 524288, 131072, mode, rows); }));
    for (uint64_t pad : {0ull, 256ull, 1024ull, 4096ull, 5120ull, 16384ull, 33792ull})
        printf("tile pad %llu: r+w %.1f us, r+w+parents(pad/4) %.1f us, r+w+parents unpadded %.1f us\n", (unsigned long long)pad,
               timeit([&] { k<<<1024, 256>>>(src, tiles, parents, 524288 + pad, 131072, 1, 1); }),
               timeit([&] { k<<<1024, 256>>>(src, tiles, parents, 524288 + pad, 131072 + pad / 4, 3, 1); }),
               timeit([&] { k<<<1024, 256>>>(src, tiles, parents, 524288 + pad, 131072, 3, 1); }));
    for (uint64_t off : {0ull, 4352ull, 69888ull, 1118464ull})  // allocations shifted against each other (bank / channel phase)
        printf("tiles + %llu, parents + 2 x that: r+w %.1f us, r+w+parents %.1f us\n", (unsigned long long)off,
               timeit([&] { k<<<1024, 256>>>(src, tiles + off, parents + 2 * off, 524288, 131072, 1, 2, 0); }),
               timeit([&] { k<<<1024, 256>>>(src, tiles + off, parents + 2 * off, 524288, 131072, 3, 2, 0); }));
    // the third stream alone, and the tile stream alone, with and without the reads
    for (int mode : {0, 2, 1, 3})
        printf("mode %d (1 = tile writes, 2 = parent writes; reads always): %.1f us\n", mode, timeit([&] { k<<<1024, 256>>>(src, tiles, parents, 524288, 131072, mode, 2, 0); }));
    for (int map : {0, 1, 2, 3, 4, 5})
        printf("workgroup map %d: r+w %.1f us, r+w+parents %.1f us\n", map, timeit([&] { k<<<1024, 256>>>(src, tiles, parents, 524288, 131072, 1, 2, map); }),
               timeit([&] { k<<<1024, 256>>>(src, tiles, parents, 524288, 131072, 3, 2, map); }));
    return 0;
}
