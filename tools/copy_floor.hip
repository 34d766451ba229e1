// How fast can the 16k job's memory traffic go at all?  A free-running copy with the job's geometry and none of its
// arithmetic, LDS or barriers: 1024 workgroups (tile-row order, like fused_main), each streams the 516 source rows
// x 1 KB of its tile's footprint out of the 16384^2 u16 raster and writes its 512 x 512 tile (1 KB rows), optionally
// a quarter-size parent write.  Build: hipcc --offload-arch=gfx950 -O3 -o tools/copy_floor.out tools/copy_floor.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k(const uint16_t* src, uint16_t* tiles, uint16_t* parents, int mode, int depth, int prot = 0) {
    // blockIdx -> XCD-contiguous work id, tile-row order
    const uint32_t total = gridDim.x, q = total / 8, xcd = blockIdx.x % 8, i = blockIdx.x / 8;
    const uint32_t work = xcd * q + i, ty = work / 32, tx = work % 32;
    const uint32_t lane16 = threadIdx.x & 63u, rsub = threadIdx.x >> 6;  // 64 lanes x 16 B = one 1 KB row; 4 rows per step
    const uint8_t* s = (const uint8_t*)src + (uint64_t(ty) * 512) * 32768 + uint64_t(tx) * 1024 + lane16 * 16;
    uint8_t* d = (uint8_t*)tiles + uint64_t(tx * 32 + ty) * 524288 + lane16 * 16;
    uint8_t* p = (uint8_t*)parents + uint64_t(tx * 32 + ty) * 131072 + lane16 * 8;
    u32x4 acc = {0, 0, 0, 0};
    for (uint32_t r = rsub; r < 512; r += 4 * depth) {
        u32x4 v[4];
#pragma unroll
        for (int j = 0; j < 4; j++) if (j < depth) v[j] = *(const u32x4*)(s + uint64_t(r + 4 * j) * 32768);
#pragma unroll
        for (int j = 0; j < 4; j++) if (j < depth) {
            if (mode & 1) *(u32x4*)(d + uint64_t(r + 4 * j) * 1024) = v[j];
            else acc += v[j];
            if ((mode & 2) && ((r + 4 * j) & 1u) == 0) *(uint2*)(p + uint64_t((r + 4 * j) >> 1) * 512) = uint2{v[j].x, v[j].y};
            // the same parent bytes as whole 1 KB rows (every 4th row of the tile writes one)
            if ((mode & 4) && ((r + 4 * j) & 3u) == 0) *(u32x4*)((uint8_t*)parents + uint64_t(tx * 32 + ty) * 131072 + uint64_t((((r + 4 * j) >> 2) + prot) & 127u) * 1024 + lane16 * 16) = v[j];
        }
    }
    if (!(mode & 1) && acc.x == 0x12345678u) tiles[0] = 1;
}
// the job's geometry with P workgroups per tile: contiguous row ranges (inter = 0) or 8-row chunks dealt round-robin
// (inter = 1); the P workgroups of a tile have consecutive work ids (same XCD, same time)
__global__ __launch_bounds__(256) void kparts(const uint16_t* src, uint16_t* tiles, uint32_t P, int inter) {
    const uint32_t total = gridDim.x, q = total / 8, xcd = blockIdx.x % 8, i = blockIdx.x / 8;
    const uint32_t work = xcd * q + i, tile = work / P, part = work % P, ty = tile / 32, tx = tile % 32;
    const uint32_t lane16 = threadIdx.x & 63u, rsub = threadIdx.x >> 6;
    const uint8_t* s = (const uint8_t*)src + (uint64_t(ty) * 512) * 32768 + uint64_t(tx) * 1024 + lane16 * 16;
    uint8_t* d = (uint8_t*)tiles + uint64_t(tx * 32 + ty) * 524288 + lane16 * 16;
    const uint32_t chunks = 64 / P;
    for (uint32_t c = 0; c < chunks; c++) {
        const uint32_t k = inter ? c * P + part : part * chunks + c;
        for (uint32_t r = k * 8 + rsub; r < k * 8 + 8; r += 4) *(u32x4*)(d + uint64_t(r) * 1024) = *(const u32x4*)(s + uint64_t(r) * 32768);
    }
}
// reference points: a plain linear copy, and the two half-way geometries
__global__ __launch_bounds__(256) void lin(const uint8_t* src, uint8_t* dst, uint64_t bytes, int mode) {
    // mode 0: linear -> linear.  mode 1: tile-geometry reads -> linear writes.  mode 2: linear reads -> tile-geometry writes
    const uint64_t per_block = bytes / gridDim.x;  // 512 KB
    const uint32_t total = gridDim.x, q = total / 8, xcd = blockIdx.x % 8, i = blockIdx.x / 8, work = xcd * q + i;
    const uint32_t ty = work / 32, tx = work % 32, lane16 = threadIdx.x & 63u, rsub = threadIdx.x >> 6;
    for (uint32_t r = rsub; r < 512; r += 4) {
        const uint64_t lin_off = uint64_t(work) * per_block + uint64_t(r) * 1024 + lane16 * 16;
        const uint64_t src_geo = (uint64_t(ty) * 512 + r) * 32768 + uint64_t(tx) * 1024 + lane16 * 16;
        const uint64_t dst_geo = uint64_t(tx * 32 + ty) * 524288 + uint64_t(r) * 1024 + lane16 * 16;
        const u32x4 v = *(const u32x4*)(src + (mode == 1 ? src_geo : lin_off));
        *(u32x4*)(dst + (mode == 2 ? dst_geo : lin_off)) = v;
    }
}
__global__ __launch_bounds__(256) void stride_copy(const u32x4* src, u32x4* dst, uint64_t n) {
    for (uint64_t i = uint64_t(blockIdx.x) * 256 + threadIdx.x; i < n; i += uint64_t(gridDim.x) * 256) dst[i] = src[i];
}
int main() {
    uint16_t *src, *tiles, *parents;
    hipMalloc(&src, 16384ull * 16384 * 2); hipMalloc(&tiles, 1024ull * 524288); hipMalloc(&parents, 1024ull * 131072);
    hipMemset(src, 1, 16384ull * 16384 * 2);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int depth : {1}) for (int mode : {0, 1, 3, 5}) {
        for (int i = 0; i < 200; i++) k<<<1024, 256>>>(src, tiles, parents, mode, depth);
        hipEventRecord(e0);
        for (int i = 0; i < 100; i++) k<<<1024, 256>>>(src, tiles, parents, mode, depth);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double bytes = 536870912.0 + ((mode & 1) ? 536870912.0 : 0) + ((mode & 6) ? 134217728.0 : 0);
        printf("loads in flight per lane %d, mode %d (%s): %.1f us  %.2f TB/s\n", depth, mode,
               mode == 0 ? "read only" : mode == 1 ? "read + tile write" : mode == 3 ? "read + tile + parent write (512-byte pieces)" : "read + tile + parent write (whole 1 KB rows)", ms * 10, bytes / (ms * 1e-5) / 1e12);
    }
    for (int prot : {0, 1, 7, 32, 64, 100}) {  // parent rows written out of phase with the tile rows
        for (int i = 0; i < 200; i++) k<<<1024, 256>>>(src, tiles, parents, 5, 1, prot);
        hipEventRecord(e0);
        for (int i = 0; i < 100; i++) k<<<1024, 256>>>(src, tiles, parents, 5, 1, prot);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("parent rows rotated by %d: %.1f us\n", prot, ms * 10);
    }
    for (int inter : {0, 1}) for (uint32_t P : {1u, 2u, 4u, 8u}) {
        for (int i = 0; i < 200; i++) kparts<<<1024 * P, 256>>>(src, tiles, P, inter);
        hipEventRecord(e0);
        for (int i = 0; i < 100; i++) kparts<<<1024 * P, 256>>>(src, tiles, P, inter);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("read + tile write, %u workgroups per tile, %s: %.1f us\n", P, inter ? "chunks dealt round-robin" : "contiguous row ranges", ms * 10);
    }
    for (int blocks : {1024, 2048, 4096, 16384}) {
        for (int i = 0; i < 200; i++) stride_copy<<<blocks, 256>>>((const u32x4*)src, (u32x4*)tiles, 536870912ull / 16);
        hipEventRecord(e0);
        for (int i = 0; i < 100; i++) stride_copy<<<blocks, 256>>>((const u32x4*)src, (u32x4*)tiles, 536870912ull / 16);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("grid-stride float4 copy, %d workgroups: %.1f us  %.2f TB/s\n", blocks, ms * 10, 2 * 536870912.0 / (ms * 1e-5) / 1e12);
    }
    for (int mode : {0, 1, 2}) {
        for (int i = 0; i < 200; i++) lin<<<1024, 256>>>((const uint8_t*)src, (uint8_t*)tiles, 536870912ull, mode);
        hipEventRecord(e0);
        for (int i = 0; i < 100; i++) lin<<<1024, 256>>>((const uint8_t*)src, (uint8_t*)tiles, 536870912ull, mode);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%s: %.1f us  %.2f TB/s\n", mode == 0 ? "linear read -> linear write (512 KB per workgroup)" : mode == 1 ? "raster-window read -> linear write" : "linear read -> tile write (x-major atlas order)",
               ms * 10, 2 * 536870912.0 / (ms * 1e-5) / 1e12);
    }
    return 0;
}
