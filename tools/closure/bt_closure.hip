// Closure record of fused_main: the two yardsticks the kernel is held against, measured IN THE PROCESS that times the
// headline, on the same stream, minutes apart at most — so that kernel, skeleton and copy floor of one lease stand next to each other:
//   (i)  a LINEAR copy of the 16k job's byte mix: 536 870 912 bytes read, 704 643 072 written (1024 finest + 320 parent tiles of 512 KiB:
//        every 16-byte vector goes to the first destination, 5 of every 16 also to a second one), grid-stride, 4 x 16 bytes in flight per
//        lane, plain and non-temporal;
//   (ii) the MEMORY SKELETON of fused_main ("V1" of git history, tools/experiments/dma_skeleton.hip, rounds 3-5): the same bytes through the same addresses in
//        the same workgroup -> tile order — LDS-DMA of the 1056-byte source rows into a 32-row ring two chunks ahead, one dword per lane
//        for the finest rows (4-byte shifted like the b = 2 apron), a quarter-size parent row per two tile rows — with 24 packed FMAs per
//        output row standing in for the arithmetic (and with none).
// Test / measurement infrastructure: never loaded by the package.  Built by `make -C tools/closure` (called from __graft_entry__.build()).
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef const uint8_t __attribute__((address_space(1))) * gbytes;
typedef uint8_t __attribute__((address_space(3))) * lbytes;

namespace {

constexpr uint64_t kSourceBytes = 16384ull * 32768ull;           // 16384^2 u16
constexpr uint64_t kFinestBytes = 1024ull * 524288ull;           // 1024 tiles of 512^2 u16
constexpr uint64_t kParentBytes = 320ull * 524288ull;            // 256 + 64 tiles
constexpr uint32_t kPitch = 32768, kRing = 32, kRowMain = 1024, kRowTail = 32;
constexpr uint32_t kTailBase = kRing * kRowMain;

// ---- (i) linear copy of the byte mix
template <bool NT>
__global__ __launch_bounds__(256) void mix_copy(const u32x4* __restrict__ src, u32x4* __restrict__ dst, u32x4* __restrict__ dst2, uint64_t nvec) {
    constexpr int U = 4;
    const uint64_t step = 256ull * U, iters = nvec / step;
    for (uint64_t it = blockIdx.x; it < iters; it += gridDim.x) {
        const uint64_t base = it * step + threadIdx.x;
        u32x4 v[U];
#pragma unroll
        for (int j = 0; j < U; j++) v[j] = NT ? __builtin_nontemporal_load(src + base + 256ull * j) : src[base + 256ull * j];
#pragma unroll
        for (int j = 0; j < U; j++) {
            if (NT) __builtin_nontemporal_store(v[j], dst + base + 256ull * j);
            else dst[base + 256ull * j] = v[j];
        }
        // 5 of every 16 row-blocks of 256 vectors (4 KiB) also go to the second stream: 5 / 16 of 512 MiB = 160 MiB
#pragma unroll
        for (int j = 0; j < U; j++) {
            const uint64_t blk = it * U + j;
            if ((blk & 15u) < 5u) {
                const uint64_t o = ((blk >> 4) * 5u + (blk & 15u)) * 256ull + threadIdx.x;
                if (NT) __builtin_nontemporal_store(v[j], dst2 + o);
                else dst2[o] = v[j];
            }
        }
    }
}

// ---- (ii) the skeleton
__device__ __forceinline__ void tile_of(uint32_t& tx, uint32_t& ty) {
    const uint32_t q = gridDim.x / 8, work = (blockIdx.x % 8) * q + blockIdx.x / 8;  // XCD-contiguous tile rows, like fused_main
    ty = work / 32;
    tx = work % 32;
}
template <int ARITH>
__device__ __forceinline__ uint32_t shade(uint32_t a0, uint32_t a1, uint32_t b0, uint32_t b1, f2& carry, float w) {
    const f2 x = {float(a0), float(b0)}, y = {float(a1), float(b1)};
    f2 h = x * 0.75f + y * 0.25f;
#pragma unroll
    for (int i = 0; i < ARITH; i++) h = __builtin_elementwise_fma(h, f2{w, w}, carry);
    const f2 v = carry * (1.0f - w) + h * w;
    carry = h;
    return (uint32_t(v.x) & 0xFFFFu) | (uint32_t(v.y) << 16);
}
__device__ __forceinline__ void dma_row(gbytes tile_base, lbytes ring, uint32_t y, uint32_t lane) {
    __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(tile_base + uint64_t(y) * kPitch + lane * 16),
                                     (void __attribute__((address_space(3)))*)(ring + (y & (kRing - 1)) * kRowMain), 16, 0, 0);
}
__device__ __forceinline__ void dma_tails(gbytes tile_base, lbytes ring, uint32_t y0, uint32_t lane) {
    if (lane < 16)
        __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(tile_base + uint64_t(y0 + (lane >> 1)) * kPitch + 1024 + (lane & 1u) * 16),
                                         (void __attribute__((address_space(3)))*)(ring + kTailBase + (y0 & (kRing - 1)) * kRowTail), 16, 0, 0);
}
template <int ARITH>
__global__ __launch_bounds__(256) void skeleton(const uint8_t* src, uint8_t* tiles, uint8_t* parents) {
    __shared__ __attribute__((aligned(16))) uint8_t lds[kRing * (kRowMain + kRowTail)];
    uint32_t tx, ty;
    tile_of(tx, ty);
    const uint32_t tid = threadIdx.x, wave = tid >> 6, lane = tid & 63u;
    const gbytes base = (gbytes)src + uint64_t(ty) * 512 * kPitch + uint64_t(tx) * 1024;
    const lbytes ring = (lbytes)lds;
    for (uint32_t y = wave; y < 18; y += 4) dma_row(base, ring, y, lane);
    if (wave == 0) {
        dma_tails(base, ring, 0, lane);
        dma_tails(base, ring, 8, lane);
        dma_tails(base, ring, 16, lane);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    uint32_t* d5 = reinterpret_cast<uint32_t*>(tiles + uint64_t(tx * 32 + ty) * 524288 + 4) + tid;
    uint32_t* d4 = reinterpret_cast<uint32_t*>(parents + uint64_t((tx / 2) * 16 + ty / 2) * 524288 + uint64_t(ty & 1u) * 262144 + (tx & 1u) * 512 + 4) + (tid >> 1);
    const uint32_t c0 = 2 * tid, c1 = 2 * tid + 2;
    auto col = [&](uint32_t c, uint32_t& off, uint32_t& stride) {
        if (c < 512) { off = c * 2; stride = kRowMain; } else { off = kTailBase + (c - 512) * 2; stride = kRowTail; }
    };
    uint32_t o00, s00, o01, s01, o10, s10, o11, s11;
    col(c0, o00, s00); col(c0 + 1, o01, s01); col(c1, o10, s10); col(c1 + 1, o11, s11);
    for (uint32_t k = 0; k < 64; k++) {
        const uint32_t y0 = 8 * k + 18 + 2 * wave;  // every wave streams two rows of the group two chunks ahead (wave 0 also the tails)
        dma_row(base, ring, y0, lane);
        dma_row(base, ring, y0 + 1, lane);
        if (wave == 0) dma_tails(base, ring, 8 * k + 24, lane);
        const uint32_t slot0 = (8 * k) & (kRing - 1);
        auto tex = [&](uint32_t r, uint32_t off, uint32_t stride) -> uint32_t {
            return *reinterpret_cast<const uint16_t*>(lds + off + ((slot0 + r) & (kRing - 1)) * stride);
        };
        f2 carry = {float(tex(0, o00, s00)), float(tex(0, o10, s10))};
        uint32_t out[8];
#pragma unroll
        for (uint32_t r = 0; r < 8; r++) out[r] = shade<ARITH>(tex(r + 1, o00, s00), tex(r + 1, o01, s01), tex(r + 1, o10, s10), tex(r + 1, o11, s11), carry, 0.125f * float(r));
#pragma unroll
        for (uint32_t r = 0; r < 8; r++) d5[(k * 8 + r) * 256] = out[r];
        if ((tid & 1u) == 0) {
#pragma unroll
            for (uint32_t r = 0; r < 4; r++) d4[(k * 4 + r) * 256] = out[2 * r] + out[2 * r + 1];
        }
        // the rows of chunk k + 1 were issued one iteration ago, before the stores of chunk k - 1: what was issued since may stay in flight
        asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
}

}  // namespace

// out_ms[0] linear copy of the byte mix, out_ms[1] the same non-temporal, out_ms[2] skeleton with 24 packed FMAs per output row, out_ms[3]
// skeleton without arithmetic; average of `reps` launches each (after 3 untimed ones) between HIP events on `stream`.  The four are
// timed in `rounds` interleaved rounds (copy, copy-nt, skeleton, skeleton-0, copy, ...) and the MINIMUM round average is reported per
// item next to the mean in out_ms[4..7].  bytes[0] / bytes[1]: bytes read / written per launch.  Returns 0 or the hipError_t.
extern "C" int bt_closure_run(void* stream_ptr, int reps, int rounds, float out_ms[8], uint64_t bytes[2]) {
    hipStream_t stream = (hipStream_t)stream_ptr;
    uint8_t *src = nullptr, *tiles = nullptr, *parents = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipError_t e = hipMalloc((void**)&src, kSourceBytes + (1 << 20));  // the last tiles' windows run 32 bytes and 18 rows past the raster
    if (e == hipSuccess) e = hipMalloc((void**)&tiles, kFinestBytes + 4096);
    if (e == hipSuccess) e = hipMalloc((void**)&parents, kParentBytes + 4096);
    if (e == hipSuccess) e = hipMemsetAsync(src, 3, kSourceBytes + (1 << 20), stream);
    if (e == hipSuccess) e = hipEventCreate(&e0);
    if (e == hipSuccess) e = hipEventCreate(&e1);
    auto launch = [&](int which) {
        const uint64_t nvec = kSourceBytes / 16;
        switch (which) {
            // (one workgroup per CU, 4 x 16 bytes in flight per lane: the fastest shape of the round-3 sweep, profiles/r03_copy_floor.txt)
            case 0: mix_copy<false><<<256, 256, 0, stream>>>((const u32x4*)src, (u32x4*)tiles, (u32x4*)parents, nvec); break;
            case 1: mix_copy<true><<<256, 256, 0, stream>>>((const u32x4*)src, (u32x4*)tiles, (u32x4*)parents, nvec); break;
            case 2: skeleton<24><<<1024, 256, 0, stream>>>(src, tiles, parents); break;
            default: skeleton<0><<<1024, 256, 0, stream>>>(src, tiles, parents); break;
        }
    };
    for (int w = 0; w < 8; w++) {
        out_ms[w] = w < 4 ? 1e30f : 0.0f;
    }
    for (int round = 0; round < rounds && e == hipSuccess; round++)
        for (int which = 0; which < 4 && e == hipSuccess; which++) {
            for (int i = 0; i < 3; i++) launch(which);
            e = hipEventRecord(e0, stream);
            for (int i = 0; i < reps; i++) launch(which);
            if (e == hipSuccess) e = hipEventRecord(e1, stream);
            if (e == hipSuccess) e = hipEventSynchronize(e1);
            float ms = 0.0f;
            if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
            if (e == hipSuccess) e = hipGetLastError();
            ms /= float(reps);
            if (ms < out_ms[which]) out_ms[which] = ms;
            out_ms[4 + which] += ms / float(rounds);
        }
    if (bytes) {
        bytes[0] = kSourceBytes;
        bytes[1] = kFinestBytes + kParentBytes;
    }
    if (e0) hipEventDestroy(e0);
    if (e1) hipEventDestroy(e1);
    if (src) hipFree(src);
    if (tiles) hipFree(tiles);
    if (parents) hipFree(parents);
    return int(e);
}
