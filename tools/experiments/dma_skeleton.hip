// Round 3 experiment: would fused_main gain from LDS-DMA staging (global_load_lds) and from loader / shader wave roles?
// A memory skeleton of the 16k job (same bytes, same addresses, same workgroup -> tile order as fused_main) in three
// staging structures, each with a selectable amount of arithmetic per output row:
//   V0  register staging, as fused_main does today: 4 x 16-byte loads per thread for chunk k + 1 issued before chunk k is
//       shaded, committed to LDS after it, one __syncthreads() per chunk (which drains loads AND stores: one counter)
//   V1  LDS-DMA issued by all four waves (two source rows each per chunk) into a ring of 32 source rows, two chunks ahead;
//       raw s_barrier, counted vmcnt: a wave never waits for its own stores of the current or the previous chunk
//   V2  LDS-DMA issued by a FIFTH wave that does nothing else (320-thread workgroups); the four shading waves never wait
//       on the memory counter at all
// Geometry: 16384^2 u16 source, 1024 tiles of 512 x 512 (1 KB rows), 64 chunks of 8 tile rows per tile; chunk k needs source
// rows 8k .. 8k+9 of the tile's 1056-byte window (1024 main + 32 tail, the tail in its own LDS area because a DMA
// instruction writes 64 lanes x 16 bytes contiguously); finest rows leave as one dword per lane, a quarter-size parent row
// per two tile rows as a dword from every even lane.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/experiments/dma_skeleton.out tools/experiments/dma_skeleton.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef const uint8_t __attribute__((address_space(1))) * gbytes;
typedef uint8_t __attribute__((address_space(3))) * lbytes;

constexpr uint32_t kPitch = 32768, kRing = 32, kRowMain = 1024, kRowTail = 32;
constexpr uint32_t kTailBase = kRing * kRowMain;  // tails of the ring rows follow the main parts

__device__ __forceinline__ void tile_of(uint32_t& tx, uint32_t& ty) {
    const uint32_t q = gridDim.x / 8, work = (blockIdx.x % 8) * q + blockIdx.x / 8;
    ty = work / 32;
    tx = work % 32;
}

// one output row of a column pair from two staged source rows: ARITH = number of packed fma rounds on top of the minimum
template <int ARITH> __device__ __forceinline__ uint32_t shade(uint32_t a0, uint32_t a1, uint32_t b0, uint32_t b1, f2& carry, float w) {
    const f2 x = {float(a0), float(b0)}, y = {float(a1), float(b1)};
    f2 h = x * 0.75f + y * 0.25f;
#pragma unroll
    for (int i = 0; i < ARITH; i++) h = __builtin_elementwise_fma(h, f2{w, w}, carry);
    const f2 v = carry * (1.0f - w) + h * w;
    carry = h;
    return (uint32_t(v.x) & 0xFFFFu) | (uint32_t(v.y) << 16);
}

// ------------------------------------------------------------------------------------------------ V0
template <int ARITH> __global__ __launch_bounds__(256) void v0(const uint8_t* src, uint8_t* tiles, uint8_t* parents, int stores) {
    __shared__ __attribute__((aligned(16))) uint8_t lds[2 * 10 * 1056 + 64];
    uint32_t tx, ty;
    tile_of(tx, ty);
    const uint32_t tid = threadIdx.x;
    const gbytes base = (gbytes)src + uint64_t(ty) * 512 * kPitch + uint64_t(tx) * 1024;
    const uint32_t shift = (stores & 32) ? 0u : 4u;
    uint32_t* d5 = reinterpret_cast<uint32_t*>(tiles + uint64_t(tx * 32 + ty) * 524288 + shift) + tid;
    uint32_t* d4 = reinterpret_cast<uint32_t*>(parents + uint64_t((tx / 2) * 16 + ty / 2) * 524288 + uint64_t(ty & 1u) * 262144 + (tx & 1u) * 512 + shift) + (tid >> 1);
    uint32_t soff[3], loff[3];
#pragma unroll
    for (int i = 0; i < 3; i++) {  // 10 rows x 66 pieces of 16 bytes = 660 pieces over 256 threads
        const uint32_t ch = tid + 256 * i, row = ch / 66, kk = ch % 66;
        soff[i] = row * kPitch + kk * 16;
        loff[i] = row * 1056 + kk * 16;
    }
    auto issue = [&](uint32_t k, u32x4 (&v)[3]) {
        const gbytes b = base + uint64_t(k) * 8 * kPitch;
#pragma unroll
        for (int i = 0; i < 3; i++) {
            const u32x4 __attribute__((address_space(1)))* p = (const u32x4 __attribute__((address_space(1)))*)(b + (tid + 256 * i < 660 ? soff[i] : soff[0]));
            if (stores & 4) v[i] = u32x4{tid, k, 3u, 4u};
            else v[i] = (stores & 16) ? __builtin_nontemporal_load(p) : *p;
        }
    };
    auto commit = [&](uint32_t k, const u32x4 (&v)[3]) {
#pragma unroll
        for (int i = 0; i < 3; i++)
            if (tid + 256 * i < 660) *reinterpret_cast<u32x4*>(lds + (k & 1u) * 10560 + loff[i]) = v[i];
    };
    u32x4 pre[3];
    issue(0, pre);
    commit(0, pre);
    __syncthreads();
    const uint32_t c0 = 2 * tid, c1 = 2 * tid + 2;  // this thread's source columns (texels), adjacent pairs overlap
    for (uint32_t k = 0; k < 64; k++) {
        if (k + 1 < 64) issue(k + 1, pre);
        const uint16_t* s = reinterpret_cast<const uint16_t*>(lds + (k & 1u) * 10560);
        f2 carry = {float(s[c0]), float(s[c1])};
        uint32_t out[8];
#pragma unroll
        for (uint32_t r = 0; r < 8; r++) {
            const uint16_t* row = s + (r + 1) * 528;
            out[r] = shade<ARITH>(row[c0], row[c0 + 1], row[c1], row[c1 + 1], carry, 0.125f * float(r));
        }
        if (stores & 1) {
#pragma unroll
            for (uint32_t r = 0; r < 8; r++) {
                if (stores & 8) __builtin_nontemporal_store(out[r], &d5[(k * 8 + r) * 256]);
                else d5[(k * 8 + r) * 256] = out[r];
            }
        }
        if ((stores & 2) && (tid & 1u) == 0) {
#pragma unroll
            for (uint32_t r = 0; r < 4; r++) {
                if (stores & 8) __builtin_nontemporal_store(out[2 * r] + out[2 * r + 1], &d4[(k * 4 + r) * 256]);
                else d4[(k * 4 + r) * 256] = out[2 * r] + out[2 * r + 1];
            }
        }
        if (k + 1 < 64) commit(k + 1, pre);
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------ V1 / V2
// DMA of source row y (relative to the tile's first row) into ring slot y % 32: one 1 KB instruction
template <bool NT = false>
__device__ __forceinline__ void dma_row(gbytes tile_base, lbytes ring, uint32_t y, uint32_t lane) {
    __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(tile_base + uint64_t(y) * kPitch + lane * 16),
                                     (void __attribute__((address_space(3)))*)(ring + (y & (kRing - 1)) * kRowMain), 16, 0, NT ? 2 : 0);
}
// the 32-byte tails of 8 consecutive rows y0 .. y0+7 (y0 a multiple of 8): lanes 0..15, two lanes per row
__device__ __forceinline__ void dma_tails(gbytes tile_base, lbytes ring, uint32_t y0, uint32_t lane) {
    if (lane < 16)
        __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(tile_base + uint64_t(y0 + (lane >> 1)) * kPitch + 1024 + (lane & 1u) * 16),
                                         (void __attribute__((address_space(3)))*)(ring + kTailBase + (y0 & (kRing - 1)) * kRowTail), 16, 0, 0);
}

template <int ARITH, bool LOADER_WAVE, bool NTL = false, bool NTS = false> __global__ __launch_bounds__(LOADER_WAVE ? 320 : 256) void v12(const uint8_t* src, uint8_t* tiles, uint8_t* parents, int stores) {
    __shared__ __attribute__((aligned(16))) uint8_t lds[kRing * (kRowMain + kRowTail)];
    uint32_t tx, ty;
    tile_of(tx, ty);
    const uint32_t tid = threadIdx.x, wave = tid >> 6, lane = tid & 63u;
    const gbytes base = (gbytes)src + uint64_t(ty) * 512 * kPitch + uint64_t(tx) * 1024;
    const lbytes ring = (lbytes)lds;
    // prologue: rows 0 .. 17 (chunks 0 and 1; the tails in groups of 8, rows 16..23 complete with the first in-loop group)
    if (LOADER_WAVE ? wave == 4 : true) {
        const uint32_t nw = LOADER_WAVE ? 1 : 4, w = LOADER_WAVE ? 0 : wave;
        for (uint32_t y = w; y < 18; y += nw) dma_row<NTL>(base, ring, y, lane);
        if (w == 0) {
            dma_tails(base, ring, 0, lane);
            dma_tails(base, ring, 8, lane);
            dma_tails(base, ring, 16, lane);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    if (LOADER_WAVE && wave == 4) {
        // ---- the loader: rows 8k+18 .. 8k+25 while chunk k is shaded; chunk k + 1 (rows <= 8k+17) must have landed at the barrier
        for (uint32_t k = 0; k < 64; k++) {
            const uint32_t y0 = 8 * k + 18;
            if (k + 2 < 64 + 1) {
#pragma unroll
                for (uint32_t i = 0; i < 8; i++) dma_row<NTL>(base, ring, y0 + i, lane);
                dma_tails(base, ring, 8 * k + 24, lane);
                asm volatile("s_waitcnt vmcnt(9)" ::: "memory");  // everything but this group
            }
            __builtin_amdgcn_s_barrier();
        }
        return;
    }
    uint32_t* d5 = reinterpret_cast<uint32_t*>(tiles + uint64_t(tx * 32 + ty) * 524288 + 4) + tid;
    uint32_t* d4 = reinterpret_cast<uint32_t*>(parents + uint64_t((tx / 2) * 16 + ty / 2) * 524288 + uint64_t(ty & 1u) * 262144 + (tx & 1u) * 512 + 4) + (tid >> 1);
    // this thread's source columns: texel c in the main part (c < 512) or in the tail area
    const uint32_t c0 = 2 * tid, c1 = 2 * tid + 2;
    auto col = [&](uint32_t c, uint32_t& off, uint32_t& stride) {
        if (c < 512) { off = c * 2; stride = kRowMain; } else { off = kTailBase + (c - 512) * 2; stride = kRowTail; }
    };
    uint32_t o00, s00, o01, s01, o10, s10, o11, s11;
    col(c0, o00, s00); col(c0 + 1, o01, s01); col(c1, o10, s10); col(c1 + 1, o11, s11);
    // (stores & 64: the OUTPUT rows are rotated by 16 chunks per tile row of the XCD: the four resident tile rows then write
    // different row offsets of their tiles at any moment; the source stream is untouched.  Timing only.)
    const uint32_t rot = (stores & 64) ? (ty & 3u) * 16u : ((stores & 128) ? (ty & 3u) * 16u + (tx & 3u) * 4u : 0u);
    for (uint32_t k = 0; k < 64; k++) {
        if (!LOADER_WAVE) {  // every wave streams two rows of the group two chunks ahead (wave 0 also the tails)
            const uint32_t y0 = 8 * k + 18 + 2 * wave;
            dma_row<NTL>(base, ring, y0, lane);
            dma_row<NTL>(base, ring, y0 + 1, lane);
            if (wave == 0) dma_tails(base, ring, 8 * k + 24, lane);
        }
        const uint32_t slot0 = (8 * k) & (kRing - 1);
        auto tex = [&](uint32_t r, uint32_t off, uint32_t stride) -> uint32_t {
            return *reinterpret_cast<const uint16_t*>(lds + off + ((slot0 + r) & (kRing - 1)) * stride);
        };
        f2 carry = {float(tex(0, o00, s00)), float(tex(0, o10, s10))};
        uint32_t out[8];
#pragma unroll
        for (uint32_t r = 0; r < 8; r++) out[r] = shade<ARITH>(tex(r + 1, o00, s00), tex(r + 1, o01, s01), tex(r + 1, o10, s10), tex(r + 1, o11, s11), carry, 0.125f * float(r));
        const uint32_t ko = (k + rot) & 63u;
        if ((stores & 1) && (stores & 256)) {
            // (256: a thread = 4 texture columns x 4 rows — row-ALIGNED 8-byte stores, two waves per 1 KB row, half the store instructions)
            typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
            u32x2* row = reinterpret_cast<u32x2*>(tiles + uint64_t(tx * 32 + ty) * 524288) + (tid & 127u);
#pragma unroll
            for (uint32_t r = 0; r < 4; r++) row[(ko * 8 + (tid >> 7) * 4 + r) * 128] = u32x2{out[2 * r], out[2 * r + 1]};
        } else if (stores & 1) {
#pragma unroll
            for (uint32_t r = 0; r < 8; r++) {
                if (NTS) __builtin_nontemporal_store(out[r], &d5[(ko * 8 + r) * 256]);
                else d5[(ko * 8 + r) * 256] = out[r];
            }
        }
        if ((stores & 2) && (tid & 1u) == 0) {
#pragma unroll
            for (uint32_t r = 0; r < 4; r++) {
                if (NTS) __builtin_nontemporal_store(out[2 * r] + out[2 * r + 1], &d4[(ko * 4 + r) * 256]);
                else d4[(ko * 4 + r) * 256] = out[2 * r] + out[2 * r + 1];
            }
        }
        if (!LOADER_WAVE) {
            // the rows of chunk k + 1 were issued one iteration ago, BEFORE the stores of chunk k - 1: everything issued since
            // may stay in flight (2 or 3 DMA + at least 8 finest stores per iteration, twice) — a lower bound keeps it safe
            if (stores & 1) asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
}

// ------------------------------------------------------------------------------------------------ V1 with 16-row chunks
// (round 4: VERDICT r03 suggested 16-row chunks at four workgroups per CU.)  One chunk ahead like the product: a ring of 36 rows
// (chunk k's 18 + the 16 new rows of chunk k + 1) = 38 KB, four workgroups per CU; 32 barriers per tile instead of 64, 16 + 8
// stores per thread behind every DMA group.
constexpr uint32_t kRing16 = 36, kTailBase16 = kRing16 * kRowMain;
template <int ARITH> __global__ __launch_bounds__(256) void v16(const uint8_t* src, uint8_t* tiles, uint8_t* parents, int stores) {
    __shared__ __attribute__((aligned(16))) uint8_t lds[kRing16 * (kRowMain + kRowTail)];
    uint32_t tx, ty;
    tile_of(tx, ty);
    const uint32_t tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63u;
    const gbytes base = (gbytes)src + uint64_t(ty) * 512 * kPitch + uint64_t(tx) * 1024;
    const lbytes ring = (lbytes)lds;
    auto dma_rows = [&](uint32_t y_begin, uint32_t count) {  // wave w moves rows w, w + 4, ... of the group; lanes 0, 1 the 32-byte tail
        for (uint32_t i = wave; i < count; i += 4) {
            const uint32_t y = y_begin + i, slot = y % kRing16;
            __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(base + uint64_t(y) * kPitch + lane * 16),
                                             (void __attribute__((address_space(3)))*)(ring + slot * kRowMain), 16, 0, 0);
            if (lane < 2)
                __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(base + uint64_t(y) * kPitch + 1024 + lane * 16),
                                                 (void __attribute__((address_space(3)))*)(ring + kTailBase16 + slot * kRowTail), 16, 0, 0);
        }
    };
    dma_rows(0, 18);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    uint32_t* d5 = reinterpret_cast<uint32_t*>(tiles + uint64_t(tx * 32 + ty) * 524288 + 4) + tid;
    uint32_t* d4 = reinterpret_cast<uint32_t*>(parents + uint64_t((tx / 2) * 16 + ty / 2) * 524288 + uint64_t(ty & 1u) * 262144 + (tx & 1u) * 512 + 4) + (tid >> 1);
    const uint32_t c0 = 2 * tid, c1 = 2 * tid + 2;
    auto col = [&](uint32_t c, uint32_t& off, uint32_t& stride) {
        if (c < 512) { off = c * 2; stride = kRowMain; } else { off = kTailBase16 + (c - 512) * 2; stride = kRowTail; }
    };
    uint32_t o00, s00, o01, s01, o10, s10, o11, s11;
    col(c0, o00, s00); col(c0 + 1, o01, s01); col(c1, o10, s10); col(c1 + 1, o11, s11);
    for (uint32_t k = 0; k < 32; k++) {
        if (k + 1 < 32 + 1) dma_rows(16 * k + 18, 16);  // (the last group runs past the tile like the 8-row skeleton's)
        const uint32_t slot0 = (16 * k) % kRing16;
        auto tex = [&](uint32_t r, uint32_t off, uint32_t stride) -> uint32_t {
            uint32_t sl = slot0 + r;
            sl -= sl >= kRing16 ? kRing16 : 0u;
            return *reinterpret_cast<const uint16_t*>(lds + off + sl * stride);
        };
        f2 carry = {float(tex(0, o00, s00)), float(tex(0, o10, s10))};
        uint32_t out[16];
#pragma unroll
        for (uint32_t r = 0; r < 16; r++) out[r] = shade<ARITH>(tex(r + 1, o00, s00), tex(r + 1, o01, s01), tex(r + 1, o10, s10), tex(r + 1, o11, s11), carry, 0.0625f * float(r));
        if (stores & 1) {
#pragma unroll
            for (uint32_t r = 0; r < 16; r++) d5[(k * 16 + r) * 256] = out[r];
        }
        if ((stores & 2) && (tid & 1u) == 0) {
#pragma unroll
            for (uint32_t r = 0; r < 8; r++) d4[(k * 8 + r) * 256] = out[2 * r] + out[2 * r + 1];
        }
        // the rows of chunk k + 1 were issued BEFORE this chunk's stores: the 16 finest stores may stay in flight
        if (stores & 1) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
}

// ------------------------------------------------------------------------------------------------ V1 with CH-row chunks, D chunks ahead
// (the trend of section 10 of the experiments file followed the other way: are SHORTER chunks better?)  Ring of R rows >= CH * (D + 1) + 2.
template <int ARITH, uint32_t CH, uint32_t D, uint32_t R> __global__ __launch_bounds__(256) void vch(const uint8_t* src, uint8_t* tiles, uint8_t* parents, int stores) {
    __shared__ __attribute__((aligned(16))) uint8_t lds[R * (kRowMain + kRowTail)];
    constexpr uint32_t kChunks = 512 / CH, kTail = R * kRowMain;
    uint32_t tx, ty;
    tile_of(tx, ty);
    const uint32_t tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63u;
    const gbytes base = (gbytes)src + uint64_t(ty) * 512 * kPitch + uint64_t(tx) * 1024;
    const lbytes ring = (lbytes)lds;
    auto dma_rows = [&](uint32_t y_begin, uint32_t count) {
        for (uint32_t i = wave; i < count; i += 4) {
            const uint32_t y = y_begin + i, slot = y % R;
            __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(base + uint64_t(y) * kPitch + lane * 16),
                                             (void __attribute__((address_space(3)))*)(ring + slot * kRowMain), 16, 0, 0);
            if (lane < 2)
                __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(base + uint64_t(y) * kPitch + 1024 + lane * 16),
                                                 (void __attribute__((address_space(3)))*)(ring + kTail + slot * kRowTail), 16, 0, 0);
        }
    };
    dma_rows(0, CH * D + 2);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    uint32_t* d5 = reinterpret_cast<uint32_t*>(tiles + uint64_t(tx * 32 + ty) * 524288 + 4) + tid;
    uint32_t* d4 = reinterpret_cast<uint32_t*>(parents + uint64_t((tx / 2) * 16 + ty / 2) * 524288 + uint64_t(ty & 1u) * 262144 + (tx & 1u) * 512 + 4) + (tid >> 1);
    const uint32_t c0 = 2 * tid, c1 = 2 * tid + 2;
    auto col = [&](uint32_t c, uint32_t& off, uint32_t& stride) {
        if (c < 512) { off = c * 2; stride = kRowMain; } else { off = kTail + (c - 512) * 2; stride = kRowTail; }
    };
    uint32_t o00, s00, o01, s01, o10, s10, o11, s11;
    col(c0, o00, s00); col(c0 + 1, o01, s01); col(c1, o10, s10); col(c1 + 1, o11, s11);
    for (uint32_t k = 0; k < kChunks; k++) {
        dma_rows(CH * (k + D) + 2, CH);  // the group D chunks ahead (the last ones run past the tile like the other skeletons')
        const uint32_t slot0 = (CH * k) % R;
        auto tex = [&](uint32_t r, uint32_t off, uint32_t stride) -> uint32_t {
            uint32_t sl = slot0 + r;
            sl -= sl >= R ? R : 0u;
            return *reinterpret_cast<const uint16_t*>(lds + off + sl * stride);
        };
        f2 carry = {float(tex(0, o00, s00)), float(tex(0, o10, s10))};
        uint32_t out[CH];
#pragma unroll
        for (uint32_t r = 0; r < CH; r++) out[r] = shade<ARITH>(tex(r + 1, o00, s00), tex(r + 1, o01, s01), tex(r + 1, o10, s10), tex(r + 1, o11, s11), carry, float(r) / float(CH));
        if (stores & 1) {
#pragma unroll
            for (uint32_t r = 0; r < CH; r++) d5[(k * CH + r) * 256] = out[r];
        }
        if ((stores & 2) && (tid & 1u) == 0) {
#pragma unroll
            for (uint32_t r = 0; r < CH / 2; r++) d4[(k * (CH / 2) + r) * 256] = out[2 * r] + out[2 * r + 1];
        }
        // chunk k + 1's rows were issued D iterations ago (D = 1: this iteration, before the stores): what was issued since may stay in flight
        if (D >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((D - 1) * (CH / 4 + CH)) : "memory");
        else if (stores & 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(CH) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
}

// ------------------------------------------------------------------------------------------------ round 5: the DRAM side
// every variant so far changed the CU side; loads-only + finest-stores-only = both, i.e. reads and writes do not
// overlap in this access pattern.  V1 (LDS-DMA by all waves, ring two chunks ahead) with
//   POLICY  the cache policy of the finest / parent stores from inline assembly: 0 plain, 1 sc0, 2 sc1, 3 sc0 sc1 (write-through at every
//           level), 4 nt, 5 sc0 sc1 nt
//   half    chip-wide read / write PHASE SEPARATION by the real-time clock (100 MHz ticks; 0 = off): a wave issues its DMA rows only
//           while (clock / half) is even and its stores only while it is odd — every workgroup of the chip reads in the same window and
//           writes in the same window, one chunk per period; `anti`: the odd XCDs run half a period late (4 XCDs read while 4 write)
template <int POLICY> __device__ __forceinline__ void store_policy(uint32_t* p, uint32_t v) {
    if (POLICY == 0) asm volatile("global_store_dword %0, %1, off" ::"v"(p), "v"(v) : "memory");
    if (POLICY == 1) asm volatile("global_store_dword %0, %1, off sc0" ::"v"(p), "v"(v) : "memory");
    if (POLICY == 2) asm volatile("global_store_dword %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
    if (POLICY == 3) asm volatile("global_store_dword %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
    if (POLICY == 4) asm volatile("global_store_dword %0, %1, off nt" ::"v"(p), "v"(v) : "memory");
    if (POLICY == 5) asm volatile("global_store_dword %0, %1, off sc0 sc1 nt" ::"v"(p), "v"(v) : "memory");
}
template <int ARITH, int POLICY> __global__ __launch_bounds__(256) void vdram(const uint8_t* src, uint8_t* tiles, uint8_t* parents, uint32_t half, uint32_t anti) {
    __shared__ __attribute__((aligned(16))) uint8_t lds[kRing * (kRowMain + kRowTail)];
    uint32_t tx, ty;
    tile_of(tx, ty);
    const uint32_t tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63u;
    const gbytes base = (gbytes)src + uint64_t(ty) * 512 * kPitch + uint64_t(tx) * 1024;
    const lbytes ring = (lbytes)lds;
    for (uint32_t y = wave; y < 18; y += 4) dma_row<false>(base, ring, y, lane);
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
    const uint32_t shift = anti ? (blockIdx.x & 1u) : 0u;  // blockIdx % 8 = the XCD
    auto wait_phase = [&](uint32_t want) {  // (wave-uniform) spin until the chip-wide window is the wanted one
        if (half == 0) return;
        while ((((uint32_t(__builtin_amdgcn_s_memrealtime()) / half) + shift) & 1u) != want) __builtin_amdgcn_s_sleep(2);
    };
    for (uint32_t k = 0; k < 64; k++) {
        wait_phase(0);
        {
            const uint32_t y0 = 8 * k + 18 + 2 * wave;
            dma_row<false>(base, ring, y0, lane);
            dma_row<false>(base, ring, y0 + 1, lane);
            if (wave == 0) dma_tails(base, ring, 8 * k + 24, lane);
        }
        const uint32_t slot0 = (8 * k) & (kRing - 1);
        auto tex = [&](uint32_t r, uint32_t off, uint32_t stride) -> uint32_t {
            return *reinterpret_cast<const uint16_t*>(lds + off + ((slot0 + r) & (kRing - 1)) * stride);
        };
        f2 carry = {float(tex(0, o00, s00)), float(tex(0, o10, s10))};
        uint32_t out[8];
#pragma unroll
        for (uint32_t r = 0; r < 8; r++) out[r] = shade<ARITH>(tex(r + 1, o00, s00), tex(r + 1, o01, s01), tex(r + 1, o10, s10), tex(r + 1, o11, s11), carry, 0.125f * float(r));
        wait_phase(1);
#pragma unroll
        for (uint32_t r = 0; r < 8; r++) store_policy<POLICY>(&d5[(k * 8 + r) * 256], out[r]);
        if ((tid & 1u) == 0) {
#pragma unroll
            for (uint32_t r = 0; r < 4; r++) store_policy<POLICY>(&d4[(k * 4 + r) * 256], out[2 * r] + out[2 * r + 1]);
        }
        asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
}

// ------------------------------------------------------------------------------------------------ round 5: strided chunks
// PARTS workgroups per tile, workgroup j shades the chunks j, j + PARTS, ...: the PARTS workgroups of a tile (dispatched back to back on one
// XCD) then write PARTS consecutive 8 KB pieces of the tile and read PARTS x 8 consecutive source rows at any moment — longer DRAM bursts,
// a quarter of the tiles in flight — at the price of PARTS generations of workgroups.  Ring: three chunks x 10 rows, two chunks ahead.
template <int ARITH, uint32_t PARTS> __global__ __launch_bounds__(256) void vstride(const uint8_t* src, uint8_t* tiles, uint8_t* parents) {
    __shared__ __attribute__((aligned(16))) uint8_t lds[30 * (kRowMain + kRowTail)];
    constexpr uint32_t kTail30 = 30 * kRowMain, kChunks = 64 / PARTS;
    const uint32_t q = gridDim.x / 8, work = (blockIdx.x % 8) * q + blockIdx.x / 8;
    const uint32_t tile = work / PARTS, part = work % PARTS, ty = tile / 32, tx = tile % 32;
    const uint32_t tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63u;
    const gbytes base = (gbytes)src + uint64_t(ty) * 512 * kPitch + uint64_t(tx) * 1024;
    const lbytes ring = (lbytes)lds;
    auto dma_chunk = [&](uint32_t i) {  // chunk i of this workgroup = tile chunk part + i * PARTS: source rows 8 k .. 8 k + 9 into slot i % 3
        const uint32_t k = part + i * PARTS, slot0 = (i % 3u) * 10u;
        for (uint32_t r = wave; r < 10u; r += 4u) {
            __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(base + uint64_t(8u * k + r) * kPitch + lane * 16),
                                             (void __attribute__((address_space(3)))*)(ring + (slot0 + r) * kRowMain), 16, 0, 0);
            if (lane < 2)
                __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(base + uint64_t(8u * k + r) * kPitch + 1024 + lane * 16),
                                                 (void __attribute__((address_space(3)))*)(ring + kTail30 + (slot0 + r) * kRowTail), 16, 0, 0);
        }
    };
    dma_chunk(0);
    if (kChunks > 1) dma_chunk(1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    uint32_t* d5 = reinterpret_cast<uint32_t*>(tiles + uint64_t(tx * 32 + ty) * 524288 + 4) + tid;
    uint32_t* d4 = reinterpret_cast<uint32_t*>(parents + uint64_t((tx / 2) * 16 + ty / 2) * 524288 + uint64_t(ty & 1u) * 262144 + (tx & 1u) * 512 + 4) + (tid >> 1);
    const uint32_t c0 = 2 * tid, c1 = 2 * tid + 2;
    auto col = [&](uint32_t c, uint32_t& off, uint32_t& stride) {
        if (c < 512) { off = c * 2; stride = kRowMain; } else { off = kTail30 + (c - 512) * 2; stride = kRowTail; }
    };
    uint32_t o00, s00, o01, s01, o10, s10, o11, s11;
    col(c0, o00, s00); col(c0 + 1, o01, s01); col(c1, o10, s10); col(c1 + 1, o11, s11);
    for (uint32_t i = 0; i < kChunks; i++) {
        if (i + 2 < kChunks) dma_chunk(i + 2);
        const uint32_t k = part + i * PARTS, slot0 = (i % 3u) * 10u;
        auto tex = [&](uint32_t r, uint32_t off, uint32_t stride) -> uint32_t { return *reinterpret_cast<const uint16_t*>(lds + off + (slot0 + r) * stride); };
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
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (a three-slot ring: the chunk after next must have landed before it is shaded two barriers on; kept simple)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
}

template <typename F> static float timeit(F f) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int i = 0; i < 30; i++) f();
    hipEventRecord(e0);
    for (int i = 0; i < 100; i++) f();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms * 10;
}

int main(int argc, char** argv) {
    uint8_t *src, *tiles, *parents;
    hipMalloc(&src, 16384ull * kPitch + (1 << 20));  // the last tiles' windows run 32 bytes and 18 rows past the raster
    hipMalloc(&tiles, 1024ull * 524288 + 4096);
    hipMalloc(&parents, 256ull * 524288 + 4096);
    hipMemset(src, 3, 16384ull * kPitch + (1 << 20));
    for (int i = 0; i < 200; i++) v0<0><<<1024, 256>>>(src, tiles, parents, 3);
    hipDeviceSynchronize();
    if (argc > 1 && !strcmp(argv[1], "stride")) {  // round 5: PARTS workgroups per tile, chunks interleaved between them
        for (int rep = 0; rep < 3; rep++) {
            printf("arith 24: V1 %6.1f | strided ring, 1 part %6.1f | 2 parts %6.1f | 4 parts %6.1f | 8 parts %6.1f us\n", timeit([&] { v12<24, false><<<1024, 256>>>(src, tiles, parents, 3); }),
                   timeit([&] { vstride<24, 1><<<1024, 256>>>(src, tiles, parents); }), timeit([&] { vstride<24, 2><<<2048, 256>>>(src, tiles, parents); }),
                   timeit([&] { vstride<24, 4><<<4096, 256>>>(src, tiles, parents); }), timeit([&] { vstride<24, 8><<<8192, 256>>>(src, tiles, parents); }));
            fflush(stdout);
        }
        return 0;
    }
    if (argc > 1 && !strcmp(argv[1], "dram")) {  // round 5: store cache policies x chip-wide read / write phase separation (promotion gate: <= 225 us at arith 24)
        const char* pol[6] = {"plain", "sc0", "sc1", "sc0 sc1", "nt", "sc0 sc1 nt"};
        for (int rep = 0; rep < 2; rep++) {
            printf("reference (compiler-issued stores): V1 arith 24 %6.1f us, arith 0 %6.1f us\n", timeit([&] { v12<24, false><<<1024, 256>>>(src, tiles, parents, 3); }),
                   timeit([&] { v12<0, false><<<1024, 256>>>(src, tiles, parents, 3); }));
            auto run = [&](int policy, uint32_t half, uint32_t anti) -> float {
                switch (policy) {
                    case 0: return timeit([&] { vdram<24, 0><<<1024, 256>>>(src, tiles, parents, half, anti); });
                    case 1: return timeit([&] { vdram<24, 1><<<1024, 256>>>(src, tiles, parents, half, anti); });
                    case 2: return timeit([&] { vdram<24, 2><<<1024, 256>>>(src, tiles, parents, half, anti); });
                    case 3: return timeit([&] { vdram<24, 3><<<1024, 256>>>(src, tiles, parents, half, anti); });
                    case 4: return timeit([&] { vdram<24, 4><<<1024, 256>>>(src, tiles, parents, half, anti); });
                    default: return timeit([&] { vdram<24, 5><<<1024, 256>>>(src, tiles, parents, half, anti); });
                }
            };
            for (int policy = 0; policy < 6; policy++) {
                printf("arith 24, stores %-10s: no phases %6.1f us |", pol[policy], run(policy, 0, 0));
                for (uint32_t half : {130u, 150u, 170u, 200u}) printf(" half %3.1f us: in phase %6.1f anti %6.1f |", half / 100.0, run(policy, half, 0), run(policy, half, 1));
                printf("\n");
                fflush(stdout);
            }
        }
        return 0;
    }
    if (argc > 1 && !strcmp(argv[1], "rows16")) {  // round 4: 16-row chunks against 8-row chunks (V1), same lease, alternating
        for (int rep = 0; rep < 3; rep++) {
            printf("arith  0: 8-row chunks %6.1f   16-row chunks %6.1f us\n", timeit([&] { v12<0, false><<<1024, 256>>>(src, tiles, parents, 3); }), timeit([&] { v16<0><<<1024, 256>>>(src, tiles, parents, 3); }));
            printf("arith 24: 8-row chunks %6.1f   16-row chunks %6.1f us\n", timeit([&] { v12<24, false><<<1024, 256>>>(src, tiles, parents, 3); }), timeit([&] { v16<24><<<1024, 256>>>(src, tiles, parents, 3); }));
            printf("arith 48: 8-row chunks %6.1f   16-row chunks %6.1f us\n", timeit([&] { v12<48, false><<<1024, 256>>>(src, tiles, parents, 3); }), timeit([&] { v16<48><<<1024, 256>>>(src, tiles, parents, 3); }));
            fflush(stdout);
        }
        return 0;
    }
    if (argc > 1 && !strcmp(argv[1], "rows")) {  // round 4: chunk height swept, same lease, alternating
        for (int rep = 0; rep < 2; rep++) {
            printf("arith  0: V1 (8 rows, 2 ahead) %6.1f | 2 rows x 4 ahead %6.1f | 4 rows x 2 ahead %6.1f | 4 rows x 4 ahead %6.1f | 8 rows x 2 ahead %6.1f | 8 x 1 %6.1f | 16 x 1 %6.1f us\n",
                   timeit([&] { v12<0, false><<<1024, 256>>>(src, tiles, parents, 3); }), timeit([&] { vch<0, 2, 4, 16><<<1024, 256>>>(src, tiles, parents, 3); }),
                   timeit([&] { vch<0, 4, 2, 16><<<1024, 256>>>(src, tiles, parents, 3); }), timeit([&] { vch<0, 4, 4, 32><<<1024, 256>>>(src, tiles, parents, 3); }),
                   timeit([&] { vch<0, 8, 2, 32><<<1024, 256>>>(src, tiles, parents, 3); }), timeit([&] { vch<0, 8, 1, 20><<<1024, 256>>>(src, tiles, parents, 3); }),
                   timeit([&] { vch<0, 16, 1, 36><<<1024, 256>>>(src, tiles, parents, 3); }));
            printf("arith 24: V1 (8 rows, 2 ahead) %6.1f | 2 rows x 4 ahead %6.1f | 4 rows x 2 ahead %6.1f | 4 rows x 4 ahead %6.1f | 8 rows x 2 ahead %6.1f | 8 x 1 %6.1f | 16 x 1 %6.1f us\n",
                   timeit([&] { v12<24, false><<<1024, 256>>>(src, tiles, parents, 3); }), timeit([&] { vch<24, 2, 4, 16><<<1024, 256>>>(src, tiles, parents, 3); }),
                   timeit([&] { vch<24, 4, 2, 16><<<1024, 256>>>(src, tiles, parents, 3); }), timeit([&] { vch<24, 4, 4, 32><<<1024, 256>>>(src, tiles, parents, 3); }),
                   timeit([&] { vch<24, 8, 2, 32><<<1024, 256>>>(src, tiles, parents, 3); }), timeit([&] { vch<24, 8, 1, 20><<<1024, 256>>>(src, tiles, parents, 3); }),
                   timeit([&] { vch<24, 16, 1, 36><<<1024, 256>>>(src, tiles, parents, 3); }));
            fflush(stdout);
        }
        return 0;
    }
    if (argc > 1 && !strcmp(argv[1], "nt")) {  // round 4: every source row read once (V1's ring) — does the non-temporal policy pay in the tile-stream pattern?
        for (int rep = 0; rep < 2; rep++) {
            printf("V1 arith  0: plain %6.1f   nt loads %6.1f   nt stores %6.1f   both %6.1f us\n", timeit([&] { v12<0, false, false, false><<<1024, 256>>>(src, tiles, parents, 3); }),
                   timeit([&] { v12<0, false, true, false><<<1024, 256>>>(src, tiles, parents, 3); }), timeit([&] { v12<0, false, false, true><<<1024, 256>>>(src, tiles, parents, 3); }),
                   timeit([&] { v12<0, false, true, true><<<1024, 256>>>(src, tiles, parents, 3); }));
            printf("V1 arith 24: plain %6.1f   nt loads %6.1f   nt stores %6.1f   both %6.1f us\n", timeit([&] { v12<24, false, false, false><<<1024, 256>>>(src, tiles, parents, 3); }),
                   timeit([&] { v12<24, false, true, false><<<1024, 256>>>(src, tiles, parents, 3); }), timeit([&] { v12<24, false, false, true><<<1024, 256>>>(src, tiles, parents, 3); }),
                   timeit([&] { v12<24, false, true, true><<<1024, 256>>>(src, tiles, parents, 3); }));
            printf("V1 arith 48: plain %6.1f   nt loads %6.1f   nt stores %6.1f   both %6.1f us\n", timeit([&] { v12<48, false, false, false><<<1024, 256>>>(src, tiles, parents, 3); }),
                   timeit([&] { v12<48, false, true, false><<<1024, 256>>>(src, tiles, parents, 3); }), timeit([&] { v12<48, false, false, true><<<1024, 256>>>(src, tiles, parents, 3); }),
                   timeit([&] { v12<48, false, true, true><<<1024, 256>>>(src, tiles, parents, 3); }));
            printf("V2 arith 24: plain %6.1f   nt loads %6.1f   both %6.1f us\n", timeit([&] { v12<24, true, false, false><<<1024, 320>>>(src, tiles, parents, 3); }),
                   timeit([&] { v12<24, true, true, false><<<1024, 320>>>(src, tiles, parents, 3); }), timeit([&] { v12<24, true, true, true><<<1024, 320>>>(src, tiles, parents, 3); }));
            fflush(stdout);
        }
        return 0;
    }
    const char* names[4] = {"loads only", "loads + finest stores", "loads + parent stores", "loads + finest + parent stores"};
    for (int stores = 0; stores < 4; stores++) {
        printf("%-32s arith 0 : V0 %6.1f  V1 %6.1f  V2 %6.1f us\n", names[stores], timeit([&] { v0<0><<<1024, 256>>>(src, tiles, parents, stores); }),
               timeit([&] { v12<0, false><<<1024, 256>>>(src, tiles, parents, stores); }), timeit([&] { v12<0, true><<<1024, 320>>>(src, tiles, parents, stores); }));
        fflush(stdout);
    }
    for (int f : {4 + 1, 4 + 1 + 32, 4 + 1 + 8, 4 + 3, 1, 1 + 32, 1 + 8, 1 + 16, 1 + 8 + 16, 3 + 8 + 16, 3 + 8 + 16 + 32})
        printf("V0 flags %2d (1 finest, 2 parents, 4 no loads, 8 nt stores, 16 nt loads, 32 aligned rows): %6.1f us\n", f, timeit([&] { v0<0><<<1024, 256>>>(src, tiles, parents, f); }));
    for (int a : {0, 24}) {
        printf("arith %2d, loads + finest + parents: dword per lane (4-byte shifted) V1 %6.1f, 8 bytes per lane row-aligned V1 %6.1f us\n", a,
               a ? timeit([&] { v12<24, false><<<1024, 256>>>(src, tiles, parents, 3); }) : timeit([&] { v12<0, false><<<1024, 256>>>(src, tiles, parents, 3); }),
               a ? timeit([&] { v12<24, false><<<1024, 256>>>(src, tiles, parents, 3 + 256); }) : timeit([&] { v12<0, false><<<1024, 256>>>(src, tiles, parents, 3 + 256); }));
    }
    for (int f : {3, 3 + 64, 3 + 128, 1, 1 + 64, 1 + 128})
        printf("V1 / V2 stores %3d (64: output rows rotated by tile row, 128: by tile row and column): %6.1f  %6.1f us\n", f,
               timeit([&] { v12<0, false><<<1024, 256>>>(src, tiles, parents, f); }), timeit([&] { v12<0, true><<<1024, 320>>>(src, tiles, parents, f); }));
    printf("%-32s arith 4 : V0 %6.1f  V1 %6.1f  V2 %6.1f us\n", names[3], timeit([&] { v0<4><<<1024, 256>>>(src, tiles, parents, 3); }),
           timeit([&] { v12<4, false><<<1024, 256>>>(src, tiles, parents, 3); }), timeit([&] { v12<4, true><<<1024, 320>>>(src, tiles, parents, 3); }));
    printf("%-32s arith 12: V0 %6.1f  V1 %6.1f  V2 %6.1f us\n", names[3], timeit([&] { v0<12><<<1024, 256>>>(src, tiles, parents, 3); }),
           timeit([&] { v12<12, false><<<1024, 256>>>(src, tiles, parents, 3); }), timeit([&] { v12<12, true><<<1024, 320>>>(src, tiles, parents, 3); }));
    printf("%-32s arith 24: V0 %6.1f  V1 %6.1f  V2 %6.1f us\n", names[3], timeit([&] { v0<24><<<1024, 256>>>(src, tiles, parents, 3); }),
           timeit([&] { v12<24, false><<<1024, 256>>>(src, tiles, parents, 3); }), timeit([&] { v12<24, true><<<1024, 320>>>(src, tiles, parents, 3); }));
    return 0;
}
