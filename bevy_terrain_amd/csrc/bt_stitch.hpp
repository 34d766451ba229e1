// Apron (stitch) arithmetic shared by the batched kernels (bt_kernels.hip) and the fused plan (bt_fused.hip, whose tail launch
// carries the cube's cross-face seam regions as extra workgroups).  Device code only; included inside namespace bt { namespace { ... } }.
#pragma once

// stitch.wgsl:12-51; PS/PT/NS/NT = +x, +y, T-1-x, T-1-y of the input coordinate
__device__ __forceinline__ uint2 project_to_side(uint32_t x, uint32_t y, uint32_t Tsz, uint32_t own, uint32_t other) {
    // two bits per output axis, low = x: 0 PS, 1 PT, 2 NS, 3 NT
    constexpr uint32_t kEven[6] = {0 | 1 << 2, 0 | 1 << 2, 3 | 0 << 2, 3 | 2 << 2, 1 | 2 << 2, 0 | 1 << 2};
    constexpr uint32_t kOdd[6] = {0 | 1 << 2, 0 | 1 << 2, 1 | 2 << 2, 1 | 0 << 2, 3 | 0 << 2, 0 | 1 << 2};
    const uint32_t index = (6u + other - own) % 6u;
    const uint32_t info = (own % 2u == 0u) ? kEven[index] : kOdd[index];
    const uint32_t v[4] = {x, y, Tsz - 1u - x, Tsz - 1u - y};
    return make_uint2(v[info & 3u], v[(info >> 2) & 3u]);
}

// apron pixel -> (layer, x, y) it copies, per stitch.wgsl:53-118
__device__ __forceinline__ void stitch_source(const TaskDev& task, uint32_t px, uint32_t py, uint32_t Tsz, uint32_t b,
                                              uint32_t c, uint32_t& layer, uint32_t& sx, uint32_t& sy) {
    const uint32_t o = b + c;
    // region ids of neighbour_index(): 0 top, 1 right, 2 bottom, 3 left, 4 TL, 5 TR, 6 BR, 7 BL
    const int rx = px < b ? -1 : (px >= o ? 1 : 0);
    const int ry = py < b ? -1 : (py >= o ? 1 : 0);
    uint32_t region;
    if (ry < 0) region = rx < 0 ? 4u : (rx > 0 ? 5u : 0u);
    else if (ry > 0) region = rx < 0 ? 7u : (rx > 0 ? 6u : 2u);
    else region = rx > 0 ? 1u : 3u;
    const uint32_t nb = task.rel_index[region];
    if (nb == 0xFFFFFFFFu) {  // repeat_data: clamp into the own centre
        layer = task.atlas_index;
        sx = min(max(px, b), o - 1u);
        sy = min(max(py, b), o - 1u);
        return;
    }
    // neighbour_data: offsets[region] = -(neighbour offset) * c
    const uint32_t qx = uint32_t(int(px) - rx * int(c));
    const uint32_t qy = uint32_t(int(py) - ry * int(c));
    const uint2 q = project_to_side(qx, qy, Tsz, task.side, task.rel_side[region]);
    layer = nb;
    sx = q.x;
    sy = q.y;
}


// ONE apron region of one tile (task.regions has exactly one bit), by one workgroup of 256 threads: the cube's cross-face seams after the
// fused plans, where a face-edge tile needs one edge + two corners of its eight regions — no thread is launched for the other five.
// kPack = 2 (R16, b even): a thread moves two horizontally adjacent pixels and stores them as one dword.
template <typename T, uint32_t kPack>
__device__ __forceinline__ void stitch_region_body(const AttachmentMeta& m, void* atlas_, const TaskDev& task) {
    const uint32_t Tsz = m.texture_size, b = m.border_size, c = m.center_size, o = b + c;
    const uint32_t region = uint32_t(__ffs(int(task.regions))) - 1u;  // 0 top, 1 right, 2 bottom, 3 left, 4 TL, 5 TR, 6 BR, 7 BL
    const uint32_t x0 = (region == 0u || region == 2u) ? b : ((region == 1u || region == 5u || region == 6u) ? o : 0u);
    const uint32_t y0 = (region == 1u || region == 3u) ? b : ((region == 2u || region == 6u || region == 7u) ? o : 0u);
    const uint32_t w = (region == 0u || region == 2u) ? c : b, h = (region == 1u || region == 3u) ? c : b;
    T* atlas = (T*)atlas_;
    for (uint32_t i = threadIdx.x; i < (w / kPack) * h; i += 256u) {
        const uint32_t px = x0 + kPack * (i % (w / kPack)), py = y0 + i / (w / kPack);
        if (py >= m.row_limit) continue;
        uint32_t v[kPack];
#pragma unroll
        for (uint32_t e = 0; e < kPack; e++) {
            uint32_t layer, sx, sy;
            stitch_source(task, px + e, py, Tsz, b, c, layer, sx, sy);
            // (the load unconditional, from a clamped address: a conditional load is a divergent block of its own with a wait behind it — kPack round
            // trips per thread instead of one)
            const bool inside = layer < m.atlas_size && sx < Tsz && sy < Tsz;
            const uint32_t t = uint32_t(atlas[uint64_t(inside ? layer : task.atlas_index) * Tsz * Tsz + uint64_t(inside ? sy : 0u) * Tsz + (inside ? sx : 0u)]);
            v[e] = inside ? t : 0u;
        }
        T* dst = atlas + uint64_t(task.atlas_index) * Tsz * Tsz + uint64_t(py) * Tsz + px;
        if constexpr (kPack == 2) *reinterpret_cast<uint32_t*>(dst) = v[0] | (v[1] << 16);
        else *dst = T(v[0]);
    }
}
