// Mutation fuzzer for the source-image decoders (bevy_terrain_amd/csrc/bt_image.cpp), built with AddressSanitizer + UBSan on the CPU:
//   fuzz_image <seed file>... -- <iterations> <rng seed>
// Every iteration takes one seed file, applies 1-8 random mutations (byte flips, random bytes, 16/32-bit boundary values, truncation,
// block duplication / removal, splices from another seed), decodes it as R16 and as Rgba8 and frees the result.  A sanitizer report or a
// signal is the failure; a decode error is the expected outcome of nearly every mutant.  Test infrastructure (tests/test_image_decode.py).
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cstdarg>
#include <string>
#include <vector>

#include "bevy_terrain_amd.h"

// the two library-internal symbols bt_image.cpp needs (defined in bt_host.cpp in the product)
#include <hip/hip_runtime_api.h>
namespace bt {
void set_error(const char*, ...) {}
bt_status hip_fail(hipError_t, const char*) { return BT_ERR_DEVICE; }
}  // namespace bt

static uint64_t rng_state;
static uint32_t rnd() {
    rng_state ^= rng_state << 13;
    rng_state ^= rng_state >> 7;
    rng_state ^= rng_state << 17;
    return uint32_t(rng_state >> 16);
}

int main(int argc, char** argv) {
    std::vector<std::vector<uint8_t>> seeds;
    int i = 1;
    for (; i < argc && strcmp(argv[i], "--"); i++) {
        FILE* f = fopen(argv[i], "rb");
        if (!f) return 2;
        std::vector<uint8_t> b;
        uint8_t tmp[4096];
        size_t got;
        while ((got = fread(tmp, 1, sizeof tmp, f)) > 0) b.insert(b.end(), tmp, tmp + got);
        fclose(f);
        seeds.push_back(b);
    }
    if (seeds.empty() || i + 2 >= argc) return 2;
    const long iterations = atol(argv[i + 1]);
    rng_state = strtoull(argv[i + 2], nullptr, 10) * 0x9E3779B97F4A7C15ull + 1;
    static const uint32_t kEdge[] = {0, 1, 2, 7, 8, 0x7F, 0x80, 0xFF, 0x100, 0x7FFF, 0x8000, 0xFFFF, 0x10000, 0x7FFFFFFF, 0x80000000u, 0xFFFFFFFFu};
    long decoded = 0;
    for (long it = 0; it < iterations; it++) {
        std::vector<uint8_t> b = seeds[rnd() % seeds.size()];
        const uint32_t muts = 1 + rnd() % 8;
        for (uint32_t m = 0; m < muts && !b.empty(); m++) {
            const size_t at = rnd() % b.size();
            switch (rnd() % 9) {
                case 0: b[at] ^= uint8_t(1u << (rnd() % 8)); break;
                case 1: b[at] = uint8_t(rnd()); break;
                case 2: {  // a 16-bit boundary value, either byte order
                    if (at + 2 > b.size()) break;
                    const uint32_t v = kEdge[rnd() % 16];
                    if (rnd() & 1) { b[at] = uint8_t(v); b[at + 1] = uint8_t(v >> 8); } else { b[at] = uint8_t(v >> 8); b[at + 1] = uint8_t(v); }
                    break;
                }
                case 3: {  // a 32-bit boundary value
                    if (at + 4 > b.size()) break;
                    const uint32_t v = kEdge[rnd() % 16];
                    for (int k = 0; k < 4; k++) b[at + k] = uint8_t((rnd() & 1) ? v >> (8 * k) : v >> (24 - 8 * k));
                    break;
                }
                case 4: b.resize(at); break;                                                    // truncate
                case 5: {  // duplicate a block
                    const size_t len = 1 + rnd() % 64;
                    if (at + len > b.size()) break;
                    std::vector<uint8_t> blk(b.begin() + at, b.begin() + at + len);
                    b.insert(b.begin() + at, blk.begin(), blk.end());
                    break;
                }
                case 6: {  // remove a block
                    const size_t len = 1 + rnd() % 64;
                    if (at + len > b.size()) break;
                    b.erase(b.begin() + at, b.begin() + at + len);
                    break;
                }
                case 7: {  // splice from another seed
                    const std::vector<uint8_t>& o = seeds[rnd() % seeds.size()];
                    const size_t from = rnd() % o.size(), len = 1 + rnd() % 128;
                    for (size_t k = 0; k < len && from + k < o.size() && at + k < b.size(); k++) b[at + k] = o[from + k];
                    break;
                }
                default: {  // a run of one byte
                    const size_t len = 1 + rnd() % 32;
                    const uint8_t v = uint8_t(rnd());
                    for (size_t k = 0; k < len && at + k < b.size(); k++) b[at + k] = v;
                }
            }
        }
        for (uint32_t format : {uint32_t(BT_FORMAT_R16), uint32_t(BT_FORMAT_RGBA8)}) {
            bt_image img;
            if (bt_image_decode(b.data(), b.size(), format, &img) == BT_OK) {
                // touch every byte the result claims to hold (ASan checks the claim)
                uint64_t sum = 0;
                const uint8_t* d = (const uint8_t*)img.data;
                for (uint64_t y = 0; y < img.height; y++)
                    for (uint64_t x = 0; x < uint64_t(img.width) * (format == BT_FORMAT_R16 ? 2 : 4); x++) sum += d[y * img.row_pitch + x];
                decoded += 1 + long(sum & 0);
                bt_image_free(&img);
            }
        }
    }
    printf("iterations %ld decoded %ld\n", iterations, decoded);
    return 0;
}
