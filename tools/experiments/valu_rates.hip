// Issue cost of the VALU instructions the Rgba8 paths are made of (gfx950), relative to v_fma_f32: 8 independent chains per
// wave, 4 waves per SIMD, every CU busy.  Also prints what v_cvt_pk_u8_f32 does with .5 cases (its rounding is not documented
// in the guides at hand).  Build: hipcc --offload-arch=gfx950 -O3 -o tools/valu_rates.out tools/experiments/valu_rates.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define OP8(fmt)                                                                                                              \
    asm volatile(fmt(0) "\n" fmt(1) "\n" fmt(2) "\n" fmt(3) "\n" fmt(4) "\n" fmt(5) "\n" fmt(6) "\n" fmt(7)                      \
                 : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7])             \
                 : "v"(a), "v"(b))
#define F_FMA(i) "v_fma_f32 %" #i ", %" #i ", %8, %9"
#define F_CVT_UB0(i) "v_cvt_f32_ubyte0 %" #i ", %" #i
#define F_CVT_UB2(i) "v_cvt_f32_ubyte2 %" #i ", %" #i
#define F_CVT_U32(i) "v_cvt_u32_f32 %" #i ", %" #i
#define F_CVT_F32(i) "v_cvt_f32_u32 %" #i ", %" #i
#define F_PERM(i) "v_perm_b32 %" #i ", %" #i ", %8, %9"
#define F_FLOOR(i) "v_floor_f32 %" #i ", %" #i
#define F_PKU8(i) "v_cvt_pk_u8_f32 %" #i ", %" #i ", %8, %9"
#define F_ANDOR(i) "v_and_or_b32 %" #i ", %" #i ", %8, %9"
#define F_LSHLOR(i) "v_lshl_or_b32 %" #i ", %" #i ", %8, %9"
#define F_MIN3(i) "v_min3_u32 %" #i ", %" #i ", %8, %9"
#define F_DPP(i) "v_mov_b32_dpp %" #i ", %" #i " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"
#define F_ADD(i) "v_add_f32 %" #i ", %" #i ", %8"
#define F_BFE(i) "v_bfe_u32 %" #i ", %" #i ", %8, %9"
#define F_SDWA(i) "v_cvt_f32_ubyte0_sdwa %" #i ", %" #i " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1"
template <int MODE>
__global__ __launch_bounds__(256) void k(uint32_t* out, int iters, uint32_t a, uint32_t b) {
    uint32_t x[8];
    for (int j = 0; j < 8; j++) x[j] = threadIdx.x * 8 + j;
    for (int i = 0; i < iters; i++) {
        if (MODE == 0) OP8(F_FMA);
        if (MODE == 1) OP8(F_CVT_UB0);
        if (MODE == 2) OP8(F_CVT_UB2);
        if (MODE == 3) OP8(F_CVT_U32);
        if (MODE == 4) OP8(F_CVT_F32);
        if (MODE == 5) OP8(F_PERM);
        if (MODE == 6) OP8(F_FLOOR);
        if (MODE == 7) OP8(F_PKU8);
        if (MODE == 8) OP8(F_ANDOR);
        if (MODE == 9) OP8(F_LSHLOR);
        if (MODE == 10) OP8(F_MIN3);
        if (MODE == 11) OP8(F_DPP);
        if (MODE == 12) OP8(F_ADD);
        if (MODE == 13) OP8(F_BFE);
        if (MODE == 14) OP8(F_SDWA);
    }
    uint32_t s = 0;
    for (int j = 0; j < 8; j++) s += x[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
// two packed instructions on 4 chains of register pairs
template <int mode>
__global__ __launch_bounds__(256) void kpk(float* out, int iters, float a) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p[4];
    for (int j = 0; j < 4; j++) p[j] = f2{float(threadIdx.x + j), float(j)};
    const f2 a2 = {a, a};
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int r = 0; r < 2; r++)
#pragma unroll
            for (int j = 0; j < 4; j++) {
                if (mode == 0) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[j]) : "v"(a2));
                else if (mode == 1) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[j]) : "v"(a2));
                else asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[j]) : "v"(a2));
            }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = p[0].x + p[1].y + p[2].x + p[3].y;
}
__global__ void pku8(const float* in, uint32_t* out, int n) {
    const int i = threadIdx.x;
    if (i < n) {
        uint32_t r = 0xAABBCCDDu;
        const float v = in[i];
        asm volatile("v_cvt_pk_u8_f32 %0, %1, 1, %0" : "+v"(r) : "v"(v));
        out[i] = r;
    }
}
static float ref_ms = 0;
template <int MODE>
void run(const char* name, uint32_t* out) {
    const int iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<1024, 256>>>(out, 100, 0x3f800100u, 0x05040100u);
    hipEventRecord(e0);
    k<MODE><<<1024, 256>>>(out, iters, 0x3f800100u, 0x05040100u);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (MODE == 0) ref_ms = ms;
    printf("%-24s %.3f ms  = %.2f x v_fma_f32\n", name, ms, ms / ref_ms);
}
int main() {
    uint32_t* out; hipMalloc(&out, 1024 * 256 * 4);
    for (int rep = 0; rep < 2; rep++) run<0>("v_fma_f32", out);
    run<1>("v_cvt_f32_ubyte0", out); run<2>("v_cvt_f32_ubyte2", out); run<3>("v_cvt_u32_f32", out); run<4>("v_cvt_f32_u32", out);
    run<5>("v_perm_b32", out); run<6>("v_floor_f32", out); run<7>("v_cvt_pk_u8_f32", out); run<8>("v_and_or_b32", out);
    run<9>("v_lshl_or_b32", out); run<10>("v_min3_u32", out); run<11>("v_mov_b32_dpp", out); run<12>("v_add_f32", out);
    run<13>("v_bfe_u32", out); run<14>("v_cvt_f32_ubyte0_sdwa", out);
    auto packed = [&](auto launch, const char* name) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        launch(100);
        hipEventRecord(e0);
        launch(20000);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-24s %.3f ms  = %.2f x v_fma_f32 per instruction (2 results each)\n", name, ms, ms / ref_ms);
    };
    packed([&](int it) { kpk<0><<<1024, 256>>>((float*)out, it, 1.0001f); }, "v_pk_add_f32");
    packed([&](int it) { kpk<1><<<1024, 256>>>((float*)out, it, 1.0001f); }, "v_pk_mul_f32");
    packed([&](int it) { kpk<2><<<1024, 256>>>((float*)out, it, 1.0001f); }, "v_pk_fma_f32");
    const float vals[] = {0.0f, 0.49f, 0.5f, 0.51f, 1.5f, 2.5f, 2.7f, 3.5f, 254.5f, 255.4f, 255.5f, 256.0f, 300.0f, -1.0f};
    const int n = sizeof(vals) / sizeof(float);
    float* din; uint32_t* dout; hipMalloc(&din, sizeof(vals)); hipMalloc(&dout, n * 4);
    hipMemcpy(din, vals, sizeof(vals), hipMemcpyHostToDevice);
    pku8<<<1, 64>>>(din, dout, n);
    uint32_t res[32]; hipMemcpy(res, dout, n * 4, hipMemcpyDeviceToHost);
    for (int i = 0; i < n; i++) printf("v_cvt_pk_u8_f32(%g, byte 1, 0xAABBCCDD) = 0x%08X\n", vals[i], res[i]);
    return 0;
}
