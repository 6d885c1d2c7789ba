// valu_probe.hip -- developer micro-benchmark: issue cost (cycles per wave64 instruction, one wave per SIMD) of the
// VALU instructions the argmin epilogue is made of, on gfx950.  s_memtime ticks are shader-clock cycles.
// build: hipcc --offload-arch=gfx950 -O3 tools/valu_probe.hip -o tools/valu_probe.bin   (run on the GPU box)
#include <hip/hip_runtime.h>
#include <stdio.h>

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

template <int KIND>
__global__ __launch_bounds__(256) void probe(float *out, int iters, unsigned long long *cyc) {
    float a = threadIdx.x * 0.5f, b = 1.5f, c = 0.25f, d = 3.0f, e = 7.0f, f = 9.0f, g = 11.f, h = 13.f;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (KIND == 0) asm volatile(REP64("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f));
        if (KIND == 1) asm volatile(REP64("v_add_f32 %0, %0, %4\n v_add_f32 %1, %1, %4\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %4\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f));
        if (KIND == 2) asm volatile(REP64("v_min_f32 %0, %0, %4\n v_min_f32 %1, %1, %4\n v_min_f32 %2, %2, %4\n v_min_f32 %3, %3, %4\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f));
        if (KIND == 3) asm volatile(REP64("v_min3_f32 %0, %0, %4, %5\n v_min3_f32 %1, %1, %4, %5\n v_min3_f32 %2, %2, %4, %5\n v_min3_f32 %3, %3, %4, %5\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f));
        if (KIND == 4) asm volatile(REP64("v_cmp_lt_f32 vcc, %0, %4\n v_cmp_lt_f32 vcc, %1, %4\n v_cmp_lt_f32 vcc, %2, %4\n v_cmp_lt_f32 vcc, %3, %4\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f) : "vcc");
        if (KIND == 5) asm volatile(REP64("v_cmp_lt_f32 s[20:21], %0, %4\n v_cmp_lt_f32 s[22:23], %1, %4\n v_cmp_lt_f32 s[24:25], %2, %4\n v_cmp_lt_f32 s[26:27], %3, %4\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f) : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");
        if (KIND == 6) asm volatile(REP64("v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %4, vcc\n v_cndmask_b32 %2, %2, %4, vcc\n v_cndmask_b32 %3, %3, %4, vcc\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f) : "vcc");
        if (KIND == 7) asm volatile(REP64("v_cndmask_b32 %0, %0, %4, s[20:21]\n v_cndmask_b32 %1, %1, %4, s[22:23]\n v_cndmask_b32 %2, %2, %4, s[24:25]\n v_cndmask_b32 %3, %3, %4, s[26:27]\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f) : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");
        if (KIND == 8) asm volatile(REP64("v_cmp_lt_f32 vcc, %0, %4\n v_cndmask_b32 %1, %1, %4, vcc\n v_cmp_lt_f32 vcc, %2, %4\n v_cndmask_b32 %3, %3, %4, vcc\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f) : "vcc");
        if (KIND == 9) asm volatile(REP64("v_pk_fma_f32 %0, %0, %4, %4\n v_pk_fma_f32 %1, %1, %4, %4\n v_pk_fma_f32 %2, %2, %4, %4\n v_pk_fma_f32 %3, %3, %4, %4\n") : "+v"(*(double *)&a), "+v"(*(double *)&c), "+v"(*(double *)&e), "+v"(*(double *)&g) : "v"(*(double *)&a));
        if (KIND == 10) asm volatile(REP64("v_min_u32 %0, %0, %4\n v_max_i32 %1, %1, %4\n v_and_b32 %2, %2, %4\n v_lshl_or_b32 %3, %3, 1, %4\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f));
        if (KIND == 11) asm volatile(REP64("v_mov_b32_dpp %0, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f));
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d + e + f + g + h;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int KIND>
void run(const char *name) {
    float *out;
    unsigned long long *cyc, hc = 0;
    (void)hipMalloc(&out, 256 * 256 * 4);
    (void)hipMalloc(&cyc, 8);
    const int iters = 200;
    probe<KIND><<<256, 256>>>(out, iters, cyc);
    (void)hipDeviceSynchronize();
    probe<KIND><<<256, 256>>>(out, iters, cyc);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(&hc, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-40s %6.2f cycles / instruction\n", name, (double)hc / (iters * 256.0));
    (void)hipFree(out);
    (void)hipFree(cyc);
}

int main() {
    run<0>("v_fma_f32");
    run<1>("v_add_f32");
    run<2>("v_min_f32");
    run<3>("v_min3_f32");
    run<4>("v_cmp_lt_f32 -> vcc");
    run<5>("v_cmp_lt_f32 -> sgpr pair");
    run<6>("v_cndmask_b32 (vcc)");
    run<7>("v_cndmask_b32 (sgpr pair)");
    run<8>("v_cmp -> v_cndmask dependent pairs");
    run<9>("v_pk_fma_f32");
    run<10>("integer min/max/and/lshl_or mix");
    run<11>("v_mov_b32_dpp quad_perm");
    return 0;
}
