// MFMA issue-rate probe (gfx950): 48 v_mfma_f32_32x32x16_bf16 per iteration on 8 accumulators, three dependency patterns;
// 512 threads per workgroup, one workgroup per CU (two waves per SIMD), as csrc/gemm_split.hip runs.
//   hipcc --offload-arch=gfx950 -O3 -o tools/mfma_chain_probe.bin tools/mfma_chain_probe.hip && tools/mfma_chain_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int PATTERN>
__global__ __launch_bounds__(512) void probe(float *out, int iters, unsigned long long *cyc) {
    f32x16 acc[8];
    for (int i = 0; i < 8; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    bf16x8 a[3], b[3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 8; ++j) { a[i][j] = (__bf16)(float)(threadIdx.x + i + j); b[i][j] = (__bf16)(float)(i * j + 1); }
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if (PATTERN == 0) {          // product outer, accumulator inner: dependent instructions 8 apart
#pragma unroll
            for (int pr = 0; pr < 6; ++pr)
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[pr % 3], b[pr / 2], acc[i], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
        } else if (PATTERN == 1) {   // two accumulators alternating, six products each, then the next pair
#pragma unroll
            for (int i = 0; i < 8; i += 2)
#pragma unroll
                for (int pr = 0; pr < 6; ++pr) {
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[pr % 3], b[pr / 2], acc[i], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    acc[i + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[pr % 3], b[pr / 2], acc[i + 1], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
        } else {                     // six products back to back on ONE accumulator
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int pr = 0; pr < 6; ++pr) {
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[pr % 3], b[pr / 2], acc[i], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int i = 0; i < 8; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int P>
static void run(const char *name, float *out, unsigned long long *cyc) {
    const int iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    probe<P><<<256, 512>>>(out, 100, cyc);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    probe<P><<<256, 512>>>(out, iters, cyc);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double mfma_per_simd = 2.0 * 48 * iters;   // two waves per SIMD
    const double tflops = 256.0 * 8 * 48 * iters * 2.0 * 32 * 32 * 16 / (ms * 1e-3) / 1e12;
    printf("%-44s %8.3f ms  %7.1f TFLOP/s bf16  s_memtime ticks/MFMA/SIMD %.2f  -> wall clock needed for 32-cycle issue: %.2f GHz\n", name, ms,
           tflops, (double)c / mfma_per_simd, mfma_per_simd * 32 / (ms * 1e-3) / 1e9);
}

int main() {
    float *out; unsigned long long *cyc;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 8);
    for (int rep = 0; rep < 2; ++rep) {
        run<0>("8 accumulators round robin", out, cyc);
        run<1>("2 accumulators alternating x6, 4 pairs", out, cyc);
        run<2>("1 accumulator x6 back to back, 8 in turn", out, cyc);
    }
    return 0;
}
