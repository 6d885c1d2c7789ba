// overlap_probe.hip -- developer micro-benchmark: do fp32 MFMAs (v_mfma_f32_32x32x2_f32, an "SGEMM" op, not XDL)
// and plain VALU work of ANOTHER wave on the same SIMD run concurrently on gfx950, or do they share the datapath?
// 512-thread workgroups = 2 waves per SIMD: waves 0-3 run a dependent MFMA chain, waves 4-7 run VALU work
// (fp32 FMAs, or integer/compare/select ops like the argmin tournament).
// build: hipcc --offload-arch=gfx950 -O3 tools/overlap_probe.hip -o gpurun_out/overlap_probe   (run on the GPU box)
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));

// mode bit 0: MFMA waves active, bit 1: VALU waves active; kind 0: v_fma_f32, 1: v_cmp+v_cndmask, 2: v_pk_fma_f32
__global__ __launch_bounds__(512) void probe(float *out, int n_mfma, int n_valu, int mode, int kind) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float s = 0.f;
    if (wave < 4) {
        if (mode & 1) {
            f32x16 acc = {0};
            float a = 0.5f + lane * 0.001f, b = 0.25f;
            for (int it = 0; it < n_mfma; ++it) {
#pragma unroll
                for (int k = 0; k < 16; ++k) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
            }
            for (int j = 0; j < 16; ++j) s += acc[j];
        }
    } else if (mode & 2) {
        float v[8];
        for (int j = 0; j < 8; ++j) v[j] = lane * 0.01f + j;
        float c = 1.0001f, d = 0.5f;
        if (kind == 0) {
            for (int it = 0; it < n_valu; ++it) {
#pragma unroll
                for (int r = 0; r < 8; ++r) {  // 64 independent-ish fma per iteration
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = __builtin_fmaf(v[j], c, d);
                }
            }
        } else if (kind == 1) {
            int idx[8] = {0, 1, 2, 3, 4, 5, 6, 7};
            for (int it = 0; it < n_valu; ++it) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {  // 4 x 8 x (cmp + 2 cndmask) = 96 instr, issue ~64 "slots" of 1.5
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int o = (j + 1 + r) & 7;
                        const bool t = v[o] < v[j];
                        asm volatile("" : "+v"(v[j]), "+v"(idx[j]));
                        v[j] = t ? v[o] + 1.0f : v[j];
                        idx[j] = t ? idx[o] : idx[j];
                    }
                }
            }
            for (int j = 0; j < 8; ++j) s += idx[j];
        } else {
            typedef float f32x2 __attribute__((ext_vector_type(2)));
            f32x2 w[4], cc = {c, c}, dd = {d, d};
            for (int j = 0; j < 4; ++j) w[j] = f32x2{v[2 * j], v[2 * j + 1]};
            for (int it = 0; it < n_valu; ++it) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {  // 64 packed fma = 128 fp32 fma per iteration
#pragma unroll
                    for (int j = 0; j < 4; ++j) w[j] = __builtin_elementwise_fma(w[j], cc, dd);
                }
            }
            for (int j = 0; j < 4; ++j) { v[2 * j] = w[j].x; v[2 * j + 1] = w[j].y; }
        }
        for (int j = 0; j < 8; ++j) s += v[j];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

static float run(float *out, int n_mfma, int n_valu, int mode, int kind) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    probe<<<256, 512>>>(out, n_mfma, n_valu, mode, kind);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    probe<<<256, 512>>>(out, n_mfma, n_valu, mode, kind);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3f;
}

int main() {
    float *out;
    hipMalloc(&out, 256 * 512 * 4);
    const int n_mfma = 2000;  // x16 MFMAs x 64 cycles = 2.05 M cycles
    const char *names[3] = {"v_fma_f32", "v_cmp+2 v_cndmask", "v_pk_fma_f32"};
    for (int kind = 0; kind < 3; ++kind) {
        for (int n_valu : {2000, 4000, 8000}) {
            const float a = run(out, n_mfma, n_valu, 1, kind);
            const float b = run(out, n_mfma, n_valu, 2, kind);
            const float c = run(out, n_mfma, n_valu, 3, kind);
            printf("%-18s n_valu %5d: MFMA only %8.1f us   VALU only %8.1f us   both %8.1f us   (sum %8.1f, max %8.1f)\n",
                   names[kind], n_valu, a, b, c, a + b, a > b ? a : b);
        }
    }
    return 0;
}
