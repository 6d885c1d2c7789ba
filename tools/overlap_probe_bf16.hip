// overlap_probe_bf16.hip -- companion of overlap_probe.hip: does a bf16 XDL MFMA chain (v_mfma_f32_32x32x16_bf16) in
// one wave overlap with plain VALU work of another wave on the same SIMD (gfx950)?  Also reports the chain's own rate.
// build: hipcc --offload-arch=gfx950 -O3 tools/overlap_probe_bf16.hip -o tools/overlap_probe_bf16.bin
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(512) void probe(float *out, int n_mfma, int n_valu, int mode, int dep) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float s = 0.f;
    if (wave < 4) {
        if (mode & 1) {
            f32x16 acc = {0}, acc2 = {0};
            bf16x8 a, b;
            for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(0.5f + lane * 0.001f + j); b[j] = (__bf16)0.25f; }
            for (int it = 0; it < n_mfma; ++it) {
                if (dep) {
#pragma unroll
                    for (int k = 0; k < 12; ++k) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
                } else {
#pragma unroll
                    for (int k = 0; k < 6; ++k) {
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
                        acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a, acc2, 0, 0, 0);
                    }
                }
            }
            for (int j = 0; j < 16; ++j) s += acc[j] + acc2[j];
        }
    } else if (mode & 2) {
        float v[8];
        for (int j = 0; j < 8; ++j) v[j] = lane * 0.01f + j;
        float c = 1.0001f, d = 0.5f;
        for (int it = 0; it < n_valu; ++it) {
#pragma unroll
            for (int r = 0; r < 8; ++r) {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = __builtin_fmaf(v[j], c, d);
            }
        }
        for (int j = 0; j < 8; ++j) s += v[j];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

static float run(float *out, int n_mfma, int n_valu, int mode, int dep) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    probe<<<256, 512>>>(out, n_mfma, n_valu, mode, dep);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    probe<<<256, 512>>>(out, n_mfma, n_valu, mode, dep);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3f;
}

int main() {
    float *out;
    (void)hipMalloc(&out, 256 * 512 * 4);
    const int n_mfma = 4000;  // x 12 MFMAs
    for (int dep = 1; dep >= 0; --dep) {
        for (int n_valu : {2000, 4000, 8000}) {
            const float a = run(out, n_mfma, n_valu, 1, dep);
            const float b = run(out, n_mfma, n_valu, 2, dep);
            const float c = run(out, n_mfma, n_valu, 3, dep);
            printf("%s bf16 chain, n_valu %5d: MFMA only %8.1f us (%.1f ns / MFMA / SIMD)   VALU only %8.1f us   both %8.1f us   "
                   "(sum %8.1f, max %8.1f)\n", dep ? "dependent  " : "2 x 6 indep", n_valu, a, a * 1e3 / (n_mfma * 12.0), b, c, a + b,
                   a > b ? a : b);
        }
    }
    return 0;
}
