// mfma_probe.hip -- developer micro-benchmark: what does a dependent v_mfma_f32_32x32x2_f32 chain really cost
// on this chip, alone, with interleaved VALU, with LDS operand reads, with 1 or 2 waves per SIMD?
// build: hipcc --offload-arch=gfx950 -O3 tools/mfma_probe.hip -o gpurun_out/mfma_probe ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int VARIANT>
__global__ __launch_bounds__(512) void probe(float *out, int iters, unsigned long long *cyc) {
    __shared__ __attribute__((aligned(16))) float lds[8192];
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = 0.001f * (i & 63);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    float a = 0.5f + lane * 0.001f, b = 0.25f;
    f32x16 acc0 = {0}, acc1 = {0};
    float best = 1e30f;
    int bidx = 0;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (VARIANT == 0) {  // 16 dependent MFMAs
#pragma unroll
            for (int k = 0; k < 16; ++k) acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc0, 0, 0, 0);
        } else if (VARIANT == 1) {  // two independent chains of 8
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, acc1, 0, 0, 0);
            }
        } else if (VARIANT == 2) {  // 16 dependent MFMAs, operands from LDS (ds_read_b128 per 4)
            const f32x4 *l4 = reinterpret_cast<const f32x4 *>(lds);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x4 v = l4[(q * 64 + lane + it) & 2047];
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(v.x, b, acc0, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(v.y, b, acc0, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(v.z, b, acc0, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(v.w, b, acc0, 0, 0, 0);
            }
        } else if (VARIANT == 3) {  // fresh chain from zero each iteration + 16-element argmin epilogue after it
            f32x16 acc = {0};
#pragma unroll
            for (int k = 0; k < 16; ++k) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b + k, acc, 0, 0, 0);
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                float d = (a + j) - acc[j];
                if (d < best) { best = d; bidx = it * 16 + j; }
            }
        } else if (VARIANT == 5) {  // TWO tiles per iteration: chains alternate A,B (independent accumulators), then
                                    // both epilogues -- no instruction sits between two MFMAs on one accumulator
            f32x16 ca = {0}, cb = {0};
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                ca = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b + k, ca, 0, 0, 0);
                cb = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a + k, cb, 0, 0, 0);
            }
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                float d = (a + j) - ca[j];
                if (d < best) { best = d; bidx = it * 32 + j; }
                float e = (b + j) - cb[j];
                if (e < best) { best = e; bidx = it * 32 + 16 + j; }
            }
        } else if (VARIANT == 6) {  // as 5, but the epilogue of the PREVIOUS pair is interleaved between the
                                    // alternating MFMAs of this pair (distance between dependent MFMAs = 2)
            f32x16 ca = {0}, cb = {0};
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                ca = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b + k, ca, 0, 0, 0);
                {
                    float d = (a + k) - acc0[k];
                    if (d < best) { best = d; bidx = it * 32 + k; }
                }
                cb = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a + k, cb, 0, 0, 0);
                {
                    float e = (b + k) - acc1[k];
                    if (e < best) { best = e; bidx = it * 32 + 16 + k; }
                }
            }
            acc0 = ca;
            acc1 = cb;
        } else if (VARIANT == 4) {  // same work, software pipelined: chain(it+1) interleaved with epilogue(it)
            f32x16 acc = {0};
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b + j, acc, 0, 0, 0);
                float d = (a + j) - acc0[j];
                if (d < best) { best = d; bidx = it * 16 + j; }
            }
            acc0 = acc;
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = best + bidx;
    for (int j = 0; j < 16; ++j) s += acc0[j] + acc1[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int VARIANT>
void run(const char *name, int threads, int iters) {
    float *out;
    unsigned long long *cyc, hc = 0;
    hipMalloc(&out, 256 * 512 * 4);
    hipMalloc(&cyc, 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    probe<VARIANT><<<256, threads>>>(out, iters, cyc);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    probe<VARIANT><<<256, threads>>>(out, iters, cyc);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(&hc, cyc, 8, hipMemcpyDeviceToHost);
    const double mfma = (double)iters * 16 * (VARIANT >= 5 ? 2 : 1);
    const int waves_per_simd = threads / 256;
    printf("%-34s waves/SIMD %d: %8.1f us  memtime %10llu ticks  %7.1f ticks/MFMA/wave  wall-ns/MFMA/SIMD %6.2f  "
           "=> %6.1f TFLOP/s\n",
           name, waves_per_simd, ms * 1e3, hc, hc / mfma, ms * 1e6 / (mfma * waves_per_simd),
           256.0 * 4 * waves_per_simd * mfma * 4096 / (ms * 1e-3) / 1e12);
    hipFree(out);
    hipFree(cyc);
}

int main() {
    const int iters = 2000;
    for (int threads : {256, 512}) {
        run<0>("16 dependent", threads, iters);
        run<1>("2 chains x 8", threads, iters);
        run<2>("16 dependent, LDS operands", threads, iters);
        run<3>("chain + epilogue (serial)", threads, iters);
        run<4>("chain || epilogue (pipelined)", threads, iters);
        run<5>("2 tiles: AB chains, then epilogues", threads, iters / 2);
        run<6>("2 tiles: AB chains || prev epilogues", threads, iters / 2);
    }
    return 0;
}
