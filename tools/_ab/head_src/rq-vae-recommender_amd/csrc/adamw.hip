// adamw.hip -- the AdamW update of every parameter of the model in ONE launch (gfx950).  Reference: train_rqvae.py:136-138
// (`AdamW(params=model.parameters(), lr, weight_decay)`: decoupled weight decay on every parameter, codebooks included; SURVEY.md
// appendix A item 16).
//
// Why not torch's fused AdamW: its multi-tensor kernel hands 64 K-element chunks to workgroups -- 1.15 M parameters are 18 chunks, 18 of
// 256 CUs work, 40-46 us per step at every batch size (11 % of the 0.35 ms hipGraph step at batch 640, 1.6 % at 100 000 rows), plus a
// second launch that increments the per-parameter step counters.  Here: 1024 elements per workgroup (1 100 workgroups); the step counter is a
// device scalar (so a captured hipGraph advances it on replay), bumped by a one-thread kernel that also forms the bias corrections.
// The arithmetic is torch's `_fused_adamw_` (aten/src/ATen/native/cuda/fused_adam_utils.cuh, ADAMW mode, no amsgrad, no maximize), in fp32:
//     p  -= lr wd p
//     m   = lerp(m, g, 1 - b1)              (= m + (1 - b1) (g - m) for 1 - b1 < 0.5)
//     v   = b2 v + (1 - b2) g g
//     p  -= (lr / (1 - b1^t)) m / (sqrt(v) / sqrt(1 - b2^t) + eps),          t = step + 1
// -- the same update as the reference's (foreach) AdamW up to the rounding of each operation (tests/test_gpu_optim.py holds both to 1e-6).
#include <math.h>

#include "rqhip_common.h"

namespace rqhip {

typedef float aw_f32x4 __attribute__((ext_vector_type(4)));

constexpr int kAwMaxJobs = 24;
constexpr int kAwBlockElems = 1024;      // 256 threads x one float4

struct AdamwJob {
    float *p, *m, *v;
    const float *g;
    long long n;
    int block0;
};
struct AdamwJobs {
    AdamwJob j[kAwMaxJobs];
    int n;
};

// step += 1 and the scalars every element needs, once per step, by one thread -- so that the update kernel only READS them (a first version
// let every workgroup read the counter and the last one to finish bump it: 1 100 returning atomics on one word are 13 us by themselves, and
// every thread evaluated two powf)
struct AdamwScalars {
    float step_size, bc2_sqrt;
};
__global__ void adamw_bump_kernel(float *__restrict__ step, AdamwScalars *__restrict__ sc, float lr, float beta1, float beta2) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const float t = step[0] + 1.0f;
    step[0] = t;
    const float bc1 = 1.0f - powf(beta1, t), bc2 = 1.0f - powf(beta2, t);
    sc->step_size = lr / bc1;
    sc->bc2_sqrt = sqrtf(bc2);
}

__global__ __launch_bounds__(256) void adamw_kernel(const AdamwJobs jobs, const AdamwScalars *__restrict__ sc, float lr, float beta1, float beta2,
                                                    float eps, float wd) {
    AdamwJob job = jobs.j[0];
#pragma unroll
    for (int i = 1; i < kAwMaxJobs; ++i)
        if (i < jobs.n && (int)blockIdx.x >= jobs.j[i].block0) job = jobs.j[i];
    const float step_size = sc->step_size, bc2_sqrt = sc->bc2_sqrt;
    const long long i0 = ((long long)((int)blockIdx.x - job.block0) * 256 + threadIdx.x) * 4;
    auto one = [&](float &p, float &m, float &v, float g) {
        p = p - (lr * wd) * p;
        m = m + (1.0f - beta1) * (g - m);
        v = beta2 * v + ((1.0f - beta2) * g) * g;
        const float denom = sqrtf(v) / bc2_sqrt + eps;
        p = p - step_size * (m / denom);
    };
    if (i0 + 3 < job.n) {
        aw_f32x4 p = *reinterpret_cast<aw_f32x4 *>(job.p + i0), m = *reinterpret_cast<aw_f32x4 *>(job.m + i0);
        aw_f32x4 v = *reinterpret_cast<aw_f32x4 *>(job.v + i0);
        const aw_f32x4 g = *reinterpret_cast<const aw_f32x4 *>(job.g + i0);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float pk = p[k], mk = m[k], vk = v[k];
            one(pk, mk, vk, g[k]);
            p[k] = pk; m[k] = mk; v[k] = vk;
        }
        *reinterpret_cast<aw_f32x4 *>(job.p + i0) = p;
        *reinterpret_cast<aw_f32x4 *>(job.m + i0) = m;
        *reinterpret_cast<aw_f32x4 *>(job.v + i0) = v;
    } else {
        for (long long i = i0; i < job.n; ++i) one(job.p[i], job.m[i], job.v[i], job.g[i]);
    }
}

}  // namespace rqhip

using namespace rqhip;

// One AdamW step over n tensors: p[i] (updated in place), g[i], m[i], v[i] of numel[i] fp32 elements each (16-byte aligned, contiguous).
// `step`: device float scalar, the number of steps taken so far (incremented by the call); `scratch`: 8 device bytes the call may overwrite.
extern "C" int rqhip_adamw_step(float *const *p, const float *const *g, float *const *m, float *const *v, const int64_t *numel, int n,
                                float *step, unsigned *scratch, float lr, float beta1, float beta2, float eps, float weight_decay,
                                rqhip_stream_t stream) {
    if (n < 0 || !step || !scratch || (n > 0 && (!p || !g || !m || !v || !numel))) {
        set_error("adamw_step: null pointer");
        return RQHIP_EARG;
    }
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    auto al16 = [](const void *q) { return (reinterpret_cast<uintptr_t>(q) & 15u) == 0; };
    AdamwScalars *sc = reinterpret_cast<AdamwScalars *>(scratch);
    hipLaunchKernelGGL(adamw_bump_kernel, dim3(1), dim3(64), 0, s, step, sc, lr, beta1, beta2);
    RQ_CHECK_LAUNCH("adamw_bump_kernel");
    int first = 0;
    while (first < n) {      // launches of at most kAwMaxJobs tensors (empty tensors are skipped)
        AdamwJobs jobs;
        jobs.n = 0;
        int blocks = 0, i = first;
        for (; i < n && jobs.n < kAwMaxJobs; ++i) {
            if (numel[i] <= 0) continue;
            if (!p[i] || !g[i] || !m[i] || !v[i] || !al16(p[i]) || !al16(g[i]) || !al16(m[i]) || !al16(v[i])) {
                set_error("adamw_step: tensor %d: null or not 16-byte aligned", i);
                return RQHIP_EARG;
            }
            AdamwJob &j = jobs.j[jobs.n++];
            j.p = p[i]; j.g = g[i]; j.m = m[i]; j.v = v[i]; j.n = numel[i]; j.block0 = blocks;
            blocks += (int)((numel[i] + kAwBlockElems - 1) / kAwBlockElems);
        }
        if (jobs.n > 0) {
            hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)blocks), dim3(256), 0, s, jobs, sc, lr, beta1, beta2, eps, weight_decay);
            RQ_CHECK_LAUNCH("adamw_kernel");
        }
        first = i;
    }
    return RQHIP_OK;
}
