// gumbel_mfma.h -- entry points of the matrix-instruction Gumbel-softmax kernels (gumbel_mfma.hip), called from the
// C ABI functions in gumbel.hip when the shape qualifies.
#pragma once
#include "rqhip_common.h"

namespace rqhip {

struct GumbelMfmaParams {
    const float *x, *cb, *U, *g_emb, *g_loss;
    int64_t *ids;
    float *emb, *loss, *g_x, *partial;
    long long B, n_tiles;
    int K;
    float temperature, beta;
};

// D == 32, K in {32, 64, 128, 256} and the given row pointers 16-byte aligned (NULL counts as aligned)
bool gumbel_mfma_supported(int D, int K, const void *x, const void *U, const void *a, const void *b);
int gumbel_mfma_forward(const GumbelMfmaParams &p, hipStream_t s);
// writes one [K,32] partial table per workgroup to p.partial (gumbel_mfma_backward_grid(B) of them) and g_x
int gumbel_mfma_backward_grid(long long B);
int gumbel_mfma_backward(const GumbelMfmaParams &p, hipStream_t s);
// rows from which the 32-rows-per-wave kernels beat the one-row-per-wave ones of gumbel.hip (fewer rows cannot fill
// the chip with 32-row tiles); rqhip_gumbel_matrix_path_min_rows overrides (developer / test switch)
long long gumbel_mfma_min_rows();
void gumbel_mfma_set_min_rows(long long n);

#ifdef __HIPCC__
// Softmax arithmetic of both Gumbel kernels (~150 -> ~50 VALU instructions per (row, code)):
//   * both logarithms of the Gumbel noise and the softmax exponential are the hardware v_log_f32 / v_exp_f32.  Measured
//     on MI355X over the whole 2^24-point grid of torch.rand (tools/log_probe.hip): t = -log(u) to 1.6e-7 relative
//     (also for u -> 1, where a sloppy inner log would be amplified by the outer one), the Gumbel value to 1.7e-6
//     absolute (|g| <= 17), exp to ~1e-7 relative above the denormal range (denormal results flush to 0);
//   * the divisions by the temperature and by the softmax sum are multiplications by one IEEE reciprocal per call /
//     per row.
// Results differ from the oracle's libm / division chain by a few 1e-7 relative (tests: rtol 2e-4 forward, 2e-3
// backward); ids come from the noise-free distances and are not affected.
// RQ_GUMBEL_LIBM (developer A/B switch, tools/gumbel_libm_ab.sh): the library's logf / expf (__ocml_log_f32 / __ocml_exp_f32,
// <= 1 ulp) in place of the hardware instructions -- what the tolerance would be bought with; measured in DESIGN.md section 4.3.
#ifndef RQ_GUMBEL_LIBM
#define RQ_GUMBEL_LIBM 0
#endif
__device__ __forceinline__ float gm_gumbel(float u) {   // -log(-log(u + 1e-20) + 1e-20), gumbel.py:10-11
#if RQ_GUMBEL_LIBM
    return -logf(-logf(u + 1e-20f) + 1e-20f);
#else
    const float t = -(__builtin_amdgcn_logf(u + 1e-20f) * 0.69314718055994530942f);
    return -(__builtin_amdgcn_logf(t + 1e-20f) * 0.69314718055994530942f);
#endif
}
__device__ __forceinline__ float gm_exp(float x) {
#if RQ_GUMBEL_LIBM
    return expf(x);
#else
    return __builtin_amdgcn_exp2f(x * 1.44269504088896340736f);
#endif
}
#endif

}  // namespace rqhip
