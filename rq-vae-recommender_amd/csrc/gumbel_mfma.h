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
// the chip with 32-row tiles); RQ_GUMBEL_MFMA_MIN_ROWS overrides (developer / test switch)
long long gumbel_mfma_min_rows();

}  // namespace rqhip
