// kmeans.hip -- the two data-parallel steps of one Lloyd iteration of the reference's codebook
// initialisation (init/kmeans.py), for gfx950.
//
//   kmeans_assign : kmeans.py:40-43.  The reference materialises a B x K x D difference tensor
//                   (655 MB at 20000 x 256 x 32); here a thread keeps its row in registers, centroids are
//                   broadcast from LDS four at a time, nothing but the int64 assignment is written.
//                   Distances use the direct-difference form (x-c)^2 summed with the oracle's two parity
//                   accumulators, and torch.min's scan rule -> bit-exact assignments.
//   kmeans_update : kmeans.py:44-59,68.  One wave per cluster walks the assignment vector in row order and
//                   adds its rows sequentially (deterministic, == oracle), divides by the count, leaves
//                   empty clusters alone (counts[k] = 0 tells the host to reseed, kmeans.py:50-54) and
//                   folds the convergence statistic max_k |c_new - c_old|^2 into one device scalar.
#include "rqhip_common.h"

namespace rqhip {

constexpr int kAssignThreads = 256;
constexpr int kAssignLdsFloats = 16 * 1024;  // 64 KiB of centroids per chunk

template <int DP>  // padded D (registers per thread)
__global__ __launch_bounds__(kAssignThreads) void kmeans_assign_kernel(const float *__restrict__ x, long long B,
                                                                       int D,
                                                                       const float *__restrict__ cent, int K,
                                                                       int Kc, int64_t *__restrict__ assign) {
    __shared__ __attribute__((aligned(16))) float cs[kAssignLdsFloats];
    const long long row = (long long)blockIdx.x * kAssignThreads + threadIdx.x;
    const bool ok = row < B;
    const long long rc = ok ? row : B - 1;
    float xr[DP];
#pragma unroll
    for (int d = 0; d < DP; ++d) xr[d] = (d < D) ? x[(size_t)rc * D + d] : 0.0f;

    float best = 0.0f;
    int bidx = 0;
    bool stop = false;  // a NaN has been taken: torch.min's scan breaks (ATen compare kernel)
    bool first = true;

    for (int kbase = 0; kbase < K; kbase += Kc) {
        const int kn = min(Kc, K - kbase);
        __syncthreads();
        for (int e = threadIdx.x; e < Kc * DP; e += kAssignThreads) {
            const int c = e / DP, d = e - c * DP;
            cs[e] = (c < kn && d < D) ? cent[(size_t)(kbase + c) * D + d] : 0.0f;
        }
        __syncthreads();
        for (int c0 = 0; c0 < kn; c0 += 4) {
            float a0[4] = {0.f, 0.f, 0.f, 0.f}, a1[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int d = 0; d < DP; d += 2) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float *cp = cs + (size_t)(c0 + u) * DP;  // rows beyond kn are zero padding (in bounds:
                    const float t0 = xr[d] - cp[d];                //  Kc is a multiple of 4)
                    const float t1 = xr[d + 1] - cp[d + 1];
                    a0[u] = a0[u] + t0 * t0;
                    a1[u] = a1[u] + t1 * t1;
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (c0 + u < kn) {
                    const float dist = a0[u] + a1[u];
                    const int k = kbase + c0 + u;
                    if (first) {
                        best = dist; bidx = k; first = false;
                        stop = dist != dist;
                    } else if (!stop && !(dist >= best)) {
                        best = dist; bidx = k;
                        stop = dist != dist;
                    }
                }
            }
        }
    }
    if (ok) assign[row] = bidx;
}

// one wave per cluster
__global__ __launch_bounds__(64) void kmeans_update_kernel(const float *__restrict__ x, long long B, int D,
                                                           const int64_t *__restrict__ assign, int K,
                                                           float *__restrict__ cent, int64_t *__restrict__ counts,
                                                           unsigned int *__restrict__ shift_bits) {
    const int k = blockIdx.x;
    const int lane = threadIdx.x;
    float acc0 = 0.0f, acc1 = 0.0f;  // d = lane, lane + 64
    const bool d0 = lane < D, d1 = lane + 64 < D;
    long long n = 0;
    for (long long base = 0; base < B; base += 64) {
        const long long i = base + lane;
        const bool m = (i < B) && (assign[i] == (int64_t)k);
        unsigned long long mask = __ballot(m);
        n += __builtin_popcountll(mask);
        while (mask) {  // rows in ascending order; up to 4 loads in flight, adds strictly in order
            long long j[4];
            int cnt = 0;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (mask) {
                    j[u] = base + __builtin_ctzll(mask);
                    mask &= mask - 1;
                    cnt = u + 1;
                } else {
                    j[u] = base;
                }
            }
            float v0[4], v1[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                v0[u] = d0 ? x[(size_t)j[u] * D + lane] : 0.0f;
                v1[u] = d1 ? x[(size_t)j[u] * D + lane + 64] : 0.0f;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (u < cnt) {
                    acc0 = acc0 + v0[u];
                    acc1 = acc1 + v1[u];
                }
            }
        }
    }
    if (lane == 0) counts[k] = n;
    if (n == 0) return;  // empty: centroid untouched, zero shift (the host reseeds it)
    const float fn = (float)n;
    float diff0 = 0.0f, diff1 = 0.0f;
    if (d0) {
        const float c = acc0 / fn;
        diff0 = c - cent[(size_t)k * D + lane];
        cent[(size_t)k * D + lane] = c;
    }
    if (d1) {
        const float c = acc1 / fn;
        diff1 = c - cent[(size_t)k * D + lane + 64];
        cent[(size_t)k * D + lane + 64] = c;
    }
    if (shift_bits) {
        // sumsq2(c_new - c_old): parity accumulators, features in ascending order (all lanes redundantly)
        float a0 = 0.0f, a1 = 0.0f;
        for (int d = 0; d < D; ++d) {
            const float v = (d < 64) ? __shfl(diff0, d, 64) : __shfl(diff1, d - 64, 64);
            const float pq = v * v;
            if (d & 1) a1 = a1 + pq; else a0 = a0 + pq;
        }
        const float sq = a0 + a1;
        if (lane == 0) atomicMax(shift_bits, __float_as_uint(sq) & 0x7fffffffu);  // NaN sorts above +Inf
    }
}

template <int DP>
static int launch_assign(const float *x, long long B, int D, const float *cent, int K, int64_t *assign,
                         hipStream_t s) {
    int Kc = (kAssignLdsFloats / DP) & ~3;
    const int Kpad = (K + 3) & ~3;
    if (Kc > Kpad) Kc = Kpad;
    const int grid = (int)((B + kAssignThreads - 1) / kAssignThreads);
    hipLaunchKernelGGL(kmeans_assign_kernel<DP>, dim3(grid), dim3(kAssignThreads), 0, s, x, B, D, cent, K, Kc,
                       assign);
    RQ_CHECK_LAUNCH("kmeans_assign_kernel");
    return 0;
}

}  // namespace rqhip

using namespace rqhip;

extern "C" int rqhip_kmeans_assign(const float *x, int64_t B, int D, const float *centroids, int K,
                                   int64_t *assign, rqhip_stream_t stream) {
    if (B < 0 || !centroids || (B > 0 && (!x || !assign))) {
        set_error("kmeans_assign: null pointer or negative B");
        return RQHIP_EARG;
    }
    if (D < 1 || D > 128 || K < 1) {
        set_error("kmeans_assign: unsupported shape D=%d K=%d (need 1<=D<=128, K>=1)", D, K);
        return RQHIP_EUNSUPPORTED;
    }
    if (B == 0) return RQHIP_OK;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (D <= 8) return launch_assign<8>(x, B, D, centroids, K, assign, s);
    if (D <= 16) return launch_assign<16>(x, B, D, centroids, K, assign, s);
    if (D <= 32) return launch_assign<32>(x, B, D, centroids, K, assign, s);
    if (D <= 64) return launch_assign<64>(x, B, D, centroids, K, assign, s);
    return launch_assign<128>(x, B, D, centroids, K, assign, s);
}

extern "C" int rqhip_kmeans_update(const float *x, int64_t B, int D, const int64_t *assign, int K,
                                   float *centroids, int64_t *counts, float *shift_sq_max,
                                   rqhip_stream_t stream) {
    if (B < 0 || !centroids || !counts || (B > 0 && (!x || !assign))) {
        set_error("kmeans_update: null pointer or negative B");
        return RQHIP_EARG;
    }
    if (D < 1 || D > 128 || K < 1) {
        set_error("kmeans_update: unsupported shape D=%d K=%d", D, K);
        return RQHIP_EUNSUPPORTED;
    }
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (shift_sq_max) RQ_RETURN_IF_HIP(hipMemsetAsync(shift_sq_max, 0, sizeof(float), s));
    hipLaunchKernelGGL(kmeans_update_kernel, dim3(K), dim3(64), 0, s, x, (long long)B, D, assign, K, centroids,
                       counts, reinterpret_cast<unsigned int *>(shift_sq_max));
    RQ_CHECK_LAUNCH("kmeans_update_kernel");
    return RQHIP_OK;
}
