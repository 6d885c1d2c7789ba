// kmeans.hip -- the two data-parallel steps of one Lloyd iteration of the reference's codebook
// initialisation (init/kmeans.py), for gfx950.
//
//   kmeans_assign : kmeans.py:40-43.  The reference materialises a B x K x D difference tensor
//                   (655 MB at 20000 x 256 x 32); here a thread keeps its row in registers, centroids are
//                   broadcast from LDS four at a time, nothing but the int64 assignment is written.
//                   Distances use the direct-difference form (x-c)^2 summed with the oracle's two parity
//                   accumulators, and torch.min's scan rule -> bit-exact assignments.
//   kmeans_update : kmeans.py:44-59,68.  One workgroup per cluster: its waves list the cluster's rows in parallel, one
//                   wave adds them sequentially in row order (deterministic, == oracle), divides by the count, leaves
//                   empty clusters alone (counts[k] = 0 tells the host to reseed, kmeans.py:50-54) and
//                   folds the convergence statistic max_k |c_new - c_old|^2 into one device scalar.
//
// rqhip_kmeans_lloyd runs a BATCH of Lloyd iterations without a host round trip: every iteration is three launches
// (assign, update, finalize) that look at a device flag first and return at once when an earlier iteration of the
// batch converged (max shift below the threshold, kmeans.py:68-69) or met an empty cluster -- whose reseed consumes the
// host's torch RNG stream (kmeans.py:50-54) and is therefore done by the host, which reads the 16-byte state once per
// batch instead of once per iteration (the per-iteration .tolist() sync and tensor bookkeeping used to cost ~30x the
// kernels' time).
#include "rqhip_common.h"

namespace rqhip {

// device state of a batched run: [0] stop flag (0 running, 1 converged, 2 an empty cluster needs the host),
// [1] iterations completed, [2] fp32 bits of the last iteration's max |c_new - c_old|^2, [3] reserved
constexpr int kStRunning = 0, kStConverged = 1, kStEmpty = 2;

constexpr int kAssignThreads = 256;
constexpr int kAssignLdsFloats = 16 * 1024;  // 64 KiB of centroids per chunk

// A workgroup takes 64 rows; thread (r = tid & 63, q = tid >> 6) scans the codes of quarter q of every staged chunk
// (four threads per row: 313 workgroups for 20 000 rows instead of 79 -- the first version left two thirds of the
// chip idle), the four candidates meet in LDS.  torch.min's scan rule: the first NaN distance wins if there is one,
// else the first index of the minimum; both survive the split because every quarter keeps its own first NaN index
// and its own first-index minimum, and the merge takes the lowest NaN index / the lexicographic (dist, index) minimum.
template <int DP>  // padded D (registers per thread)
__global__ __launch_bounds__(kAssignThreads) void kmeans_assign_kernel(const float *__restrict__ x, long long B,
                                                                       int D,
                                                                       const float *__restrict__ cent, int K,
                                                                       int Kc, int64_t *__restrict__ assign,
                                                                       int *__restrict__ state) {
    if (state) {
        if (state[0] != kStRunning) return;
        if (blockIdx.x == 0 && threadIdx.x == 0) state[2] = 0;   // this iteration's shift maximum starts at 0
    }
    __shared__ __attribute__((aligned(16))) float cs[kAssignLdsFloats];
    __shared__ float m_best[4][64];
    __shared__ int m_idx[4][64], m_nan[4][64];
    const int r = threadIdx.x & 63, q = threadIdx.x >> 6;
    const long long row = (long long)blockIdx.x * 64 + r;
    const bool ok = row < B;
    const long long rc = ok ? row : B - 1;
    float xr[DP];
#pragma unroll
    for (int d = 0; d < DP; ++d) xr[d] = (d < D) ? x[(size_t)rc * D + d] : 0.0f;

    float best = __builtin_inff();
    int bidx = 0x7fffffff, nanidx = 0x7fffffff;

    for (int kbase = 0; kbase < K; kbase += Kc) {
        const int kn = min(Kc, K - kbase);
        __syncthreads();
        for (int e = threadIdx.x; e < Kc * DP; e += kAssignThreads) {
            const int c = e / DP, d = e - c * DP;
            cs[e] = (c < kn && d < D) ? cent[(size_t)(kbase + c) * D + d] : 0.0f;
        }
        __syncthreads();
        // quarter q takes the groups of four codes c0 = 4 (4 g + q): ascending within the thread
        for (int c0 = 4 * q; c0 < kn; c0 += 16) {
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
                    if (dist != dist) {
                        if (k < nanidx) nanidx = k;
                    } else if (dist < best || (dist == best && k < bidx)) {
                        best = dist;
                        bidx = k;
                    }
                }
            }
        }
    }
    m_best[q][r] = best;
    m_idx[q][r] = bidx;
    m_nan[q][r] = nanidx;
    __syncthreads();
    if (q == 0 && ok) {
#pragma unroll
        for (int u = 1; u < 4; ++u) {
            const float ob = m_best[u][r];
            const int oi = m_idx[u][r], on = m_nan[u][r];
            if (on < nanidx) nanidx = on;
            if (ob < best || (ob == best && oi < bidx)) {
                best = ob;
                bidx = oi;
            }
        }
        // (every distance +Inf: torch.min keeps index 0)
        assign[row] = nanidx != 0x7fffffff ? nanidx : (bidx != 0x7fffffff ? bidx : 0);
    }
}

// One workgroup per cluster, two phases.
//   list: the workgroup's waves split the assignment vector into contiguous ranges and write the rows of cluster k they
//         find, in ascending order, to per-wave lists in LDS (a parallel scan: 1/16 of the vector per wave -- the first
//         version was ONE wave per cluster walking all B assignments, O(K B) and latency-bound: 60 us per iteration at
//         20 000 x 256, 13 % of all GPU time of a bench run);
//   sum : wave 0 adds the listed rows strictly in ascending row order (the ranges are ascending, so the concatenation
//         of the lists is), sixteen row loads in flight -- the same sequential sum as before, bit for bit (== oracle).
// A wave whose range holds more rows of the cluster than its list takes (collapsed clusterings) raises a flag and wave 0
// falls back to the old single-wave walk for this cluster.
constexpr int kUpdThreads = 1024, kUpdWaves = kUpdThreads / 64, kUpdCap = 512;   // 16 x 512 x 4 B = 32 KiB of lists

__global__ __launch_bounds__(kUpdThreads) void kmeans_update_kernel(const float *__restrict__ x, long long B, int D,
                                                                    const int64_t *__restrict__ assign, int K,
                                                                    float *__restrict__ cent, int64_t *__restrict__ counts,
                                                                    unsigned int *__restrict__ shift_bits,
                                                                    const int *__restrict__ state,
                                                                    float *__restrict__ sums) {
    if (state && state[0] != kStRunning) return;
    __shared__ int list[kUpdWaves][kUpdCap];
    __shared__ int n_w[kUpdWaves];
    __shared__ int overflow;
    const int k = blockIdx.x;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (threadIdx.x == 0) overflow = B > 0x7fffffffLL;   // (the lists hold 32-bit row numbers)
    __syncthreads();
    // ---- list ---------------------------------------------------------------------------------------------------------
    {
        const long long batches = (B + 63) / 64, per = (batches + kUpdWaves - 1) / kUpdWaves;
        const long long b0 = wave * per, b1 = (b0 + per < batches) ? b0 + per : batches;
        int n = 0;
        for (long long bb = b0; bb < b1; bb += 4) {
            int64_t a4[4];   // four 64-row batches of assignments in flight
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const long long i = (bb + u) * 64 + lane;
                a4[u] = (bb + u < b1 && i < B) ? assign[i] : (int64_t)-1;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const bool m = a4[u] == (int64_t)k;
                const unsigned long long mask = __ballot(m);
                const int before = __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
                if (m && n + before < kUpdCap) list[wave][n + before] = (int)((bb + u) * 64 + lane);
                n += __builtin_popcountll(mask);
            }
        }
        if (lane == 0) {
            n_w[wave] = n;
            if (n > kUpdCap) overflow = 1;
        }
    }
    __syncthreads();
    if (wave != 0) return;
    // ---- sum (wave 0): lane = feature d and d + 64 ---------------------------------------------------------------------
    float acc0 = 0.0f, acc1 = 0.0f;  // d = lane, lane + 64
    const bool d0 = lane < D, d1 = lane + 64 < D;
    long long n = 0;
    if (!overflow) {
        for (int w = 0; w < kUpdWaves; ++w) {
            const int nw = n_w[w];
            n += nw;
            for (int i0 = 0; i0 < nw; i0 += 16) {
                float v0[16], v1[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    const int j = list[w][(i0 + u < nw) ? i0 + u : i0];
                    v0[u] = d0 ? x[(size_t)j * D + lane] : 0.0f;
                    v1[u] = d1 ? x[(size_t)j * D + lane + 64] : 0.0f;
                }
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    if (i0 + u < nw) {
                        acc0 = acc0 + v0[u];
                        acc1 = acc1 + v1[u];
                    }
                }
            }
        }
    } else {
    for (long long base4 = 0; base4 < B; base4 += 256) {
      // the assignments of four 64-row batches are fetched together (the scan is latency-bound otherwise)
      int64_t a4[4];
#pragma unroll
      for (int u4 = 0; u4 < 4; ++u4) {
          const long long i = base4 + 64 * u4 + lane;
          a4[u4] = (i < B) ? assign[i] : (int64_t)-1;
      }
#pragma unroll
      for (int u4 = 0; u4 < 4; ++u4) {
        const long long base = base4 + 64 * u4;
        const bool m = a4[u4] == (int64_t)k;
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
    }
    }
    if (sums) {  // row-sharded run: this rank's per-cluster sums and count, [K, D+1]; the means are formed after the
        float *o = sums + (size_t)k * (D + 1);   // all-reduce by kmeans_apply_sums_kernel
        if (d0) o[lane] = acc0;
        if (d1) o[lane + 64] = acc1;
        if (lane == 0) o[D] = (float)n;
        // (a rank without rows launches no assign kernel: start this iteration's shift maximum here as well)
        if (k == 0 && lane == 0 && state) const_cast<int *>(state)[2] = 0;
        return;
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

// Row-sharded k-means: after the all-reduce of [K, D+1] (sums || counts) every rank forms the
// same means; one wave per cluster, same arithmetic as the tail of kmeans_update_kernel.
__global__ __launch_bounds__(64) void kmeans_apply_sums_kernel(const float *__restrict__ sums, int K, int D,
                                                               float *__restrict__ cent, int64_t *__restrict__ counts,
                                                               unsigned int *__restrict__ shift_bits,
                                                               const int *__restrict__ state) {
    if (state && state[0] != kStRunning) return;
    const int k = blockIdx.x, lane = threadIdx.x;
    const float *src = sums + (size_t)k * (D + 1);
    const float fn = src[D];
    if (lane == 0) counts[k] = (int64_t)fn;
    if (!(fn > 0.0f)) return;
    const bool d0 = lane < D, d1 = lane + 64 < D;
    float diff0 = 0.0f, diff1 = 0.0f;
    if (d0) {
        const float c = src[lane] / fn;
        diff0 = c - cent[(size_t)k * D + lane];
        cent[(size_t)k * D + lane] = c;
    }
    if (d1) {
        const float c = src[lane + 64] / fn;
        diff1 = c - cent[(size_t)k * D + lane + 64];
        cent[(size_t)k * D + lane + 64] = c;
    }
    float a0 = 0.0f, a1 = 0.0f;
    for (int d = 0; d < D; ++d) {
        const float v = (d < 64) ? __shfl(diff0, d, 64) : __shfl(diff1, d - 64, 64);
        const float pq = v * v;
        if (d & 1) a1 = a1 + pq; else a0 = a0 + pq;
    }
    if (lane == 0) atomicMax(shift_bits, __float_as_uint(a0 + a1) & 0x7fffffffu);
}

// end of one batched iteration: count it, raise the stop flag on an empty cluster or on convergence (kmeans.py:68-69:
// torch.norm(...).max() < threshold, i.e. sqrt(max |dc|^2) < threshold in fp32)
__global__ __launch_bounds__(256) void kmeans_finalize_kernel(const int64_t *__restrict__ counts, int K,
                                                              float stop_threshold, int *__restrict__ state) {
    if (state[0] != kStRunning) return;
    __shared__ int any_empty;
    if (threadIdx.x == 0) any_empty = 0;
    __syncthreads();
    for (int k = threadIdx.x; k < K; k += 256)
        if (counts[k] == 0) any_empty = 1;
    __syncthreads();
    if (threadIdx.x == 0) {
        state[1] = state[1] + 1;
        const float shift = __builtin_sqrtf(__uint_as_float((unsigned)state[2]));
        if (any_empty) state[0] = kStEmpty;
        else if (shift < stop_threshold) state[0] = kStConverged;   // a NaN shift never converges, like the reference
    }
}

template <int DP>
static int launch_assign(const float *x, long long B, int D, const float *cent, int K, int64_t *assign,
                         int *state, hipStream_t s) {
    int Kc = (kAssignLdsFloats / DP) & ~15;   // groups of four codes per thread quarter: a multiple of 16
    const int Kpad = (K + 15) & ~15;
    if (Kc > Kpad) Kc = Kpad;
    const int grid = (int)((B + 63) / 64);
    hipLaunchKernelGGL(kmeans_assign_kernel<DP>, dim3(grid), dim3(kAssignThreads), 0, s, x, B, D, cent, K, Kc,
                       assign, state);
    RQ_CHECK_LAUNCH("kmeans_assign_kernel");
    return 0;
}

static int assign_dispatch(const float *x, long long B, int D, const float *cent, int K, int64_t *assign, int *state,
                           hipStream_t s) {
    if (D <= 8) return launch_assign<8>(x, B, D, cent, K, assign, state, s);
    if (D <= 16) return launch_assign<16>(x, B, D, cent, K, assign, state, s);
    if (D <= 32) return launch_assign<32>(x, B, D, cent, K, assign, state, s);
    if (D <= 64) return launch_assign<64>(x, B, D, cent, K, assign, state, s);
    return launch_assign<128>(x, B, D, cent, K, assign, state, s);
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
    return assign_dispatch(x, B, D, centroids, K, assign, nullptr, s);
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
    if (shift_sq_max)
        if (int rc = fill_words(shift_sq_max, 0u, sizeof(float), s)) return rc;
    hipLaunchKernelGGL(kmeans_update_kernel, dim3(K), dim3(kUpdThreads), 0, s, x, (long long)B, D, assign, K, centroids,
                       counts, reinterpret_cast<unsigned int *>(shift_sq_max), (const int *)nullptr, (float *)nullptr);
    RQ_CHECK_LAUNCH("kmeans_update_kernel");
    return RQHIP_OK;
}

extern "C" int rqhip_kmeans_partial_sums(const float *x, int64_t B, int D, const float *centroids, int K,
                                         int64_t *assign, float *sums, int *state, rqhip_stream_t stream) {
    if (B < 0 || !centroids || !sums || !state || (B > 0 && (!x || !assign))) {
        set_error("kmeans_partial_sums: null pointer or negative B");
        return RQHIP_EARG;
    }
    if (D < 1 || D > 128 || K < 1) {
        set_error("kmeans_partial_sums: unsupported shape D=%d K=%d", D, K);
        return RQHIP_EUNSUPPORTED;
    }
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (B > 0) {
        int rc = assign_dispatch(x, B, D, centroids, K, assign, state, s);
        if (rc) return rc;
    }
    // (a rank without rows still contributes zeros; the early-exit flag keeps stale sums from mattering)
    hipLaunchKernelGGL(kmeans_update_kernel, dim3(K), dim3(kUpdThreads), 0, s, x, (long long)B, D, assign, K,
                       (float *)nullptr, (int64_t *)nullptr, (unsigned int *)nullptr, (const int *)state, sums);
    RQ_CHECK_LAUNCH("kmeans_update_kernel");
    return RQHIP_OK;
}

extern "C" int rqhip_kmeans_apply_sums(const float *sums, int K, int D, float *centroids, int64_t *counts, int *state,
                                       float stop_threshold, rqhip_stream_t stream) {
    if (!sums || !centroids || !counts || !state || K < 1 || D < 1 || D > 128) {
        set_error("kmeans_apply_sums: bad arguments");
        return RQHIP_EARG;
    }
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(kmeans_apply_sums_kernel, dim3(K), dim3(64), 0, s, sums, K, D, centroids, counts,
                       reinterpret_cast<unsigned int *>(state + 2), (const int *)state);
    RQ_CHECK_LAUNCH("kmeans_apply_sums_kernel");
    hipLaunchKernelGGL(kmeans_finalize_kernel, dim3(1), dim3(256), 0, s, counts, K, stop_threshold, state);
    RQ_CHECK_LAUNCH("kmeans_finalize_kernel");
    return RQHIP_OK;
}

extern "C" int rqhip_kmeans_lloyd(const float *x, int64_t B, int D, float *centroids, int K, int64_t *assign,
                                  int64_t *counts, int *state, int n_iters, float stop_threshold,
                                  rqhip_stream_t stream) {
    if (B <= 0 || !x || !centroids || !assign || !counts || !state || n_iters < 0) {
        set_error("kmeans_lloyd: null pointer, empty x or negative iteration count");
        return RQHIP_EARG;
    }
    if (D < 1 || D > 128 || K < 1) {
        set_error("kmeans_lloyd: unsupported shape D=%d K=%d", D, K);
        return RQHIP_EUNSUPPORTED;
    }
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    for (int it = 0; it < n_iters; ++it) {
        int rc = assign_dispatch(x, B, D, centroids, K, assign, state, s);
        if (rc) return rc;
        hipLaunchKernelGGL(kmeans_update_kernel, dim3(K), dim3(kUpdThreads), 0, s, x, (long long)B, D, assign, K, centroids,
                           counts, reinterpret_cast<unsigned int *>(state + 2), (const int *)state, (float *)nullptr);
        RQ_CHECK_LAUNCH("kmeans_update_kernel");
        hipLaunchKernelGGL(kmeans_finalize_kernel, dim3(1), dim3(256), 0, s, counts, K, stop_threshold, state);
        RQ_CHECK_LAUNCH("kmeans_finalize_kernel");
    }
    return RQHIP_OK;
}
