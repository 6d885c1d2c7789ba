// gumbel.hip -- one Gumbel-softmax quantisation level, forward and backward, for gfx950.
//
// Replaces the training branch of the reference's Quantize.forward for QuantizeForwardMode.GUMBEL_SOFTMAX
// (modules/quantize.py:112-117,128,131-136,157 with distributions/gumbel.py:8-20) and what autograd derives
// from it (closed form in oracle/rq_oracle.c:rqo_gumbel_backward).  The reference materialises dist, the
// noise, the logits, the weights [B,K] and runs two extra GEMMs; here a wave owns a row: the K distances,
// logits and weights live in registers (K/64 per lane), the codebook sits transposed in LDS ([d][k], row
// stride K+1: conflict-free both for lanes-over-k and lanes-over-d), and the only HBM traffic is x, the
// uniform noise U (the dominant 4K bytes per row) and the outputs.  The dense codebook gradient (every row
// touches every code) is accumulated in an LDS table per workgroup and reduced across workgroups in fixed
// order.  This file is the general one-row-per-wave implementation (VALU FMAs); for the shipped latent width
// (D == 32, K <= 256) and batches of >= 4096 rows the entry points below hand over to gumbel_mfma.hip.
//
// Limits: D <= 128 and the LDS footprint below must fit 160 KiB (K*D <= ~16k floats for backward);
// otherwise RQHIP_EUNSUPPORTED.  Transcendentals are the hardware v_log_f32 / v_exp_f32 (gumbel_mfma.h): results
// match the oracle to ~1e-6 relative, not bit for bit; ids (noise-free argmin) are exact.
#include "gumbel_mfma.h"
#include "rqhip_common.h"

namespace rqhip {

constexpr int kGThreads = 256;
constexpr int kGWaves = kGThreads / 64;
constexpr int kGMaxPerLane = 16;  // K <= 1024

struct GumbelParams {
    const float *x, *cb, *U, *g_emb, *g_loss;
    int64_t *ids;
    float *emb, *loss, *g_x, *partial;
    long long B;
    int D, K, Kpad;  // Kpad = K rounded up to 64
    float temperature, beta;
};

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) v = fmaxf(v, __shfl_xor(v, m, 64));
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) v = v + __shfl_xor(v, m, 64);
    return v;
}

// LDS carve (floats): Ct [D][K+1] | csq [Kpad] | per wave: w [Kpad], dd [Kpad], xs [128], ge [128], es [128]
// backward adds: gC [K][D+1]
__host__ __device__ inline size_t gumbel_lds_floats(int D, int K, int Kpad, bool backward) {
    size_t n = (size_t)D * (K + 1) + Kpad + (size_t)kGWaves * (2 * (size_t)Kpad + 3 * 128);
    if (backward) n += (size_t)K * (D + 1);
    return n;
}

// REGACC (backward, K <= 256, D <= 32): the dense codebook gradient -- every row touches every code -- is summed
// in registers, lane owns codes lane + 64 g (g < 4) x all 32 features = 128 accumulators, and only meets the other
// three waves of the workgroup in LDS once, after the row loop.  The first version did one ds_add_f32 per
// (row, code, feature): 819 M LDS float atomics for 100 000 rows at ~2 cycles per lane and CU = 6 ms per level.
constexpr int kGAccPerLane = 4, kGAccD = 32;

template <bool BACKWARD, bool REGACC>
__global__ __launch_bounds__(kGThreads) void gumbel_kernel(const GumbelParams p) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int D = p.D, K = p.K, Kpad = p.Kpad, KS = K + 1;
    const float inv_t = 1.0f / p.temperature;
    float *Ct = sm;
    float *csq = Ct + (size_t)D * KS;
    float *wave_base = csq + Kpad;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float *ws = wave_base + (size_t)wave * (2 * Kpad + 3 * 128);
    float *dds = ws + Kpad, *xs = dds + Kpad, *ges = xs + 128, *es = ges + 128;
    float *gC = wave_base + (size_t)kGWaves * (2 * Kpad + 3 * 128);  // [K][D+1], backward only

    for (int e = threadIdx.x; e < K * D; e += kGThreads) {
        const int k = e / D, d = e - k * D;
        Ct[(size_t)d * KS + k] = p.cb[e];
    }
    if (BACKWARD && !REGACC)
        for (int e = threadIdx.x; e < K * (D + 1); e += kGThreads) gC[e] = 0.0f;
    __syncthreads();
    float gacc[REGACC ? kGAccPerLane : 1][REGACC ? kGAccD : 1];
#pragma unroll
    for (int g = 0; g < (REGACC ? kGAccPerLane : 1); ++g)
#pragma unroll
        for (int d = 0; d < (REGACC ? kGAccD : 1); ++d) gacc[g][d] = 0.0f;
    for (int k = threadIdx.x; k < Kpad; k += kGThreads) {
        float v = __builtin_inff();
        if (k < K) {  // sumsq2 of code k (parity accumulators)
            float a0 = 0.0f, a1 = 0.0f;
            for (int d = 0; d < D; ++d) {
                const float c = Ct[(size_t)d * KS + k];
                const float q = c * c;
                if (d & 1) a1 = a1 + q; else a0 = a0 + q;
            }
            v = a0 + a1;
        }
        csq[k] = v;
    }
    __syncthreads();

    const int per_lane = Kpad / 64;
    const long long gw = (long long)blockIdx.x * kGWaves + wave, nw = (long long)gridDim.x * kGWaves;
    for (long long row = gw; row < p.B; row += nw) {
        // row into LDS (broadcast source) -- lanes cover d and d + 64
        if (lane < D) xs[lane] = p.x[(size_t)row * D + lane];
        if (lane + 64 < D) xs[lane + 64] = p.x[(size_t)row * D + lane + 64];
        __builtin_amdgcn_wave_barrier();
        float a0 = 0.0f, a1 = 0.0f;
#pragma unroll 8
        for (int d = 0; d < D; ++d) {
            const float v = xs[d];
            const float q = v * v;
            if (d & 1) a1 = a1 + q; else a0 = a0 + q;
        }
        const float xsq = a0 + a1;

        float dist[kGMaxPerLane], y[kGMaxPerLane];
        float lbest = __builtin_inff();
        int lidx = 0x7fffffff, nanidx = 0x7fffffff;
#pragma unroll
        for (int g = 0; g < kGMaxPerLane; ++g) {
            if (g < per_lane) {
                const int k = lane + 64 * g;
                float acc = 0.0f;
                const float *ck = Ct + (k < K ? k : 0);
#pragma unroll 8
                for (int d = 0; d < D; ++d) acc = __builtin_fmaf(xs[d], ck[(size_t)d * KS], acc);
                const float t = xsq + csq[k];
                const float dv = (k < K) ? t - 2.0f * acc : __builtin_inff();
                dist[g] = dv;
                if (k < K) {
                    if (dv != dv) nanidx = min(nanidx, k);
                    else if (dv < lbest || (dv == lbest && k < lidx)) { lbest = dv; lidx = k; }
                }
            }
        }
        // noise-free argmin with torch.min semantics (first NaN wins, else first minimum)
#pragma unroll
        for (int m = 1; m < 64; m <<= 1) {
            const int on = __shfl_xor(nanidx, m, 64);
            const float ob = __shfl_xor(lbest, m, 64);
            const int oi = __shfl_xor(lidx, m, 64);
            nanidx = min(nanidx, on);
            if (ob < lbest || (ob == lbest && oi < lidx)) { lbest = ob; lidx = oi; }
        }
        const int id = nanidx != 0x7fffffff ? nanidx : lidx;

        // gumbel logits, softmax over k (gumbel.py:11,18-19)
        float mx = -__builtin_inff();
#pragma unroll
        for (int g = 0; g < kGMaxPerLane; ++g) {
            if (g < per_lane) {
                const int k = lane + 64 * g;
                float yy = -__builtin_inff();
                if (k < K) {
                    const float u = p.U[(size_t)row * K + k];
                    const float gn = gm_gumbel(u);
                    yy = ((-dist[g]) + gn) * inv_t;
                }
                y[g] = yy;
                mx = fmaxf(mx, yy);
            }
        }
        mx = wave_max(mx);
        float zs = 0.0f;
#pragma unroll
        for (int g = 0; g < kGMaxPerLane; ++g) {
            if (g < per_lane) {
                const int k = lane + 64 * g;
                const float ev = (k < K) ? gm_exp(y[g] - mx) : 0.0f;
                y[g] = ev;
                zs = zs + ev;
            }
        }
        const float Z = wave_sum(zs);
        const float rz = 1.0f / Z;
#pragma unroll
        for (int g = 0; g < kGMaxPerLane; ++g) {
            if (g < per_lane) {
                y[g] = y[g] * rz;  // weights
                ws[lane + 64 * g] = y[g];
            }
        }
        __builtin_amdgcn_wave_barrier();

        // emb = w @ C: lanes over d (and d + 64), chain over k ascending
        float e0 = 0.0f, e1 = 0.0f;
        {
            const float *c0 = Ct + (size_t)(lane < D ? lane : 0) * KS;
            const float *c1 = Ct + (size_t)(lane + 64 < D ? lane + 64 : 0) * KS;
#pragma unroll 8
            for (int k = 0; k < K; ++k) {
                const float wk = ws[k];
                e0 = __builtin_fmaf(wk, c0[k], e0);
                e1 = __builtin_fmaf(wk, c1[k], e1);
            }
        }
        if (lane < D) es[lane] = e0;
        if (lane + 64 < D) es[lane + 64] = e1;
        __builtin_amdgcn_wave_barrier();
        // quantize loss on (x, emb): parity accumulators over d, every lane redundantly
        float s0 = 0.0f, s1 = 0.0f;
#pragma unroll 8
        for (int d = 0; d < D; ++d) {
            const float df = xs[d] - es[d];
            const float q = df * df;
            if (d & 1) s1 = s1 + q; else s0 = s0 + q;
        }
        const float ssum = s0 + s1;

        if (!BACKWARD) {
            if (lane < D) p.emb[(size_t)row * D + lane] = e0;
            if (lane + 64 < D) p.emb[(size_t)row * D + lane + 64] = e1;
            if (lane == 0) {
                p.ids[row] = id;
                p.loss[row] = ssum + p.beta * ssum;
            }
        } else {
            const float gl = p.g_loss ? p.g_loss[row] : 0.0f;
            // ge = g_emb + 2 (emb - x) gl
            if (lane < D) {
                const float ga = p.g_emb ? p.g_emb[(size_t)row * D + lane] : 0.0f;
                ges[lane] = ga + (2.0f * (e0 - xs[lane])) * gl;
            }
            if (lane + 64 < D) {
                const float ga = p.g_emb ? p.g_emb[(size_t)row * D + lane + 64] : 0.0f;
                ges[lane + 64] = ga + (2.0f * (e1 - xs[lane + 64])) * gl;
            }
            __builtin_amdgcn_wave_barrier();
            // dw_k = ge . C_k ; sw = sum_k w_k dw_k
            float dw[kGMaxPerLane];
            float swl = 0.0f;
#pragma unroll
            for (int g = 0; g < kGMaxPerLane; ++g) {
                if (g < per_lane) {
                    const int k = lane + 64 * g;
                    float acc = 0.0f;
                    const float *ck = Ct + (k < K ? k : 0);
#pragma unroll 8
                    for (int d = 0; d < D; ++d) acc = __builtin_fmaf(ges[d], ck[(size_t)d * KS], acc);
                    dw[g] = (k < K) ? acc : 0.0f;
                    swl = __builtin_fmaf(y[g], dw[g], swl);
                }
            }
            const float sw = wave_sum(swl);
            float sddl = 0.0f;
#pragma unroll
            for (int g = 0; g < kGMaxPerLane; ++g) {
                if (g < per_lane) {
                    const int k = lane + 64 * g;
                    const float dy = (y[g] * (dw[g] - sw)) * inv_t;
                    const float ddk = (k < K) ? -dy : 0.0f;
                    dw[g] = ddk;
                    dds[k] = ddk;
                    sddl = sddl + ddk;
                }
            }
            const float sdd = wave_sum(sddl);
            __builtin_amdgcn_wave_barrier();
            // g_x: lanes over d
            {
                const float *c0 = Ct + (size_t)(lane < D ? lane : 0) * KS;
                const float *c1 = Ct + (size_t)(lane + 64 < D ? lane + 64 : 0) * KS;
                float g0 = (lane < D) ? (2.0f * xs[lane]) * sdd : 0.0f;
                float g1 = (lane + 64 < D) ? (2.0f * xs[lane + 64]) * sdd : 0.0f;
#pragma unroll 8
                for (int k = 0; k < K; ++k) {
                    const float m2 = -2.0f * dds[k];
                    g0 = __builtin_fmaf(m2, c0[k], g0);
                    g1 = __builtin_fmaf(m2, c1[k], g1);
                }
                if (lane < D) p.g_x[(size_t)row * D + lane] = g0 + ((2.0f * p.beta) * (xs[lane] - e0)) * gl;
                if (lane + 64 < D)
                    p.g_x[(size_t)row * D + lane + 64] = g1 + ((2.0f * p.beta) * (xs[lane + 64] - e1)) * gl;
            }
            // dense codebook gradient: gC[k][d] += w_k ge_d + 2 dd_k (C[k][d] - x_d), lanes over k
            if (REGACC) {
#pragma unroll
                for (int d = 0; d < kGAccD; ++d) {
                    if (d < D) {
                        const float xd = xs[d], gd = ges[d];
#pragma unroll
                        for (int g = 0; g < kGAccPerLane; ++g) {
                            const int k = lane + 64 * g;
                            if (g < per_lane && k < K) {
                                const float v = __builtin_fmaf(2.0f * dw[g], Ct[(size_t)d * KS + k] - xd, y[g] * gd);
                                gacc[g][d] = gacc[g][d] + v;
                            }
                        }
                    }
                }
            } else {
#pragma unroll
                for (int g = 0; g < kGMaxPerLane; ++g) {
                    if (g < per_lane) {
                        const int k = lane + 64 * g;
                        if (k < K) {
                            const float wk = y[g], d2 = 2.0f * dw[g];
                            float *row_g = gC + (size_t)k * (D + 1);
                            for (int d = 0; d < D; ++d) {
                                const float v = __builtin_fmaf(d2, Ct[(size_t)d * KS + k] - xs[d], wk * ges[d]);
                                atomicAdd(row_g + d, v);  // ds_add_f32: 4 waves share the table
                            }
                        }
                    }
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
    }

    if (BACKWARD && REGACC) {
        // the four waves' register tables meet in the workgroup's LDS table, one wave at a time (plain stores and
        // read-modify-writes between barriers, wave order fixed: no atomics, deterministic)
        for (int w = 0; w < kGWaves; ++w) {
            __syncthreads();
            if (wave == w) {
#pragma unroll
                for (int g = 0; g < kGAccPerLane; ++g) {
                    const int k = lane + 64 * g;
                    if (k < K) {
#pragma unroll
                        for (int d = 0; d < kGAccD; ++d)
                            if (d < D) {
                                float *cell = gC + (size_t)k * (D + 1) + d;
                                *cell = (w == 0) ? gacc[g][d] : *cell + gacc[g][d];
                            }
                    }
                }
            }
        }
    }
    if (BACKWARD) {
        __syncthreads();
        float *out = p.partial + (size_t)blockIdx.x * K * D;
        for (int e = threadIdx.x; e < K * D; e += kGThreads) {
            const int k = e / D, d = e - k * D;
            out[e] = gC[(size_t)k * (D + 1) + d];
        }
    }
}

__global__ void gumbel_reduce_kernel(const float *__restrict__ partial, int G, int n, float *__restrict__ out) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    float s = 0.0f;
    for (int g = 0; g < G; ++g) s = s + partial[(size_t)g * n + j];
    out[j] = s;
}

static int gumbel_grid(long long B) {
    long long want = (B + kGWaves - 1) / kGWaves;
    long long cap = (long long)cu_count() * 2;
    if (want < 1) want = 1;
    return (int)(want < cap ? want : cap);
}

static int check_shape(const char *who, int64_t B, int D, int K, bool backward) {
    if (B < 0 || D < 1 || K < 1) {
        set_error("%s: bad shape B=%lld D=%d K=%d", who, (long long)B, D, K);
        return RQHIP_EARG;
    }
    const int Kpad = (K + 63) & ~63;
    if (D > 128 || Kpad > 64 * kGMaxPerLane ||
        gumbel_lds_floats(D, K, Kpad, backward) * sizeof(float) > 160 * 1024) {
        set_error("%s: D=%d K=%d exceeds what the Gumbel kernels keep on chip (D<=128, K<=1024, LDS %zu B > 160 KiB)",
                  who, D, K, gumbel_lds_floats(D, K, Kpad, backward) * sizeof(float));
        return RQHIP_EUNSUPPORTED;
    }
    return 0;
}

}  // namespace rqhip

using namespace rqhip;

extern "C" int64_t rqhip_gumbel_matrix_path_min_rows(int64_t set_to) {
    const int64_t before = (int64_t)gumbel_mfma_min_rows();
    if (set_to > 0) gumbel_mfma_set_min_rows((long long)set_to);
    return before;
}

extern "C" int rqhip_gumbel_forward(const float *x, int64_t B, int D, const float *codebook, int K, const float *U,
                                    float temperature, float beta, int64_t *ids, float *emb, float *loss,
                                    rqhip_stream_t stream) {
    int rc = check_shape("gumbel_forward", B, D, K, false);
    if (rc) return rc;
    if (!codebook || (B > 0 && (!x || !U || !ids || !emb || !loss))) {
        set_error("gumbel_forward: null pointer");
        return RQHIP_EARG;
    }
    if (B == 0) return RQHIP_OK;
    if (B >= gumbel_mfma_min_rows() && gumbel_mfma_supported(D, K, x, U, emb, nullptr) &&
        (reinterpret_cast<uintptr_t>(codebook) & 15u) == 0) {
        GumbelMfmaParams mp = {};
        mp.x = x; mp.cb = codebook; mp.U = U; mp.ids = ids; mp.emb = emb; mp.loss = loss;
        mp.B = B; mp.K = K; mp.temperature = temperature; mp.beta = beta;
        return gumbel_mfma_forward(mp, reinterpret_cast<hipStream_t>(stream));
    }
    GumbelParams p = {};
    p.x = x; p.cb = codebook; p.U = U; p.ids = ids; p.emb = emb; p.loss = loss;
    p.B = B; p.D = D; p.K = K; p.Kpad = (K + 63) & ~63; p.temperature = temperature; p.beta = beta;
    const size_t lds = gumbel_lds_floats(D, K, p.Kpad, false) * sizeof(float);
    static LdsGrant attr_fwd;
    RQ_RETURN_IF_HIP(attr_fwd.ensure(reinterpret_cast<const void *>(gumbel_kernel<false, false>), 160 * 1024));
    hipLaunchKernelGGL((gumbel_kernel<false, false>), dim3(gumbel_grid(B)), dim3(kGThreads), lds,
                       reinterpret_cast<hipStream_t>(stream), p);
    RQ_CHECK_LAUNCH("gumbel_kernel<fwd>");
    return RQHIP_OK;
}

extern "C" size_t rqhip_gumbel_backward_workspace_bytes(int64_t B, int D, int K) {
    if (B <= 0 || D <= 0 || K <= 0) return 16;
    return (size_t)cu_count() * 2 * (size_t)K * D * sizeof(float);
}

extern "C" int rqhip_gumbel_backward(const float *x, int64_t B, int D, const float *codebook, int K, const float *U,
                                     float temperature, float beta, const float *g_emb, const float *g_loss,
                                     float *g_x, float *g_codebook, void *workspace, size_t workspace_bytes,
                                     rqhip_stream_t stream) {
    int rc = check_shape("gumbel_backward", B, D, K, true);
    if (rc) return rc;
    if (!codebook || !g_codebook || (B > 0 && (!x || !U || !g_x))) {
        set_error("gumbel_backward: null pointer");
        return RQHIP_EARG;
    }
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (B == 0) {
        if (int rc = fill_words(g_codebook, 0u, sizeof(float) * (size_t)K * D, s)) return rc;
        return RQHIP_OK;
    }
    const bool mfma = B >= gumbel_mfma_min_rows() && gumbel_mfma_supported(D, K, x, U, g_emb, g_x) &&
                      (reinterpret_cast<uintptr_t>(codebook) & 15u) == 0;
    const int grid = mfma ? gumbel_mfma_backward_grid(B) : gumbel_grid(B);
    if (!workspace || workspace_bytes < (size_t)grid * K * D * sizeof(float)) {
        set_error("gumbel_backward: workspace too small");
        return RQHIP_EWORKSPACE;
    }
    if (mfma) {
        GumbelMfmaParams mp = {};
        mp.x = x; mp.cb = codebook; mp.U = U; mp.g_emb = g_emb; mp.g_loss = g_loss; mp.g_x = g_x;
        mp.partial = reinterpret_cast<float *>(workspace);
        mp.B = B; mp.K = K; mp.temperature = temperature; mp.beta = beta;
        rc = gumbel_mfma_backward(mp, s);
        if (rc) return rc;
        const int n = K * D;
        hipLaunchKernelGGL(gumbel_reduce_kernel, dim3((n + 255) / 256), dim3(256), 0, s, mp.partial, grid, n, g_codebook);
        RQ_CHECK_LAUNCH("gumbel_reduce_kernel");
        return RQHIP_OK;
    }
    GumbelParams p = {};
    p.x = x; p.cb = codebook; p.U = U; p.g_emb = g_emb; p.g_loss = g_loss; p.g_x = g_x;
    p.partial = reinterpret_cast<float *>(workspace);
    p.B = B; p.D = D; p.K = K; p.Kpad = (K + 63) & ~63; p.temperature = temperature; p.beta = beta;
    const size_t lds = gumbel_lds_floats(D, K, p.Kpad, true) * sizeof(float);
    static LdsGrant grant_plain, grant_acc;
    RQ_RETURN_IF_HIP(grant_plain.ensure(reinterpret_cast<const void *>(gumbel_kernel<true, false>), 160 * 1024));
    RQ_RETURN_IF_HIP(grant_acc.ensure(reinterpret_cast<const void *>(gumbel_kernel<true, true>), 160 * 1024));
    if (K <= 64 * kGAccPerLane && D <= kGAccD)
        hipLaunchKernelGGL((gumbel_kernel<true, true>), dim3(grid), dim3(kGThreads), lds, s, p);
    else
        hipLaunchKernelGGL((gumbel_kernel<true, false>), dim3(grid), dim3(kGThreads), lds, s, p);
    RQ_CHECK_LAUNCH("gumbel_kernel<bwd>");
    const int n = K * D;
    hipLaunchKernelGGL(gumbel_reduce_kernel, dim3((n + 255) / 256), dim3(256), 0, s, p.partial, grid, n, g_codebook);
    RQ_CHECK_LAUNCH("gumbel_reduce_kernel");
    return RQHIP_OK;
}
