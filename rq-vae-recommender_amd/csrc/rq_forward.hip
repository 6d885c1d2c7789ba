// rq_forward.hip -- fused residual-quantisation forward for gfx950 (MI355X).
//
// Replaces, per batch row, the level loop of RqVae.get_semantic_ids (reference modules/rqvae.py:118-139)
// and each level's Quantize.forward (modules/quantize.py:104-163): distance, argmin, codeword gather,
// STE / rotation-trick / eval output, quantize loss, residual subtraction -- plus the emb-sum and
// emb-norm consumers of rqvae.py:146,158.  The B x K distance matrix never exists in memory.
//
// Mapping to the hardware
//   * distance: dist[i,k] = (|x_i|^2 + |c_k|^2) - (2 x_i).c_k with the dot product on the fp32 matrix
//     pipe: v_mfma_f32_32x32x2_f32 computes a 32(codes) x 32(items) tile, k = 2 feature dims per
//     instruction, as an exact fp32 FMA chain in d order (== oracle's dot2x_chain).  Codes are the A
//     operand so that after the MFMA every lane owns ONE item (column) and 16 codes (rows): the argmin
//     over codes is a per-lane running minimum, no cross-lane traffic until one final lane/lane+32
//     exchange per level.
//   * a wave owns 32 items; lane (i = lane&31, h = lane>>5) keeps the item's residual features of parity h
//     (d = 2*kk + h) in KSTEPS registers for all L levels -- the residual never leaves registers.
//   * codebooks are staged in LDS, permuted to [d-quad q][parity h][code c][4] so that a lane's A operands
//     for four consecutive MFMAs are one conflict-free ds_read_b128.  All L levels stay resident when
//     they fit in the 160 KiB LDS (3 x 256 x 32: 99 KiB); otherwise one chunk of one level at a time.
//   * 512-thread workgroups = 2 waves per SIMD: one wave's VALU epilogue (16 distances per lane per
//     tile) overlaps the other wave's MFMAs.
//
// Arithmetic is bit-identical to oracle/rq_oracle.c (tests/test_gpu_parity.py).
#include "rqhip_common.h"
#include "rq_rowmath.h"

namespace rqhip {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int kWgThreads = 512;
constexpr int kWavesPerWg = kWgThreads / RQ_WAVE;
constexpr int kLdsBudget = 160 * 1024;

struct RqFwdParams {
    const float *res0;
    const float *cb;      // [L,K,D]
    const float *csq;     // [L,Kp]  (workspace)
    const float *csqmax;  // [L]     (workspace) NaN-propagating max of csq per level
    int64_t *ids;
    float *embs, *residuals, *emb_sum, *loss, *embs_norm;
    long long B;
    long long n_tiles;    // ceil(B/32)
    int n_iter;           // tiles per wave (grid-stride)
    int D, L, K, Kp;
    int Kc;               // codes per LDS buffer (multiple of 32)
    int nchunks;          // chunks per level (1 when resident)
    int resident;         // all levels staged once
    float beta;
};

// ---- codebook squared norms (quantize.py:115), once per call -------------------------------------
// csq[l,k] = sumsq2(C[l,k,:]); csqmax[l] = max_k csq[l,k] (NaN if any is NaN).  grid = L, block = 256.
__global__ void rq_csq_kernel(const float *__restrict__ cb, int L, int K, int Kp, int D,
                              float *__restrict__ csq, float *__restrict__ csqmax) {
    const int l = blockIdx.x;
    const float *c = cb + (size_t)l * K * D;
    float m = 0.0f;
    bool nan = false;
    for (int k = threadIdx.x; k < Kp; k += blockDim.x) {
        float v = __builtin_inff();
        if (k < K) {
            float a0 = 0.0f, a1 = 0.0f;
            for (int d = 0; d < D; ++d) {
                float x = c[(size_t)k * D + d];
                float p = x * x;
                if (d & 1) a1 = a1 + p; else a0 = a0 + p;
            }
            v = a0 + a1;
            if (v != v) nan = true; else if (v > m) m = v;
        }
        csq[(size_t)l * Kp + k] = v;
    }
    __shared__ float sm[256];
    __shared__ int sn[256];
    sm[threadIdx.x] = m;
    sn[threadIdx.x] = nan;
    __syncthreads();
    for (int s = blockDim.x / 2; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) {
            sm[threadIdx.x] = fmaxf(sm[threadIdx.x], sm[threadIdx.x + s]);
            sn[threadIdx.x] |= sn[threadIdx.x + s];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) csqmax[l] = sn[0] ? __builtin_nanf("") : sm[0];
}

// ---- LDS staging ------------------------------------------------------------------------------------
// buffer = [image: KSTEPS*2*Kc floats as float4[(q*2+h)*Kc + c]][csq: Kc floats]
// image float4 (q,h,c), element j  =  C[kbase+c][d = 2*(4q+j)+h]   (0 beyond K or D)
template <int KSTEPS>
__device__ __forceinline__ void stage_codes(float *buf, const float *__restrict__ cb_l,
                                            const float *__restrict__ csq_l, int kbase, int Kc, int K,
                                            int D) {
    const int tid = threadIdx.x;
    if ((D & 3) == 0) {
        constexpr int d4n = KSTEPS / 2;  // float4 groups per padded row
        const int total = Kc * d4n;
        for (int e = tid; e < total; e += kWgThreads) {
            const int c = e / d4n, d4 = e - c * d4n;
            const int k = kbase + c;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (k < K && 4 * d4 < D) v = *reinterpret_cast<const f32x4 *>(cb_l + (size_t)k * D + 4 * d4);
            const int q = d4 >> 1, j = (d4 & 1) * 2;
            f32x2 ev = {v.x, v.z}, od = {v.y, v.w};
            *reinterpret_cast<f32x2 *>(buf + ((size_t)(q * 2 + 0) * Kc + c) * 4 + j) = ev;
            *reinterpret_cast<f32x2 *>(buf + ((size_t)(q * 2 + 1) * Kc + c) * 4 + j) = od;
        }
    } else {
        constexpr int Dp = KSTEPS * 2;
        const int total = Kc * Dp;
        for (int e = tid; e < total; e += kWgThreads) {
            const int c = e / Dp, d = e - c * Dp;
            const int k = kbase + c;
            float v = (k < K && d < D) ? cb_l[(size_t)k * D + d] : 0.0f;
            const int kk = d >> 1, h = d & 1;
            buf[((size_t)((kk >> 2) * 2 + h) * Kc + c) * 4 + (kk & 3)] = v;
        }
    }
    float *csq_s = buf + (size_t)KSTEPS * 2 * Kc;
    for (int c = tid; c < Kc; c += kWgThreads)
        csq_s[c] = (kbase + c < K) ? csq_l[kbase + c] : __builtin_inff();
}

// ---- exact torch.min semantics for rows whose distances may be non-finite (rare) ----------------------
// Wave-cooperative: every lane scans codes k = lane, lane+64, ...; result = index of the first NaN
// distance if any, else the first index of the minimum (quantize.py:128 / ATen min kernel).
template <int KSTEPS>
__device__ __forceinline__ int slow_argmin_row(const float (&r)[KSTEPS], int j, float xsq_j,
                                            const float *__restrict__ cb_l,
                                            const float *__restrict__ csq_l, int K, int D) {
    const int lane = threadIdx.x & 63;
    float x0[KSTEPS], x1[KSTEPS];
#pragma unroll
    for (int kk = 0; kk < KSTEPS; ++kk) {
        x0[kk] = __shfl(r[kk], j, 64);
        x1[kk] = __shfl(r[kk], j + 32, 64);
    }
    int nanidx = 0x7fffffff, lidx = 0x7fffffff;
    float lbest = __builtin_inff();
    for (int k = lane; k < K; k += 64) {
        const float *c = cb_l + (size_t)k * D;
        float acc = 0.0f;
#pragma unroll
        for (int kk = 0; kk < KSTEPS; ++kk) {
            if (2 * kk < D) acc = __builtin_fmaf(2.0f * x0[kk], c[2 * kk], acc);
            if (2 * kk + 1 < D) acc = __builtin_fmaf(2.0f * x1[kk], c[2 * kk + 1], acc);
        }
        const float t = xsq_j + csq_l[k];
        const float dist = t - acc;
        if (dist != dist) {
            nanidx = min(nanidx, k);
        } else if (dist < lbest || (dist == lbest && k < lidx)) {
            lbest = dist;
            lidx = k;
        }
    }
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) {
        const int on = __shfl_xor(nanidx, m, 64);
        const float ob = __shfl_xor(lbest, m, 64);
        const int oi = __shfl_xor(lidx, m, 64);
        nanidx = min(nanidx, on);
        if (ob < lbest || (ob == lbest && oi < lidx)) {
            lbest = ob;
            lidx = oi;
        }
    }
    return nanidx != 0x7fffffff ? nanidx : lidx;
}

template <int KSTEPS, int MODE>
__global__ __launch_bounds__(kWgThreads) void rq_forward_kernel(const RqFwdParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float *smem = reinterpret_cast<float *>(smem_raw);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int il = lane & 31;
    const int h = lane >> 5;
    const int D = p.D, K = p.K, Kc = p.Kc, L = p.L;
    const size_t buf_floats = (size_t)Kc * (KSTEPS * 2 + 1);
    const long long total_waves = (long long)gridDim.x * kWavesPerWg;
    const long long gw = (long long)blockIdx.x * kWavesPerWg + wave;

    if (p.resident) {
        for (int l = 0; l < L; ++l)
            stage_codes<KSTEPS>(smem + l * buf_floats, p.cb + (size_t)l * K * D, p.csq + (size_t)l * p.Kp, 0, Kc,
                                K, D);
        __syncthreads();
    }

    for (int it = 0; it < p.n_iter; ++it) {
        const long long tile = (long long)it * total_waves + gw;
        const bool active = tile < p.n_tiles;
        if (p.resident && !active) break;
        const long long row = tile * 32 + il;
        const bool row_ok = active && row < p.B;
        const long long rowc = row_ok ? row : (p.B - 1);

        float r[KSTEPS], es[KSTEPS];
#pragma unroll
        for (int kk = 0; kk < KSTEPS; ++kk) {
            const int d = 2 * kk + h;
            r[kk] = (d < D) ? p.res0[(size_t)rowc * D + d] : 0.0f;
            es[kk] = 0.0f;
        }
        float lsum = 0.0f;

        for (int l = 0; l < L; ++l) {
            const float *cb_l = p.cb + (size_t)l * K * D;
            const float *csq_l = p.csq + (size_t)l * p.Kp;

            // |x|^2 (quantize.py:114): parity accumulators, multiply and add separately rounded
            const float xsq = pair_sumsq<KSTEPS>(r);

            float best = __builtin_inff();
            int bidx = 0;

            for (int ch = 0; ch < p.nchunks; ++ch) {
                const int kbase = ch * Kc;
                const float *buf = smem + (p.resident ? l * buf_floats : 0);
                if (!p.resident) {
                    __syncthreads();  // previous chunk fully consumed
                    stage_codes<KSTEPS>(smem, cb_l, csq_l, kbase, Kc, K, D);
                    __syncthreads();
                }
                if (active) {
                    const f32x4 *img = reinterpret_cast<const f32x4 *>(buf);
                    const float *csq_s = buf + (size_t)KSTEPS * 2 * Kc;
                    const int ntiles = Kc / 32;

                    for (int t = 0; t < ntiles; ++t) {
                        f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int q = 0; q < KSTEPS / 4; ++q) {
                            const f32x4 a4 = img[(size_t)(q * 2 + h) * Kc + t * 32 + il];
                            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, r[4 * q + 0] + r[4 * q + 0], acc, 0, 0, 0);
                            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, r[4 * q + 1] + r[4 * q + 1], acc, 0, 0, 0);
                            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, r[4 * q + 2] + r[4 * q + 2], acc, 0, 0, 0);
                            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, r[4 * q + 3] + r[4 * q + 3], acc, 0, 0, 0);
                        }
                        // acc[j] belongs to code  t*32 + 8*(j>>2) + 4*h + (j&3)  and item il
                        const float *cq = csq_s + t * 32 + 4 * h;
                        const int kt = kbase + t * 32 + 4 * h;
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const f32x4 c4 = *reinterpret_cast<const f32x4 *>(cq + 8 * g);
#pragma unroll
                            for (int jj = 0; jj < 4; ++jj) {
                                const float tt = xsq + c4[jj];            // (|x|^2 + |c|^2)
                                const float dist = tt - acc[4 * g + jj];  //   - (2x).c
                                const int k = kt + 8 * g + jj;
                                if (dist < best) {
                                    best = dist;
                                    bidx = k;
                                }
                            }
                        }
                    }
                }
            }

            if (active) {
                // lanes (il,0) and (il,1) scanned disjoint code subsets: keep the smaller, ties -> lower index
                {
                    const float ob = shfl_xor32(best);
                    const int oi = shfl_xor32(bidx);
                    if (ob < best || (ob == best && oi < bidx)) {
                        best = ob;
                        bidx = oi;
                    }
                }
                // rows whose distances can be Inf/NaN take torch's exact scan
                const float guard = xsq + p.csqmax[l];
                const bool bad = !(guard < __builtin_inff());
                unsigned long long badmask = __ballot(bad) & 0xffffffffull;
                while (badmask) {
                    const int j = __builtin_ctzll(badmask);
                    badmask &= badmask - 1;
                    const float xj = __shfl(xsq, j, 64);
                    const int res = slow_argmin_row<KSTEPS>(r, j, xj, cb_l, csq_l, K, D);
                    if (il == j) bidx = res;
                }

                // codeword gather (quantize.py:101-102) for this lane's feature parity
                float e[KSTEPS];
#pragma unroll
                for (int kk = 0; kk < KSTEPS; ++kk) {
                    const int d = 2 * kk + h;
                    e[kk] = (d < D) ? cb_l[(size_t)bidx * D + d] : 0.0f;
                }
                // QuantizeLoss (loss.py:38-41): both terms equal sum((x-emb)^2)
                float sa = 0.0f;
#pragma unroll
                for (int kk = 0; kk < KSTEPS; ++kk) {
                    const float df = r[kk] - e[kk];
                    sa = sa + df * df;
                }
                const float s = pair_sum(sa);
                const float lv = s + p.beta * s;
                lsum = (l == 0) ? lv : lsum + lv;

                float o[KSTEPS];
                level_output<KSTEPS, MODE>(r, e, xsq, o);

                const float onorm = __builtin_sqrtf(pair_sumsq<KSTEPS>(o));

                if (row_ok) {
                    if (h == 0) {
                        p.ids[(size_t)l * p.B + row] = (int64_t)bidx;
                        if (p.embs_norm) p.embs_norm[(size_t)row * L + l] = onorm;
                    }
                    const size_t base = ((size_t)l * p.B + row) * D;
#pragma unroll
                    for (int kk = 0; kk < KSTEPS; ++kk) {
                        const int d = 2 * kk + h;
                        if (d < D) {
                            if (p.residuals) p.residuals[base + d] = r[kk];
                            if (p.embs) p.embs[base + d] = o[kk];
                        }
                    }
                }
#pragma unroll
                for (int kk = 0; kk < KSTEPS; ++kk) {
                    es[kk] = (l == 0) ? o[kk] : es[kk] + o[kk];
                    r[kk] = r[kk] - o[kk];  // rqvae.py:130
                }
            }
        }

        if (row_ok) {
            if (h == 0 && p.loss) p.loss[row] = lsum;
            if (p.emb_sum) {
#pragma unroll
                for (int kk = 0; kk < KSTEPS; ++kk) {
                    const int d = 2 * kk + h;
                    if (d < D) p.emb_sum[(size_t)row * D + d] = es[kk];
                }
            }
        }
    }
}

template <int KSTEPS>
static int launch_mode(const RqFwdParams &p, int mode, int grid, size_t lds, hipStream_t s) {
    auto go = [&](auto kern) -> int {
        RQ_RETURN_IF_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        profile_begin(s);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(kWgThreads), lds, s, p);
        profile_end(s);
        RQ_CHECK_LAUNCH("rq_forward_kernel");
        return 0;
    };
    switch (mode) {
        case RQHIP_MODE_EVAL: return go(rq_forward_kernel<KSTEPS, RQHIP_MODE_EVAL>);
        case RQHIP_MODE_STE: return go(rq_forward_kernel<KSTEPS, RQHIP_MODE_STE>);
        case RQHIP_MODE_ROTATION: return go(rq_forward_kernel<KSTEPS, RQHIP_MODE_ROTATION>);
    }
    set_error("rq_forward: unsupported mode %d", mode);
    return RQHIP_EARG;
}

}  // namespace rqhip

using namespace rqhip;

static inline int pad32(int k) { return (k + 31) & ~31; }

extern "C" size_t rqhip_rq_forward_workspace_bytes(int L, int K) {
    if (L <= 0 || K <= 0) return 0;
    return ((size_t)L * pad32(K) + (size_t)L) * sizeof(float);
}

extern "C" int rqhip_rq_forward(const float *res0, int64_t B, int D, const float *codebooks, int L, int K,
                                int mode, float beta, int64_t *ids, float *embs, float *residuals,
                                float *emb_sum, float *loss, float *embs_norm, void *workspace,
                                size_t workspace_bytes, rqhip_stream_t stream) {
    if (B < 0 || !codebooks || (B > 0 && (!res0 || !ids))) {
        set_error("rq_forward: null pointer or negative B");
        return RQHIP_EARG;
    }
    if (D < 1 || D > 128 || K < 1 || K > 65536 || L < 1 || L > 16) {
        set_error("rq_forward: unsupported shape D=%d K=%d L=%d (need 1<=D<=128, 1<=K<=65536, 1<=L<=16)", D, K, L);
        return RQHIP_EUNSUPPORTED;
    }
    if (mode != RQHIP_MODE_EVAL && mode != RQHIP_MODE_STE && mode != RQHIP_MODE_ROTATION) {
        set_error("rq_forward: mode %d is not EVAL/STE/ROTATION (Gumbel has its own entry point)", mode);
        return RQHIP_EARG;
    }
    if (!workspace || workspace_bytes < rqhip_rq_forward_workspace_bytes(L, K)) {
        set_error("rq_forward: workspace too small (%zu < %zu)", workspace_bytes,
                  rqhip_rq_forward_workspace_bytes(L, K));
        return RQHIP_EWORKSPACE;
    }
    if (B == 0) return RQHIP_OK;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int Kp = pad32(K);
    float *csq = reinterpret_cast<float *>(workspace);
    float *csqmax = csq + (size_t)L * Kp;
    hipLaunchKernelGGL(rq_csq_kernel, dim3(L), dim3(256), 0, s, codebooks, L, K, Kp, D, csq, csqmax);
    RQ_CHECK_LAUNCH("rq_csq_kernel");

    const int ksteps = ksteps_for(D);
    const int Dp = ksteps * 2;
    RqFwdParams p;
    p.res0 = res0; p.cb = codebooks; p.csq = csq; p.csqmax = csqmax;
    p.ids = ids; p.embs = embs; p.residuals = residuals; p.emb_sum = emb_sum; p.loss = loss;
    p.embs_norm = embs_norm;
    p.B = B; p.n_tiles = (B + 31) / 32; p.D = D; p.L = L; p.K = K; p.Kp = Kp; p.beta = beta;
    const size_t level_bytes = (size_t)Kp * (Dp + 1) * sizeof(float);
    if (level_bytes * L <= (size_t)kLdsBudget) {
        p.resident = 1; p.Kc = Kp; p.nchunks = 1;
    } else {
        p.resident = 0;
        int kc = (int)(((size_t)kLdsBudget / 2) / ((size_t)(Dp + 1) * sizeof(float)));  // <= 80 KiB: 2 WG/CU
        kc &= ~31;
        if (kc > Kp) kc = Kp;
        if (kc < 32) kc = 32;
        p.Kc = kc; p.nchunks = (Kp + kc - 1) / kc;
    }
    const size_t lds = (size_t)p.Kc * (Dp + 1) * sizeof(float) * (p.resident ? L : 1);
    const int cus = cu_count();
    const int wg_per_cu = (lds * 2 <= (size_t)kLdsBudget) ? 2 : 1;
    long long want = (p.n_tiles + kWavesPerWg - 1) / kWavesPerWg;
    long long cap = (long long)cus * wg_per_cu;
    const int grid = (int)(want < cap ? want : cap);
    const long long total_waves = (long long)grid * kWavesPerWg;
    p.n_iter = (int)((p.n_tiles + total_waves - 1) / total_waves);

    switch (ksteps) {
        case 4: return launch_mode<4>(p, mode, grid, lds, s);
        case 8: return launch_mode<8>(p, mode, grid, lds, s);
        case 16: return launch_mode<16>(p, mode, grid, lds, s);
        case 32: return launch_mode<32>(p, mode, grid, lds, s);
        default: return launch_mode<64>(p, mode, grid, lds, s);
    }
}
