// rq_forward_valu.hip -- the LDS / vector-ALU form of the residual-quantisation forward (gfx950), D = 32.
//
// BASELINE.json's north_star describes the hot path as "codebooks staged in LDS ... per-row argmin" and allows the matrix
// (MFMA) form "only if rocprof shows it wins over the LDS path".  This file IS that LDS path, built to be measured
// against csrc/rq_forward.hip (rqhip_rq_forward_ex, RQHIP_FWD_SCAN_VALU; tools/ab_scan.py, profiles/r03_scan_ab.*):
// same arithmetic, same bits (tests/test_gpu_parity.py::test_valu_scan_equals_the_oracle), no matrix instruction.
//
// Reference code replaced: modules/quantize.py:104-163 per level, the level loop of modules/rqvae.py:118-139.
//
// Mapping: one LANE owns one row at a time for all L levels (their 32 residual features live in registers); the codes of a
// level are read from LDS as wave-wide broadcasts (every lane the same address: no bank conflicts), two codes at a time
// so that one v_pk_fma_f32 advances both codes' dot products: the LDS image interleaves code pairs,
// img[(k >> 1) * 64 + 2 d + (k & 1)] = C[k][d], and a ds_read_b128 yields {c_k[d], c_k+1[d], c_k[d+1], c_k+1[d+1]}.
// Each half of the packed accumulator is ONE fp32 FMA chain over d = 0..31 -- oracle/rq_oracle.c:dot_chain -- so the
// distances, the strict-< ascending scan and everything behind them are bit-identical to the oracle and to the MFMA
// kernels.  Per 2 codes: 32 packed FMAs against 16 LDS broadcasts per wave.  One persistent workgroup per CU (the
// codebooks fill its LDS), sized so that the CU's share of the batch is one pass (100 000 rows: 391 rows -> 448 threads).
#include "rqhip_common.h"
#include "rq_rowmath.h"

namespace rqhip {

typedef float vf2 __attribute__((ext_vector_type(2)));
typedef float vf4 __attribute__((ext_vector_type(4)));

constexpr int kValuMaxThreads = 512;
constexpr int kValuRows = 1;      // rows per lane and pass
constexpr int kValuD = 32;

struct RqValuParams {
    const float *res0, *cb, *csq, *csqmax;
    int64_t *ids;
    float *embs, *residuals, *emb_sum, *loss, *embs_norm;
    long long B;
    int L, K, Kp;      // Kp: K rounded up to even
    int csq_stride;
    int resident;
    float beta;
};

// parity-split sum of squares, multiply and add separately rounded (oracle sumsq2)
__device__ __forceinline__ float flat_sumsq(const float (&v)[kValuD]) {
    float a0 = 0.0f, a1 = 0.0f;
#pragma unroll
    for (int d = 0; d < kValuD; d += 2) {
        a0 = a0 + v[d] * v[d];
        a1 = a1 + v[d + 1] * v[d + 1];
    }
    return a0 + a1;
}

__device__ __forceinline__ void valu_stage(float *smem, int level_floats, int l0, int nl, const RqValuParams &p) {
    const int per = p.Kp * kValuD;
    for (int e = threadIdx.x; e < nl * per; e += blockDim.x) {
        const int li = e / per, rem = e - li * per;
        const int k = rem >> 5, d = rem & 31;
        const float v = (k < p.K) ? p.cb[((size_t)(l0 + li) * p.K + k) * kValuD + d] : 0.0f;
        smem[li * level_floats + (k >> 1) * 64 + 2 * d + (k & 1)] = v;
    }
    for (int e = threadIdx.x; e < nl * p.Kp; e += blockDim.x) {
        const int li = e / p.Kp, k = e - li * p.Kp;
        smem[li * level_floats + per + k] = (k < p.K) ? p.csq[(size_t)(l0 + li) * p.csq_stride + k] : __builtin_inff();
    }
}

template <int MODE>
__global__ __launch_bounds__(kValuMaxThreads) void rq_forward_valu_kernel(const RqValuParams p) {
    extern __shared__ __attribute__((aligned(16))) float vsmem[];
    constexpr int D = kValuD, R = kValuRows;
    const int level_floats = p.Kp * (D + 1);
    const int L = p.L, K = p.K;
    if (p.resident) {
        valu_stage(vsmem, level_floats, 0, L, p);
        __syncthreads();
    }
    const long long stride = (long long)gridDim.x * blockDim.x;
    // (every thread of the workgroup runs the same number of passes: the per-level staging of non-resident launches
    // synchronises the whole workgroup)
    const long long n_pass = (p.B + stride * R - 1) / (stride * R);
    for (long long pass = 0; pass < n_pass; ++pass) {
    const long long row0 = pass * stride * R + (long long)blockIdx.x * blockDim.x + threadIdx.x;

    float r[R][D], es[R][D], lsum[R];
    bool ok[R];
#pragma unroll
    for (int rr = 0; rr < R; ++rr) {
        const long long row = row0 + (long long)rr * stride;
        ok[rr] = row < p.B;
        const vf4 *src = reinterpret_cast<const vf4 *>(p.res0 + (size_t)(ok[rr] ? row : p.B - 1) * D);
#pragma unroll
        for (int j = 0; j < D / 4; ++j) {
            const vf4 q = src[j];
            r[rr][4 * j] = q.x; r[rr][4 * j + 1] = q.y; r[rr][4 * j + 2] = q.z; r[rr][4 * j + 3] = q.w;
        }
        lsum[rr] = 0.0f;
    }

    for (int l = 0; l < L; ++l) {
        if (!p.resident) {
            __syncthreads();
            valu_stage(vsmem, level_floats, l, 1, p);
            __syncthreads();
        }
        const float *img = vsmem + (p.resident ? l * level_floats : 0);
        const float *csq_s = img + p.Kp * D;
        const float csqmax_l = p.csqmax[l];
        float xsq[R], best[R];
        int bidx[R];
#pragma unroll
        for (int rr = 0; rr < R; ++rr) {
            xsq[rr] = flat_sumsq(r[rr]);
            best[rr] = __builtin_inff();
            bidx[rr] = 0x7fffffff;
        }
        // ---- the scan: two codes per iteration, strict '<' in ascending code order == first-index argmin -----------
        for (int kp = 0; kp < p.Kp / 2; ++kp) {
            const vf4 *c4 = reinterpret_cast<const vf4 *>(img + (size_t)kp * 64);
            vf2 acc[R];
#pragma unroll
            for (int rr = 0; rr < R; ++rr) acc[rr] = vf2{0.0f, 0.0f};
#pragma unroll
            for (int j = 0; j < D / 2; ++j) {
                const vf4 c = c4[j];     // {c_k[2j], c_k+1[2j], c_k[2j+1], c_k+1[2j+1]}
#pragma unroll
                for (int rr = 0; rr < R; ++rr) {
                    acc[rr] = __builtin_elementwise_fma(vf2{r[rr][2 * j], r[rr][2 * j]}, vf2{c.x, c.y}, acc[rr]);
                    acc[rr] = __builtin_elementwise_fma(vf2{r[rr][2 * j + 1], r[rr][2 * j + 1]}, vf2{c.z, c.w}, acc[rr]);
                }
            }
            const vf2 cs = *reinterpret_cast<const vf2 *>(csq_s + 2 * kp);
#pragma unroll
            for (int rr = 0; rr < R; ++rr) {
                const vf2 tt = vf2{xsq[rr], xsq[rr]} + cs;                                         // quantize.py:113-115
                const vf2 dd = __builtin_elementwise_fma(vf2{-2.0f, -2.0f}, acc[rr], tt);         // tt - 2 dot (2 dot exact)
                if (dd.x < best[rr]) { best[rr] = dd.x; bidx[rr] = 2 * kp; }
                if (dd.y < best[rr]) { best[rr] = dd.y; bidx[rr] = 2 * kp + 1; }
            }
        }
        // ---- per row: exact non-finite path, gather, loss, output, residual update -----------------------------------
#pragma unroll
        for (int rr = 0; rr < R; ++rr) {
            const long long row = row0 + (long long)rr * stride;
            const float *cb_l = p.cb + (size_t)l * K * D;
            if (bidx[rr] == 0x7fffffff) bidx[rr] = 0;      // every distance +Inf: torch.min keeps index 0
            if (!(xsq[rr] + csqmax_l < 1.0e38f)) {
                // rows whose distances may be Inf / NaN: torch.min's scan rule (oracle argmin_torch), in-lane
                float bst = 0.0f;
                int bi = 0;
                bool stop = false;
                for (int k = 0; k < K && !stop; ++k) {
                    float a = 0.0f;
                    for (int d = 0; d < D; ++d) a = __builtin_fmaf(r[rr][d], cb_l[(size_t)k * D + d], a);
                    const float t = (xsq[rr] + p.csq[(size_t)l * p.csq_stride + k]) - 2.0f * a;
                    if (k == 0) { bst = t; bi = 0; stop = t != t; }
                    else if (!(t >= bst)) { bst = t; bi = k; stop = t != t; }
                }
                bidx[rr] = bi;
            }
            float e[D], o[D];
            const vf4 *ev = reinterpret_cast<const vf4 *>(cb_l + (size_t)bidx[rr] * D);
#pragma unroll
            for (int j = 0; j < D / 4; ++j) {
                const vf4 q = ev[j];
                e[4 * j] = q.x; e[4 * j + 1] = q.y; e[4 * j + 2] = q.z; e[4 * j + 3] = q.w;
            }
            float df[D];
#pragma unroll
            for (int d = 0; d < D; ++d) df[d] = r[rr][d] - e[d];
            const float s = flat_sumsq(df);
            const float lv = s + p.beta * s;
            lsum[rr] = (l == 0) ? lv : lsum[rr] + lv;
#pragma unroll
            for (int d = 0; d < D; ++d) o[d] = (MODE == RQHIP_MODE_EVAL) ? e[d] : r[rr][d] + (e[d] - r[rr][d]);
            if (ok[rr]) {
                p.ids[(size_t)l * p.B + row] = (int64_t)bidx[rr];
                if (p.embs_norm) p.embs_norm[(size_t)row * L + l] = __builtin_sqrtf(flat_sumsq(o));
                if (p.residuals) {
                    vf4 *dst = reinterpret_cast<vf4 *>(p.residuals + ((size_t)l * p.B + row) * D);
#pragma unroll
                    for (int j = 0; j < D / 4; ++j) dst[j] = vf4{r[rr][4 * j], r[rr][4 * j + 1], r[rr][4 * j + 2], r[rr][4 * j + 3]};
                }
                if (p.embs) {
                    vf4 *dst = reinterpret_cast<vf4 *>(p.embs + ((size_t)l * p.B + row) * D);
#pragma unroll
                    for (int j = 0; j < D / 4; ++j) dst[j] = vf4{o[4 * j], o[4 * j + 1], o[4 * j + 2], o[4 * j + 3]};
                }
            }
#pragma unroll
            for (int d = 0; d < D; ++d) {
                es[rr][d] = (l == 0) ? o[d] : es[rr][d] + o[d];
                r[rr][d] = r[rr][d] - o[d];   // rqvae.py:130
            }
        }
    }
#pragma unroll
    for (int rr = 0; rr < R; ++rr) {
        const long long row = row0 + (long long)rr * stride;
        if (!ok[rr]) continue;
        if (p.loss) p.loss[row] = lsum[rr];
        if (p.emb_sum) {
            vf4 *dst = reinterpret_cast<vf4 *>(p.emb_sum + (size_t)row * D);
#pragma unroll
            for (int j = 0; j < D / 4; ++j) dst[j] = vf4{es[rr][4 * j], es[rr][4 * j + 1], es[rr][4 * j + 2], es[rr][4 * j + 3]};
        }
    }
    }  // passes
}

// host side, called by rqhip_rq_forward_ex (rq_forward.hip) after the codebook norms are in the workspace
int launch_rq_forward_valu(const float *res0, int64_t B, int D, const float *codebooks, int L, int K, int mode, float beta,
                           int64_t *ids, float *embs, float *residuals, float *emb_sum, float *loss, float *embs_norm,
                           const float *csq, int csq_stride, const float *csqmax, hipStream_t s) {
    auto al16 = [](const void *q) { return (reinterpret_cast<uintptr_t>(q) & 15u) == 0; };
    if (D != kValuD || K > 1024 || (mode != RQHIP_MODE_EVAL && mode != RQHIP_MODE_STE) || !al16(res0) || !al16(codebooks) ||
        !al16(embs) || !al16(residuals) || !al16(emb_sum)) {
        set_error("rq_forward (VALU scan): needs D = 32, K <= 1024, EVAL / STE and 16-byte aligned rows");
        return RQHIP_EUNSUPPORTED;
    }
    RqValuParams p;
    p.res0 = res0; p.cb = codebooks; p.csq = csq; p.csqmax = csqmax; p.csq_stride = csq_stride;
    p.ids = ids; p.embs = embs; p.residuals = residuals; p.emb_sum = emb_sum; p.loss = loss; p.embs_norm = embs_norm;
    p.B = B; p.L = L; p.K = K; p.Kp = (K + 1) & ~1; p.beta = beta;
    const size_t level_bytes = (size_t)p.Kp * (kValuD + 1) * sizeof(float);
    p.resident = level_bytes * L <= 150 * 1024;
    const size_t lds = level_bytes * (p.resident ? L : 1);
    // one workgroup per CU; its share of the batch in one pass when that fits 512 threads
    const int cus = cu_count();
    const long long share = (B + cus - 1) / cus;
    int nt = (int)(((share + 63) / 64) * 64);
    if (nt > kValuMaxThreads) nt = kValuMaxThreads;
    if (nt < 64) nt = 64;
    const int grid = (int)((B + nt - 1) / nt < cus ? (B + nt - 1) / nt : cus);
    auto go = [&](auto kern) -> int {
        static LdsGrant grant;
        RQ_RETURN_IF_HIP(grant.ensure(reinterpret_cast<const void *>(kern), 160 * 1024));
        profile_begin(s, RQHIP_PROF_RQ_FORWARD, (double)p.B * p.L * (2.0 * kValuD * p.K + 5.0 * kValuD), (double)p.B * (8.0 * kValuD + 12.0 * p.L + 4.0));
        hipLaunchKernelGGL(kern, dim3(grid), dim3(nt), lds, s, p);
        profile_end(s);
        RQ_CHECK_LAUNCH("rq_forward_valu_kernel");
        return 0;
    };
    return mode == RQHIP_MODE_EVAL ? go(rq_forward_valu_kernel<RQHIP_MODE_EVAL>) : go(rq_forward_valu_kernel<RQHIP_MODE_STE>);
}

}  // namespace rqhip
