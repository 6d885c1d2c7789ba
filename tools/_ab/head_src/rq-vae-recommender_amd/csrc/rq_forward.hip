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
//     instruction, as an exact fp32 FMA chain in d order (== oracle's dot_chain; the factor 2 is applied to
//     the finished dot product, which is exact).  Codes are the A
//     operand so that after the MFMA every lane owns ONE item (column) and 16 codes (rows): the argmin
//     over codes is a per-lane running minimum, no cross-lane traffic until one final lane/lane+32
//     exchange per level.
//   * a wave owns 32 items; lane (i = lane&31, h = lane>>5) keeps the item's residual features of parity h
//     (d = 2*kk + h) in KSTEPS registers for all L levels -- the residual never leaves registers.
//   * codebooks are staged in LDS, permuted to [d-quad q][parity h][code c][4] so that a lane's A operands
//     for four consecutive MFMAs are one conflict-free ds_read_b128.  All L levels stay resident when
//     they fit in the 160 KiB LDS (3 x 256 x 32: 99 KiB); otherwise one chunk of one level at a time.
//   * 768-thread workgroups = 3 waves per SIMD (512 / 256 threads for D = 64 / 128), one workgroup per CU.  More
//     waves hide latencies only: the fp32 MFMA and ordinary VALU instructions share the SIMD's datapath on gfx950
//     (tools/overlap_probe.hip), so the epilogue per 32 codes is kept to ~57 VALU instructions (packed add / fma,
//     a min tree and a top-down walk for the index, see scan_codes).
//   * rows are loaded / stored as float4 half-rows and brought into the pair layout with v_permlane32_swap
//     (rq_rowmath.h).
//   * cooperative tiles (rq_tile<COOP = true>): four waves, one per SIMD, split the codes of ONE row tile and merge
//     their argmin candidates through LDS -- for small batches (<= 4 row tiles per CU) and for the partly filled
//     last round of a big batch.
//
//   * filtered scan (FILT, D = 32 without margins): the distances of the scan come from a 3-term bf16 split of the fp32
//     operands on v_mfma_f32_32x32x16_bf16; rows whose two smallest approximate distances are within the error bound
//     are re-scanned exactly.  Same ids as the fp32 scan, bit for bit (see stage_codes_bf16 / rq_tile).
//
// Arithmetic is bit-identical to oracle/rq_oracle.c (tests/test_gpu_parity.py).
#include "rqhip_common.h"
#include <stdlib.h>
#include "rq_rowmath.h"

namespace rqhip {

#ifdef RQ_TIMING
// developer-only phase timestamps of wave 0 / workgroup 0 (build with EXTRA=-DRQ_TIMING; tools/phase_timing.py)
__device__ unsigned long long rq_dbg[256];
// per-wave trace (100 MHz constant clock): [workgroup*16 + wave][slot]; slot 0 = kernel entry, 1 = staged,
// 2.. = end of each tile this wave processed
__device__ unsigned long long rq_trace[4096 * 16 * 8];
#define RQ_TRACE(slot)                                                                               \
    do {                                                                                             \
        if ((threadIdx.x & 63) == 0 && blockIdx.x < 4096 && (slot) < 8)                              \
            rq_trace[((size_t)blockIdx.x * 16 + (threadIdx.x >> 6)) * 8 + (slot)] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)
#define RQ_STAMP(i)                                                                         \
    do {                                                                                    \
        if (blockIdx.x == 0 && threadIdx.x == 0 && (i) < 256) rq_dbg[(i)] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define RQ_STAMP(i) do { } while (0)
#define RQ_TRACE(slot) do { } while (0)
#endif

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int kCoopWaves = 4;   // waves sharing one cooperative row tile: one per SIMD
constexpr int kCoopSteps = 32;  // LDS counters, used round-robin by (cooperative tile, level)
// floats of LDS behind the staged codebooks: candidates [2][4 waves][32] x (value, index, runner-up) + the counters
constexpr int kCoopCandFloats = 3 * kCoopWaves * 32;   // one parity buffer
constexpr int kCoopLdsFloats = 2 * kCoopCandFloats + kCoopSteps;

// workgroup size is a template parameter of the kernel (NT): more waves per SIMD hide the VALU epilogue and the
// per-level tail of one wave behind the MFMAs of the others, as far as the register budget of KSTEPS allows
constexpr int kLdsBudget = 160 * 1024;

struct RqFwdParams {
    const float *res0;
    const float *cb;      // [L,K,D]
    const float *csq;     // [L,Kp]  (workspace)
    const float *csqmax;  // [L]     (workspace) NaN-propagating max of csq per level
    int64_t *ids;
    float *embs, *residuals, *emb_sum, *loss, *embs_norm;
    float *tie_margin;    // [L,B] or nullptr: relative top-2 distance margin of every level's argmin
    long long B;
    long long n_tiles;    // ceil(B/32)
    int n_iter;           // tiles per wave (grid-stride)
    int D, L, K, Kp;
    int Kc;               // codes per LDS buffer (multiple of 32)
    int nchunks;          // chunks per level (1 when resident)
    int resident;         // all levels staged once
    long long coop_first; // first row tile of the cooperative tail (== n_tiles when there is none)
    float beta;
    // filtered scan: per-group score maxima in LDS (GroupMax): tiles per group, groups per level, bf16 storage
    int tpg, ngroups, gm16;
    // filtered scan with all levels resident: the codebook norms are formed by the kernel itself while it stages the codes
    // (no rq_csq_kernel launch in front: 7-12 us of every call); fp32 copy in LDS for the exact re-decision
    int incsq;
    // ---- the RQ <-> MLP seam (rq_seam_kernel only; SURVEY.md section 8 row f2, first clause) ----
    const float *sm_h;        // [B, 128] rows in front of the input GEMM (res0 = h . w_in^T), or nullptr: rows come from res0
    const float *sm_hmask;    // [B, 128] or nullptr: h is taken as h where hmask > 0 else 0 (the ReLU backward of a data gradient)
    const float *sm_win;      // input weight, [D, 128] row-major (sm_win_t: [128, D], used transposed)
    float *sm_res0_out;       // [B, D] the input GEMM's result (or nullptr)
    const float *sm_wout;     // output weight, [128, D] row-major (sm_wout_t: [D, 128], used transposed); nullptr: no output GEMM
    const float *sm_omask;    // [B, 128]: epilogue 3 keeps out where omask > 0
    float *sm_out;            // [B, 128]
    unsigned *sm_rowmax;      // [4][B] bit patterns of the largest |value| of every row's four 32-column blocks, or nullptr
    unsigned *sm_colmax;      // [128] column maxima (atomic maxima: zeroed by the caller), or nullptr
    int sm_win_t, sm_wout_t, sm_epi;   // epilogue: 0 store, 1 ReLU, 3 mask
};

// rows of a tile as lane (il, h) fetches them: full-width kernels take their half of the row as float4s ("raw", see
// rows_to_pairs), the others their features d = 2 kk + h one by one
template <int KSTEPS, bool FULLD>
__device__ __forceinline__ void load_tile_rows(const RqFwdParams &p, long long tile, int il, int h, int D, float (&v)[KSTEPS]) {
    const long long row = tile * 32 + il;
    const long long rowc = (tile < p.n_tiles && row < p.B) ? row : (p.B - 1);
    if (FULLD) {
        // (the lane's half-row offset is laundered: otherwise `p.res0 + h * KSTEPS` is formed once per wave as a 64-bit
        // per-lane pointer that lives across every scan -- with the one of the emb_sum store, the two register pairs the
        // filtered kernel spilled in rounds 4-5, profiles/r05_rq_forward_spills.txt)
        int hv = h;
        asm volatile("" : "+v"(hv));
        const f32x4 *src = reinterpret_cast<const f32x4 *>(p.res0 + ((size_t)rowc * D + hv * KSTEPS));
#pragma unroll
        for (int j = 0; j < KSTEPS / 4; ++j) {
            const f32x4 q = src[j];
            v[4 * j + 0] = q.x; v[4 * j + 1] = q.y; v[4 * j + 2] = q.z; v[4 * j + 3] = q.w;
        }
    } else {
        const float *src = p.res0 + (size_t)rowc * D + h;
#pragma unroll
        for (int kk = 0; kk < KSTEPS; ++kk) v[kk] = (2 * kk + h < D) ? src[2 * kk] : 0.0f;
    }
}

// ---- codebook squared norms (quantize.py:115), once per call -------------------------------------
// csq[l,k] = sumsq2(C[l,k,:]); csqmax[l] = max_k csq[l,k] (NaN if any is NaN).  grid = L, block = a power of two <= 1024 (one code per thread up to K = 1024).
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
            if ((D & 3) == 0 && (reinterpret_cast<uintptr_t>(c) & 15u) == 0) {
                // whole rows as 16-byte loads, all in flight before the first add (the dword form walked a row with 128-byte strides between
                // lanes: 20.8 us for 4 x 1024 codes in front of every micro-batch of configuration 4); same two parity chains, same bits
                const f32x4 *row = reinterpret_cast<const f32x4 *>(c + (size_t)k * D);
                for (int d4 = 0; d4 < D / 4; d4 += 8) {
                    f32x4 x[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) x[u] = (d4 + u < D / 4) ? row[d4 + u] : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        if (d4 + u < D / 4) {
                            a0 = a0 + x[u].x * x[u].x;
                            a1 = a1 + x[u].y * x[u].y;
                            a0 = a0 + x[u].z * x[u].z;
                            a1 = a1 + x[u].w * x[u].w;
                        }
                    }
                }
            } else {
                for (int d = 0; d < D; ++d) {
                    float x = c[(size_t)k * D + d];
                    float p = x * x;
                    if (d & 1) a1 = a1 + p; else a0 = a0 + p;
                }
            }
            v = a0 + a1;
            if (v != v) nan = true; else if (v > m) m = v;
        }
        csq[(size_t)l * Kp + k] = v;
    }
    __shared__ float sm[1024];
    __shared__ int sn[1024];
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
// `nbuf` consecutive buffers are filled from `nbuf` consecutive codebooks (resident mode: all levels at once,
// so that a thread has up to kStageBatch independent 16-byte loads in flight before its first LDS write).
#ifndef RQ_STAGE_BATCH
#define RQ_STAGE_BATCH 4   // (developer A/B: tools/ab_build.sh)
#endif
constexpr int kStageBatch = RQ_STAGE_BATCH;

template <int KSTEPS, int NT>
__device__ __forceinline__ void stage_codes(float *buf0, int buf_floats, int nbuf, const float *__restrict__ cb0,
                                            const float *__restrict__ csq0, int csq_stride, int kbase, int Kc, int K,
                                            int D) {
    const int tid = threadIdx.x;
    if ((D & 3) == 0) {
        // thread -> (code c, float4 group d4) with d4 fixed for the thread's whole walk; c advances by a constant
        // step, wrapping into the next buffer (no divisions, 32-bit offsets: L*K*D <= 2^27)
        constexpr int d4n = KSTEPS / 2;            // float4 groups per padded row (power of two)
        constexpr int cstep = NT / d4n;            // codes covered per sweep of the workgroup
        const int d4 = tid & (d4n - 1);
        const bool dok = 4 * d4 < D;
        const int q = d4 >> 1, j = (d4 & 1) * 2;
        const int lds_even = ((q * 2 + 0) * Kc) * 4 + j, lds_odd = ((q * 2 + 1) * Kc) * 4 + j;
        int c = tid / d4n, bi = 0;
        while (c >= Kc) { c -= Kc; ++bi; }
        while (bi < nbuf) {
            f32x4 v[kStageBatch];
            int cc[kStageBatch], bb[kStageBatch];
#pragma unroll
            for (int u = 0; u < kStageBatch; ++u) {
                cc[u] = c; bb[u] = bi;
                v[u] = f32x4{0.f, 0.f, 0.f, 0.f};
                const int k = kbase + c;
                if (bi < nbuf && k < K && dok)
                    v[u] = *reinterpret_cast<const f32x4 *>(cb0 + (bi * K + k) * D + 4 * d4);
                c += cstep;
                while (c >= Kc && bi < nbuf) { c -= Kc; ++bi; }
            }
#pragma unroll
            for (int u = 0; u < kStageBatch; ++u) {
                if (bb[u] < nbuf) {
                    float *buf = buf0 + bb[u] * buf_floats + cc[u] * 4;
                    f32x2 ev = {v[u].x, v[u].z}, od = {v[u].y, v[u].w};
                    *reinterpret_cast<f32x2 *>(buf + lds_even) = ev;
                    *reinterpret_cast<f32x2 *>(buf + lds_odd) = od;
                }
            }
        }
    } else {
        constexpr int Dp = KSTEPS * 2;
        const int per_buf = Kc * Dp;
        const int total = per_buf * nbuf;
        for (int e = tid; e < total; e += NT) {
            const int bi = e / per_buf, er = e - bi * per_buf;
            const int c = er / Dp, d = er - c * Dp;
            const int k = kbase + c;
            float v = (k < K && d < D) ? cb0[(bi * K + k) * D + d] : 0.0f;
            const int kk = d >> 1, h = d & 1;
            buf0[bi * buf_floats + (((kk >> 2) * 2 + h) * Kc + c) * 4 + (kk & 3)] = v;
        }
    }
    {
        int c = tid, bi = 0;
        while (c >= Kc) { c -= Kc; ++bi; }
        while (bi < nbuf) {
            buf0[bi * buf_floats + KSTEPS * 2 * Kc + c] = (kbase + c < K) ? csq0[bi * csq_stride + kbase + c] : __builtin_inff();
            c += NT;
            while (c >= Kc && bi < nbuf) { c -= Kc; ++bi; }
        }
    }
}

// ---- bf16-split image for the filtered scan (FILT kernels, D = 32 and 64) ---------------------------------------
// Only the ARGMIN of the distances is an output; ids, the gathered codeword and the loss are computed from it exactly.
// The FILT kernels therefore scan with an approximate score and prove, row by row, that the approximation cannot have
// changed the answer.  score_k = x.c_k - |c_k|^2 / 2  (so  d_k = |x|^2 - 2 score_k : the argmin of the distance is the
// argmax of the score), accumulated entirely on the bf16 matrix cores:
//     x = xh + xl + rho_x,  c = ch + cl + rho_c  (bf16 pieces)      x.c ~ xh.ch + xh.cl + xl.ch
//     -|c|^2/2 = qh + qm + ql  EXACTLY (three bf16 pieces of the fp32 value the oracle's csq holds)
// as 3 D/16 + 1 chained v_mfma_f32_32x32x16_bf16 per 32 codes x 32 rows (fp32 accumulation; products of two bf16 are
// exact in fp32): no VALU arithmetic at all between the matrix pipe and the (max, runner-up, index) tournament.  The XDL
// cores run beside the VALU instead of sharing its datapath as the fp32 MFMA does, and cost 32 cycles each instead of 64.
// Rows whose best and runner-up scores are closer than the bound `filt_threshold` (below; derivation in DESIGN.md
// section 4.1, pinned by tests/test_filter_bound.py) are re-decided exactly by `exact_argmin_groups`.
//
// buffer = [image: 4 S blocks of Kc 16-byte elements, S = D/16][q: Kc 8-byte elements]
//   block (plane * S + s) * 2 + h, element c = 8 bf16: j-th = plane (hi / lo) of C[kbase + c][d = 2 (8 s + j) + h]
//   i.e. lane (il, h) finds, for K-step s, the same features 2 kk + h, kk = 8 s + j, that its row registers r[kk] hold;
//   q element c = {qh, qm, ql, 0}: the A operand (k slots 0..3 of lane (c, 0)) of the last matrix instruction, whose B
//   operand is 1 in k slots 0..2 and 0 elsewhere.  Codes beyond K get q = -3e38: they never win a finite row.
typedef __bf16 rq_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 rq_bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 rq_bf16x2 __attribute__((ext_vector_type(2)));

// v = hi + lo + rho with hi = bf16(v) (round to nearest even), lo = bf16(v - hi) (the difference is exact in fp32):
// |lo| <= 2^-8 |v|, |rho| <= 2^-17 |v|  (bf16 keeps 8 significant bits: unit round-off 2^-8, half an ulp of lo is 2^-17 |v|)
__device__ __forceinline__ void bf16_split(float v, __bf16 &hi, __bf16 &lo) {
    hi = (__bf16)v;
    lo = (__bf16)(v - (float)hi);
}
// two values at once: one v_cvt_pk_bf16_f32 per plane
__device__ __forceinline__ void bf16_split2(float a, float b, rq_bf16x2 &hi, rq_bf16x2 &lo) {
    hi = __builtin_convertvector(f32x2{a, b}, rq_bf16x2);
    const unsigned u = __builtin_bit_cast(unsigned, hi);
    const float ha = __builtin_bit_cast(float, u << 16), hb = __builtin_bit_cast(float, u & 0xffff0000u);
    lo = __builtin_convertvector(f32x2{a - ha, b - hb}, rq_bf16x2);
}

// The too-close-to-call threshold of the filtered scan, in SCORE units (half the distance gap): a row is re-decided
// exactly unless best - runner_up > filt_threshold.  kFiltC1 / kFiltC2 are exported through rqhip_filter_bound() and
// checked against the derivation by tests/test_filter_bound.py:
//   |score~_k - (x.c_k - csq_k/2)| <= [2^-15 (dropped split terms) + 99 * 2^-23 * 1.016 (accumulation)] |x||c_k|
//                                     + 3 * 2^-24 csq_k (the three q pieces enter last)
//   |d_k - (xsq + csq_k - 2 x.c_k)| <= 2^-18 |x||c_k| + 3 * 2^-24 (xsq + csq_k)          (the oracle's own roundings)
//   two codes, distance units:  gap needed  <= 1.773e-4 |x| max|c| + 1.08e-6 (xsq + max csq)
//   threshold used (distance units)          2^-12    |x| max|c| + 2^-19   (xsq + max csq)     headroom 1.38 / 1.77
// The numbers above are D = 32's (3 D + 3 = 99 accumulation steps).  At D = 64 (195 steps) the derived bound is 2.313e-4
// |x| max|c|, which 2^-12 would cover with 6 % to spare -- and that margin rests on the hardware assumption (H) of DESIGN.md
// 4.1 (at most 2^-23 relative error per accumulation step, measured on a few operand sets by tests/test_gpu_filter_bound.py).
// D = 64 therefore uses 2^-11 (headroom 2.1: a few more rows re-decided exactly, never a wrong id).
constexpr float kFiltC1 = 2.44140625e-4f;       // 2^-12, distance units (D = 32)
constexpr float kFiltC1D64 = 4.8828125e-4f;     // 2^-11 (D = 64)
constexpr float kFiltC2 = 1.9073486328125e-6f;  // 2^-19
template <int KSTEPS>
__device__ __forceinline__ float filt_threshold(float xsq, float csqmax) {
    constexpr float c1 = KSTEPS > 16 ? kFiltC1D64 : kFiltC1;
    return (0.5f * c1) * __builtin_sqrtf(xsq * csqmax) + (0.5f * kFiltC2) * (xsq + csqmax);
}

template <int KSTEPS, int NT>
__device__ __forceinline__ void stage_codes_bf16(float *buf0, int buf_floats, int nbuf, const float *__restrict__ cb0,
                                                 const float *__restrict__ csq0, int csq_stride, int kbase, int Kc, int K,
                                                 float *csq_lds = nullptr, unsigned *csqmax_bits = nullptr, int level0 = 0) {
    constexpr int D = 2 * KSTEPS, S = KSTEPS / 8, d4n = D / 4, cstep = NT / d4n;
    const int tid = threadIdx.x;
    const int d4 = tid & (d4n - 1);
    const int s_blk = d4 >> 2, j0 = 2 * (d4 & 3);
    int c = tid / d4n, bi = 0;
    while (c >= Kc) { c -= Kc; ++bi; }
    while (bi < nbuf) {
        f32x4 v[kStageBatch];
        int cc[kStageBatch], bb[kStageBatch];
#pragma unroll
        for (int u = 0; u < kStageBatch; ++u) {
            cc[u] = c; bb[u] = bi;
            v[u] = f32x4{0.f, 0.f, 0.f, 0.f};
            const int k = kbase + c;
            if (bi < nbuf && k < K) v[u] = *reinterpret_cast<const f32x4 *>(cb0 + (bi * K + k) * D + 4 * d4);
            c += cstep;
            while (c >= Kc && bi < nbuf) { c -= Kc; ++bi; }
        }
#pragma unroll
        for (int u = 0; u < kStageBatch; ++u) {
            if (bb[u] < nbuf) {
                __bf16 *img = reinterpret_cast<__bf16 *>(buf0 + bb[u] * buf_floats);
                rq_bf16x2 hxz, lxz, hyw, lyw;
                bf16_split2(v[u].x, v[u].z, hxz, lxz);   // features 4 d4, 4 d4 + 2 (h = 0)
                bf16_split2(v[u].y, v[u].w, hyw, lyw);   // features 4 d4 + 1, 4 d4 + 3 (h = 1)
                auto at = [&](int plane, int h) { return img + ((size_t)(((plane * S + s_blk) * 2 + h) * Kc + cc[u])) * 8 + j0; };
                *reinterpret_cast<rq_bf16x2 *>(at(0, 0)) = hxz;
                *reinterpret_cast<rq_bf16x2 *>(at(0, 1)) = hyw;
                *reinterpret_cast<rq_bf16x2 *>(at(1, 0)) = lxz;
                *reinterpret_cast<rq_bf16x2 *>(at(1, 1)) = lyw;
            }
        }
    }
    {
        // -|c|^2 / 2 as three bf16 pieces.  csq_lds != nullptr: the norms are formed HERE, one thread per code, with
        // rq_csq_kernel's arithmetic (parity accumulators, multiply and add separately rounded: oracle sumsq2) from the
        // fp32 code rows in L2; they also go to csq_lds (fp32, for the exact re-decision) and into the level's maximum
        // csqmax_bits[level] (unsigned order == float order for values >= 0; a NaN norm sorts above +Inf and poisons the
        // level's guard, as rq_csq_kernel's NaN-propagating maximum does).
        int c2 = tid, b2 = 0;
        while (c2 >= Kc) { c2 -= Kc; ++b2; }
        while (b2 < nbuf) {
            float cs;
            if (kbase + c2 >= K) {
                cs = __builtin_inff();
            } else if (csq_lds) {
                const f32x4 *row = reinterpret_cast<const f32x4 *>(cb0 + (size_t)(b2 * K + kbase + c2) * D);
                f32x4 v[d4n];
#pragma unroll
                for (int j = 0; j < d4n; ++j) v[j] = row[j];
                float a0 = 0.0f, a1 = 0.0f;
#pragma unroll
                for (int j = 0; j < d4n; ++j) {
                    a0 = a0 + v[j].x * v[j].x;
                    a1 = a1 + v[j].y * v[j].y;
                    a0 = a0 + v[j].z * v[j].z;
                    a1 = a1 + v[j].w * v[j].w;
                }
                cs = a0 + a1;
                atomicMax(csqmax_bits + level0 + b2, (cs != cs) ? 0x7fc00000u : __float_as_uint(cs));
            } else {
                cs = csq0[b2 * csq_stride + kbase + c2];
            }
            if (csq_lds) csq_lds[b2 * Kc + c2] = cs;
            const float q = (kbase + c2 < K) ? -0.5f * cs : -3.0e38f;
            const __bf16 qh = (__bf16)q;
            const float r1 = q - (float)qh;          // exact
            const __bf16 qm = (__bf16)r1;
            const __bf16 ql = (__bf16)(r1 - (float)qm);   // exact again; the remainder fits 8 bits: qh + qm + ql == q
            *reinterpret_cast<rq_bf16x4 *>(buf0 + b2 * buf_floats + 2 * KSTEPS * Kc + 2 * c2) = rq_bf16x4{qh, qm, ql, (__bf16)0.0f};
            c2 += NT;
            while (c2 >= Kc && b2 < nbuf) { c2 -= Kc; ++b2; }
        }
    }
}

// ---- exact torch.min semantics for rows whose distances may be non-finite (rare) ----------------------
// Wave-cooperative: every lane scans codes k = lane, lane+64, ...; result = index of the first NaN
// distance if any, else the first index of the minimum (quantize.py:128 / ATen min kernel).
template <int KSTEPS>
__device__ __forceinline__ int slow_argmin_row(const float (&r)[KSTEPS], int j, float xsq_j,
                                               const float *__restrict__ cb_l, const float *__restrict__ csq_l, int K,
                                               int D) {
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
            if (2 * kk < D) acc = __builtin_fmaf(x0[kk], c[2 * kk], acc);
            if (2 * kk + 1 < D) acc = __builtin_fmaf(x1[kk], c[2 * kk + 1], acc);
        }
        const float t = xsq_j + csq_l[k];
        const float dist = t - 2.0f * acc;
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

// The exact decision of the filtered kernels (full-width rows, D = 2 KSTEPS): the oracle's distance -- one fp32 FMA chain
// over d = 0, 1, 2, ..., (xsq + csq) - 2 dot -- for the codes of the groups named by `gmask` (group g = codes
// [g gsz, (g + 1) gsz), gsz a multiple of 32), torch.min's rule over them: the first NaN distance if any, else the first
// index of the minimum (quantize.py:128).  Wave-cooperative: row j's features are broadcast through scalar registers
// (v_readlane), every lane takes one code per pass (its row as 16-byte loads from L2), two 32-code half-groups per
// pass.  With gmask = all groups this is the full exact scan (rows that may hold Inf / NaN, cooperative tiles).
template <int KSTEPS>
__device__ __forceinline__ int exact_argmin_groups(const float (&r)[KSTEPS], int j, float xsq_j, unsigned gmask, int gsz,
                                                   const float *__restrict__ cb_l, const float *__restrict__ csq_l, int K) {
    constexpr int D = 2 * KSTEPS;
    const int lane = threadIdx.x & 63;
    float x0[KSTEPS], x1[KSTEPS];   // wave-uniform: features 2 kk and 2 kk + 1 of row j
#pragma unroll
    for (int kk = 0; kk < KSTEPS; ++kk) {
        x0[kk] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, r[kk]), j));
        x1[kk] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, r[kk]), j + 32));
    }
    int nanidx = 0x7fffffff, lidx = 0x7fffffff;
    float lbest = __builtin_inff();
    const int hpg = gsz >> 5;   // 32-code half-units per group
    int g = 0, sub = hpg;
    auto next_base = [&]() -> int {
        if (sub >= hpg) {
            if (!gmask) return -1;
            g = __builtin_ctz(gmask);
            gmask &= gmask - 1;
            sub = 0;
        }
        return g * gsz + 32 * (sub++);
    };
    for (;;) {
        const int b0 = next_base();
        if (b0 < 0) break;
        const int b1 = next_base();
        const int base = lane < 32 ? b0 : b1;
        const int k = base + (lane & 31);
        if (base >= 0 && k < K) {
            const f32x4 *c = reinterpret_cast<const f32x4 *>(cb_l + (size_t)k * D);
            f32x4 v[KSTEPS / 2];
#pragma unroll
            for (int q = 0; q < KSTEPS / 2; ++q) v[q] = c[q];
            const float cs = csq_l[k];
            float acc = 0.0f;
#pragma unroll
            for (int q = 0; q < KSTEPS / 2; ++q) {
                acc = __builtin_fmaf(x0[2 * q], v[q].x, acc);
                acc = __builtin_fmaf(x1[2 * q], v[q].y, acc);
                acc = __builtin_fmaf(x0[2 * q + 1], v[q].z, acc);
                acc = __builtin_fmaf(x1[2 * q + 1], v[q].w, acc);
            }
            const float t = xsq_j + cs;
            const float dist = t - 2.0f * acc;
            if (dist != dist) {
                nanidx = min(nanidx, k);
            } else if (dist < lbest || (dist == lbest && k < lidx)) {
                lbest = dist;
                lidx = k;
            }
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

// ---- the hot loop: distances of 32 items against Kc staged codes, running argmin -------------------------
// Per 32-code tile: KSTEPS dependent MFMAs (exact fp32 FMA chain over d), then 16 distances per lane:
// distance = (|x|^2 + |c|^2) - (2x).c, strict '<' in ascending code order == first-index ties.  The MFMAs of
// one wave overlap the VALU epilogue of the other wave on the same SIMD; interleaving them inside one wave
// measured slower (an instruction between two MFMAs on one accumulator costs ~40 cycles, tools/mfma_probe.hip).
__device__ __forceinline__ float rq_min(float a, float b) {  // plain v_min_f32: no sNaN canonicalisation moves
    float r;
    asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ f32x2 rq_pk_add(f32x2 a, f32x2 b) {  // two IEEE fp32 adds in one issue slot
    f32x2 r;
    asm("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float rq_min3(float a, float b, float c) {
    float r;
    asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ float rq_max(float a, float b) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// (smallest, runner-up) of the union of two sets given each set's (smallest, runner-up); duplicates count
__device__ __forceinline__ void rq_merge2(float lo1, float hi1, float lo2, float hi2, float &lo, float &hi) {
    lo = rq_min(lo1, lo2);
    hi = rq_min3(rq_max(lo1, lo2), hi1, hi2);
}


// MARGIN: also track `second`, the smallest distance over all codes but the winner (the runner-up of the argmin;
// a duplicate of the minimum counts), for the tie-margin output: a (min, runner-up) tournament over the lane's 16
// distances, 26 VALU instructions more per 32 codes than the plain minimum tree.
template <int KSTEPS, bool MARGIN>
__device__ __forceinline__ void scan_codes(const f32x4 *__restrict__ img, const float *__restrict__ csq_s, int Kc,
                                           int kbase, int il, int h, const float (&x)[KSTEPS], float xsq,
                                           float &best, int &bidx, float &second, int t_begin = 0, int t_step = 1) {
    constexpr int KQ = KSTEPS / 4;
    const int ntiles = Kc / 32;
    // The code operands (A) of the NEXT group of four matrix instructions are fetched from LDS before the current
    // group is issued: a wave issues in order and stalls on each dependent MFMA, so a load placed after a group
    // (where the compiler puts it to save four registers) exposes the LDS latency once per group.
    const f32x2 xsq2 = {xsq, xsq};
    auto lda = [&](int t, int q) { return img[(size_t)(q * 2 + h) * Kc + t * 32 + il]; };
    f32x4 cur = lda(t_begin < ntiles ? t_begin : 0, 0);
    for (int t = t_begin; t < ntiles; t += t_step) {
        const int tn = (t + t_step < ntiles) ? t + t_step : t;
        const float *cq = csq_s + t * 32 + 4 * h;
        f32x4 c4[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) c4[g] = *reinterpret_cast<const f32x4 *>(cq + 8 * g);
        f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        f32x2 dp[8];  // distances, two per packed instruction (v_pk_add_f32 / v_pk_fma_f32: same IEEE results)
#pragma unroll
        for (int q = 0; q < KQ; ++q) {
            const f32x4 nxt = (q + 1 < KQ) ? lda(t, q + 1) : lda(tn, 0);
            if (q == KQ - 1) {
                // |x|^2 + |c|^2 (quantize.py:113-115), issued while the previous group still occupies the pipe
#pragma unroll
                for (int pr = 0; pr < 8; ++pr)
                    dp[pr] = rq_pk_add(xsq2, f32x2{c4[pr >> 1][2 * (pr & 1)], c4[pr >> 1][2 * (pr & 1) + 1]});
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 4; ++i)
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(cur[i], x[4 * q + i], acc, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            cur = nxt;
        }
        // acc[j]: code = 32 t + 8 (j>>2) + 4 h + (j&3), item = il.   dist = (|x|^2 + |c|^2) - 2 (x.c): the
        // doubling is exact, so one FMA gives the separately rounded  tt - (2 acc)  (quantize.py:113-117).
        float d[16];
#pragma unroll
        for (int pr = 0; pr < 8; ++pr) {
            const f32x2 v = __builtin_elementwise_fma(f32x2{-2.0f, -2.0f}, f32x2{acc[2 * pr], acc[2 * pr + 1]}, dp[pr]);
            d[2 * pr] = v.x;
            d[2 * pr + 1] = v.y;
        }
        // Argmin of the lane's 16 distances, first index on ties.  fp32 MFMAs and ordinary VALU instructions share
        // the SIMD's datapath on gfx950 (tools/overlap_probe.hip: the two never overlap), so every VALU instruction
        // here costs matrix time.  A minimum tree (11 min/min3) followed by a top-down walk ("is the minimum in
        // the left half?" -- 4 compares, 11 selects) needs ~2/3 of the issue cycles of a compare-and-select
        // tournament that drags the index along (15 compares, 30 selects).
        float a01, a45, b01, b45, a03, a47, b03, b47, a07, b07, tmin;
        if (MARGIN) {
            float lo1[8], hi1[8], lo2[4], hi2[4], hi3a, hi3b, t2;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                lo1[i] = rq_min(d[2 * i], d[2 * i + 1]);
                hi1[i] = rq_max(d[2 * i], d[2 * i + 1]);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) rq_merge2(lo1[2 * i], hi1[2 * i], lo1[2 * i + 1], hi1[2 * i + 1], lo2[i], hi2[i]);
            rq_merge2(lo2[0], hi2[0], lo2[1], hi2[1], a07, hi3a);
            rq_merge2(lo2[2], hi2[2], lo2[3], hi2[3], b07, hi3b);
            rq_merge2(a07, hi3a, b07, hi3b, tmin, t2);
            a01 = lo1[0]; a45 = lo1[2]; b01 = lo1[4]; b45 = lo1[6];
            a03 = lo2[0]; a47 = lo2[1]; b03 = lo2[2]; b47 = lo2[3];
            // runner-up over everything scanned so far (uses `best` before this tile's update)
            second = rq_min3(second, t2, rq_max(best, tmin));
        } else {
            a01 = rq_min(d[0], d[1]); a45 = rq_min(d[4], d[5]);
            b01 = rq_min(d[8], d[9]); b45 = rq_min(d[12], d[13]);
            a03 = rq_min3(a01, d[2], d[3]); a47 = rq_min3(a45, d[6], d[7]);
            b03 = rq_min3(b01, d[10], d[11]); b47 = rq_min3(b45, d[14], d[15]);
            a07 = rq_min(a03, a47); b07 = rq_min(b03, b47);
            tmin = rq_min(a07, b07);
        }
        const bool c3 = a07 != tmin;                       // not in elements 0..7
        const float q03 = c3 ? b03 : a03;
        const bool c2 = q03 != tmin;                       // not in the first quarter of that half
        const float s01 = c3 ? b01 : a01, s45 = c3 ? b45 : a45;
        const float p01 = c2 ? s45 : s01;
        const bool c1 = p01 != tmin;                       // not in the first pair of that quarter
        const float t0 = c3 ? d[8] : d[0], t2 = c3 ? d[10] : d[2], t4 = c3 ? d[12] : d[4], t6 = c3 ? d[14] : d[6];
        const float u0 = c2 ? t4 : t0, u2 = c2 ? t6 : t2;
        const float e0 = c1 ? u2 : u0;
        const bool c0 = e0 != tmin;                        // not the first element of that pair
        // element j = 8 c3 + 4 c2 + 2 c1 + c0 holds code offset 8 (j >> 2) + (j & 3) = 16 c3 + 8 c2 + 2 c1 + c0
        const int slot = (c3 ? 16 : 0) | (c2 ? 8 : 0) | (c1 ? 2 : 0) | (c0 ? 1 : 0);
        const int cand = kbase + t * 32 + 4 * h + slot;
        const bool better = tmin < best;
        best = better ? tmin : best;
        bidx = better ? cand : bidx;
    }
}

__device__ __forceinline__ float rq_max3(float a, float b, float c) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
// (largest, runner-up) of the union of two sets given each set's (largest, runner-up); duplicates count
__device__ __forceinline__ void rq_merge2max(float w1, float l1, float w2, float l2, float &w, float &l) {
    w = rq_max(w1, w2);
    l = rq_max3(rq_min(w1, w2), l1, l2);
}

// Running per-group maxima of a lane's scores, parked in LDS for the exact re-decision (gm[group][lane], one wave's
// slice): `grun` collects the tiles of the current group, `gcnt` counts them, `g` is the group being filled.
struct GroupMax {
    float *slot;     // this lane's word of group 0; groups are 64 words apart
    float grun;
    int gcnt, g, tpg, as16;
    __device__ __forceinline__ void flush() {
        if (as16) {
            // bf16 rounded towards +inf (the test `gm >= best - Th` must never miss a group)
            const unsigned u = __builtin_bit_cast(unsigned, grun);
            const unsigned up = (u + (((int)u >= 0) ? 0xffffu : 0u)) >> 16;
            reinterpret_cast<unsigned short *>(slot)[(size_t)g * 64] = (unsigned short)up;
        } else {
            slot[(size_t)g * 64] = grun;
        }
        g = __builtin_amdgcn_readfirstlane(g + 1);   // (wave-uniform counters: keep them in scalar registers)
        gcnt = 0;
        grun = -__builtin_inff();
    }
    __device__ __forceinline__ void add(float tmax) {
        grun = rq_max(grun, tmax);
        gcnt = __builtin_amdgcn_readfirstlane(gcnt + 1);
        if (gcnt == tpg) flush();
    }
    __device__ __forceinline__ float read(int gi) const {
        if (as16) {
            const unsigned v = reinterpret_cast<const unsigned short *>(slot)[(size_t)gi * 64];
            return __builtin_bit_cast(float, v << 16);
        }
        return slot[(size_t)gi * 64];
    }
};

// scores of 32 codes x 32 rows: the instruction chain the error bound (filt_threshold) is derived for -- the split
// products first, the three pieces of -|c|^2/2 LAST (their rounding is then relative to the finished score only)
template <int S>
__device__ __forceinline__ f32x16 split_scores(const rq_bf16x8 (&a)[2 * S], const rq_bf16x8 &aq, const rq_bf16x8 (&xh)[S],
                                               const rq_bf16x8 (&xl)[S], const rq_bf16x8 &ones) {
    f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < S; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s], xh[s], acc, 0, 0, 0);       // ch . xh
#pragma unroll
    for (int s = 0; s < S; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[S + s], xh[s], acc, 0, 0, 0);   // cl . xh
#pragma unroll
    for (int s = 0; s < S; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s], xl[s], acc, 0, 0, 0);       // ch . xl
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(aq, ones, acc, 0, 0, 0);                                       // - |c|^2 / 2
}
// B operand of the last instruction: 1 in k slots 0..2 (lanes h = 0 hold k = 0..7), 0 elsewhere
__device__ __forceinline__ rq_bf16x8 split_ones(int h) {
    const __bf16 one = (h == 0) ? (__bf16)1.0f : (__bf16)0.0f, zero = (__bf16)0.0f;
    return rq_bf16x8{one, one, one, zero, zero, zero, zero, zero};
}
// a full-width row in pair layout (lane (il, h): r[kk] = feature 2 kk + h) as bf16 hi / lo planes, K-step s = kk in [8 s, 8 s + 8)
template <int KSTEPS>
__device__ __forceinline__ void split_row(const float (&r)[KSTEPS], rq_bf16x8 (&xh)[KSTEPS / 8], rq_bf16x8 (&xl)[KSTEPS / 8]) {
#pragma unroll
    for (int sx = 0; sx < KSTEPS / 8; ++sx)
#pragma unroll
        for (int j = 0; j < 8; j += 2) {
            rq_bf16x2 hi, lo;
            bf16_split2(r[8 * sx + j], r[8 * sx + j + 1], hi, lo);
            xh[sx][j] = hi[0]; xh[sx][j + 1] = hi[1];
            xl[sx][j] = lo[0]; xl[sx][j + 1] = lo[1];
        }
}

// S = 2 (D = 32): per tile only the maximum, the index walk and the in-tile runner-up once per level (finish_split); S = 4: per tile
__host__ __device__ constexpr bool split_walk_deferred(int S) { return S == 2; }

// The filtered scan of Kc staged codes (see stage_codes_bf16): scores x.c - |c|^2/2 of 32 rows against 32 codes per
// tile entirely on the bf16 matrix cores, then the (best, runner-up, index) tournament on the 16 scores of the lane --
// the mirror image (max for min) of scan_codes<.., MARGIN = true>, without any arithmetic in front of it.
template <int S, bool GROUPS, int NS>
__device__ __forceinline__ void scan_codes_split(const rq_bf16x8 *__restrict__ img, const rq_bf16x4 *__restrict__ qimg, int Kc,
                                                 int kbase, int il, int h, const rq_bf16x8 (&xh)[S], const rq_bf16x8 (&xl)[S],
                                                 float &best, int &bcode, float &second, float (&saved)[NS], GroupMax &gm,
                                                 int t_begin = 0, int t_step = 1) {
    const int ntiles = Kc / 32;
    auto lda = [&](int t, int blk) { return img[(size_t)(blk * 2 + h) * Kc + t * 32 + il]; };   // blk = plane * S + s
    auto ldq = [&](int t) {
        const rq_bf16x4 q = qimg[t * 32 + il];
        return rq_bf16x8{q[0], q[1], q[2], q[3], (__bf16)0.0f, (__bf16)0.0f, (__bf16)0.0f, (__bf16)0.0f};
    };
    constexpr bool kDeferredWalk = split_walk_deferred(S);
    const rq_bf16x8 ones = split_ones(h);
    const int t0 = t_begin < ntiles ? t_begin : 0;
    rq_bf16x8 a[2 * S], aq;
#pragma unroll
    for (int b = 0; b < 2 * S; ++b) a[b] = lda(t0, b);
    aq = ldq(t0);
    for (int t = t_begin; t < ntiles; t += t_step) {
        const int tn = (t + t_step < ntiles) ? t + t_step : t;
        const f32x16 acc = split_scores<S>(a, aq, xh, xl, ones);
        // the next tile's code operands, into the same registers: in flight during the VALU epilogue below
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int b = 0; b < 2 * S; ++b) a[b] = lda(tn, b);
        aq = ldq(tn);
        if constexpr (kDeferredWalk) {
            // acc[j]: code = 32 t + 8 (j>>2) + 4 h + (j&3), item = il.  Per tile only the lane's MAXIMUM is formed (eight v_max3) and
            // merged into the running (best, second-best) of the TILE maxima; the 16 scores of the tile that holds the best are
            // copied aside when it changes.  Which of them it is (first occurrence) and the runner-up inside that tile are found
            // once per level by finish_split -- 30 instead of 65 VALU instructions per 32 codes.  Strict `>` over ascending tiles
            // keeps the first of equal maxima.
            const float m0 = rq_max3(acc[0], acc[1], acc[2]), m1 = rq_max3(acc[3], acc[4], acc[5]), m2 = rq_max3(acc[6], acc[7], acc[8]);
            const float m3 = rq_max3(acc[9], acc[10], acc[11]), m4 = rq_max3(acc[12], acc[13], acc[14]);
            const float tmax = rq_max(rq_max3(m0, m1, acc[15]), rq_max3(m2, m3, m4));
            second = rq_max(second, rq_min(best, tmax));    // second-best over the tile maxima (uses `best` before this tile's update)
            if (GROUPS) gm.add(tmax);
            const bool better = tmax > best;
            best = better ? tmax : best;
            bcode = better ? kbase + t * 32 : bcode;
#pragma unroll
            for (int j = 0; j < 16; ++j) saved[j] = better ? acc[j] : saved[j];
        } else {
            // D = 64 (S = 4: 256 registers, two waves per SIMD): the tournament of round 3, the index carried per tile -- the 16 saved
            // scores of the deferred form spill there (measured 103 vs 99 us at 100 000 x 3 x 256 x 64)
            // acc[j]: code = 32 t + 8 (j>>2) + 4 h + (j&3), item = il.  (max, runner-up) tree and top-down walk for the index
            float w1[8], l1[8], w2[4], l2[4], l3a, l3b, t2, a07, b07, tmax;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                w1[i] = rq_max(acc[2 * i], acc[2 * i + 1]);
                l1[i] = rq_min(acc[2 * i], acc[2 * i + 1]);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) rq_merge2max(w1[2 * i], l1[2 * i], w1[2 * i + 1], l1[2 * i + 1], w2[i], l2[i]);
            rq_merge2max(w2[0], l2[0], w2[1], l2[1], a07, l3a);
            rq_merge2max(w2[2], l2[2], w2[3], l2[3], b07, l3b);
            rq_merge2max(a07, l3a, b07, l3b, tmax, t2);
            const float a01 = w1[0], a45 = w1[2], b01 = w1[4], b45 = w1[6];
            const float a03 = w2[0], b03 = w2[2];
            second = rq_max3(second, t2, rq_min(best, tmax));
            if (GROUPS) gm.add(tmax);
            const bool c3 = a07 != tmax;
            const float q03 = c3 ? b03 : a03;
            const bool c2 = q03 != tmax;
            const float s01 = c3 ? b01 : a01, s45 = c3 ? b45 : a45;
            const float p01 = c2 ? s45 : s01;
            const bool c1 = p01 != tmax;
            const float u0a = c3 ? acc[8] : acc[0], u2a = c3 ? acc[10] : acc[2], u4a = c3 ? acc[12] : acc[4], u6a = c3 ? acc[14] : acc[6];
            const float u0 = c2 ? u4a : u0a, u2 = c2 ? u6a : u2a;
            const float e0 = c1 ? u2 : u0;
            const bool c0 = e0 != tmax;
            const int slot = (c3 ? 16 : 0) | (c2 ? 8 : 0) | (c1 ? 2 : 0) | (c0 ? 1 : 0);
            const int cand = kbase + t * 32 + 4 * h + slot;
            const bool better = tmax > best;
            best = better ? tmax : best;
            bcode = better ? cand : bcode;   // (the winner's code itself: finish_split is not called)
        }
    }
}

// The level's finish of the filtered scan for one lane: `saved` = the 16 scores of the tile that holds `best` (codes
// bcode + 8 (j>>2) + 4 h + (j&3)).  second <- max(second-best tile maximum, runner-up inside the best tile) (a duplicate of the
// maximum counts), bidx <- the code of the FIRST slot that equals best.  bcode < 0: the lane scanned nothing.
template <int NS>
__device__ __forceinline__ void finish_split(float best, int bcode, const float (&saved)[NS], int h, float &second, int &bidx) {
    static_assert(NS == 16, "the deferred walk keeps the 16 scores of one tile");
    if (bcode < 0) return;
    float w1[8], l1[8], w2[4], l2[4], l3a, l3b, t2, a07, b07, tmax;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        w1[i] = rq_max(saved[2 * i], saved[2 * i + 1]);
        l1[i] = rq_min(saved[2 * i], saved[2 * i + 1]);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) rq_merge2max(w1[2 * i], l1[2 * i], w1[2 * i + 1], l1[2 * i + 1], w2[i], l2[i]);
    rq_merge2max(w2[0], l2[0], w2[1], l2[1], a07, l3a);
    rq_merge2max(w2[2], l2[2], w2[3], l2[3], b07, l3b);
    rq_merge2max(a07, l3a, b07, l3b, tmax, t2);
    second = rq_max(second, t2);
    const float a01 = w1[0], a45 = w1[2], b01 = w1[4], b45 = w1[6];
    const float a03 = w2[0], b03 = w2[2];
    const bool c3 = a07 != tmax;
    const float q03 = c3 ? b03 : a03;
    const bool c2 = q03 != tmax;
    const float s01 = c3 ? b01 : a01, s45 = c3 ? b45 : a45;
    const float p01 = c2 ? s45 : s01;
    const bool c1 = p01 != tmax;
    const float u0a = c3 ? saved[8] : saved[0], u2a = c3 ? saved[10] : saved[2], u4a = c3 ? saved[12] : saved[4], u6a = c3 ? saved[14] : saved[6];
    const float u0 = c2 ? u4a : u0a, u2 = c2 ? u6a : u2a;
    const float e0 = c1 ? u2 : u0;
    const bool c0 = e0 != tmax;
    const int slot = (c3 ? 16 : 0) | (c2 ? 8 : 0) | (c1 ? 2 : 0) | (c0 ? 1 : 0);
    bidx = bcode + 4 * h + slot;
    (void)best;
}

// One 32-row tile through all L levels.
//   COOP = false: the calling wave owns the tile and scans every staged code itself.
//   COOP = true : the first kCoopWaves (4: one per SIMD) waves of the workgroup work on the SAME tile: wave w scans code
//                 tiles w, w+4, ... of each level, the per-item (distance, index) candidates meet in LDS, and every
//                 one of the four then finishes the level redundantly (gather, loss, output: cheap next to a scan;
//                 no hand-over of the next residual), wave 0 stores.  They synchronise through an LDS counter per
//                 level, not s_barrier, so the other waves of the workgroup are not involved.  Used (a) for small
//                 batches (at most four row tiles per CU, e.g. the reference's batch 640), where one wave per tile
//                 would leave most SIMDs empty, and (b) for the partly filled last round of a big batch, whose
//                 tiles would otherwise each put a whole extra tile on one SIMD (+17 us for 53 of 3125 tiles).
// FULLD: D == 2*KSTEPS, no feature-tail predicates anywhere (the shipped widths 16/32/64 and 8, 128)
// FILT (KSTEPS = 16 / 32, FULLD): the scan runs on the bf16-split image (stage_codes_bf16 / scan_codes_split) and rows
//       that are too close to call are re-decided exactly; never together with MARGIN (the margins are exact quantities).
//       `best` / `second` then hold SCORES (larger is better), not distances.  next_tile >= 0: its rows are fetched into
//       `rn` once the last level's scan is over (a prefetch issued before the tile would keep KSTEPS registers alive
//       across every scan: the filtered kernel spilled 20 of them).
template <int KSTEPS, int MODE, bool FULLD, int NT, bool COOP, bool MARGIN, bool FILT, int RES, bool SEAM = false>
__device__ __forceinline__ void rq_tile(const RqFwdParams &p, float *smem, const float *csqmax_s, float *cand_s,
                                        float *gm_s, float *csqf_s, long long tile, float (&r)[KSTEPS], int D, int buf_floats,
                                        int phase, long long next_tile, float (&rn)[KSTEPS], float *es_ret = nullptr) {
    constexpr int KQ = KSTEPS / 4;
    constexpr int S = FILT ? KSTEPS / 8 : 1;   // 16-wide K steps of the bf16 matrix instruction
    constexpr bool TRACK2 = MARGIN || FILT;   // the runner-up is tracked
    // RES: 1 = all levels staged once, 0 = one chunk of one level at a time, -1 = decided per launch (p.resident).  The
    // filtered kernels fix it at compile time: the two forms share little, and the staging addresses of the form that is
    // not running cost registers.
    const bool resident = RES < 0 ? (p.resident != 0) : (RES == 1);
    static_assert(!FILT || ((KSTEPS == 16 || KSTEPS == 32) && FULLD && !MARGIN), "filtered scan: D = 32 / 64, no margins");
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int il = lane & 31, h = lane >> 5;
    const int K = p.K, Kc = p.Kc, L = p.L;
    const bool active = tile < p.n_tiles;
    const long long row = tile * 32 + il;
    const bool row_ok = active && row < p.B;
    const bool writer = row_ok && (!COOP || wave == 0);
    const size_t level_stride = (size_t)p.B * D;

    float es[KSTEPS];
#pragma unroll
    for (int kk = 0; kk < KSTEPS; ++kk) es[kk] = 0.0f;
    float lsum = 0.0f;
    // (output addresses are formed where they are used, once per level: running per-lane pointers cost ten registers
    // that are alive across every scan)
    RQ_STAMP(1);

    for (int l = 0; l < L; ++l) {
        RQ_STAMP(2 + 8 * l);
        // |x|^2 (quantize.py:114): parity accumulators, multiply and add separately rounded
        const float xsq = pair_sumsq<KSTEPS>(r);

        float best = FILT ? -__builtin_inff() : __builtin_inff(), second = best;
        int bidx = 0x7fffffff;
        // filtered scan: the scores of the tile that holds the lane's best (see scan_codes_split / finish_split)
        float saved[(FILT && split_walk_deferred(S)) ? 16 : 1];
        int bcode = -1;
        if constexpr (FILT && split_walk_deferred(S)) {
#pragma unroll
            for (int j = 0; j < 16; ++j) saved[j] = -__builtin_inff();
        }
        // filtered scan: this level's input rows as bf16 hi / lo planes, K-step s = features 2 (8 s + j) + h
        rq_bf16x8 xh[S], xl[S];
        GroupMax gm;
        if constexpr (FILT) {
            split_row<KSTEPS>(r, xh, xl);
            gm.as16 = p.gm16;
            gm.slot = p.gm16 ? reinterpret_cast<float *>(reinterpret_cast<unsigned short *>(gm_s) + (size_t)wave * 8 * 64 + lane)
                             : gm_s + (size_t)wave * 8 * 64 + lane;
            gm.grun = -__builtin_inff();
            gm.gcnt = 0; gm.g = 0; gm.tpg = p.tpg;
        }

        const float *buf = smem + (resident ? l * buf_floats : 0);
        for (int ch = 0; ch < p.nchunks; ++ch) {
            const int kbase = ch * Kc;
            if (!resident) {
                __syncthreads();  // previous chunk fully consumed
                if constexpr (FILT)
                    stage_codes_bf16<KSTEPS, NT>(smem, buf_floats, 1, p.cb + (size_t)l * K * D, p.csq + (size_t)l * p.Kp, p.Kp,
                                                 kbase, Kc, K, p.incsq ? csqf_s : nullptr,
                                                 const_cast<unsigned *>(reinterpret_cast<const unsigned *>(csqmax_s)), l);
                else
                    stage_codes<KSTEPS, NT>(smem, buf_floats, 1, p.cb + (size_t)l * K * D, p.csq + (size_t)l * p.Kp, p.Kp,
                                            kbase, Kc, K, D);
                __syncthreads();
            }
            if (active) {
                if constexpr (FILT)
                    scan_codes_split<S, !COOP, (split_walk_deferred(S) ? 16 : 1)>(reinterpret_cast<const rq_bf16x8 *>(buf),
                                               reinterpret_cast<const rq_bf16x4 *>(buf + KSTEPS * 2 * Kc), Kc, kbase, il, h, xh,
                                               xl, best, bcode, second, saved, gm, COOP ? wave : 0, COOP ? kCoopWaves : 1);
                else
                    scan_codes<KSTEPS, MARGIN>(reinterpret_cast<const f32x4 *>(buf), buf + KSTEPS * 2 * Kc, Kc, kbase, il, h,
                                               r, xsq, best, bidx, second, COOP ? wave : 0, COOP ? kCoopWaves : 1);
            }
        }

        // (read after the scans: with in-kernel norms a non-resident level's maximum is complete only after its staging barrier)
        const float csqmax_l = csqmax_s[l];
        if constexpr (FILT) {
            if (active) {
                if constexpr (split_walk_deferred(S)) finish_split(best, bcode, saved, h, second, bidx);
                else bidx = bcode < 0 ? bidx : bcode;
            }
            if (!COOP && gm.gcnt) gm.flush();   // a partly filled last group
            if (l == L - 1 && next_tile >= 0) load_tile_rows<KSTEPS, FULLD>(p, next_tile, il, h, D, rn);
        }
        RQ_STAMP(3 + 8 * l);
        // lanes (il,0) and (il,1) scanned disjoint code subsets: keep the smaller, ties -> lower index
        // (lexicographic (distance, index) minimum == first-index argmin over the union; FILT: larger score)
        if (active) {
            const float ob = shfl_xor32(best);
            const int oi = shfl_xor32(bidx);
            if constexpr (FILT) {
                second = rq_max3(second, shfl_xor32(second), rq_min(best, ob));
                if (ob > best || (ob == best && oi < bidx)) {
                    best = ob;
                    bidx = oi;
                }
            } else {
                if (TRACK2) second = rq_min3(second, shfl_xor32(second), rq_max(best, ob));
                if (ob < best || (ob == best && oi < bidx)) {
                    best = ob;
                    bidx = oi;
                }
            }
        }
        if (COOP) {
            // the four waves' candidates meet in LDS, double-buffered by level parity; `phase` counts levels across
            // consecutive cooperative tiles so that parity and counter index keep advancing.  Counter slot
            // (phase+l) mod kCoopSteps has been bumped by every wave once it has published its candidates.
            const int step = phase + l;
            float *cv = cand_s + (step & 1) * kCoopCandFloats;
            int *ci = reinterpret_cast<int *>(cv + kCoopWaves * 32);
            float *c2 = cv + 2 * kCoopWaves * 32;
            int *cnt = reinterpret_cast<int *>(cand_s + 2 * kCoopCandFloats);
            if (h == 0) {
                cv[wave * 32 + il] = best;
                ci[wave * 32 + il] = bidx;
                if (TRACK2) c2[wave * 32 + il] = second;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            // (counters are never reset: the r-th reuse of a slot waits for kCoopWaves * (r + 1))
            const int want = kCoopWaves * (step / kCoopSteps + 1);
            if (lane == 0) __hip_atomic_fetch_add(&cnt[step & (kCoopSteps - 1)], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            while (__hip_atomic_load(&cnt[step & (kCoopSteps - 1)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < want)
                __builtin_amdgcn_s_sleep(1);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            best = FILT ? -__builtin_inff() : __builtin_inff();
            second = best;
            bidx = 0x7fffffff;
#pragma unroll
            for (int w = 0; w < kCoopWaves; ++w) {
                const float ov = cv[w * 32 + il];
                const int oi = ci[w * 32 + il];
                if constexpr (FILT) {
                    second = rq_max3(second, c2[w * 32 + il], rq_min(best, ov));
                    if (ov > best || (ov == best && oi < bidx)) {
                        best = ov;
                        bidx = oi;
                    }
                } else {
                    if (TRACK2) second = rq_min3(second, c2[w * 32 + il], rq_max(best, ov));
                    if (ov < best || (ov == best && oi < bidx)) {
                        best = ov;
                        bidx = oi;
                    }
                }
            }
        }
        const bool do_tail = active;
        if (do_tail) {
            if (bidx == 0x7fffffff) bidx = 0;  // every distance was +Inf: torch.min keeps index 0
            // rows whose distances can be Inf/NaN take torch's exact scan
            // fast path only when no distance term can overflow (then fma(-2,acc,tt) == tt - 2*acc exactly)
            const float guard = xsq + csqmax_l;
            bool bad = !(guard < 1.0e38f);
            unsigned cmask = 0;          // filtered scan: this lane's candidate groups (rows that are too close to call)
            bool full_scan = bad;        // rows that need torch.min's rule over every code
            if constexpr (FILT) {
                // Too close to call?  best / second are scores; Th is half the distance-gap bound (filt_threshold).  Rows of
                // vanishing magnitude (the bound's sqrt underflows; bf16 pieces may be flushed) and any NaN go exact too.
                const float scale2 = xsq * csqmax_l;
                const float Th = filt_threshold<KSTEPS>(xsq, csqmax_l);
                // ... except rows that are EXACTLY zero (every feature +-0: a dead encoder output, a padded row): all their piece
                // products are 0 and the score is the three exact pieces of -|c|^2/2, i.e. exact -- the scan's answer stands
                // unless two norms are within Th, as for any other row.  (Without this, a batch of zero latents paid the full
                // exact scan for every row: 47 -> 476 us for 64 rows at D = 64, profiles/r04_fwd_c3_probe.txt.)
                bool vanishing = !(scale2 > 1.0e-30f);
                if (vanishing && xsq == 0.0f) {
                    unsigned any = 0u;
#pragma unroll
                    for (int kk = 0; kk < KSTEPS; ++kk) any |= __builtin_bit_cast(unsigned, r[kk]) & 0x7fffffffu;
                    any |= (unsigned)shfl_xor32((int)any);
                    if (any == 0u && csqmax_l > 0.0f && csqmax_l < 1.0e38f) vanishing = false;
                }
                full_scan = bad || vanishing || COOP;
#ifndef RQ_FILT_NOSLOW   // (developer timing build, tools/ab_build.sh: how fast is the scan without its exact re-checks?)
                const bool close = !((best - second) > Th);
                bad = bad || vanishing || close;
                if (!COOP) {
                    // groups of codes that may hold the exact argmin: every code whose score is within Th of the best
                    if (__ballot(close && !full_scan)) {
                        const float thr = best - Th;
                        for (int gi = 0; gi < p.ngroups; ++gi) cmask |= (gm.read(gi) >= thr) ? (1u << gi) : 0u;
                    }
                }
#else
                (void)Th;
#endif
            }
            unsigned long long badmask = __ballot(bad) & 0xffffffffull;
            if (badmask) {
                const float *cb_l = p.cb + (size_t)l * K * D;
                const float *csq_l = (FILT && p.incsq) ? csqf_s + (resident ? l * Kc : 0) : p.csq + (size_t)l * p.Kp;
                while (badmask) {
                    const int j = __builtin_ctzll(badmask);
                    badmask &= badmask - 1;
                    const float xj = __shfl(xsq, j, 64);
                    int res;
                    if constexpr (FILT) {
                        const unsigned all = p.ngroups >= 32 ? 0xffffffffu : ((1u << p.ngroups) - 1u);
                        const bool fs = __builtin_amdgcn_readlane((int)full_scan, j) != 0;
                        const unsigned gmask = fs ? all
                                                  : ((unsigned)__builtin_amdgcn_readlane((int)cmask, j) |
                                                     (unsigned)__builtin_amdgcn_readlane((int)cmask, j + 32));
                        res = exact_argmin_groups<KSTEPS>(r, j, xj, gmask, 32 * p.tpg, cb_l, csq_l, K);
                    } else {
                        res = slow_argmin_row<KSTEPS>(r, j, xj, cb_l, csq_l, K, p.D);
                    }
                    if (il == j) bidx = res;
                }
            }
            if (MARGIN) {
                // relative top-2 margin of this level's argmin (see include/rqhip.h); 0 for exact-scan rows
                const float cwin = resident ? buf[KSTEPS * 2 * Kc + bidx] : p.csq[(size_t)l * p.Kp + bidx];
                float m = (second - best) / (xsq + cwin);
                if (bad || m != m) m = 0.0f;
                if (writer && h == 0) p.tie_margin[(size_t)l * p.B + row] = m;
            }

            RQ_STAMP(4 + 8 * l);
            // codeword gather (quantize.py:101-102) for this lane's feature parity: from the staged LDS image
            // when the whole level is resident, else from global memory (L2)
            float e[KSTEPS];
            if (resident && !FILT) {
                const f32x4 *img = reinterpret_cast<const f32x4 *>(buf) + h * Kc + bidx;
#pragma unroll
                for (int q = 0; q < KQ; ++q) {
                    const f32x4 v = img[q * 2 * Kc];
                    e[4 * q + 0] = v.x; e[4 * q + 1] = v.y; e[4 * q + 2] = v.z; e[4 * q + 3] = v.w;
                }
            } else if (FULLD) {
                int hv = h;   // (laundered: `p.cb + h * KSTEPS` as a per-lane pointer across the scans was spilled, see load_tile_rows)
                asm volatile("" : "+v"(hv));
                load_pair_row_vec<KSTEPS>(p.cb + ((size_t)l * K + bidx) * D, hv, e);
            } else {
                const float *src = p.cb + ((size_t)l * K + bidx) * D + h;
#pragma unroll
                for (int kk = 0; kk < KSTEPS; ++kk) e[kk] = (2 * kk + h < D) ? src[2 * kk] : 0.0f;
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

            RQ_STAMP(5 + 8 * l);
            float o[KSTEPS];
            level_output<KSTEPS, MODE>(r, e, xsq, o);

            RQ_STAMP(6 + 8 * l);
            if (writer) {
                // (the row number is laundered through an empty asm: otherwise the compiler forms all these addresses
                // once per tile, keeps them across the scans of every level and spills them -- 22 MB of scratch
                // traffic per 100 000-row launch in round 2)
                long long rowv = row;
                asm volatile("" : "+v"(rowv));
                if (h == 0) p.ids[(size_t)l * p.B + rowv] = (int64_t)bidx;
                if (p.embs_norm) {  // uniform branch: the sqrt sequence is skipped when norms are not requested
                    const float onorm = __builtin_sqrtf(pair_sumsq<KSTEPS>(o));
                    if (h == 0) p.embs_norm[(size_t)rowv * L + l] = onorm;
                }
                // (full-width kernels store float4 row chunks: pointer to the row; otherwise to this lane's first feature)
                if (p.residuals) {
                    float *resid_ptr = p.residuals + l * level_stride + (size_t)rowv * D + (FULLD ? 0 : h);
                    if (FULLD) {
                        store_pair_row<KSTEPS>(resid_ptr, h, r);
                    } else {
#pragma unroll
                        for (int kk = 0; kk < KSTEPS; ++kk)
                            if (2 * kk + h < D) resid_ptr[2 * kk] = r[kk];
                    }
                }
                if (p.embs) {
                    float *embs_ptr = p.embs + l * level_stride + (size_t)rowv * D + (FULLD ? 0 : h);
                    if (FULLD) {
                        store_pair_row<KSTEPS>(embs_ptr, h, o);
                    } else {
#pragma unroll
                        for (int kk = 0; kk < KSTEPS; ++kk)
                            if (2 * kk + h < D) embs_ptr[2 * kk] = o[kk];
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

    RQ_STAMP(100);
    if constexpr (SEAM) {    // the sum of the levels' outputs stays in registers for the output GEMM (seam_out)
#pragma unroll
        for (int kk = 0; kk < KSTEPS; ++kk) es_ret[kk] = es[kk];
    }
    if (writer) {
        long long rowv = row;
        asm volatile("" : "+v"(rowv));
        if (h == 0 && p.loss) p.loss[rowv] = lsum;
        if (p.emb_sum) {
            if (FULLD) {
                int hv = h;   // (laundered like the row: see load_tile_rows)
                asm volatile("" : "+v"(hv));
                store_pair_row<KSTEPS>(p.emb_sum + (size_t)rowv * D, hv, es);
            } else {
                float *dst = p.emb_sum + (size_t)rowv * D + h;
#pragma unroll
                for (int kk = 0; kk < KSTEPS; ++kk)
                    if (2 * kk + h < D) dst[2 * kk] = es[kk];
            }
        }
    }
    RQ_STAMP(101);
}

template <int KSTEPS, int MODE, bool FULLD, int NT, bool MARGIN, bool FILT = false, int RES = -1>
__global__ __launch_bounds__(NT) void rq_forward_kernel(const RqFwdParams p) {
    const bool resident = RES < 0 ? (p.resident != 0) : (RES == 1);
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float *smem = reinterpret_cast<float *>(smem_raw);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int il = lane & 31;
    const int h = lane >> 5;
    const int D = FULLD ? 2 * KSTEPS : p.D;
    const int K = p.K, Kc = p.Kc, L = p.L;
    const int buf_floats = Kc * (KSTEPS * 2 + (FILT ? 2 : 1));   // image + (q pieces : csq) per code
    constexpr int kWavesPerWg = NT / RQ_WAVE;
    const long long total_waves = (long long)gridDim.x * kWavesPerWg;
    // round `it`: waves are enumerated wave-major (wave w of every workgroup before wave w+1), so a partly
    // filled last round spreads over all CUs instead of filling the first workgroups only
    const long long wave_slot = (long long)wave * gridDim.x + blockIdx.x;

    auto load_rows = [&](long long tile, float(&v)[KSTEPS]) { load_tile_rows<KSTEPS, FULLD>(p, tile, il, h, D, v); };
    auto unpack_rows = [&](const float(&raw)[KSTEPS], float(&v)[KSTEPS]) {
        if (FULLD) {
            rows_to_pairs<KSTEPS>(raw, v);
        } else {
#pragma unroll
            for (int kk = 0; kk < KSTEPS; ++kk) v[kk] = raw[kk];
        }
    };

    RQ_STAMP(0);
    RQ_TRACE(0);
    float rn[KSTEPS];  // rows of the NEXT tile, fetched one tile ahead
    load_rows(wave_slot, rn);
    // per-level max codebook norm (Inf/NaN guard) lives in LDS: no global load inside the level loop, so the
    // in-order vmcnt counter never makes a level wait for the previous level's stores
    float *csqmax_s = smem + (resident ? L : 1) * buf_floats;
    float *cand_s = csqmax_s + 16;  // cooperative-tile candidates and counters (kCoopLdsFloats)
    float *gm_s = cand_s + kCoopLdsFloats;   // filtered scan: per-wave group maxima (GroupMax), 8 x 64 words per wave
    // fp32 codebook norms formed in the kernel (p.incsq), behind the group maxima: [levels resident][Kc]
    float *csqf_s = gm_s + (size_t)kWavesPerWg * 8 * 64 / (FILT && p.gm16 ? 2 : 1);
    const bool incsq = FILT && p.incsq;
    if (tid < 16) csqmax_s[tid] = incsq ? 0.0f : (tid < L ? p.csqmax[tid] : 0.0f);   // (bit pattern 0: atomicMax identity)
    if (tid < kCoopSteps) reinterpret_cast<int *>(cand_s + 2 * kCoopCandFloats)[tid] = 0;
    RQ_STAMP(200);
    if (resident) {
        if constexpr (FILT) {
            if (incsq) __syncthreads();   // the level maxima are zeroed before anybody raises them
            stage_codes_bf16<KSTEPS, NT>(smem, buf_floats, L, p.cb, p.csq, p.Kp, 0, Kc, K, incsq ? csqf_s : nullptr,
                                         reinterpret_cast<unsigned *>(csqmax_s), 0);
        } else {
            stage_codes<KSTEPS, NT>(smem, buf_floats, L, p.cb, p.csq, p.Kp, 0, Kc, K, D);
        }
    }
    RQ_STAMP(201);
    __syncthreads();
    RQ_STAMP(202);
    RQ_TRACE(1);
    int trace_slot = 2;
    (void)trace_slot;

    // full rounds: one tile per wave
    for (int it = 0; it < p.n_iter; ++it) {
        const long long tile = (long long)it * total_waves + wave_slot;
        const bool active = tile < p.coop_first;
        if (resident && !active) break;
        float r[KSTEPS];
        unpack_rows(rn, r);
        const long long next = (it + 1 < p.n_iter) ? tile + total_waves : -1;
        if (!FILT && next >= 0) load_rows(next, rn);   // (filtered kernels fetch it after the last level's scan)
        rq_tile<KSTEPS, MODE, FULLD, NT, false, MARGIN, FILT, RES>(p, smem, csqmax_s, cand_s, gm_s, csqf_s,
                                                                   active ? tile : p.n_tiles, r, D, buf_floats, 0,
                                                                   FILT ? next : -1, rn);
        RQ_TRACE(trace_slot);
        ++trace_slot;
    }
    // cooperative tiles (resident mode only), one per workgroup at a time, by the first four waves -- the oldest
    // wave of each SIMD, which finishes its own tile of a full round first
    if (resident && wave < kCoopWaves) {
        int phase = 0;
        // (fetching the next cooperative tile's rows under the current one was measured: no gain)
        for (long long tile = p.coop_first + blockIdx.x; tile < p.n_tiles; tile += gridDim.x) {
            float raw[KSTEPS], r[KSTEPS];
            load_rows(tile, raw);
            unpack_rows(raw, r);
            rq_tile<KSTEPS, MODE, FULLD, NT, true, MARGIN, FILT, RES>(p, smem, csqmax_s, cand_s, gm_s, csqf_s, tile, r, D, buf_floats,
                                                                      phase, -1, raw);
            phase += L;
            RQ_TRACE(trace_slot);
            ++trace_slot;
        }
    }
}

// ---- the RQ <-> MLP seam: 128 -> D GEMM, all levels, D -> 128 GEMM as ONE row-local launch ---------------------------------
// SURVEY.md section 8 row f2, first clause; reference modules/rqvae.py:118-154 (encoder's last Linear, the level loop, embs.sum, the
// decoder's first Linear + ReLU of modules/encoder.py:25-38).  A wave owns 32 rows from the encoder's 128-wide hidden activation to
// the decoder's: res0 never leaves the registers between the GEMM that forms it and the scans, the sum of the levels' outputs goes
// from the last level straight into the output GEMM.  Both GEMMs run on the fp32 matrix pipe with the weight's output features as
// the A operand, i.e. every output is ONE fp32 FMA chain over the input features in ascending order -- the arithmetic of the
// distance scan, restated by oracle/rq_oracle.c:rqo_linear_chain -- so the result does not depend on tiling, batch size or launch
// form: the same kernel with L = 0 and one GEMM switched off IS the stand-alone 128 -> D / D -> 128 layer (the data gradients of
// the backward, with the ReLU backward applied on load / in the epilogue), bit-identical to its fused use.
// Weight images in LDS, behind everything rq_forward_kernel keeps there: the A-operand layout of stage_codes (float4 (q, h, n):
// element j = W[n][d = 2 (4 q + j) + h]).
constexpr int kSeamH = 128;                          // width of the hidden activation on either side
constexpr int kSeamWFloats = kSeamH * 32;            // one weight image: 128 x 32 (either orientation) fp32, 16 KB
constexpr int kSeamLdsFloats = 2 * kSeamWFloats + kSeamH;   // both images + the column maxima of the output

// image[(q * 2 + hh) * n_out + n][j] = W(n, d) for d = 2 (4 q + j) + hh < n_in, W(n, d) = t ? w[d * n_out + n] : w[n * n_in + d]
template <int NT>
__device__ __forceinline__ void seam_stage_weight(float *img, const float *__restrict__ w, int n_out, int n_in, int t) {
    // walk the SOURCE contiguously (coalesced reads; the LDS writes scatter, once per workgroup); all of a thread's loads are issued
    // before its first LDS write (one memory latency per workgroup, not one per trip: a stand-alone launch at batch 640 is two
    // workgroups whose whole life is this staging, one 64-deep chain of matrix instructions and a store)
    constexpr int kPer = (kSeamH * 32 + NT - 1) / NT;
    const int total = n_out * n_in;
    float v[kPer];
#pragma unroll
    for (int u = 0; u < kPer; ++u) {
        const int e = threadIdx.x + u * NT;
        v[u] = e < total ? w[e] : 0.0f;
    }
#pragma unroll
    for (int u = 0; u < kPer; ++u) {
        const int e = threadIdx.x + u * NT;
        if (e < total) {
            const int n = t ? e % n_out : e / n_in, d = t ? e / n_out : e % n_in;
            const int kk = d >> 1, hh = d & 1;
            img[(((kk >> 2) * 2 + hh) * n_out + n) * 4 + (kk & 3)] = v[u];
        }
    }
}

// rows of tile `tile` of h [B, 128] -> res0 = h . w_in^T in pair layout (r[kk] = feature 2 kk + h of row il), stored to sm_res0_out
// this lane's half (64 floats) of row il of tile `tile` of h, masked by hmask when given
__device__ __forceinline__ void seam_load_rows(const RqFwdParams &p, long long tile, int il, int h, float (&raw)[kSeamH / 2]) {
    constexpr int KH = kSeamH / 2;                       // 64 K steps of the 32x32x2 instruction
    const long long row = tile * 32 + il;
    const long long rowc = (tile < p.n_tiles && row < p.B) ? row : (p.B - 1);
    int hv = h;
    asm volatile("" : "+v"(hv));                         // (per-lane row pointers are formed here, not hoisted across tiles)
    const f32x4 *src = reinterpret_cast<const f32x4 *>(p.sm_h + ((size_t)rowc * kSeamH + hv * KH));
#pragma unroll
    for (int j = 0; j < KH / 4; ++j) {
        const f32x4 q = src[j];
        raw[4 * j + 0] = q.x; raw[4 * j + 1] = q.y; raw[4 * j + 2] = q.z; raw[4 * j + 3] = q.w;
    }
    if (p.sm_hmask) {   // threshold_backward(h, mask, 0): 0 where mask <= 0 (a NaN mask keeps the value, as torch does)
        const f32x4 *msk = reinterpret_cast<const f32x4 *>(p.sm_hmask + ((size_t)rowc * kSeamH + hv * KH));
#pragma unroll
        for (int j = 0; j < KH / 4; ++j) {
            const f32x4 m = msk[j];
            raw[4 * j + 0] = m.x <= 0.0f ? 0.0f : raw[4 * j + 0];
            raw[4 * j + 1] = m.y <= 0.0f ? 0.0f : raw[4 * j + 1];
            raw[4 * j + 2] = m.z <= 0.0f ? 0.0f : raw[4 * j + 2];
            raw[4 * j + 3] = m.w <= 0.0f ? 0.0f : raw[4 * j + 3];
        }
    }
}

template <int KSTEPS>
__device__ __forceinline__ void seam_in(const RqFwdParams &p, const float *win_s, long long tile, int il, int h, bool store,
                                        const float (&raw)[kSeamH / 2], float (&r)[KSTEPS]) {
    static_assert(KSTEPS == 16, "the seam kernels are built for D = 32");
    constexpr int KH = kSeamH / 2;
    const long long row = tile * 32 + il;
    int hv = h;
    asm volatile("" : "+v"(hv));
    float x[KH];
    rows_to_pairs<KH>(raw, x);                           // x[kk] = feature 2 kk + h of row il
    const f32x4 *img = reinterpret_cast<const f32x4 *>(win_s);
    f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    f32x4 cur = img[(size_t)h * 32 + il];
#pragma unroll
    for (int q = 0; q < KH / 4; ++q) {
        const f32x4 nxt = img[(size_t)(((q + 1 < KH / 4) ? q + 1 : q) * 2 + h) * 32 + il];
#pragma unroll
        for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(cur[i], x[4 * q + i], acc, 0, 0, 0);
        cur = nxt;
    }
    // acc[j]: output feature 8 (j >> 2) + 4 h + (j & 3) of row il
    if (store && p.sm_res0_out && tile < p.n_tiles && row < p.B) {
        f32x4 *dst = reinterpret_cast<f32x4 *>(p.sm_res0_out + (size_t)row * (2 * KSTEPS) + 4 * hv);
#pragma unroll
        for (int g = 0; g < 4; ++g) dst[2 * g] = f32x4{acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
    }
    // ... into pair layout: lane (il, 0) keeps its even features and takes the partner's, lane (il, 1) the odd ones
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int g = i >> 1, sx = i & 1;
        rq_swap32(acc[2 * i], acc[2 * i + 1], r[4 * g + sx], r[4 * g + 2 + sx]);
    }
}

// out[row][n] = epilogue(sum_d es[d] W_out(n, d)) for the 32-column blocks blk0, blk0 + bstep, ... < 4 of the 128 output columns
template <int KSTEPS>
__device__ __forceinline__ void seam_out(const RqFwdParams &p, const float *wout_s, unsigned *colmax_s, long long tile, int il, int h,
                                         const float (&es)[KSTEPS], int blk0, int bstep) {
    static_assert(KSTEPS == 16, "the seam kernels are built for D = 32");
    const long long row = tile * 32 + il;
    const bool row_ok = tile < p.n_tiles && row < p.B;
    const long long rowc = row_ok ? row : (p.B - 1);
    const f32x4 *img = reinterpret_cast<const f32x4 *>(wout_s);
    int hv = h;
    asm volatile("" : "+v"(hv));
    // (the ReLU-backward mask of the whole row is requested in front of the first matrix instruction: a load behind the stores of the block
    // before it waits for those stores -- stores count in vmcnt -- and for its own memory latency, four times per tile)
    f32x4 msk[kSeamH / 32][4];
    if (p.sm_epi == 3) {
#pragma unroll
        for (int b = 0; b < kSeamH / 32; ++b)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                msk[b][g] = *reinterpret_cast<const f32x4 *>(p.sm_omask + (size_t)rowc * kSeamH + 32 * b + 4 * hv + 8 * g);
    }
#pragma unroll
    for (int blk = 0; blk < kSeamH / 32; ++blk) {
        if (blk < blk0 || (blk - blk0) % bstep != 0) continue;
        f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < KSTEPS / 4; ++q) {
            const f32x4 a = img[(size_t)(q * 2 + h) * kSeamH + 32 * blk + il];
#pragma unroll
            for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], es[4 * q + i], acc, 0, 0, 0);
        }
        // acc[j]: column 32 blk + 8 (j >> 2) + 4 h + (j & 3) of row il
        const size_t at = (size_t)rowc * kSeamH + 32 * blk + 4 * hv;
        unsigned rmx = 0u;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f32x4 v = {acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
            if (p.sm_epi == 1) {          // (a NaN stays a NaN, as torch.relu)
                v.x = v.x < 0.0f ? 0.0f : v.x; v.y = v.y < 0.0f ? 0.0f : v.y;
                v.z = v.z < 0.0f ? 0.0f : v.z; v.w = v.w < 0.0f ? 0.0f : v.w;
            } else if (p.sm_epi == 3) {   // threshold_backward(out, omask, 0)
                const f32x4 m = msk[blk][g];
                v.x = m.x <= 0.0f ? 0.0f : v.x; v.y = m.y <= 0.0f ? 0.0f : v.y;
                v.z = m.z <= 0.0f ? 0.0f : v.z; v.w = m.w <= 0.0f ? 0.0f : v.w;
            }
            if (row_ok) *reinterpret_cast<f32x4 *>(p.sm_out + at + 8 * g) = v;
            if (p.sm_rowmax || p.sm_colmax) {
                const float vc[4] = {v.x, v.y, v.z, v.w};   // (scalars: `v[c]` inside this loop read element 0 for every c with this compiler)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const unsigned bits = row_ok ? (__builtin_bit_cast(unsigned, vc[c]) & 0x7fffffffu) : 0u;
                    rmx = bits > rmx ? bits : rmx;
                    if (p.sm_colmax) {
                        // maximum over the 32 rows of the tile: lanes 0 .. 31 (h = 0) / 32 .. 63 (h = 1) hold the same column
                        unsigned m = bits;
                        m = max(m, (unsigned)__builtin_amdgcn_update_dpp(0, (int)m, 0x111, 0xf, 0xf, true));   // row_shr:1
                        m = max(m, (unsigned)__builtin_amdgcn_update_dpp(0, (int)m, 0x112, 0xf, 0xf, true));   // row_shr:2
                        m = max(m, (unsigned)__builtin_amdgcn_update_dpp(0, (int)m, 0x114, 0xf, 0xf, true));   // row_shr:4
                        m = max(m, (unsigned)__builtin_amdgcn_update_dpp(0, (int)m, 0x118, 0xf, 0xf, true));   // row_shr:8
                        m = max(m, (unsigned)__builtin_amdgcn_update_dpp(0, (int)m, 0x142, 0xa, 0xf, true));   // row_bcast:15 -> rows 1, 3
                        if (il == 31 && m != 0u) atomicMax(&colmax_s[32 * blk + 8 * g + 4 * h + c], m);
                    }
                }
            }
        }
        if (p.sm_rowmax) {
            const unsigned o = (unsigned)shfl_xor32((int)rmx);
            rmx = o > rmx ? o : rmx;
            if (row_ok && h == 0) p.sm_rowmax[(size_t)blk * p.B + row] = rmx;
        }
    }
}

// The seam kernel: rq_forward_kernel's filtered, all-levels-resident form at D = 32 (KSTEPS = 16, 768 threads) with the input GEMM in
// front of every tile and the output GEMM behind it; L = 0 skips the quantisation (and the codebook staging) altogether.
template <int MODE>
__global__ __launch_bounds__(768) void rq_seam_kernel(const RqFwdParams p) {
    constexpr int KSTEPS = 16, NT = 768, D = 32;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float *smem = reinterpret_cast<float *>(smem_raw);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, il = lane & 31, h = lane >> 5;
    const int K = p.K, Kc = p.Kc, L = p.L;
    const int buf_floats = Kc * (KSTEPS * 2 + 2);
    constexpr int kWavesPerWg = NT / RQ_WAVE;
    const long long total_waves = (long long)gridDim.x * kWavesPerWg;
    const long long wave_slot = (long long)wave * gridDim.x + blockIdx.x;
    float *csqmax_s = smem + L * buf_floats;
    float *cand_s = csqmax_s + 16;
    float *gm_s = cand_s + kCoopLdsFloats;
    float *csqf_s = gm_s + (size_t)kWavesPerWg * 8 * 64 / (p.gm16 ? 2 : 1);
    float *win_s = csqf_s + (size_t)L * Kc;
    float *wout_s = win_s + kSeamWFloats;
    unsigned *colmax_s = reinterpret_cast<unsigned *>(wout_s + kSeamWFloats);
    if (tid < 16) csqmax_s[tid] = 0.0f;
    if (tid < kCoopSteps) reinterpret_cast<int *>(cand_s + 2 * kCoopCandFloats)[tid] = 0;
    if (tid < kSeamH) colmax_s[tid] = 0u;
    if (p.sm_h) seam_stage_weight<NT>(win_s, p.sm_win, D, kSeamH, p.sm_win_t);
    if (p.sm_wout) seam_stage_weight<NT>(wout_s, p.sm_wout, kSeamH, D, p.sm_wout_t);
    if (L > 0) {
        __syncthreads();   // the level maxima are zeroed before anybody raises them
        stage_codes_bf16<KSTEPS, NT>(smem, buf_floats, L, p.cb, p.csq, p.Kp, 0, Kc, K, csqf_s, reinterpret_cast<unsigned *>(csqmax_s), 0);
    }
    __syncthreads();

    float rn[KSTEPS];   // (no row prefetch across tiles here: a 128-wide row is 64 registers per lane)
    // (requesting the first tile's rows before the staging was tried: 64 registers that the allocator then keeps across the whole tile
    // loop -- 158 spilled)
    auto tile_rows = [&](long long tile, float (&r)[KSTEPS], bool store) {
        if (p.sm_h) {
            float hraw[kSeamH / 2];
            seam_load_rows(p, tile, il, h, hraw);
            seam_in<KSTEPS>(p, win_s, tile, il, h, store, hraw, r);
        } else {
            float raw[KSTEPS];
            load_tile_rows<KSTEPS, true>(p, tile, il, h, D, raw);
            rows_to_pairs<KSTEPS>(raw, r);
        }
    };
    for (int it = 0; it < p.n_iter; ++it) {
        const long long tile = (long long)it * total_waves + wave_slot;
        if (tile >= p.coop_first) break;
        float r[KSTEPS], es[KSTEPS];
        tile_rows(tile, r, true);
        if (L > 0) {
            rq_tile<KSTEPS, MODE, true, NT, false, false, true, 1, true>(p, smem, csqmax_s, cand_s, gm_s, csqf_s, tile, r, D, buf_floats, 0,
                                                                         -1, rn, es);
        } else {
#pragma unroll
            for (int kk = 0; kk < KSTEPS; ++kk) es[kk] = r[kk];
        }
        if (p.sm_wout) seam_out<KSTEPS>(p, wout_s, colmax_s, tile, il, h, es, 0, 1);
    }
    // cooperative tiles (L > 0 only): the four waves of a tile each form res0 (wave 0 stores it) and, behind the levels, one 32-column
    // block of the output
    if (L > 0 && wave < kCoopWaves) {
        int phase = 0;
        for (long long tile = p.coop_first + blockIdx.x; tile < p.n_tiles; tile += gridDim.x) {
            float r[KSTEPS], es[KSTEPS];
            tile_rows(tile, r, wave == 0);
            rq_tile<KSTEPS, MODE, true, NT, true, false, true, 1, true>(p, smem, csqmax_s, cand_s, gm_s, csqf_s, tile, r, D, buf_floats,
                                                                        phase, -1, rn, es);
            phase += L;
            if (p.sm_wout) seam_out<KSTEPS>(p, wout_s, colmax_s, tile, il, h, es, wave, kCoopWaves);
        }
    }
    if (p.sm_colmax) {
        __syncthreads();
        if (tid < kSeamH && colmax_s[tid]) atomicMax(p.sm_colmax + tid, colmax_s[tid]);
    }
}

#ifdef RQ_TIMING
extern "C" int rqhip_debug_read(unsigned long long *out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(rqhip::rq_dbg), sizeof(unsigned long long) * 256);
}
extern "C" int rqhip_debug_trace(unsigned long long *out, int clear) {
    const size_t bytes = sizeof(unsigned long long) * 4096 * 16 * 8;
    if (clear) {
        void *ptr = nullptr;
        hipError_t e = hipGetSymbolAddress(&ptr, HIP_SYMBOL(rqhip::rq_trace));
        if (e != hipSuccess) return (int)e;
        return (int)hipMemset(ptr, 0, bytes);
    }
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(rqhip::rq_trace), bytes);
}
#endif

// ---- the scores the filtered scan ranks by, written out (tests only) -------------------------------------------------
// One wave per 32 rows; 64 codes staged at a time with the product kernels' own staging, split and instruction chain.
template <int KSTEPS>
__global__ __launch_bounds__(64) void filter_scores_kernel(const float *__restrict__ x, long long B, const float *__restrict__ cb,
                                                           const float *__restrict__ csq, int K, float *__restrict__ scores) {
    constexpr int S = KSTEPS / 8, D = 2 * KSTEPS, Kc = 64;
    __shared__ __attribute__((aligned(16))) float buf[Kc * (2 * KSTEPS + 2)];
    const int lane = threadIdx.x & 63, il = lane & 31, h = lane >> 5;
    const long long row = (long long)blockIdx.x * 32 + il;
    float r[KSTEPS];
    load_pair_row_vec<KSTEPS>(x + (size_t)(row < B ? row : B - 1) * D, h, r);
    rq_bf16x8 xh[S], xl[S];
    split_row<KSTEPS>(r, xh, xl);
    const rq_bf16x8 ones = split_ones(h);
    for (int kbase = 0; kbase < K; kbase += Kc) {
        __syncthreads();
        stage_codes_bf16<KSTEPS, 64>(buf, Kc * (2 * KSTEPS + 2), 1, cb, csq, 0, kbase, Kc, K);
        __syncthreads();
        const rq_bf16x8 *img = reinterpret_cast<const rq_bf16x8 *>(buf);
        const rq_bf16x4 *qimg = reinterpret_cast<const rq_bf16x4 *>(buf + 2 * KSTEPS * Kc);
        for (int t = 0; t < Kc / 32; ++t) {
            rq_bf16x8 a[2 * S];
#pragma unroll
            for (int b = 0; b < 2 * S; ++b) a[b] = img[(size_t)(b * 2 + h) * Kc + t * 32 + il];
            const rq_bf16x4 q = qimg[t * 32 + il];
            const rq_bf16x8 aq = {q[0], q[1], q[2], q[3], (__bf16)0.0f, (__bf16)0.0f, (__bf16)0.0f, (__bf16)0.0f};
            const f32x16 acc = split_scores<S>(a, aq, xh, xl, ones);
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int k = kbase + t * 32 + 8 * (j >> 2) + 4 * h + (j & 3);
                if (row < B && k < K) scores[(size_t)row * K + k] = acc[j];
            }
        }
    }
}

// threads per workgroup by register appetite: <= 168 VGPRs at KSTEPS <= 16 (3 waves/SIMD), 256 at 32, 512 at 64
template <int KSTEPS>
struct WgThreads { static constexpr int value = KSTEPS <= 16 ? 768 : KSTEPS == 32 ? 512 : 256; };

int launch_rq_forward_valu(const float *res0, int64_t B, int D, const float *codebooks, int L, int K, int mode, float beta,
                           int64_t *ids, float *embs, float *residuals, float *emb_sum, float *loss, float *embs_norm,
                           const float *csq, int csq_stride, const float *csqmax, hipStream_t s);   // rq_forward_valu.hip

// does this launch take the filtered scan?  D = 32 / 64, full-width aligned rows, no margins, not forced to fp32
static bool filtered_launch(const RqFwdParams &p, unsigned flags) {
    auto al16 = [](const void *q) { return (reinterpret_cast<uintptr_t>(q) & 15u) == 0; };
    return (p.D == 32 || p.D == 64) && !p.tie_margin && !(flags & RQHIP_FWD_SCAN_FP32) && al16(p.res0) && al16(p.cb) &&
           al16(p.embs) && al16(p.residuals) && al16(p.emb_sum);
}

template <int KSTEPS>
static int launch_mode(const RqFwdParams &p, int mode, int grid, size_t lds, unsigned flags, hipStream_t s) {
    constexpr int NT = WgThreads<KSTEPS>::value;
    auto go = [&](auto kern) -> int {
        // raise the dynamic-LDS limit once per instantiation (not a stream operation: keep it out of the per-call
        // path and out of hipGraph captures); the limit only ever grows
        static LdsGrant grant;
        RQ_RETURN_IF_HIP(grant.ensure(reinterpret_cast<const void *>(kern), (int)kLdsBudget));
        // (bench only; the launch covers every level: SURVEY 8d's per-row figures x B)
        profile_begin(s, RQHIP_PROF_RQ_FORWARD, (double)p.B * p.L * (2.0 * p.D * p.K + 5.0 * p.D), (double)p.B * (8.0 * p.D + 12.0 * p.L + 4.0));
        hipLaunchKernelGGL(kern, dim3(grid), dim3(NT), lds, s, p);
        profile_end(s);
        RQ_CHECK_LAUNCH("rq_forward_kernel");
        return 0;
    };
    // full-width kernels move rows as float4s: every row pointer must be 16-byte aligned (rows are 8*KSTEPS bytes)
    auto al16 = [](const void *q) { return (reinterpret_cast<uintptr_t>(q) & 15u) == 0; };
    const bool full = p.D == 2 * KSTEPS && al16(p.res0) && al16(p.cb) && al16(p.embs) && al16(p.residuals) && al16(p.emb_sum);
    const bool margin = p.tie_margin != nullptr;
    // D = 32 / 64, aligned rows, no margins wanted: the filtered scan (bf16-split matrix products + exact re-decision of
    // the rows that are too close to call).  RQHIP_FWD_SCAN_FP32 keeps the all-fp32 scan (bench / test A/B).
    if constexpr (KSTEPS == 16 || KSTEPS == 32) {
        if (filtered_launch(p, flags)) {
#define RQ_GOF(MODE_)                                                                       \
    return p.resident ? go(rq_forward_kernel<KSTEPS, MODE_, true, NT, false, true, 1>)     \
                      : go(rq_forward_kernel<KSTEPS, MODE_, true, NT, false, true, 0>)
            switch (mode) {
                case RQHIP_MODE_EVAL: RQ_GOF(RQHIP_MODE_EVAL);
                case RQHIP_MODE_STE: RQ_GOF(RQHIP_MODE_STE);
                case RQHIP_MODE_ROTATION: RQ_GOF(RQHIP_MODE_ROTATION);
            }
#undef RQ_GOF
        }
    }
#define RQ_GO(MODE_)                                                                                              \
    return margin ? (full ? go(rq_forward_kernel<KSTEPS, MODE_, true, NT, true>)                                  \
                          : go(rq_forward_kernel<KSTEPS, MODE_, false, NT, true>))                                \
                  : (full ? go(rq_forward_kernel<KSTEPS, MODE_, true, NT, false>)                                 \
                          : go(rq_forward_kernel<KSTEPS, MODE_, false, NT, false>))
    switch (mode) {
        case RQHIP_MODE_EVAL: RQ_GO(RQHIP_MODE_EVAL);
        case RQHIP_MODE_STE: RQ_GO(RQHIP_MODE_STE);
        case RQHIP_MODE_ROTATION: RQ_GO(RQHIP_MODE_ROTATION);
    }
#undef RQ_GO
    set_error("rq_forward: unsupported mode %d", mode);
    return RQHIP_EARG;
}

}  // namespace rqhip

using namespace rqhip;

static inline bool f_resident_small(int resident, long long n_tiles, long long cap) {
    return resident && n_tiles <= 4 * cap;
}

static inline int pad32(int k) { return (k + 63) & ~63; }  // code tiles are processed in pairs

extern "C" size_t rqhip_rq_forward_workspace_bytes(int L, int K) {
    if (L <= 0 || K <= 0) return 0;
    return ((size_t)L * pad32(K) + (size_t)L) * sizeof(float);
}

extern "C" int rqhip_filter_scores(const float *x, int64_t B, int D, const float *codebook, int K, float *scores,
                                   void *workspace, size_t workspace_bytes, rqhip_stream_t stream) {
    if (B < 0 || K < 1 || K > 65536 || (B > 0 && (!x || !codebook || !scores))) {
        set_error("filter_scores: bad arguments");
        return RQHIP_EARG;
    }
    auto al16 = [](const void *q) { return (reinterpret_cast<uintptr_t>(q) & 15u) == 0; };
    if ((D != 32 && D != 64) || !al16(x) || !al16(codebook)) {
        set_error("filter_scores: the filtered scan exists for D = 32 / 64 and 16-byte aligned rows only (D = %d)", D);
        return RQHIP_EUNSUPPORTED;
    }
    if (!workspace || workspace_bytes < rqhip_rq_forward_workspace_bytes(1, K)) {
        set_error("filter_scores: workspace too small");
        return RQHIP_EWORKSPACE;
    }
    if (B == 0) return RQHIP_OK;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int Kp = pad32(K);
    float *csq = reinterpret_cast<float *>(workspace);
    hipLaunchKernelGGL(rq_csq_kernel, dim3(1), dim3(256), 0, s, codebook, 1, K, Kp, D, csq, csq + Kp);
    RQ_CHECK_LAUNCH("rq_csq_kernel");
    const int grid = (int)((B + 31) / 32);
    if (D == 32) hipLaunchKernelGGL(filter_scores_kernel<16>, dim3(grid), dim3(64), 0, s, x, (long long)B, codebook, csq, K, scores);
    else hipLaunchKernelGGL(filter_scores_kernel<32>, dim3(grid), dim3(64), 0, s, x, (long long)B, codebook, csq, K, scores);
    RQ_CHECK_LAUNCH("filter_scores_kernel");
    return RQHIP_OK;
}

extern "C" void rqhip_filter_bound_d(int D, float *c1, float *c2) {
    if (c1) *c1 = D > 32 ? kFiltC1D64 : kFiltC1;
    if (c2) *c2 = kFiltC2;
}

extern "C" void rqhip_filter_bound(float *c1, float *c2) {
    if (c1) *c1 = kFiltC1;
    if (c2) *c2 = kFiltC2;
}

extern "C" int rqhip_rq_forward(const float *res0, int64_t B, int D, const float *codebooks, int L, int K,
                                int mode, float beta, int64_t *ids, float *embs, float *residuals,
                                float *emb_sum, float *loss, float *embs_norm, float *tie_margin, void *workspace,
                                size_t workspace_bytes, rqhip_stream_t stream) {
    return rqhip_rq_forward_ex(res0, B, D, codebooks, L, K, mode, beta, ids, embs, residuals, emb_sum, loss, embs_norm,
                               tie_margin, workspace, workspace_bytes, 0u, stream);
}

extern "C" int rqhip_rq_forward_ex(const float *res0, int64_t B, int D, const float *codebooks, int L, int K,
                                   int mode, float beta, int64_t *ids, float *embs, float *residuals,
                                   float *emb_sum, float *loss, float *embs_norm, float *tie_margin, void *workspace,
                                   size_t workspace_bytes, unsigned flags, rqhip_stream_t stream) {
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
    if (flags & ~(RQHIP_FWD_SCAN_FP32 | RQHIP_FWD_SCAN_VALU | RQHIP_FWD_NO_COOP_TAIL)) {
        set_error("rq_forward: unknown flags 0x%x", flags);
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
    auto launch_csq = [&]() -> int {
        int threads = 256;
        while (threads < Kp && threads < 1024) threads <<= 1;
        hipLaunchKernelGGL(rq_csq_kernel, dim3(L), dim3(threads), 0, s, codebooks, L, K, Kp, D, csq, csqmax);
        RQ_CHECK_LAUNCH("rq_csq_kernel");
        return 0;
    };

    if (flags & RQHIP_FWD_SCAN_VALU) {
        if (int rc = launch_csq()) return rc;
        if (tie_margin) {
            set_error("rq_forward (VALU scan): tie_margin is not available");
            return RQHIP_EUNSUPPORTED;
        }
        return launch_rq_forward_valu(res0, B, D, codebooks, L, K, mode, beta, ids, embs, residuals, emb_sum, loss, embs_norm,
                                      csq, Kp, csqmax, s);
    }
    const int ksteps = ksteps_for(D);
    const int Dp = ksteps * 2;
    RqFwdParams p;
    p.res0 = res0; p.cb = codebooks; p.csq = csq; p.csqmax = csqmax;
    p.ids = ids; p.embs = embs; p.residuals = residuals; p.emb_sum = emb_sum; p.loss = loss;
    p.embs_norm = embs_norm; p.tie_margin = tie_margin;
    p.B = B; p.n_tiles = (B + 31) / 32; p.D = D; p.L = L; p.K = K; p.Kp = Kp; p.beta = beta;
    p.tpg = 1; p.ngroups = 0; p.gm16 = 0; p.incsq = 0;
    p.sm_h = nullptr; p.sm_wout = nullptr;
    const int waves_per_wg = (ksteps <= 16 ? 768 : ksteps == 32 ? 512 : 256) / RQ_WAVE;
    const bool filt = filtered_launch(p, flags);
    // per code: the operand image (Dp words) + its squared norm (fp32 scan) or the three bf16 pieces of -|c|^2/2 (filtered)
    const size_t code_bytes = (size_t)(Dp + (filt ? 2 : 1)) * sizeof(float);
    const size_t fixed_bytes = 64 + kCoopLdsFloats * sizeof(float);
    // filtered scan: 8 group maxima per lane and wave, fp32 -- or bf16 (rounded up) when LDS is short
    const size_t gm32 = filt ? (size_t)waves_per_wg * 8 * 64 * sizeof(float) : 0;
    size_t gm_bytes = gm32;
    // filtered scan with every level resident: + an fp32 copy of the norms, which the kernel then forms itself while it
    // stages (4 bytes per code).  Launches that stage level by level keep rq_csq_kernel in front: they would redo the
    // norms at every level of every round of row tiles (measured: 211 -> 228 us on the 125 000 x 4 x 1024 micro-batch).
    const size_t csq_copy = filt ? sizeof(float) : 0;
    const size_t level_bytes = (size_t)Kp * code_bytes;
    if ((level_bytes + (size_t)Kp * csq_copy) * L + fixed_bytes + gm_bytes <= (size_t)kLdsBudget) {
        p.resident = 1; p.Kc = Kp; p.nchunks = 1;
    } else {
        p.resident = 0;
        // one workgroup per CU (its waves fill the register file), so a chunk may use the whole LDS
        if (filt && level_bytes + fixed_bytes + gm32 > (size_t)kLdsBudget && level_bytes + fixed_bytes + gm32 / 2 <= (size_t)kLdsBudget) {
            p.gm16 = 1;   // K = 1024 at D = 32: the whole level fits beside half-width group maxima
            gm_bytes = gm32 / 2;
        }
        int kc = (int)(((size_t)kLdsBudget - fixed_bytes - gm_bytes) / code_bytes);
        kc &= ~63;
        if (kc > Kp) kc = Kp;
        if (kc < 64) kc = 64;
        p.Kc = kc; p.nchunks = (Kp + kc - 1) / kc;
    }
    if (filt) {
        const int tiles = p.nchunks * (p.Kc / 32);   // code tiles a wave scans per level (padding included)
        p.tpg = (tiles + 7) / 8;
        p.ngroups = (tiles + p.tpg - 1) / p.tpg;
    }
    // ... unless the launch is ONE round of row tiles and the level is one chunk (a small batch at D = 64: the reference's
    // rqvae_ml32m.gin, batch 64): every workgroup then stages every level exactly once, the norms cost it nothing extra, and the
    // launch in front (10 us of a 250 us training step, strided row reads) goes away.
    const bool one_round = p.n_tiles <= (long long)cu_count() * waves_per_wg;
    const bool chunk_with_norms = filt && !p.resident && p.nchunks == 1 && one_round &&
                                  (size_t)p.Kc * (code_bytes + csq_copy) + fixed_bytes + gm_bytes <= (size_t)kLdsBudget;
    p.incsq = filt && (p.resident || chunk_with_norms);
    if (!p.incsq)
        if (int rc = launch_csq()) return rc;
    const size_t lds = (size_t)p.Kc * (code_bytes + (p.incsq ? csq_copy : 0)) * (p.resident ? L : 1) + fixed_bytes + gm_bytes;
    const int cus = cu_count();
    const int wg_per_cu = 1;  // 768 / 512 / 256 threads at <= 168 / 256 / 512 VGPRs: one workgroup fills a CU
    long long want = (p.n_tiles + waves_per_wg - 1) / waves_per_wg;
    long long cap = (long long)cus * wg_per_cu;
    // small batches (at most four row tiles per CU): every tile is cooperative, one workgroup per tile at a time
    const bool all_coop = f_resident_small(p.resident, p.n_tiles, cap);
    if (all_coop) want = p.n_tiles;
    const int grid = (int)(want < cap ? want : cap);
    const long long total_waves = (long long)grid * waves_per_wg;
    // A partly filled last round (53 of 3125 tiles at 100 000 rows) puts a whole extra tile on one SIMD of each CU it
    // lands on; when it is at most one tile per workgroup it is done cooperatively instead, a quarter per SIMD.  Rounds
    // are counted per SIMD (4 per CU: waves are enumerated wave-major, four consecutive waves of a workgroup sit on its
    // four SIMDs), not per wave slot: at D = 64 (8 waves per workgroup) 3125 tiles are 3 x 1024 + 53, and the 53 used to
    // run as ordinary tiles of a fourth SIMD round on 53 CUs.
    p.coop_first = all_coop ? 0 : p.n_tiles;
    if (!all_coop && p.resident && !(flags & RQHIP_FWD_NO_COOP_TAIL)) {
        const long long rem = p.n_tiles % ((long long)grid * 4);
        if (rem > 0 && rem <= grid) p.coop_first = p.n_tiles - rem;
    }
    p.n_iter = (int)((p.coop_first + total_waves - 1) / total_waves);

    switch (ksteps) {
        case 4: return launch_mode<4>(p, mode, grid, lds, flags, s);
        case 8: return launch_mode<8>(p, mode, grid, lds, flags, s);
        case 16: return launch_mode<16>(p, mode, grid, lds, flags, s);
        case 32: return launch_mode<32>(p, mode, grid, lds, flags, s);
        default: return launch_mode<64>(p, mode, grid, lds, flags, s);
    }
}

// ---- the RQ <-> MLP seam (rq_seam_kernel) ---------------------------------------------------------------------------------------
// LDS of a launch: rq_forward_kernel's filtered resident layout for L levels of Kp codes + the two weight images + the column maxima;
// gm16 (bf16 group maxima, rounded up) when the fp32 form does not fit.  Returns 0 when even that does not fit.
static size_t seam_lds_bytes(int L, int Kp, int *gm16) {
    const size_t code_bytes = (size_t)(32 + 2) * sizeof(float) + sizeof(float);   // bf16 image + q pieces + the fp32 norm
    const size_t fixed = 64 + kCoopLdsFloats * sizeof(float) + (size_t)kSeamLdsFloats * sizeof(float);
    const size_t gm32 = (size_t)(768 / RQ_WAVE) * 8 * 64 * sizeof(float);
    const size_t levels = (size_t)L * Kp * code_bytes;
    *gm16 = 0;
    if (levels + fixed + gm32 <= (size_t)kLdsBudget) return levels + fixed + gm32;
    *gm16 = 1;
    if (levels + fixed + gm32 / 2 <= (size_t)kLdsBudget) return levels + fixed + gm32 / 2;
    return 0;
}

extern "C" int rqhip_rq_seam_supported(int D, int H, int L, int K) {
    if (D != 32 || H != kSeamH || L < 0 || L > 16 || (L > 0 && (K < 1 || K > 65536))) return 0;
    int gm16;
    return seam_lds_bytes(L, L > 0 ? pad32(K) : 0, &gm16) != 0 ? 1 : 0;
}

extern "C" int rqhip_rq_seam(const rqhip_seam_args *a, rqhip_stream_t stream) {
    if (!a) {
        set_error("rq_seam: null argument block");
        return RQHIP_EARG;
    }
    if (a->B < 0 || !rqhip_rq_seam_supported(a->D, a->H, a->L, a->K)) {
        set_error("rq_seam: unsupported shape (D = %d must be 32, H = %d must be %d, L = %d levels of K = %d codes must fit the LDS "
                  "beside the two weight images: rqhip_rq_seam_supported)", a->D, a->H, kSeamH, a->L, a->K);
        return RQHIP_EUNSUPPORTED;
    }
    auto al16 = [](const void *q) { return (reinterpret_cast<uintptr_t>(q) & 15u) == 0; };
    const bool pre = a->h != nullptr, post = a->w_out != nullptr;
    if ((pre && !a->w_in) || (!pre && a->B > 0 && !a->res0) || (post && !a->out) || (a->L > 0 && (!a->codebooks || (a->B > 0 && !a->ids))) ||
        (!pre && a->L == 0 && !post) || (post && a->out_epilogue == RQHIP_EPI_MASK && !a->out_mask) ||
        (post && a->out_epilogue != RQHIP_EPI_STORE && a->out_epilogue != RQHIP_EPI_RELU && a->out_epilogue != RQHIP_EPI_MASK)) {
        set_error("rq_seam: inconsistent arguments (h needs w_in; no h needs res0; w_out needs out; levels need codebooks and ids; "
                  "RQHIP_EPI_MASK needs out_mask; something must be asked for)");
        return RQHIP_EARG;
    }
    if (a->L > 0 && a->mode != RQHIP_MODE_EVAL && a->mode != RQHIP_MODE_STE && a->mode != RQHIP_MODE_ROTATION) {
        set_error("rq_seam: mode %d is not EVAL/STE/ROTATION", a->mode);
        return RQHIP_EARG;
    }
    if (!al16(a->h) || !al16(a->h_mask) || !al16(a->res0) || !al16(a->res0_out) || !al16(a->codebooks) || !al16(a->emb_sum) ||
        !al16(a->out) || !al16(a->out_mask)) {
        set_error("rq_seam: row pointers must be 16-byte aligned");
        return RQHIP_EARG;
    }
    if (a->B == 0) return RQHIP_OK;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    RqFwdParams p;
    p.res0 = a->res0; p.cb = a->codebooks; p.csq = nullptr; p.csqmax = nullptr;
    p.ids = a->ids; p.embs = nullptr; p.residuals = nullptr; p.emb_sum = a->L > 0 ? a->emb_sum : nullptr;
    p.loss = a->L > 0 ? a->loss : nullptr; p.embs_norm = a->L > 0 ? a->embs_norm : nullptr; p.tie_margin = nullptr;
    p.B = a->B; p.n_tiles = (a->B + 31) / 32; p.D = 32; p.L = a->L; p.K = a->K; p.Kp = a->L > 0 ? pad32(a->K) : 0; p.beta = a->beta;
    p.resident = 1; p.Kc = p.Kp; p.nchunks = 1; p.incsq = 1;
    int gm16 = 0;
    const size_t lds = seam_lds_bytes(p.L, p.Kp, &gm16);
    p.gm16 = gm16;
    const int tiles = p.Kc / 32;
    p.tpg = tiles > 0 ? (tiles + 7) / 8 : 1;
    p.ngroups = tiles > 0 ? (tiles + p.tpg - 1) / p.tpg : 0;
    p.sm_h = a->h; p.sm_hmask = pre ? a->h_mask : nullptr; p.sm_win = a->w_in; p.sm_win_t = a->w_in_transposed ? 1 : 0;
    p.sm_res0_out = pre ? a->res0_out : nullptr;
    p.sm_wout = a->w_out; p.sm_wout_t = a->w_out_transposed ? 1 : 0; p.sm_epi = a->out_epilogue; p.sm_omask = a->out_mask;
    p.sm_out = a->out; p.sm_rowmax = post ? a->out_row_max : nullptr; p.sm_colmax = post ? a->out_col_max : nullptr;
    // grid and cooperative tiles: rq_forward's rules (one workgroup of 12 waves per CU; small batches and the partly filled last round
    // of a big one are worked on by four waves per tile) -- with levels only: a bare GEMM tile is short
    const int cus = cu_count();
    const int waves_per_wg = 768 / RQ_WAVE;
    long long want = (p.n_tiles + waves_per_wg - 1) / waves_per_wg;
    const long long cap = cus;
    const bool all_coop = p.L > 0 && f_resident_small(1, p.n_tiles, cap);
    // (a bare GEMM launch spreads its tiles over the CUs first -- waves are enumerated wave-major: at batch 640, 20 tiles packed into
    // two workgroups put three 64-deep chains of fp32 matrix instructions on every SIMD they used, 12 us for a 6 us job)
    if (all_coop || p.L == 0) want = p.n_tiles;
    const int grid = (int)(want < cap ? want : cap);
    const long long total_waves = (long long)grid * waves_per_wg;
    p.coop_first = all_coop ? 0 : p.n_tiles;
    if (!all_coop && p.L > 0) {
        const long long rem = p.n_tiles % ((long long)grid * 4);
        if (rem > 0 && rem <= grid) p.coop_first = p.n_tiles - rem;
    }
    p.n_iter = (int)((p.coop_first + total_waves - 1) / total_waves);
    auto go = [&](auto kern) -> int {
        static LdsGrant grant;
        RQ_RETURN_IF_HIP(grant.ensure(reinterpret_cast<const void *>(kern), (int)kLdsBudget));
        const double gemm = 2.0 * (double)p.B * 32.0 * kSeamH;
        profile_begin(s, RQHIP_PROF_SEAM, (pre ? gemm : 0.0) + (post ? gemm : 0.0) + (double)p.B * p.L * (2.0 * 32 * p.K + 5.0 * 32),
                      (double)p.B * 4.0 * ((pre ? kSeamH + (a->h_mask ? kSeamH : 0) + (a->res0_out ? 32 : 0) : 32) +
                                           (post ? kSeamH + (a->out_epilogue == RQHIP_EPI_MASK ? kSeamH : 0) : 0) +
                                           (p.L > 0 ? 2 * p.L + (a->emb_sum ? 32 : 0) + 1 + p.L : 0)));
        hipLaunchKernelGGL(kern, dim3(grid), dim3(768), lds, s, p);
        profile_end(s);
        RQ_CHECK_LAUNCH("rq_seam_kernel");
        return 0;
    };
    switch (p.L > 0 ? a->mode : RQHIP_MODE_EVAL) {
        case RQHIP_MODE_EVAL: return go(rq_seam_kernel<RQHIP_MODE_EVAL>);
        case RQHIP_MODE_STE: return go(rq_seam_kernel<RQHIP_MODE_STE>);
        default: return go(rq_seam_kernel<RQHIP_MODE_ROTATION>);
    }
}
