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

// ---- bf16-split image for the filtered scan (FILT kernels, D = 32) ---------------------------------------------
// Only the ARGMIN of the distances is an output; ids, the gathered codeword and the loss are computed from it exactly.
// The FILT kernels therefore scan with approximate distances -- x = xh + xl + (rest), c = ch + cl + (rest) in bf16,
// x.c ~ xh.ch + xh.cl + xl.ch as three chains of v_mfma_f32_32x32x16_bf16 (fp32 accumulation; products of bf16 are
// exact in fp32) on the matrix cores proper, which run 5x faster than the f32-input form and beside the VALU epilogue
// instead of sharing its datapath -- and hand every row whose two smallest approximate distances are closer than a
// bound on the approximation error to the exact scan below (slow_argmin_row, the oracle's fp32 chain).  Same ids, bit
// for bit; see rq_tile for the bound.
// buffer = [image: 8 blocks of Kc 16-byte elements][csq: Kc floats]  (as large as the fp32 image)
//   block (plane * 2 + s) * 2 + h, element c = 8 bf16: j-th = plane (hi / lo) of C[kbase + c][d = 2 (8 s + j) + h]
// i.e. lane (il, h) finds, for K-step s, the same features 2 kk + h, kk = 8 s + j, that its row registers r[kk] hold.
typedef __bf16 rq_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 rq_bf16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void bf16_split(float v, __bf16 &hi, __bf16 &lo) {
    hi = (__bf16)v;                    // round to nearest even
    lo = (__bf16)(v - (float)hi);      // exact difference, rounded again: |v - hi - lo| <= 2^-18 |v|
}

template <int NT>
__device__ __forceinline__ void stage_codes_bf16(float *buf0, int buf_floats, int nbuf, const float *__restrict__ cb0,
                                                 const float *__restrict__ csq0, int csq_stride, int kbase, int Kc, int K) {
    constexpr int D = 32, d4n = D / 4, cstep = NT / d4n;
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
                __bf16 hx, lx, hy, ly, hz, lz, hw, lw;
                bf16_split(v[u].x, hx, lx); bf16_split(v[u].y, hy, ly);
                bf16_split(v[u].z, hz, lz); bf16_split(v[u].w, hw, lw);
                auto at = [&](int plane, int h) { return img + ((size_t)(((plane * 2 + s_blk) * 2 + h) * Kc + cc[u])) * 8 + j0; };
                *reinterpret_cast<rq_bf16x2 *>(at(0, 0)) = rq_bf16x2{hx, hz};   // features 4 d4, 4 d4 + 2 (h = 0)
                *reinterpret_cast<rq_bf16x2 *>(at(0, 1)) = rq_bf16x2{hy, hw};   // features 4 d4 + 1, 4 d4 + 3 (h = 1)
                *reinterpret_cast<rq_bf16x2 *>(at(1, 0)) = rq_bf16x2{lx, lz};
                *reinterpret_cast<rq_bf16x2 *>(at(1, 1)) = rq_bf16x2{ly, lw};
            }
        }
    }
    {
        int c2 = tid, b2 = 0;
        while (c2 >= Kc) { c2 -= Kc; ++b2; }
        while (b2 < nbuf) {
            buf0[b2 * buf_floats + 32 * Kc + c2] = (kbase + c2 < K) ? csq0[b2 * csq_stride + kbase + c2] : __builtin_inff();
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

// The same scan for D = 32 with every code row fetched as eight 16-byte loads (the dword version above is latency-bound:
// 32 dependent-issue loads per code).  The filtered kernels call it for ~0.4 % of the rows, so it has to be cheap; the
// FMA chain runs over d = 0, 1, 2, ... exactly as above.
__device__ __forceinline__ int slow_argmin_row32(const float (&r)[16], int j, float xsq_j, const float *__restrict__ cb_l,
                                                 const float *__restrict__ csq_l, int K) {
    const int lane = threadIdx.x & 63;
    float x0[16], x1[16];
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
        x0[kk] = __shfl(r[kk], j, 64);
        x1[kk] = __shfl(r[kk], j + 32, 64);
    }
    int nanidx = 0x7fffffff, lidx = 0x7fffffff;
    float lbest = __builtin_inff();
    for (int k = lane; k < K; k += 64) {
        const f32x4 *c = reinterpret_cast<const f32x4 *>(cb_l + (size_t)k * 32);
        f32x4 v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = c[q];
        const float cs = csq_l[k];
        float acc = 0.0f;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
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

// The filtered scan of Kc staged codes (see stage_codes_bf16): same (best, index, runner-up) tournament as
// scan_codes<16, true>, on distances whose dot product is the three-term bf16 split.
__device__ __forceinline__ void scan_codes_bf16(const rq_bf16x8 *__restrict__ img, const float *__restrict__ csq_s, int Kc,
                                                int kbase, int il, int h, const rq_bf16x8 (&xh)[2],
                                                const rq_bf16x8 (&xl)[2], float xsq, float &best, int &bidx,
                                                float &second, int t_begin = 0, int t_step = 1) {
    const int ntiles = Kc / 32;
    const f32x2 xsq2 = {xsq, xsq};
    auto lda = [&](int t, int blk) { return img[(size_t)(blk * 2 + h) * Kc + t * 32 + il]; };   // blk = plane * 2 + s
    const int t0 = t_begin < ntiles ? t_begin : 0;
    rq_bf16x8 a0 = lda(t0, 0), a1 = lda(t0, 1), a2 = lda(t0, 2), a3 = lda(t0, 3);
    for (int t = t_begin; t < ntiles; t += t_step) {
        const int tn = (t + t_step < ntiles) ? t + t_step : t;
        const float *cq = csq_s + t * 32 + 4 * h;
        f32x4 c4[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) c4[g] = *reinterpret_cast<const f32x4 *>(cq + 8 * g);
        f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, xh[0], acc, 0, 0, 0);   // ch . xh
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, xh[1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, xh[0], acc, 0, 0, 0);   // cl . xh
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3, xh[1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, xl[0], acc, 0, 0, 0);   // ch . xl
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, xl[1], acc, 0, 0, 0);
        // the next tile's code operands, into the same registers: in flight during the VALU epilogue below
        __builtin_amdgcn_sched_barrier(0);
        a0 = lda(tn, 0); a1 = lda(tn, 1); a2 = lda(tn, 2); a3 = lda(tn, 3);
        f32x2 dp[8];
#pragma unroll
        for (int pr = 0; pr < 8; ++pr)
            dp[pr] = rq_pk_add(xsq2, f32x2{c4[pr >> 1][2 * (pr & 1)], c4[pr >> 1][2 * (pr & 1) + 1]});
        float d[16];
#pragma unroll
        for (int pr = 0; pr < 8; ++pr) {
            const f32x2 v = __builtin_elementwise_fma(f32x2{-2.0f, -2.0f}, f32x2{acc[2 * pr], acc[2 * pr + 1]}, dp[pr]);
            d[2 * pr] = v.x;
            d[2 * pr + 1] = v.y;
        }
        // (min, runner-up) tree and top-down walk for the index: see scan_codes
        float lo1[8], hi1[8], lo2[4], hi2[4], hi3a, hi3b, t2, a07, b07, tmin;
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
        const float a01 = lo1[0], a45 = lo1[2], b01 = lo1[4], b45 = lo1[6];
        const float a03 = lo2[0], a47 = lo2[1], b03 = lo2[2], b47 = lo2[3];
        (void)a47; (void)b47;
        second = rq_min3(second, t2, rq_max(best, tmin));
        const bool c3 = a07 != tmin;
        const float q03 = c3 ? b03 : a03;
        const bool c2 = q03 != tmin;
        const float s01 = c3 ? b01 : a01, s45 = c3 ? b45 : a45;
        const float p01 = c2 ? s45 : s01;
        const bool c1 = p01 != tmin;
        const float u0a = c3 ? d[8] : d[0], u2a = c3 ? d[10] : d[2], u4a = c3 ? d[12] : d[4], u6a = c3 ? d[14] : d[6];
        const float u0 = c2 ? u4a : u0a, u2 = c2 ? u6a : u2a;
        const float e0 = c1 ? u2 : u0;
        const bool c0 = e0 != tmin;
        const int slot = (c3 ? 16 : 0) | (c2 ? 8 : 0) | (c1 ? 2 : 0) | (c0 ? 1 : 0);
        const int cand = kbase + t * 32 + 4 * h + slot;
        const bool better = tmin < best;
        best = better ? tmin : best;
        bidx = better ? cand : bidx;
    }
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
// FILT (KSTEPS = 16, FULLD): the scan runs on the bf16-split image (stage_codes_bf16 / scan_codes_bf16) and rows that are
//       too close to call go through the exact scan; never together with MARGIN (the margins are exact quantities).
template <int KSTEPS, int MODE, bool FULLD, int NT, bool COOP, bool MARGIN, bool FILT>
__device__ __forceinline__ void rq_tile(const RqFwdParams &p, float *smem, const float *csqmax_s, float *cand_s,
                                        long long tile, float (&r)[KSTEPS], int D, int buf_floats, int phase) {
    constexpr int KQ = KSTEPS / 4;
    constexpr bool TRACK2 = MARGIN || FILT;   // the runner-up distance is tracked
    static_assert(!FILT || (KSTEPS == 16 && FULLD && !MARGIN), "filtered scan: D = 32 only, no margins");
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
    // running output pointers of this lane (advanced per level: no 64-bit multiplies inside the level loop)
    int64_t *ids_ptr = p.ids + row;
    float *norm_ptr = p.embs_norm ? p.embs_norm + (size_t)row * L : nullptr;
    // (full-width kernels store float4 row chunks: pointer to the row; otherwise to this lane's first feature)
    float *embs_ptr = p.embs ? p.embs + (size_t)row * D + (FULLD ? 0 : h) : nullptr;
    float *resid_ptr = p.residuals ? p.residuals + (size_t)row * D + (FULLD ? 0 : h) : nullptr;
    RQ_STAMP(1);

    for (int l = 0; l < L; ++l) {
        RQ_STAMP(2 + 8 * l);
        const float csqmax_l = csqmax_s[l];

        // |x|^2 (quantize.py:114): parity accumulators, multiply and add separately rounded
        const float xsq = pair_sumsq<KSTEPS>(r);

        float best = __builtin_inff(), second = __builtin_inff();
        int bidx = 0x7fffffff;
        // filtered scan: this level's input rows as bf16 hi / lo planes, K-step s = features 2 (8 s + j) + h
        rq_bf16x8 xh[2], xl[2];
        if constexpr (FILT) {
#pragma unroll
            for (int sx = 0; sx < 2; ++sx)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    __bf16 hi, lo;
                    bf16_split(r[8 * sx + j], hi, lo);
                    xh[sx][j] = hi;
                    xl[sx][j] = lo;
                }
        }

        const float *buf = smem + (p.resident ? l * buf_floats : 0);
        for (int ch = 0; ch < p.nchunks; ++ch) {
            const int kbase = ch * Kc;
            if (!p.resident) {
                __syncthreads();  // previous chunk fully consumed
                if constexpr (FILT)
                    stage_codes_bf16<NT>(smem, buf_floats, 1, p.cb + (size_t)l * K * D, p.csq + (size_t)l * p.Kp, p.Kp,
                                         kbase, Kc, K);
                else
                    stage_codes<KSTEPS, NT>(smem, buf_floats, 1, p.cb + (size_t)l * K * D, p.csq + (size_t)l * p.Kp, p.Kp,
                                            kbase, Kc, K, D);
                __syncthreads();
            }
            if (active) {
                if constexpr (FILT)
                    scan_codes_bf16(reinterpret_cast<const rq_bf16x8 *>(buf), buf + KSTEPS * 2 * Kc, Kc, kbase, il, h, xh, xl,
                                    xsq, best, bidx, second, COOP ? wave : 0, COOP ? kCoopWaves : 1);
                else
                    scan_codes<KSTEPS, MARGIN>(reinterpret_cast<const f32x4 *>(buf), buf + KSTEPS * 2 * Kc, Kc, kbase, il, h,
                                               r, xsq, best, bidx, second, COOP ? wave : 0, COOP ? kCoopWaves : 1);
            }
        }

        RQ_STAMP(3 + 8 * l);
        // lanes (il,0) and (il,1) scanned disjoint code subsets: keep the smaller, ties -> lower index
        // (lexicographic (distance, index) minimum == first-index argmin over the union)
        if (active) {
            const float ob = shfl_xor32(best);
            const int oi = shfl_xor32(bidx);
            if (TRACK2) second = rq_min3(second, shfl_xor32(second), rq_max(best, ob));
            if (ob < best || (ob == best && oi < bidx)) {
                best = ob;
                bidx = oi;
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
            best = __builtin_inff();
            second = __builtin_inff();
            bidx = 0x7fffffff;
#pragma unroll
            for (int w = 0; w < kCoopWaves; ++w) {
                const float ov = cv[w * 32 + il];
                const int oi = ci[w * 32 + il];
                if (TRACK2) second = rq_min3(second, c2[w * 32 + il], rq_max(best, ov));
                if (ov < best || (ov == best && oi < bidx)) {
                    best = ov;
                    bidx = oi;
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
            if constexpr (FILT) {
                // Too close to call?  |d~ - d| for any code, d the oracle's fp32 distance and d~ the scanned one, is at
                // most 2 |x.c - (xh.ch + xh.cl + xl.ch)| + rounding: the dropped terms are <= 3 * 2^-18 sum|x_d c_d|, the
                // fp32 accumulation of 96 exact products <= ~2^-16.4 sum|x_d c_d|, the oracle's own chain 2^-19; with
                // sum|x_d c_d| <= |x| |c| that is < 2^-14.5 |x| |c| (measured maximum on config-2-like data: 2^-16.3,
                // tools/bf16_filter_study.py).  The argmin of d~ is the argmin of d whenever the two smallest d~ differ
                // by more than twice that (plus one ulp of d); the test uses 2^-12 |x| max|c| (2.8 times the bound, 20 times the measured
                // maximum) + 2^-20 (|x|^2 + max|c|^2) and also sends rows of
                // vanishing magnitude (bf16 denormals may be flushed) and any NaN to the exact scan.
                const float scale2 = xsq * csqmax_l;
                // (second term: the final rounding of d and d~ themselves, half an ulp of |d| <= 2 (|x|^2 + max|c|^2) each
                // -- it dominates when the row is much larger than every code or the reverse, where neighbouring codes'
                // distances differ by a few ulps only)
                const float T = 2.4414062e-4f * __builtin_sqrtf(scale2) + 9.5367432e-7f * guard;   // 2^-12, 2^-20
#ifndef RQ_FILT_NOSLOW   // (developer timing build, tools/ab_build.sh: how fast is the scan without its exact re-checks?)
                bad = bad || !((second - best) > T) || !(scale2 > 1.0e-30f);
#else
                (void)T;
#endif
            }
            unsigned long long badmask = __ballot(bad) & 0xffffffffull;
            if (badmask) {
                const float *cb_l = p.cb + (size_t)l * K * D;
                const float *csq_l = p.csq + (size_t)l * p.Kp;
                while (badmask) {
                    const int j = __builtin_ctzll(badmask);
                    badmask &= badmask - 1;
                    const float xj = __shfl(xsq, j, 64);
                    int res;
                    if constexpr (FILT) res = slow_argmin_row32(r, j, xj, cb_l, csq_l, K);
                    else res = slow_argmin_row<KSTEPS>(r, j, xj, cb_l, csq_l, K, p.D);
                    if (il == j) bidx = res;
                }
            }
            if (MARGIN) {
                // relative top-2 margin of this level's argmin (see include/rqhip.h); 0 for exact-scan rows
                const float cwin = p.resident ? buf[KSTEPS * 2 * Kc + bidx] : p.csq[(size_t)l * p.Kp + bidx];
                float m = (second - best) / (xsq + cwin);
                if (bad || m != m) m = 0.0f;
                if (writer && h == 0) p.tie_margin[(size_t)l * p.B + row] = m;
            }

            RQ_STAMP(4 + 8 * l);
            // codeword gather (quantize.py:101-102) for this lane's feature parity: from the staged LDS image
            // when the whole level is resident, else from global memory (L2)
            float e[KSTEPS];
            if (p.resident && !FILT) {
                const f32x4 *img = reinterpret_cast<const f32x4 *>(buf) + h * Kc + bidx;
#pragma unroll
                for (int q = 0; q < KQ; ++q) {
                    const f32x4 v = img[q * 2 * Kc];
                    e[4 * q + 0] = v.x; e[4 * q + 1] = v.y; e[4 * q + 2] = v.z; e[4 * q + 3] = v.w;
                }
            } else {
                const float *src = p.cb + ((size_t)l * K + bidx) * D + h;
#pragma unroll
                for (int kk = 0; kk < KSTEPS; ++kk) e[kk] = (FULLD || 2 * kk + h < D) ? src[2 * kk] : 0.0f;
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
                if (h == 0) *ids_ptr = (int64_t)bidx;
                if (norm_ptr) {  // uniform branch: the sqrt sequence is skipped when norms are not requested
                    const float onorm = __builtin_sqrtf(pair_sumsq<KSTEPS>(o));
                    if (h == 0) norm_ptr[l] = onorm;
                }
                if (resid_ptr) {
                    if (FULLD) {
                        store_pair_row<KSTEPS>(resid_ptr, h, r);
                    } else {
#pragma unroll
                        for (int kk = 0; kk < KSTEPS; ++kk)
                            if (2 * kk + h < D) resid_ptr[2 * kk] = r[kk];
                    }
                }
                if (embs_ptr) {
                    if (FULLD) {
                        store_pair_row<KSTEPS>(embs_ptr, h, o);
                    } else {
#pragma unroll
                        for (int kk = 0; kk < KSTEPS; ++kk)
                            if (2 * kk + h < D) embs_ptr[2 * kk] = o[kk];
                    }
                }
            }
            ids_ptr += p.B;
            if (resid_ptr) resid_ptr += level_stride;
            if (embs_ptr) embs_ptr += level_stride;
#pragma unroll
            for (int kk = 0; kk < KSTEPS; ++kk) {
                es[kk] = (l == 0) ? o[kk] : es[kk] + o[kk];
                r[kk] = r[kk] - o[kk];  // rqvae.py:130
            }
        }
    }

    RQ_STAMP(100);
    if (writer) {
        if (h == 0 && p.loss) p.loss[row] = lsum;
        if (p.emb_sum) {
            if (FULLD) {
                store_pair_row<KSTEPS>(p.emb_sum + (size_t)row * D, h, es);
            } else {
                float *dst = p.emb_sum + (size_t)row * D + h;
#pragma unroll
                for (int kk = 0; kk < KSTEPS; ++kk)
                    if (2 * kk + h < D) dst[2 * kk] = es[kk];
            }
        }
    }
    RQ_STAMP(101);
}

template <int KSTEPS, int MODE, bool FULLD, int NT, bool MARGIN, bool FILT = false>
__global__ __launch_bounds__(NT) void rq_forward_kernel(const RqFwdParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float *smem = reinterpret_cast<float *>(smem_raw);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int il = lane & 31;
    const int h = lane >> 5;
    const int D = FULLD ? 2 * KSTEPS : p.D;
    const int K = p.K, Kc = p.Kc, L = p.L;
    const int buf_floats = Kc * (KSTEPS * 2 + 1);
    constexpr int kWavesPerWg = NT / RQ_WAVE;
    const long long total_waves = (long long)gridDim.x * kWavesPerWg;
    // round `it`: waves are enumerated wave-major (wave w of every workgroup before wave w+1), so a partly
    // filled last round spreads over all CUs instead of filling the first workgroups only
    const long long wave_slot = (long long)wave * gridDim.x + blockIdx.x;

    // rows of a tile as this lane fetches them: full-width kernels take their half of the row as float4s ("raw",
    // see rows_to_pairs), the others their features d = 2 kk + h one by one
    auto load_rows = [&](long long tile, float(&v)[KSTEPS]) {
        const long long row = tile * 32 + il;
        const long long rowc = (tile < p.n_tiles && row < p.B) ? row : (p.B - 1);
        if (FULLD) {
            const f32x4 *src = reinterpret_cast<const f32x4 *>(p.res0 + (size_t)rowc * D + h * KSTEPS);
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
    };
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
    float *csqmax_s = smem + (p.resident ? L : 1) * buf_floats;
    float *cand_s = csqmax_s + 16;  // cooperative-tile candidates and counters (kCoopLdsFloats)
    if (tid < L) csqmax_s[tid] = p.csqmax[tid];
    if (tid < kCoopSteps) reinterpret_cast<int *>(cand_s + 2 * kCoopCandFloats)[tid] = 0;
    RQ_STAMP(200);
    if (p.resident) {
        if constexpr (FILT) stage_codes_bf16<NT>(smem, buf_floats, L, p.cb, p.csq, p.Kp, 0, Kc, K);
        else stage_codes<KSTEPS, NT>(smem, buf_floats, L, p.cb, p.csq, p.Kp, 0, Kc, K, D);
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
        if (p.resident && !active) break;
        float r[KSTEPS];
        unpack_rows(rn, r);
        if (it + 1 < p.n_iter) load_rows(tile + total_waves, rn);
        rq_tile<KSTEPS, MODE, FULLD, NT, false, MARGIN, FILT>(p, smem, csqmax_s, cand_s, active ? tile : p.n_tiles, r, D, buf_floats, 0);
        RQ_TRACE(trace_slot);
        ++trace_slot;
    }
    // cooperative tiles (resident mode only), one per workgroup at a time, by the first four waves -- the oldest
    // wave of each SIMD, which finishes its own tile of a full round first
    if (wave < kCoopWaves) {
        int phase = 0;
        // (fetching the next cooperative tile's rows under the current one was measured: no gain)
        for (long long tile = p.coop_first + blockIdx.x; tile < p.n_tiles; tile += gridDim.x) {
            float raw[KSTEPS], r[KSTEPS];
            load_rows(tile, raw);
            unpack_rows(raw, r);
            rq_tile<KSTEPS, MODE, FULLD, NT, true, MARGIN, FILT>(p, smem, csqmax_s, cand_s, tile, r, D, buf_floats, phase);
            phase += L;
            RQ_TRACE(trace_slot);
            ++trace_slot;
        }
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

// threads per workgroup by register appetite: <= 168 VGPRs at KSTEPS <= 16 (3 waves/SIMD), 256 at 32, 512 at 64
template <int KSTEPS>
struct WgThreads { static constexpr int value = KSTEPS <= 16 ? 768 : KSTEPS == 32 ? 512 : 256; };

template <int KSTEPS>
static int launch_mode(const RqFwdParams &p, int mode, int grid, size_t lds, hipStream_t s) {
    constexpr int NT = WgThreads<KSTEPS>::value;
    auto go = [&](auto kern) -> int {
        // raise the dynamic-LDS limit once per instantiation (not a stream operation: keep it out of the per-call
        // path and out of hipGraph captures); the limit only ever grows
        static LdsGrant grant;
        RQ_RETURN_IF_HIP(grant.ensure(reinterpret_cast<const void *>(kern), (int)kLdsBudget));
        profile_begin(s);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(NT), lds, s, p);
        profile_end(s);
        RQ_CHECK_LAUNCH("rq_forward_kernel");
        return 0;
    };
    // full-width kernels move rows as float4s: every row pointer must be 16-byte aligned (rows are 8*KSTEPS bytes)
    auto al16 = [](const void *q) { return (reinterpret_cast<uintptr_t>(q) & 15u) == 0; };
    const bool full = p.D == 2 * KSTEPS && al16(p.res0) && al16(p.embs) && al16(p.residuals) && al16(p.emb_sum);
    const bool margin = p.tie_margin != nullptr;
    // D = 32, aligned rows, no margins wanted: the filtered scan (bf16-split matrix products + exact scan of the rows that
    // are too close to call).  RQ_NO_FILTER=1 keeps the all-fp32 scan (developer A/B switch, read once).
    static const bool use_filter = getenv("RQ_NO_FILTER") == nullptr;
    if constexpr (KSTEPS == 16) {
        if (full && !margin && use_filter) {
            switch (mode) {
                case RQHIP_MODE_EVAL: return go(rq_forward_kernel<16, RQHIP_MODE_EVAL, true, NT, false, true>);
                case RQHIP_MODE_STE: return go(rq_forward_kernel<16, RQHIP_MODE_STE, true, NT, false, true>);
                case RQHIP_MODE_ROTATION: return go(rq_forward_kernel<16, RQHIP_MODE_ROTATION, true, NT, false, true>);
            }
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

extern "C" int rqhip_rq_forward(const float *res0, int64_t B, int D, const float *codebooks, int L, int K,
                                int mode, float beta, int64_t *ids, float *embs, float *residuals,
                                float *emb_sum, float *loss, float *embs_norm, float *tie_margin, void *workspace,
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
    p.embs_norm = embs_norm; p.tie_margin = tie_margin;
    p.B = B; p.n_tiles = (B + 31) / 32; p.D = D; p.L = L; p.K = K; p.Kp = Kp; p.beta = beta;
    const size_t level_bytes = (size_t)Kp * (Dp + 1) * sizeof(float);
    if (level_bytes * L + 64 + kCoopLdsFloats * sizeof(float) <= (size_t)kLdsBudget) {
        p.resident = 1; p.Kc = Kp; p.nchunks = 1;
    } else {
        p.resident = 0;
        // one workgroup per CU (its waves fill the register file), so a chunk may use the whole LDS
        int kc = (int)(((size_t)kLdsBudget - 64 - kCoopLdsFloats * sizeof(float)) / ((size_t)(Dp + 1) * sizeof(float)));
        kc &= ~63;
        if (kc > Kp) kc = Kp;
        if (kc < 64) kc = 64;
        p.Kc = kc; p.nchunks = (Kp + kc - 1) / kc;
    }
    const size_t lds = (size_t)p.Kc * (Dp + 1) * sizeof(float) * (p.resident ? L : 1) + 16 * sizeof(float) +
                       (size_t)kCoopLdsFloats * sizeof(float);
    const int cus = cu_count();
    const int wg_per_cu = 1;  // 768 / 512 / 256 threads at <= 168 / 256 / 512 VGPRs: one workgroup fills a CU
    const int waves_per_wg = (ksteps <= 16 ? 768 : ksteps == 32 ? 512 : 256) / RQ_WAVE;
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
    static const bool coop_tail = getenv("RQ_NO_COOP_TAIL") == nullptr;  // developer A/B switch, read once
    if (!all_coop && p.resident && coop_tail) {
        const long long rem = p.n_tiles % ((long long)grid * 4);
        if (rem > 0 && rem <= grid) p.coop_first = p.n_tiles - rem;
    }
    p.n_iter = (int)((p.coop_first + total_waves - 1) / total_waves);

    switch (ksteps) {
        case 4: return launch_mode<4>(p, mode, grid, lds, s);
        case 8: return launch_mode<8>(p, mode, grid, lds, s);
        case 16: return launch_mode<16>(p, mode, grid, lds, s);
        case 32: return launch_mode<32>(p, mode, grid, lds, s);
        default: return launch_mode<64>(p, mode, grid, lds, s);
    }
}
