// wgrad_split.hip -- the weight gradient of a bias-free Linear(+ReLU) layer on the bf16 matrix cores, without narrowing
// the arithmetic (gfx950).  SURVEY.md section 8 row f2; reference modules/encoder.py:25-38 (autograd of relu(x W^T)).
//
// dW[n,k] = sum_m g_pre[m,n] x[m,k] reduces over the M = 100 000 batch rows into a small output.  wgrad.hip does it with
// v_mfma_f32_32x32x2_f32 and is bound by that pipe (505 us of matrix time for the 512 x 768 layer at the fp32 peak).
// Here every fp32 operand is split into THREE bf16 pieces, v = h + m + l EXACTLY (8 + 8 + 8 significant bits), and the
// product is formed from the six piece products that matter,
//     g x  ~  gh xh + gh xm + gm xh + gh xl + gl xh + gm xm          (dropped: gm xl + gl xm + gl xl <= 2^-23 |g x|),
// each a v_mfma_f32_32x32x16_bf16 (products of two bf16 are exact in fp32; accumulation in fp32): six instructions of 32
// cycles do the work of eight of 64 cycles -- 2.7x less matrix time -- and the dropped terms are below one fp32 rounding
// of the product itself.  What it gives up: the summation order is the matrix pipe's, not a chain the oracle can restate,
// so dW is reproducible run to run (fixed instruction order, fixed reduction tree) but checked against fp64 with a
// tolerance (tests/test_gpu_wgrad.py: no less exact than the library's fp32 GEMM), not bit for bit.
// RQHIP_WGRAD_FP32 (rqhip_linear_wgrad_ex) keeps wgrad.hip's oracle-exact kernel.
//
// Round 4, the product path (NP = 2, rqhip_linear_wgrad_f16): TWO fp16 pieces per operand and the three products hh + hm + mh
// (csrc/gemm_split.hip, RQHIP_SPLIT_F16X2).  The reduction runs over the batch rows, so the exact power-of-two scale has to be
// constant along them: every COLUMN of g_pre and of x is multiplied by 2^-e, e = exponent of the column's largest |value| - 14 (from
// the epilogue that wrote the matrix, or rqhip_maxima), and dW[n, k] is multiplied back by 2^(e_n + e_k).  Half the matrix
// instructions, two thirds of the LDS bytes; error against fp64 below the three-piece kernel's (tests/test_gpu_wgrad.py).
//
// Mapping
//   * A = g_pre^T (32 n x 16 m), B = x (16 m x 32 k): a lane's operand is EIGHT CONSECUTIVE ROWS of one column, so the
//     rows are transposed on the way into LDS: a staging thread takes 4 rows x 4 columns (four 16-byte loads per tensor,
//     coalesced along the row), applies the ReLU mask, splits, and writes per column and piece one 8-byte half of the
//     16-byte element [piece][row octet][position of the column] -- which a lane then fetches with one ds_read_b128;
//     the column order in LDS is permuted so that both the strided writes and the reads are free of bank conflicts.
//   * a workgroup owns an Nt x Kt block of dW and a contiguous range of rows, streams 16-row stages through a
//     double-buffered LDS image (98 KB at 256 x 256), a wave owns (32 TA) x (32 TB); per stage and wave: 3 (TA + TB)
//     operand reads, 6 TA TB matrix instructions.  Row ranges are reduced by wgrad.hip's balanced tree (second kernel).
//   * g_pre is written back once (the k-slab-0 workgroups) for the data-gradient GEMM that follows, as in wgrad.hip.
#include "rqhip_common.h"

namespace rqhip {

typedef float ws_f32x16 __attribute__((ext_vector_type(16)));
typedef float ws_f32x4 __attribute__((ext_vector_type(4)));
typedef float ws_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 ws_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 ws_bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned ws_u32x2 __attribute__((ext_vector_type(2)));

constexpr int kWsRows = 16;   // rows per LDS stage = one K step of the matrix instruction

// developer-only phase-skipping probes (tools/ab_build.sh <name> wgrad_split.hip -DWS_PROBE=<bits>; results are WRONG with any bit
// set): 1 no split / LDS writes, 2 no global loads, 4 no matrix instructions, 8 no LDS reads, 16 no stage barriers
#ifndef WS_PROBE
#define WS_PROBE 0
#endif

struct WgradSplitParams {
    const float *g, *y, *x;
    float *gm, *out;
    long long M;
    int N, K;
    int nslab_n, nslab_k, msplit;
    long long n_chunks;   // ceil(M / 32): the row ranges are cut on wgrad.hip's 32-row granules
    const unsigned *g_max, *x_max;   // NP == 2: [N], [K] bit patterns of the columns' largest |value| (rqhip_maxima / an epilogue)
};

// (a, b) -> three dwords, each the packed bf16 pieces {piece(a), piece(b)}; a = h + m + l exactly (likewise b)
__device__ __forceinline__ void ws_split2(float a, float b, unsigned &h, unsigned &m, unsigned &l) {
    const ws_bf16x2 hh = __builtin_convertvector(ws_f32x2{a, b}, ws_bf16x2);
    h = __builtin_bit_cast(unsigned, hh);
    const float ra = a - __builtin_bit_cast(float, h << 16), rb = b - __builtin_bit_cast(float, h & 0xffff0000u);
    const ws_bf16x2 mm = __builtin_convertvector(ws_f32x2{ra, rb}, ws_bf16x2);
    m = __builtin_bit_cast(unsigned, mm);
    const float sa = ra - __builtin_bit_cast(float, m << 16), sb = rb - __builtin_bit_cast(float, m & 0xffff0000u);
    const ws_bf16x2 ll = __builtin_convertvector(ws_f32x2{sa, sb}, ws_bf16x2);
    l = __builtin_bit_cast(unsigned, ll);
}

typedef _Float16 ws_f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 ws_f16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void ws_split2_f16(float a, float b, unsigned &h, unsigned &m) {
    const ws_f16x2 hh = __builtin_convertvector(ws_f32x2{a, b}, ws_f16x2);
    h = __builtin_bit_cast(unsigned, hh);
    const ws_f32x2 hf = __builtin_convertvector(hh, ws_f32x2);
    const ws_f16x2 mm = __builtin_convertvector(ws_f32x2{a - hf.x, b - hf.y}, ws_f16x2);
    m = __builtin_bit_cast(unsigned, mm);
}
// the scale exponent of a column whose largest |value| has these bits (csrc/gemm_split.hip:gs_exp_of_bits: the scaled maximum lies in
// [2^14, 2^15), the top of fp16's range); 0 for 0 / inf / nan
__device__ __forceinline__ int ws_exp_of_bits(unsigned b) {
    b &= 0x7fffffffu;
    const int e = (int)(b >> 23);
    return (b == 0u || e == 255) ? 0 : (e == 0 ? -126 : e - 127) - 14;
}

// Loads and waits placed by hand for the pipelined part of the main loop (see there): an asm load is invisible to the compiler's
// s_waitcnt pass, ws_await<N> is the wait -- "at most N loads still in flight" -- and ties the four registers of a set to itself.
__device__ __forceinline__ ws_f32x4 ws_load4(const float *p) {
    ws_f32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(p));
    return v;
}
template <int N>
__device__ __forceinline__ void ws_await(ws_f32x4 (&r)[4]) {
    asm volatile("s_waitcnt vmcnt(%4)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]) : "n"(N) : "memory");
}

// TA x TB tiles per wave, WA x WB waves per workgroup; MASK: y given; NP: pieces per operand (3 bf16 / 2 fp16 under column scales)
// (the body of a workgroup: `block` is its number inside ITS launch or, in the job-table launch below, inside its job)
template <int TA, int TB, int WA, int WB, bool MASK, int NP>
__device__ __forceinline__ void wgrad_split_body(const WgradSplitParams &p, const int block) {
    constexpr int Nt = 32 * TA * WA, Kt = 32 * TB * WB, NT = 64 * WA * WB;
    constexpr int UNITS = Nt + Kt;                 // staging units of 4 rows x 4 columns per stage: Nt for g, Kt for x
    constexpr int UQ = (UNITS + NT - 1) / NT;      // per thread
    // LDS image of one operand: [piece][row octet][position] x 16 bytes, column c = 4 q + g at position g * S + q with
    // S = W / 4 + 4.  A staging lane owns columns 4 cq .. 4 cq + 3, so for a fixed g the 64 lanes of a wave write 64
    // CONSECUTIVE positions (with [column] order they wrote 64 bytes apart: a 16-way bank conflict on every one of the 12
    // ds_write_b64 -- 1.7 us of the 3.2 us a stage took); the 16 lanes an operand read serves per cycle (columns
    // c0 .. c0 + 15) hit positions g * S + q0 .. q0 + 3, g < 4: distinct modulo 16 because S = 4 (mod 16).
    constexpr int SG = Nt / 4 + 4, SX = Kt / 4 + 4;
    constexpr int PART_G = NP * 2 * 4 * SG * 4, PART_X = NP * 2 * 4 * SX * 4;   // dwords
    extern __shared__ __attribute__((aligned(16))) char ws_smem[];
    unsigned *sbuf = reinterpret_cast<unsigned *>(ws_smem);      // [2][PART_G + PART_X]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int il = lane & 31, h = lane >> 5;
    const int wa = wave / WB, wb = wave % WB;
    const int nslabs = p.nslab_n * p.nslab_k;
    int slab, split;
    if ((p.msplit & 7) == 0) {   // slabs of one row range back to back on one XCD (they share its strips in that L2)
        const int xcd = block & 7, j = block >> 3;
        slab = j % nslabs;
        split = (j / nslabs) * 8 + xcd;
    } else {
        slab = block % nslabs;
        split = block / nslabs;
    }
    const int slab_n = slab / p.nslab_k, slab_k = slab % p.nslab_k;
    const int n0 = slab_n * Nt, k0 = slab_k * Kt;
    const long long r_begin = (p.n_chunks * split / p.msplit) * 32;
    long long r_end = (p.n_chunks * (split + 1) / p.msplit) * 32;
    if (r_end > p.M) r_end = p.M;
    const long long n_stage = (r_end - r_begin + kWsRows - 1) / kWsRows;
    const bool write_back = MASK && p.gm != nullptr && slab_k == 0;

    ws_f32x16 acc[TA][TB];
#pragma unroll
    for (int t = 0; t < TA; ++t)
#pragma unroll
        for (int u = 0; u < TB; ++u)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][u][r] = 0.0f;

    ws_f32x4 rv0[UQ][4], rv1[UQ][4], ry[MASK ? UQ : 1][4];   // (rv1: the second staging set of the two-stage prefetch, see the main loop)
    int ue[NP == 2 ? UQ : 1][4];   // NP == 2: exponents of the four columns of each of this thread's staging units
#pragma unroll
    for (int q = 0; q < (NP == 2 ? UQ : 0); ++q) {
        const int u = tid + q * NT;
        const bool isg = u < Nt;
        const int idx = isg ? u : u - Nt, W = isg ? Nt : Kt;
        const int cq = idx % (W / 4);
        const unsigned *mxp = isg ? p.g_max : p.x_max;
        const int c0 = isg ? n0 : k0;
#pragma unroll
        for (int cc = 0; cc < 4; ++cc)
            ue[q][cc] = (mxp && ((UNITS % NT == 0) || u < UNITS)) ? ws_exp_of_bits(mxp[c0 + 4 * cq + cc]) : 0;
    }
    // Staging loads.  A thread's unit (4 rows x 4 columns) sits at the same place of every stage, 16 rows further down: the
    // four row pointers are formed ONCE and advanced by 16 rows per stage (one 64-bit add each) -- recomputing `(row0 + 4 rq + j)
    // * ld + c0 + 4 cq` with its 64-bit multiplies, the row test and the predication for every load of every stage was ~100
    // integer instructions per stage next to the 64 of the split itself.  Whole stages (every row inside the range) take this
    // path with plain loads; only the range's last, partial stage takes `fetch_tail` (predicated, addresses from scratch).
    const long long n_full = (r_end - r_begin) / kWsRows;           // stages whose 16 rows all lie inside [r_begin, r_end)
    const float *cur[UQ][4];
    const float *ycur[MASK ? UQ : 1][4];
    size_t step16[UQ];
#pragma unroll
    for (int q = 0; q < UQ; ++q) {
        const int u = tid + q * NT;
        const bool live = (UNITS % NT == 0) || u < UNITS;
        const bool isg = u < Nt;
        const int idx = live ? (isg ? u : u - Nt) : 0, W = isg ? Nt : Kt;
        const int cq = idx % (W / 4), rq = idx / (W / 4);
        const int ld = isg ? p.N : p.K, c0 = isg ? n0 : k0;
        step16[q] = (size_t)kWsRows * ld;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const size_t off = (size_t)(r_begin + 4 * rq + j) * ld + c0 + 4 * cq;
            cur[q][j] = (isg ? p.g : p.x) + off;
            if (MASK) ycur[q][j] = p.y + (isg ? off : (size_t)0);
        }
    }
    auto fetch_tail = [&](long long stage, ws_f32x4 (&rv)[UQ][4]) {
        const long long row0 = r_begin + stage * kWsRows;
#pragma unroll
        for (int q = 0; q < UQ; ++q) {
            const int u = tid + q * NT;
            const bool live = (UNITS % NT == 0) || u < UNITS;
            const bool isg = u < Nt;                          // (wave-uniform: Nt is a multiple of 64)
            const int idx = live ? (isg ? u : u - Nt) : 0, W = isg ? Nt : Kt;
            const int cq = idx % (W / 4), rq = idx / (W / 4);
            const float *base = isg ? p.g : p.x;
            const int ld = isg ? p.N : p.K, c0 = isg ? n0 : k0;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const long long row = row0 + 4 * rq + j;
                const bool ok = live && row < r_end;
                const size_t off = (size_t)(ok ? row : 0) * ld + c0 + 4 * cq;
                rv[q][j] = ok ? *reinterpret_cast<const ws_f32x4 *>(base + off) : ws_f32x4{0.f, 0.f, 0.f, 0.f};
                if (MASK) ry[q][j] = (ok && isg) ? *reinterpret_cast<const ws_f32x4 *>(p.y + off) : ws_f32x4{1.f, 1.f, 1.f, 1.f};
            }
        }
    };
    // (stages are fetched in order 0, 1, 2, ...: `cur` always points at the next one)
    auto fetch = [&](long long stage, ws_f32x4 (&rv)[UQ][4]) {
        if (stage >= n_full) {
            fetch_tail(stage, rv);
            return;
        }
#pragma unroll
        for (int q = 0; q < UQ; ++q) {
            const int u = tid + q * NT;
            const bool live = (UNITS % NT == 0) || u < UNITS;
            const bool isg = u < Nt;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if constexpr ((WS_PROBE & 2) != 0) {
                    rv[q][j] = ws_f32x4{1.f + (float)stage, 2.f, 3.f, 4.f};
                    if (MASK) ry[q][j] = ws_f32x4{1.f, 1.f, 1.f, 1.f};
                } else {
                    const ws_f32x4 v = *reinterpret_cast<const ws_f32x4 *>(cur[q][j]);
                    rv[q][j] = live ? v : ws_f32x4{0.f, 0.f, 0.f, 0.f};
                    if (MASK) {
                        const ws_f32x4 yv = *reinterpret_cast<const ws_f32x4 *>(ycur[q][j]);
                        ry[q][j] = (live && isg) ? yv : ws_f32x4{1.f, 1.f, 1.f, 1.f};
                    }
                }
                cur[q][j] += step16[q];
                if (MASK && isg) ycur[q][j] += step16[q];
            }
        }
    };
    auto stash = [&](long long stage, int buf, ws_f32x4 (&rv)[UQ][4]) {
        const long long row0 = r_begin + stage * kWsRows;
        unsigned *dst = sbuf + buf * (PART_G + PART_X);
        if (WS_PROBE & 1) return;
#pragma unroll
        for (int q = 0; q < UQ; ++q) {
            const int u = tid + q * NT;
            if (UNITS % NT != 0 && u >= UNITS) continue;
            const bool isg = u < Nt;
            const int idx = isg ? u : u - Nt, W = isg ? Nt : Kt;
            const int cq = idx % (W / 4), rq = idx / (W / 4);
            if (MASK && isg) {   // threshold_backward(gy, y, 0): 0 where y <= 0
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    rv[q][j].x = ry[q][j].x <= 0.0f ? 0.0f : rv[q][j].x;
                    rv[q][j].y = ry[q][j].y <= 0.0f ? 0.0f : rv[q][j].y;
                    rv[q][j].z = ry[q][j].z <= 0.0f ? 0.0f : rv[q][j].z;
                    rv[q][j].w = ry[q][j].w <= 0.0f ? 0.0f : rv[q][j].w;
                }
                if (write_back) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const long long row = row0 + 4 * rq + j;
                        if (row < r_end) *reinterpret_cast<ws_f32x4 *>(p.gm + (size_t)row * p.N + n0 + 4 * cq) = rv[q][j];
                    }
                }
            }
            // element [piece][octet = rq >> 1][position of the column] is 16 bytes = rows 8 octet .. 8 octet + 7; this unit
            // fills the half (rq & 1) of it for its four columns and every piece
            unsigned *part = dst + (isg ? 0 : PART_G);
            const int oct = rq >> 1, half = rq & 1, S4 = 4 * (isg ? SG : SX);
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) {
                unsigned h01, m01, l01 = 0u, h23, m23, l23 = 0u;
                if constexpr (NP == 2) {
                    ws_split2_f16(ldexpf(rv[q][0][cc], -ue[q][cc]), ldexpf(rv[q][1][cc], -ue[q][cc]), h01, m01);
                    ws_split2_f16(ldexpf(rv[q][2][cc], -ue[q][cc]), ldexpf(rv[q][3][cc], -ue[q][cc]), h23, m23);
                } else {
                    ws_split2(rv[q][0][cc], rv[q][1][cc], h01, m01, l01);
                    ws_split2(rv[q][2][cc], rv[q][3][cc], h23, m23, l23);
                }
                const int pos = cc * (S4 / 4) + cq;      // column 4 cq + cc
                *reinterpret_cast<ws_u32x2 *>(part + ((0 * 2 + oct) * S4 + pos) * 4 + 2 * half) = ws_u32x2{h01, h23};
                *reinterpret_cast<ws_u32x2 *>(part + ((1 * 2 + oct) * S4 + pos) * 4 + 2 * half) = ws_u32x2{m01, m23};
                if (NP == 3) *reinterpret_cast<ws_u32x2 *>(part + ((2 * 2 + oct) * S4 + pos) * 4 + 2 * half) = ws_u32x2{l01, l23};
            }
        }
    };

    // Pipeline.  (1) Loads run two stages ahead: the rows of stage c + 2 are requested in iteration c and consumed
    // (masked, split, written to LDS) in iteration c + 1.  (2) The two waves of a SIMD work OUT OF PHASE inside the one
    // barrier interval of a stage: the first half of the workgroup's waves (one per SIMD) stages first and multiplies
    // second, the other half multiplies first and stages second -- so each SIMD always has one wave on the matrix
    // cores and one on the VALU / LDS.  A wave issues in order, and with every wave in the same phase (all split, then
    // all multiply) nothing overlapped: 3.2 us per 16-row stage measured, the SUM of 1.3 us of matrix time and the
    // staging.  Both phases read buffer `buf` and write buffer `buf ^ 1`, which every wave left at the last barrier.
    const bool stage_first = wave < (WA * WB) / 2;
    auto multiply = [&](int buf) {
        const ws_bf16x8 *gA = reinterpret_cast<const ws_bf16x8 *>(sbuf + buf * (PART_G + PART_X));
        const ws_bf16x8 *xB = reinterpret_cast<const ws_bf16x8 *>(sbuf + buf * (PART_G + PART_X) + PART_G);
        ws_bf16x8 a[TA][NP], b[TB][NP];
#pragma unroll
        for (int t = 0; t < TA; ++t)
#pragma unroll
            for (int pc = 0; pc < NP; ++pc)
                a[t][pc] = (WS_PROBE & 8) ? ws_bf16x8{(__bf16)(float)(t + buf), 1, 2, 3, 4, 5, 6, 7}
                                          : gA[(pc * 2 + h) * (4 * SG) + (il & 3) * SG + ((wa * 32 * TA + 32 * t + il) >> 2)];
#pragma unroll
        for (int u = 0; u < TB; ++u)
#pragma unroll
            for (int pc = 0; pc < NP; ++pc)
                b[u][pc] = (WS_PROBE & 8) ? ws_bf16x8{(__bf16)(float)(u + buf), 1, 2, 3, 4, 5, 6, 7}
                                          : xB[(pc * 2 + h) * (4 * SX) + (il & 3) * SX + ((wb * 32 * TB + 32 * u + il) >> 2)];
        // smallest products first (their sum is formed before it meets the large ones)
#pragma unroll
        for (int t = 0; t < TA; ++t)
#pragma unroll
            for (int u = 0; u < TB; ++u) {
                ws_f32x16 c16 = acc[t][u];
                if constexpr ((WS_PROBE & 4) != 0) {
                    c16[0] += (float)a[t][0][0] + (float)b[u][0][1] + (float)a[t][NP - 1][2] + (float)b[u][NP - 1][3];
                } else if constexpr (NP == 2) {
                    c16 = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(ws_f16x8, a[t][1]), __builtin_bit_cast(ws_f16x8, b[u][0]), c16, 0, 0, 0);   // m h
                    c16 = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(ws_f16x8, a[t][0]), __builtin_bit_cast(ws_f16x8, b[u][1]), c16, 0, 0, 0);   // h m
                    c16 = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(ws_f16x8, a[t][0]), __builtin_bit_cast(ws_f16x8, b[u][0]), c16, 0, 0, 0);   // h h
                } else {
                    c16 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[t][1], b[u][1], c16, 0, 0, 0);   // m m
                    c16 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[t][NP - 1], b[u][0], c16, 0, 0, 0);   // l h
                    c16 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[t][0], b[u][NP - 1], c16, 0, 0, 0);   // h l
                    c16 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[t][1], b[u][0], c16, 0, 0, 0);   // m h
                    c16 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[t][0], b[u][1], c16, 0, 0, 0);   // h m
                    c16 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[t][0], b[u][0], c16, 0, 0, 0);   // h h
                }
                acc[t][u] = c16;
            }
    };
    constexpr bool kTwoSets = !MASK && UQ == 1 && WS_PROBE == 0;
    if constexpr (!kTwoSets) {
        if (n_stage > 0) {
            fetch(0, rv0);
            stash(0, 0, rv0);
            if (n_stage > 1) fetch(1, rv0);
        }
        __syncthreads();
        for (long long c = 0; c < n_stage; ++c) {
            const int buf = (int)(c & 1);
            if (stage_first) {
                if (c + 1 < n_stage) stash(c + 1, buf ^ 1, rv0);
                if (c + 2 < n_stage) fetch(c + 2, rv0);
            }
            multiply(buf);
            if (!stage_first) {
                if (c + 1 < n_stage) stash(c + 1, buf ^ 1, rv0);
                if (c + 2 < n_stage) fetch(c + 2, rv0);
            }
            if (!(WS_PROBE & 16)) __syncthreads();
        }
    } else {
        // The product path (f16 pieces, no mask, one staging unit per thread): TWO staging sets, the rows of stage c + 3 requested
        // while stage c is multiplied -- a load has two barrier intervals to arrive.  With one set (rounds 3-4: requested in iteration
        // c, split in iteration c + 1) the phase-skipping probe without global loads ran in 200 us against 313 for dW [512, 768]
        // (profiles/r05_wgrad_prefetch.txt): a third of the kernel was waiting for its rows.  In the steady part the loads and waits
        // are placed by hand (ws_load4 / ws_await<4>: "all but the other set's four loads have arrived") because the compiler's wait
        // placement covers every load in flight once the refill sits behind a branch; that part is straight-line per wave role --
        // two copies of the loop, same barriers -- so that no register of a set is copied while its load is in flight.  Whole stages
        // only (plain loads); the range's first / last stages and short ranges take the compiler-managed form below.
        auto load_set = [&](ws_f32x4 (&r)[UQ][4]) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                r[0][j] = ws_load4(cur[0][j]);
                cur[0][j] += step16[0];
            }
        };
        long long c = 0;
        if (n_full >= 5) {
            load_set(rv0);                    // stage 0
            load_set(rv1);                    // stage 1
            ws_await<4>(rv0[0]);
            stash(0, 0, rv0);
            load_set(rv0);                    // stage 2
            __syncthreads();
            if (stage_first) {
                for (; c + 4 < n_full; c += 2) {
                    ws_await<4>(rv1[0]);
                    stash(c + 1, 1, rv1);
                    load_set(rv1);            // stage c + 3
                    multiply(0);
                    __syncthreads();
                    ws_await<4>(rv0[0]);
                    stash(c + 2, 0, rv0);
                    load_set(rv0);            // stage c + 4
                    multiply(1);
                    __syncthreads();
                }
                ws_await<0>(rv1[0]);
                ws_await<0>(rv0[0]);
            } else {
                for (; c + 4 < n_full; c += 2) {
                    multiply(0);
                    ws_await<4>(rv1[0]);
                    stash(c + 1, 1, rv1);
                    load_set(rv1);            // stage c + 3
                    __syncthreads();
                    multiply(1);
                    ws_await<4>(rv0[0]);
                    stash(c + 2, 0, rv0);
                    load_set(rv0);            // stage c + 4
                    __syncthreads();
                }
                ws_await<0>(rv1[0]);
                ws_await<0>(rv0[0]);
            }
        } else {
            if (n_stage > 0) fetch(0, rv0);
            if (n_stage > 1) fetch(1, rv1);
            if (n_stage > 0) stash(0, 0, rv0);
            if (n_stage > 2) fetch(2, rv0);
            __syncthreads();
        }
        // here: LDS buffer 0 holds stage c (c even), rv1 stage c + 1, rv0 stage c + 2, `cur` points at stage c + 3
        for (; c < n_stage; c += 2) {
            if (stage_first) {
                if (c + 1 < n_stage) stash(c + 1, 1, rv1);
                if (c + 3 < n_stage) fetch(c + 3, rv1);
            }
            multiply(0);
            if (!stage_first) {
                if (c + 1 < n_stage) stash(c + 1, 1, rv1);
                if (c + 3 < n_stage) fetch(c + 3, rv1);
            }
            __syncthreads();
            if (c + 1 >= n_stage) break;
            if (stage_first) {
                if (c + 2 < n_stage) stash(c + 2, 0, rv0);
                if (c + 4 < n_stage) fetch(c + 4, rv0);
            }
            multiply(1);
            if (!stage_first) {
                if (c + 2 < n_stage) stash(c + 2, 0, rv0);
                if (c + 4 < n_stage) fetch(c + 4, rv0);
            }
            __syncthreads();
        }
    }

    // partial block -> workspace (or dW itself when there is a single row range).  acc[t][u][r]:
    // n = n0 + wave's base + 32 t + 8 (r >> 2) + 4 h + (r & 3),  k = k0 + wave's base + 32 u + il
    // (NP == 2: the column exponents are fetched with two + TA * 4 vector loads up front -- one scalar load per stored element,
    // each under its own null test, made 256 basic blocks of this epilogue, every one waiting for its own load)
    float *dst = p.out + (size_t)split * p.N * p.K;
    int ek[TB];
#pragma unroll
    for (int u = 0; u < TB; ++u) ek[u] = (NP == 2 && p.x_max) ? ws_exp_of_bits(p.x_max[k0 + wb * 32 * TB + 32 * u + il]) : 0;
    // (every exponent load in front of the first store: stores count in vmcnt like loads, so a load behind the stores of the block before
    // it made each 32-row block wait for those 32 stores to be acknowledged -- four write latencies at the end of every workgroup)
    typedef unsigned ws_u32x4 __attribute__((ext_vector_type(4)));
    ws_u32x4 gb[TA][4];
#pragma unroll
    for (int t = 0; t < TA; ++t)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4)
            gb[t][g4] = (NP == 2 && p.g_max) ? *reinterpret_cast<const ws_u32x4 *>(p.g_max + n0 + wa * 32 * TA + 32 * t + 8 * g4 + 4 * h)
                                             : ws_u32x4{0u, 0u, 0u, 0u};
#pragma unroll
    for (int t = 0; t < TA; ++t) {
#pragma unroll
        for (int u = 0; u < TB; ++u)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = n0 + wa * 32 * TA + 32 * t + 8 * (r >> 2) + 4 * h + (r & 3);
                const int k = k0 + wb * 32 * TB + 32 * u + il;
                float v = acc[t][u][r];
                if constexpr (NP == 2) v = ldexpf(v, ws_exp_of_bits(gb[t][r >> 2][r & 3]) + ek[u]);   // undo the column scales (exact)
                dst[(size_t)n * p.K + k] = v;
            }
    }
}

template <int TA, int TB, int WA, int WB, bool MASK, int NP>
__global__ __launch_bounds__(64 * WA * WB) void wgrad_split_kernel(const WgradSplitParams p) {
    wgrad_split_body<TA, TB, WA, WB, MASK, NP>(p, (int)blockIdx.x);
}

// ---- several layers in ONE launch (round 6): the weight gradients of up to four layers whose dW is tiled 256 x 256, all cut into the SAME
// number of row ranges.  What a launch costs beyond its rows is its partial blocks -- one 256 KB block per workgroup, written at the end of the
// kernel and read again by the reduction: 67 MB + 67 MB whatever the layer -- and a launch fills the chip only by cutting its rows until
// tiles x ranges reaches the CU count (dW [512, 768]: 6 x 40; dW [256, 512]: 2 x 128).  Two layers in one launch fill it with HALF the ranges
// each (8 tiles x 32 ranges), i.e. half the partial-block bytes of the two launches, ranges four times as long for the small layer, one
// prologue / flush phase instead of two.  Same kernel body, same bits for a given number of ranges.
constexpr int kWsMaxJobs = 4;
struct WgradSplitJobs {
    WgradSplitParams p[kWsMaxJobs];
    int first[kWsMaxJobs];     // first workgroup of job j (a multiple of 8 whenever the range count is)
    int n;
};
template <int TA, int TB, int WA, int WB, int NP>
__global__ __launch_bounds__(64 * WA * WB) void wgrad_split_jobs_kernel(const WgradSplitJobs jobs) {
    // (uniform selects of scalar kernel arguments: indexing the array with a run-time value would move it to scratch)
    WgradSplitParams p = jobs.p[0];
    int first = 0;
#pragma unroll
    for (int j = 1; j < kWsMaxJobs; ++j)
        if (j < jobs.n && (int)blockIdx.x >= jobs.first[j]) {
            p = jobs.p[j];
            first = jobs.first[j];
        }
    wgrad_split_body<TA, TB, WA, WB, false, NP>(p, (int)blockIdx.x - first);
}

// ... and the half-size tiles (dW tiled 128 x 256 or 256 x 128: the layers next to the 128-wide hidden activation) of two layers in one launch:
// alone, such a layer is ONE tile cut into 256 row ranges of 390 rows -- 24 stages per workgroup between a prologue and a 128 KB flush.
// cfg[j]: 1 = 128 x 256 tiles, 2 = 256 x 128 (wgrad_split_cfg); same threads and LDS bytes either way.
struct WgradSplitJobsMixed {
    WgradSplitJobs jobs;
    int cfg[kWsMaxJobs];
};
template <int NP>
__global__ __launch_bounds__(512) void wgrad_split_jobs_mixed_kernel(const WgradSplitJobsMixed m) {
    WgradSplitParams p = m.jobs.p[0];
    int first = 0, cfg = m.cfg[0];
#pragma unroll
    for (int j = 1; j < kWsMaxJobs; ++j)
        if (j < m.jobs.n && (int)blockIdx.x >= m.jobs.first[j]) {
            p = m.jobs.p[j];
            first = m.jobs.first[j];
            cfg = m.cfg[j];
        }
    if (cfg == 1) wgrad_split_body<2, 2, 2, 4, false, NP>(p, (int)blockIdx.x - first);
    else wgrad_split_body<2, 2, 4, 2, false, NP>(p, (int)blockIdx.x - first);
}

// shapes the split kernel tiles: both dimensions multiples of 128 (every large layer of the 768-512-256-128 MLPs)
int wgrad_split_cfg(int N, int K) {
    if (N % 256 == 0 && K % 256 == 0) return 0;
    if (N % 128 == 0 && K % 256 == 0) return 1;
    if (N % 256 == 0 && K % 128 == 0) return 2;
    return -1;
}

template <int TA, int TB, int WA, int WB, int NP>
static int wgrad_split_go(const WgradSplitParams &p, bool mask, hipStream_t s) {
    constexpr int Nt = 32 * TA * WA, Kt = 32 * TB * WB;
    const size_t lds = (size_t)2 * (4 * (Nt / 4 + 4) + 4 * (Kt / 4 + 4)) * NP * 2 * 16;
    auto go = [&](auto kern) -> int {
        static LdsGrant grant;
        RQ_RETURN_IF_HIP(grant.ensure(reinterpret_cast<const void *>(kern), 160 * 1024));
        hipLaunchKernelGGL(kern, dim3(p.nslab_n * p.nslab_k * p.msplit), dim3(64 * WA * WB), lds, s, p);
        RQ_CHECK_LAUNCH("wgrad_split_kernel");
        return 0;
    };
    return mask ? go(wgrad_split_kernel<TA, TB, WA, WB, true, NP>) : go(wgrad_split_kernel<TA, TB, WA, WB, false, NP>);
}

// called by rqhip_linear_wgrad_ex / rqhip_linear_wgrad_f16 (wgrad.hip), which own the plan (row ranges, workspace) and the
// reduce kernel.  g_max / x_max: the column maxima of the fp16 path (both non-null), or nullptr for the three-piece bf16 path.
int launch_wgrad_split(int cfg, const float *g, const float *y, const float *x, long long M, int N, int K, float *gm, float *out,
                       int nslab_n, int nslab_k, int msplit, const unsigned *g_max, const unsigned *x_max, hipStream_t s) {
    WgradSplitParams p;
    p.g = g; p.y = y; p.x = x; p.gm = gm; p.out = out;
    p.M = M; p.N = N; p.K = K; p.nslab_n = nslab_n; p.nslab_k = nslab_k; p.msplit = msplit;
    p.n_chunks = (M + 31) / 32;
    p.g_max = g_max;
    p.x_max = x_max;
    const bool mask = y != nullptr;
    if (g_max && x_max) {
        switch (cfg) {
            case 0: return wgrad_split_go<4, 2, 2, 4, 2>(p, mask, s);    // 256 x 256, 8 waves of 128 x 64
            case 1: return wgrad_split_go<2, 2, 2, 4, 2>(p, mask, s);    // 128 x 256, 8 waves of 64 x 64
            default: return wgrad_split_go<2, 2, 4, 2, 2>(p, mask, s);   // 256 x 128
        }
    }
    switch (cfg) {
        case 0: return wgrad_split_go<4, 2, 2, 4, 3>(p, mask, s);
        case 1: return wgrad_split_go<2, 2, 2, 4, 3>(p, mask, s);
        default: return wgrad_split_go<2, 2, 4, 2, 3>(p, mask, s);
    }
}

// The job-table launch: every job 256 x 256 tiles, f16x2 arithmetic, no mask, `msplit` row ranges each; out_j = job j's partial blocks
// [msplit][N_j][K_j] (or dW itself when msplit == 1).
// half == 0: every job 256 x 256 tiles; half == 1: every job 128 x 256 or 256 x 128 tiles (wgrad_split_cfg 1 / 2)
int launch_wgrad_split_jobs(int n, const float *const *g, const float *const *x, long long M, const int *N, const int *K, float *const *out,
                            int msplit, const unsigned *const *g_max, const unsigned *const *x_max, int half, hipStream_t s) {
    WgradSplitJobsMixed m;
    WgradSplitJobs &jobs = m.jobs;
    jobs.n = n;
    int blocks = 0;
    for (int j = 0; j < kWsMaxJobs; ++j) {
        const int i = j < n ? j : 0;
        const int cfg = half ? wgrad_split_cfg(N[i], K[i]) : 0;
        const int Nt = cfg == 1 ? 128 : 256, Kt = cfg == 2 ? 128 : 256;
        WgradSplitParams &p = jobs.p[j];
        p.g = g[i]; p.y = nullptr; p.x = x[i]; p.gm = nullptr; p.out = out[i];
        p.M = M; p.N = N[i]; p.K = K[i]; p.nslab_n = N[i] / Nt; p.nslab_k = K[i] / Kt; p.msplit = msplit;
        p.n_chunks = (M + 31) / 32;
        p.g_max = g_max[i]; p.x_max = x_max[i];
        m.cfg[j] = cfg;
        jobs.first[j] = blocks;
        if (j < n) blocks += p.nslab_n * p.nslab_k * msplit;
    }
    constexpr int NP = 2;
    static LdsGrant grant, grant_half;
    if (half) {
        constexpr int Nt = 128, Kt = 256;      // (256 x 128: the same sum)
        const size_t lds = (size_t)2 * (4 * (Nt / 4 + 4) + 4 * (Kt / 4 + 4)) * NP * 2 * 16;
        auto kern = wgrad_split_jobs_mixed_kernel<NP>;
        RQ_RETURN_IF_HIP(grant_half.ensure(reinterpret_cast<const void *>(kern), 160 * 1024));
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(512), lds, s, m);
        RQ_CHECK_LAUNCH("wgrad_split_jobs_mixed_kernel");
        return 0;
    }
    constexpr int TA = 4, TB = 2, WA = 2, WB = 4, Nt = 32 * TA * WA, Kt = 32 * TB * WB;
    const size_t lds = (size_t)2 * (4 * (Nt / 4 + 4) + 4 * (Kt / 4 + 4)) * NP * 2 * 16;
    auto kern = wgrad_split_jobs_kernel<TA, TB, WA, WB, NP>;
    RQ_RETURN_IF_HIP(grant.ensure(reinterpret_cast<const void *>(kern), 160 * 1024));
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(64 * WA * WB), lds, s, jobs);
    RQ_CHECK_LAUNCH("wgrad_split_jobs_kernel");
    return 0;
}

}  // namespace rqhip
