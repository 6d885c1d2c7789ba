// gemm_split.hip -- the activation GEMMs of the encoder / decoder MLPs on the bf16 matrix cores without narrowing the
// arithmetic (gfx950).  SURVEY.md section 8 row f2; reference modules/encoder.py:25-38 (`relu(x W^T)` forward) and its
// autograd (`g W` data gradient).
//
//   C[M, Nc] = A[M, R] . B[Nc, R]^T      A: fp32 activations (x, or the masked gradient g_pre), streamed from HBM
//                                         B: a weight matrix (W for the forward, W^T for the data gradient), small
//   optional ReLU epilogue (the forward of every layer but the last).
//
// Same arithmetic as csrc/wgrad_split.hip: every fp32 value is the exact sum of three bf16 pieces h + m + l, the product
// is formed from the six piece products that matter (dropped terms <= 2^-23 of a product, below fp32's own rounding of
// it), each piece product is exact in fp32 and accumulates in fp32 inside v_mfma_f32_32x32x16_bf16.  The library's fp32
// GEMMs run these tall-skinny shapes at the fp32 matrix peak (443-582 us for the 78.6 GFLOP layers); six bf16
// instructions of 32 cycles do the work of eight fp32 instructions of 64.
//
// Mapping
//   * both operands are consumed along the reduction index as they lie in memory (a lane's operand = 8 consecutive r of
//     one row of A / one row of B): no transposition.  The weight is split ONCE per step by `weight_planes_kernel` into
//     the stage-major image [R/16][piece][half][Nc] x 16 bytes, so a workgroup's B stage is three contiguous 4 KB runs.
//   * tile = 256 rows x 256 columns (8 waves of 128 x 64) for whole rounds of the chip, 64 x 256 (8 waves of 32 x 64) for
//     what is left over; layers with 128 (mod 256) output columns take 256 x 128 tiles (8 waves of 64 x 64) and 128 x 128
//     for the leftover; 16-deep stages through a double-buffered LDS image, loads two stages ahead.  Row tiles are handed out by an atomic counter (persistent workgroups: 100 000 rows are 782 x Nc/256
//     tiles on 512 slots; a static round-robin would leave the last round a tenth full).
//   * results do not depend on which workgroup computes a tile: bit-reproducible run to run.
#include "rqhip_common.h"

namespace rqhip {

typedef float gs_f32x16 __attribute__((ext_vector_type(16)));
typedef float gs_f32x4 __attribute__((ext_vector_type(4)));
typedef float gs_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 gs_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 gs_bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned gs_u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned gs_u32x4 __attribute__((ext_vector_type(4)));

#ifndef GS_WAVES   // 4 (developer A/B builds): one wave per SIMD with 128-column wave tiles and a 512-register budget
#define GS_WAVES 8
#endif
#ifndef GS_PC      // 1 (developer A/B builds): four more waves (one per SIMD) do ALL the staging -- global requests, the split, the
#define GS_PC 0    // LDS writes -- and the eight tile waves only read the LDS and multiply (12 waves: 170 registers each)
#endif
#ifndef GS_F16     // 1 (developer A/B builds, TIMING ONLY: no row scaling yet): two fp16 pieces per operand and the three products
#define GS_F16 0   // hh + hm + mh instead of three bf16 pieces and six products (tools/fp16_split_study.py)
#endif
#ifndef GS_BL2     // 1 (with GS_PC): the tile waves take their weight-image operands straight from L2 into registers (refilled for
#define GS_BL2 0   // the next stage right after their last use); the image never passes through the LDS or the staging waves
#endif
constexpr int kGsWaves = GS_WAVES, kGsUB = 16 / GS_WAVES;   // a wave's tile is (32 TA) x (32 UB): UB = 2 (8 waves) or 4 (4 waves)
constexpr int kGsStageWaves = GS_PC ? 4 : kGsWaves;         // waves that stage (GS_PC: waves kGsWaves .. kGsWaves + 3)
constexpr int kGsThreads = 64 * (kGsWaves + (GS_PC ? kGsStageWaves : 0)), kGsStageThreads = 64 * kGsStageWaves;
constexpr int kGsNP = GS_F16 ? 2 : 3;   // pieces per operand
constexpr int kGsK = 16;   // tile: COLS = 256 or 128 columns, (waves / (COLS / (32 UB))) * 32 * TA rows
#ifndef GS_XCD_GROUP
#define GS_XCD_GROUP 0
#endif
#ifndef GS_PHASE   // 1 (developer A/B builds): the two waves of a SIMD out of phase -- measured 5-7 % SLOWER
#define GS_PHASE 0
#endif
#ifndef GS_PROBE   // developer builds (tools/ab_build.sh): phase-skipping bit mask, results are WRONG with any bit set --
#define GS_PROBE 0 // 1: no split arithmetic, 2: no LDS writes at all, 4: operand reads once per tile, 16: no global loads (32: none of A, 64: none of the weight image, 128: A always from the first two stages = cache hits,
                   // 256: a tile's A stage is one contiguous 16 KB block (wrong data, same bytes), 512: no result stores)
#endif

#ifdef GS_TIMING
// developer-only s_memtime stamps of every wave of workgroup 0 during its FIRST tile (tools/gemm_timing.py --build):
// slot 0 tile start, 1 prologue barrier passed, then per half-iteration i (stage i multiplied): 2 + 4 i staged,
// 3 + 4 i requests issued, 4 + 4 i multiplied, 5 + 4 i barrier passed
__device__ unsigned long long gs_dbg[8 * 64];
__device__ unsigned gs_dbg_armed = 1;
#define GS_STAMP(i)                                                                                                   \
    do {                                                                                                              \
        if (gs_timed && (threadIdx.x & 63) == 0 && threadIdx.x < 512 && (i) < 64) gs_dbg[(threadIdx.x >> 6) * 64 + (i)] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define GS_STAMP(i) do { } while (0)
#endif

__device__ __forceinline__ void gs_split2(float a, float b, unsigned &h, unsigned &m, unsigned &l) {
    const gs_bf16x2 hh = __builtin_convertvector(gs_f32x2{a, b}, gs_bf16x2);
    h = __builtin_bit_cast(unsigned, hh);
    const float ra = a - __builtin_bit_cast(float, h << 16), rb = b - __builtin_bit_cast(float, h & 0xffff0000u);
    const gs_bf16x2 mm = __builtin_convertvector(gs_f32x2{ra, rb}, gs_bf16x2);
    m = __builtin_bit_cast(unsigned, mm);
    const float sa = ra - __builtin_bit_cast(float, m << 16), sb = rb - __builtin_bit_cast(float, m & 0xffff0000u);
    const gs_bf16x2 ll = __builtin_convertvector(gs_f32x2{sa, sb}, gs_bf16x2);
    l = __builtin_bit_cast(unsigned, ll);
}

#if GS_F16
typedef _Float16 gs_f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 gs_f16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void gs_split2_f16(float a, float b, unsigned &h, unsigned &m) {
    const gs_f16x2 hh = __builtin_convertvector(gs_f32x2{a, b}, gs_f16x2);
    h = __builtin_bit_cast(unsigned, hh);
    const gs_f32x2 hf = __builtin_convertvector(hh, gs_f32x2);
    const gs_f16x2 mm = __builtin_convertvector(gs_f32x2{a - hf.x, b - hf.y}, gs_f16x2);
    m = __builtin_bit_cast(unsigned, mm);
}
#endif

#if GS_F16
// Exact power-of-two scaling (PROTOTYPE, not validated on the GPU yet): fp16 pieces only carry 11 + 11 bits between 6.1e-5 and
// 65 504, so every row of A and every weight row is multiplied by 2^-e, e = exponent of its largest |value| (the scaled row then
// has its maximum in [1, 2)), and the result is multiplied back by 2^(e_row + e_column) in the epilogue -- all three exact.
__device__ __forceinline__ int gs_exp_of(float m) {   // floor(log2 m) of a positive finite float; 0 for 0 / inf / nan
    const unsigned b = __builtin_bit_cast(unsigned, m) & 0x7fffffffu;
    const int e = (int)(b >> 23);
    return (b == 0u || e == 255) ? 0 : (e == 0 ? -126 : e - 127);
}
// exps[m] for the rows of A [M, R] (16-byte aligned rows): four rows per 256-thread block, one wave per row
__global__ __launch_bounds__(256) void row_exps_kernel(const float *__restrict__ A, long long M, int R, int *__restrict__ exps) {
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const int lane = threadIdx.x & 63;
    float m = 0.0f;
    const gs_f32x4 *src = reinterpret_cast<const gs_f32x4 *>(A + (size_t)row * R);
    for (int i = lane; i < R / 4; i += 64) {
        const gs_f32x4 v = src[i];
        m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));   // (fmaxf drops NaNs: they stay NaN in the product)
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if (lane == 0) exps[row] = gs_exp_of(m);
}
// exps[n] for the rows of B: src[n][r] (transpose == 0) or src[r][n] (transpose == 1); one wave per n
__global__ __launch_bounds__(64) void weight_exps_kernel(const float *__restrict__ src, int Nc, int R, int transpose, int *__restrict__ exps) {
    const int n = blockIdx.x, lane = threadIdx.x;
    float m = 0.0f;
    for (int r = lane; r < R; r += 64) m = fmaxf(m, fabsf(transpose ? src[(size_t)r * Nc + n] : src[(size_t)n * R + r]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if (lane == 0) exps[n] = gs_exp_of(m);
}
#endif

// planes[s][piece][half][n] (16 bytes: r = 16 s + 8 half + j, j < 8) of src[n][r] (transpose == 0, src is [Nc, R]) or of
// src[r][n] (transpose == 1, src is [R, Nc]: the data gradient multiplies by W, i.e. B = W^T).  One thread per element.
__global__ __launch_bounds__(256) void weight_planes_kernel(const float *__restrict__ src, int Nc, int R, int transpose,
                                                            unsigned *__restrict__ planes) {
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;     // (n, r pair): r = 2 rp, 2 rp + 1
    const long long total = (long long)Nc * (R / 2);
    if (e < 4) planes[(size_t)(R / kGsK) * 2 * kGsNP * Nc * 4 + e] = 0u;        // the tile dispenser behind the image
    if (e >= total) return;
    const int n = (int)(e % Nc), rp = (int)(e / Nc), r = 2 * rp;
    const float a = transpose ? src[(size_t)r * Nc + n] : src[(size_t)n * R + r];
    const float b = transpose ? src[(size_t)(r + 1) * Nc + n] : src[(size_t)n * R + r + 1];
    unsigned h, m, l = 0u;
#if GS_F16
    {   // the exponents lie behind the tile dispenser (16 words): see rqhip_weight_planes
        const int e_n = reinterpret_cast<const int *>(planes + (size_t)(R / kGsK) * 2 * kGsNP * Nc * 4 + 16)[n];
        gs_split2_f16(ldexpf(a, -e_n), ldexpf(b, -e_n), h, m);
    }
#else
    gs_split2(a, b, h, m, l);
#endif
    const int s = r >> 4, half = (r >> 3) & 1, j2 = (r & 7) >> 1;      // dword j2 of the 16-byte element
    const size_t base = ((size_t)(s * kGsNP) * 2 + half) * Nc + n;
    planes[(base + 0 * 2 * (size_t)Nc) * 4 + j2] = h;
    planes[(base + 1 * 2 * (size_t)Nc) * 4 + j2] = m;
    if (kGsNP == 3) planes[(base + 2 * 2 * (size_t)Nc) * 4 + j2] = l;
}

struct GemmSplitParams {
    const float *A;          // [M, R]
    const unsigned *planes;  // weight image, see weight_planes_kernel
    float *C;                // [M, Nc]
    long long M;
    int R, Nc;
    int n_col_tiles;
    // tiles 0 .. n_big - 1 are 256 rows high (rows [0, 256 rt_big)), the rest 64 rows high (from row 256 rt_big on)
    unsigned n_big, n_tiles;
    int rt_big;
    unsigned *counter;       // [0] tile dispenser, [1] workgroups that have left; both zero between launches
    // EPI == 2 (the last decoder layer fused with the reconstruction loss): C receives (2 (A.B^T - X)) * row_scale, and
    // rowsum[ct][m] the squared error of row m over column tile ct
    const float *X;
    float *rowsum;
    float row_scale;
#if GS_F16
    const int *a_exp;        // [M] exponents of the rows of A (row_exps_kernel), or nullptr: A is used as it is
    const int *b_exp;        // [Nc] exponents of the weight rows (inside the image, written by rqhip_weight_planes)
    int a_is_max;            // a_exp holds the bit patterns of the rows' largest |value| (another GEMM's rowmax_out) instead
    unsigned *rowmax_out;    // [M] or nullptr: unsigned max of the bit patterns of |C[row, :]| (zeroed by the caller; the maximum
                             // does not depend on the order of the atomics) -- the row exponents of the GEMM that reads C next
#endif
};

// one output tile of ROWS x COLS: 8 waves of (32 TA) x 64, WN = COLS / 64 of them side by side
// EPI: 0 = store, 1 = ReLU, 2 = reconstruction loss (see GemmSplitParams)
template <int EPI, int TA, int COLS>
__device__ __forceinline__ void gs_tile(const GemmSplitParams &p, unsigned *sbuf, long long m0, int n0) {
    constexpr int kGsCols = COLS, UB = kGsUB, WN = COLS / (32 * UB), WM = kGsWaves / WN;
    constexpr int ROWS = WM * 32 * TA, AQ = (ROWS * 4 + kGsStageThreads - 1) / kGsStageThreads;   // float4s of A per staging thread and stage
    constexpr int PA = kGsNP * 2 * ROWS * 4, PB = GS_BL2 ? 0 : kGsNP * 2 * kGsCols * 4;    // dwords per stage image
    constexpr int BQ = (2 * kGsNP * kGsCols + kGsStageThreads - 1) / kGsStageThreads;      // 16-byte elements of B per staging thread and stage
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // GS_PC: waves 0 .. 7 multiply (`tiler`), waves 8 .. 11 stage (`stager`); otherwise every wave does both
    const bool stager = GS_PC ? __builtin_amdgcn_readfirstlane(wave) >= kGsWaves : true;
    const bool tiler = GS_PC ? !stager : true;
    const int tid = GS_PC ? (int)threadIdx.x - 64 * kGsWaves : (int)threadIdx.x;   // staging thread number (negative: a tile wave)
    const int il = lane & 31, h = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;
#ifdef GS_TIMING
    const bool gs_timed = blockIdx.x == 0 && TA >= 3 && m0 < 256 * (long long)gridDim.x && gs_dbg_armed;   // the workgroup's first big tile
#endif
    const int n_stage = p.R / kGsK;
    constexpr int APASS = kGsStageThreads / 4;             // rows one staging pass of the workgroup covers
    // staging roles: A -- thread (row = tid >> 2 (+ APASS q), kq = tid & 3) owns 4 consecutive r of one row; B -- three
    // 16-byte elements of the stage's weight image per thread
    const int arow = (tid < 0 ? 0 : tid) >> 2, akq = tid & 3;
    bool a_live[AQ], arow_ok[AQ];
    const float *asrc[AQ];
#pragma unroll
    for (int q = 0; q < AQ; ++q) {
        a_live[q] = arow + APASS * q < ROWS;
        const long long arow_g = m0 + arow + APASS * q;
        arow_ok[q] = a_live[q] && arow_g < p.M;
        asrc[q] = p.A + (size_t)(arow_ok[q] ? arow_g : 0) * ((GS_PROBE & 256) ? 16 : p.R) + 4 * akq;
    }
#if GS_F16
    int a_e[AQ];             // exponents of this thread's rows of A
#pragma unroll
    for (int q = 0; q < AQ; ++q) {
        a_e[q] = (p.a_exp && arow_ok[q]) ? p.a_exp[m0 + arow + APASS * q] : 0;
        if (p.a_is_max) a_e[q] = gs_exp_of(__builtin_bit_cast(float, a_e[q]));
    }
#endif

    gs_f32x16 acc[TA][UB];
#if !GS_PC   // (GS_PC zeroes them in the tile waves' own branch, so that the accumulators are not live beside the staging registers)
#pragma unroll
    for (int t = 0; t < TA; ++t)
#pragma unroll
        for (int u = 0; u < UB; ++u)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][u][r] = 0.0f;
#endif

    gs_f32x4 ra0[AQ], ra1[AQ];   // rows of A: two stages in flight (requested two iterations before they are split)
    gs_u32x4 rb[BQ];
    // (both fetches are UNCONDITIONAL: past the last stage they re-read it.  A load inside `if (stage < n_stage)` makes the
    // compiler's s_waitcnt insertion assume it may not have been issued, and the wait for the weight image then drains it)
    auto fetchA = [&](int stage, gs_f32x4 *dst) {
        stage = stage < n_stage ? stage : n_stage - 1;
        if ((GS_PROBE & 16) && stage > 1) return;
#pragma unroll
        for (int q = 0; q < AQ; ++q)
            if (!((GS_PROBE & 32) && stage > 1))
                dst[q] = !arow_ok[q] ? gs_f32x4{0.f, 0.f, 0.f, 0.f}   // (non-temporal loads of A: +3 ... +4 %)
                         : *reinterpret_cast<const gs_f32x4 *>(asrc[q] + ((GS_PROBE & 128) ? (stage & 1) : stage) * ((GS_PROBE & 256) ? 256 * 1024 : kGsK));
    };
    auto fetchB = [&](int stage) {
        if (GS_BL2) return;
        stage = stage < n_stage ? stage : n_stage - 1;
        if ((GS_PROBE & 16) && stage > 1) return;
        // stage image: [piece][half][Nc] 16-byte elements; this tile's part is columns n0 .. n0 + 255 of each of the six
        // (piece, half) rows: element e = tid + 512 q  ->  (ph = e / COLS, col = e % COLS)
        const gs_u32x4 *img = reinterpret_cast<const gs_u32x4 *>(p.planes) + (size_t)stage * 2 * kGsNP * p.Nc + n0;
#pragma unroll
        for (int q = 0; q < BQ; ++q) {
            const int e = tid + kGsStageThreads * q;
            if ((2 * kGsNP * kGsCols) % kGsStageThreads != 0 && e >= 2 * kGsNP * kGsCols) continue;
            if (!((GS_PROBE & 64) && stage > 1)) rb[q] = img[(size_t)(e / kGsCols) * p.Nc + (e % kGsCols)];
        }
    };
    auto stash = [&](int buf, const gs_f32x4 *ra) {
        if ((GS_PROBE & 2) && buf) return;
        unsigned *dA = sbuf + buf * (PA + PB), *dB = dA + PA;
#pragma unroll
        for (int q = 0; q < AQ; ++q) {
            if (!a_live[q]) continue;
            unsigned h01, m01, l01, h23, m23, l23;
            if (GS_PROBE & 1) {
                h01 = __builtin_bit_cast(unsigned, ra[q].x); m01 = __builtin_bit_cast(unsigned, ra[q].y); l01 = h01 ^ m01;
                h23 = __builtin_bit_cast(unsigned, ra[q].z); m23 = __builtin_bit_cast(unsigned, ra[q].w); l23 = h23 ^ m23;
            } else {
#if GS_F16
            gs_split2_f16(ldexpf(ra[q].x, -a_e[q]), ldexpf(ra[q].y, -a_e[q]), h01, m01);
            gs_split2_f16(ldexpf(ra[q].z, -a_e[q]), ldexpf(ra[q].w, -a_e[q]), h23, m23);
            l01 = l23 = 0u;
#else
            gs_split2(ra[q].x, ra[q].y, h01, m01, l01);
            gs_split2(ra[q].z, ra[q].w, h23, m23, l23);
#endif
            }
            // element [piece][half = akq >> 1][row] is 16 bytes = r 8 half .. 8 half + 7; this thread fills its half (akq & 1)
            unsigned *d = dA + (((akq >> 1) * ROWS) + arow + APASS * q) * 4 + 2 * (akq & 1);
            *reinterpret_cast<gs_u32x2 *>(d + 0 * 2 * ROWS * 4) = gs_u32x2{h01, h23};
            *reinterpret_cast<gs_u32x2 *>(d + 1 * 2 * ROWS * 4) = gs_u32x2{m01, m23};
            if (kGsNP == 3) *reinterpret_cast<gs_u32x2 *>(d + 2 * 2 * ROWS * 4) = gs_u32x2{l01, l23};
        }
#pragma unroll
        for (int q = 0; q < BQ; ++q) {
            const int e = tid + kGsStageThreads * q;
            if (GS_BL2 || ((2 * kGsNP * kGsCols) % kGsStageThreads != 0 && e >= 2 * kGsNP * kGsCols)) continue;
            *reinterpret_cast<gs_u32x4 *>(dB + (size_t)e * 4) = rb[q];   // [ph][col] order == the image's
        }
    };
#if GS_BL2
    // this lane's image operands of the current stage, [column half][piece]; element (piece, half h, column) of stage s is
    // planes[((s 6 + piece 2 + h) Nc + column) x 16 bytes]
    gs_bf16x8 breg[UB][kGsNP];
    auto load_b = [&](int stage, int u) {
        stage = stage < n_stage ? stage : n_stage - 1;
        const gs_bf16x8 *img = reinterpret_cast<const gs_bf16x8 *>(p.planes) + ((size_t)stage * 2 * kGsNP + h) * p.Nc + n0 + wn * 32 * UB + 32 * u + il;
#pragma unroll
        for (int pc = 0; pc < kGsNP; ++pc) breg[u][pc] = img[(size_t)pc * 2 * p.Nc];
    };
    auto multiply = [&](int buf, int next_stage) {
#else
    auto multiply = [&](int buf) {
#endif
        const gs_bf16x8 *aA = reinterpret_cast<const gs_bf16x8 *>(sbuf + buf * (PA + PB));
        const gs_bf16x8 *bB = reinterpret_cast<const gs_bf16x8 *>(sbuf + buf * (PA + PB) + PA);
        // column half outer, row block inner: 12 + 12 operand registers live (the row blocks are read once per column half:
        // 30 instead of 18 LDS reads per stage, which the LDS has room for -- tools/gemm_probe.py, GS_PROBE 4)
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            gs_bf16x8 b[kGsNP];
#pragma unroll
            for (int pc = 0; pc < kGsNP; ++pc)
#if GS_BL2
                b[pc] = breg[u][pc];
#else
                b[pc] = bB[(((GS_PROBE & 4) ? 0 : pc) * 2 + h) * kGsCols + wn * 32 * UB + ((GS_PROBE & 4) ? 0 : 32 * u) + il];
#endif
#pragma unroll
            for (int t = 0; t < TA; ++t) {
                gs_bf16x8 a[kGsNP];
#pragma unroll
                for (int pc = 0; pc < kGsNP; ++pc)
                    a[pc] = aA[(((GS_PROBE & 4) ? 0 : pc) * 2 + h) * ROWS + wm * 32 * TA + ((GS_PROBE & 4) ? 0 : 32 * t) + il];
                gs_f32x16 c16 = acc[t][u];
                // the weight columns take the instruction's ROW role: the accumulator is the tile transposed, a lane holds
                // four consecutive columns of one output row per register quad -> 16-byte result stores
#if GS_F16
                {
                    gs_f16x8 fa[2], fb[2];
#pragma unroll
                    for (int pc = 0; pc < 2; ++pc) { fa[pc] = __builtin_bit_cast(gs_f16x8, a[pc]); fb[pc] = __builtin_bit_cast(gs_f16x8, b[pc]); }
                    c16 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb[0], fa[1], c16, 0, 0, 0);   // m h (smallest first)
                    c16 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb[1], fa[0], c16, 0, 0, 0);   // h m
                    c16 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb[0], fa[0], c16, 0, 0, 0);   // h h
                }
#else
                c16 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[1], a[1], c16, 0, 0, 0);   // m m (smallest first)
                c16 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[0], a[2], c16, 0, 0, 0);   // l h
                c16 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[2], a[0], c16, 0, 0, 0);   // h l
                c16 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[0], a[1], c16, 0, 0, 0);   // m h
                c16 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[1], a[0], c16, 0, 0, 0);   // h m
                c16 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[0], a[0], c16, 0, 0, 0);   // h h
#endif
                acc[t][u] = c16;
                // GS_PC, 128 accumulators: 170 registers hold them and ONE block's operands; keep the scheduler from hoisting
                // the next block's LDS reads above this block's matrix instructions (it spills the accumulators otherwise)
                if (GS_PC && TA * UB * 16 > 96) __builtin_amdgcn_sched_barrier(0);
            }
            // GS_PC, 96 accumulators: one column half's operands (12 + 36 registers) at a time
            if (GS_PC && TA * UB * 16 <= 96) __builtin_amdgcn_sched_barrier(0);
#if GS_BL2
            load_b(next_stage, u);                 // this column half's operands of the next stage, into the registers just used
            __builtin_amdgcn_sched_barrier(0);
#endif
        }
    };

    // Order of the requests inside an iteration: the weight image FIRST, then the A rows.  Loads complete in order
    // (vmcnt): the next iteration waits for the image it stages, and with the A rows requested before it that wait also
    // drained the A rows of the stage after -- their latency had ONE iteration to hide in, not two (phase-skipping probes:
    // the kernel ran 86 us faster without its global loads although every one of them is issued 1-2 iterations early).
    // (A third register set and LDS buffer -- A rows three iterations ahead -- do not fit: 8 more registers spill inside
    // the stage loop, 535 vs 476 us.)
    // A rows are requested TWO iterations before they are split (scattered 64-byte pieces of 256 rows: their latency is
    // longer than one iteration's matrix work -- tools/gemm_probe.py: the kernel ran 17 % faster without them, 10 % with
    // cache hits), the weight image (L2-resident) one iteration before.
    GS_STAMP(0);
    if (stager) {
        fetchA(0, ra0);
        fetchB(0);
        fetchA(1, ra1);
        stash(0, ra0);
        fetchB(1);
        fetchA(2, ra0);
    }
#if GS_BL2
    if (tiler) {
#pragma unroll
        for (int u = 0; u < UB; ++u) load_b(0, u);
    }
#endif
    __syncthreads();
    GS_STAMP(1);
#if GS_PC
#if GS_BL2
#define GS_MUL(buf, next) multiply(buf, next)
#else
#define GS_MUL(buf, next) multiply(buf)
#endif
    // each role runs its own loop (registers of the two roles are then never live together); both pass the same barriers
    if (stager) {
        for (int c = 0; c < (n_stage & ~1); c += 2) {
            stash(1, ra1);                             // stage c + 1
            fetchB(c + 2);
            fetchA(c + 3, ra1);
            __syncthreads();
            stash(0, ra0);                             // stage c + 2
            fetchB(c + 3);
            fetchA(c + 4, ra0);
            __syncthreads();
        }
        if (n_stage & 1) __syncthreads();
    } else {
#pragma unroll
        for (int t = 0; t < TA; ++t)
#pragma unroll
            for (int u = 0; u < UB; ++u)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][u][r] = 0.0f;
        for (int c = 0; c < (n_stage & ~1); c += 2) {
            GS_MUL(0, c + 1);
            GS_STAMP(4 + 4 * c);
            __syncthreads();
            GS_STAMP(5 + 4 * c);
            GS_MUL(1, c + 2);
            GS_STAMP(8 + 4 * c);
            __syncthreads();
            GS_STAMP(9 + 4 * c);
        }
        if (n_stage & 1) {
            GS_MUL(0, n_stage);
            __syncthreads();
        }
    }
#else
    // Every wave stages the next stage first, then multiplies the current one.  GS_PHASE = 1 (developer builds) puts the two
    // waves of a SIMD OUT OF PHASE inside the barrier interval (one half of the waves stages first, the other multiplies
    // first -- what helped csrc/wgrad_split.hip): measured 5-7 % slower here at every shape.
    const bool stage_first = GS_PHASE ? __builtin_amdgcn_readfirstlane(wave) < 4 : true;   // (scalar: a real branch)
    const int n_pair = n_stage & ~1;
    for (int c = 0; c < n_pair; c += 2) {
        if (stage_first) {
            stash(1, ra1);                             // stage c + 1
            GS_STAMP(2 + 4 * c);
            fetchB(c + 2);
            fetchA(c + 3, ra1);
            GS_STAMP(3 + 4 * c);
        }
        multiply(0);
        GS_STAMP(4 + 4 * c);
        if (!stage_first) {
            stash(1, ra1);
            fetchB(c + 2);
            fetchA(c + 3, ra1);
        }
        if (!(GS_PROBE & 1024)) __syncthreads();
        GS_STAMP(5 + 4 * c);
        if (stage_first) {
            stash(0, ra0);                             // stage c + 2 (past the end: the last stage again, never multiplied)
            GS_STAMP(6 + 4 * c);
            fetchB(c + 3);
            fetchA(c + 4, ra0);
            GS_STAMP(7 + 4 * c);
        }
        multiply(1);
        GS_STAMP(8 + 4 * c);
        if (!stage_first) {
            stash(0, ra0);
            fetchB(c + 3);
            fetchA(c + 4, ra0);
        }
        if (!(GS_PROBE & 1024)) __syncthreads();
        GS_STAMP(9 + 4 * c);
    }
    if (n_stage & 1) {                             // the last stage of an odd count lies in buffer 0
        multiply(0);
        if (!(GS_PROBE & 1024)) __syncthreads();
    }
#endif

    // acc[t][u][r]: row = m0 + 32 TA wm + 32 t + il,  column = n0 + 64 wn + 32 u + 8 (r >> 2) + 4 h + (r & 3)
    // 32 16-byte stores per lane and tile (the untransposed accumulator needed 128 dword stores; same time: what the result
    // costs is its write traffic, 7-11 % of the kernel while every CU reaches its epilogue in the same phase of a round --
    // tools/gemm_probe.py, GS_PROBE 512)
    float rowsq[TA];
#if GS_F16
    float rowmx[TA];
#pragma unroll
    for (int t = 0; t < TA; ++t) rowmx[t] = 0.0f;
#endif
    // EPI == 2 reads x beside every result it stores.  The compiler may not move a load above a store that could alias
    // it, so with load / compute / store per 16 bytes every one of the 32 loads of a lane waited for its own latency AND
    // for the store before it (the recon GEMM ran 547 us against 440 for the plain one).  The x values of a whole row
    // block are therefore requested first, and those of the next block before this block's stores.
    gs_f32x4 xv[2][UB * 4];
    auto load_x = [&](int t, gs_f32x4 *dst) {
        const long long row = m0 + 32 * TA * wm + 32 * t + il;
        const float *xs = p.X + (size_t)(row < p.M ? row : p.M - 1) * p.Nc + n0 + 32 * UB * wn + 4 * h;
#pragma unroll
        for (int u = 0; u < UB; ++u)
#pragma unroll
            for (int g = 0; g < 4; ++g) dst[u * 4 + g] = *reinterpret_cast<const gs_f32x4 *>(xs + 32 * u + 8 * g);
    };
    if (EPI == 2 && tiler && !(GS_PROBE & 512)) load_x(0, xv[0]);
#pragma unroll
    for (int t = 0; t < TA; ++t) {
        const long long row = m0 + 32 * TA * wm + 32 * t + il;
        rowsq[t] = 0.0f;
        if (EPI == 2 && tiler && t + 1 < TA && !(GS_PROBE & 512)) load_x(t + 1, xv[(t + 1) & 1]);
        if (tiler && row < p.M && !(GS_PROBE & 512)) {
            float *dst = p.C + (size_t)row * p.Nc + n0 + 32 * UB * wn + 4 * h;
#pragma unroll
            for (int u = 0; u < UB; ++u)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    gs_f32x4 v = {acc[t][u][4 * g], acc[t][u][4 * g + 1], acc[t][u][4 * g + 2], acc[t][u][4 * g + 3]};
#if GS_F16
                    {   // undo the row and column scales (exact)
                        int er = p.a_exp ? p.a_exp[row] : 0;
                        if (p.a_is_max) er = gs_exp_of(__builtin_bit_cast(float, er));
                        const int *ec = p.b_exp + n0 + 32 * UB * wn + 4 * h + 32 * u + 8 * g;
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] = ldexpf(v[j], er + ec[j]);
                    }
#endif
                    if (EPI == 1) {   // (a NaN stays a NaN, as torch.relu)
                        v.x = v.x < 0.0f ? 0.0f : v.x; v.y = v.y < 0.0f ? 0.0f : v.y;
                        v.z = v.z < 0.0f ? 0.0f : v.z; v.w = v.w < 0.0f ? 0.0f : v.w;
                    }
                    if (EPI == 2) {   // as csrc/recon_loss.hip: d = x_hat - x, loss += d d, gradient (2 d) row_scale
                        const gs_f32x4 x4 = xv[t & 1][u * 4 + g];
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const float d = v[j] - x4[j];
                            rowsq[t] = rowsq[t] + d * d;
                            v[j] = (2.0f * d) * p.row_scale;
                        }
                    }
#if GS_F16
                    if (EPI != 2) rowmx[t] = fmaxf(fmaxf(rowmx[t], fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
#endif
                    *reinterpret_cast<gs_f32x4 *>(dst + 32 * u + 8 * g) = v;   // (non-temporal stores: +2 ... +36 %)
                }
        }
    }
#if GS_F16
    if (EPI != 2 && p.rowmax_out && tiler) {   // this wave's 32 UB columns of a row: both lane halves, then one atomic per row
#pragma unroll
        for (int t = 0; t < TA; ++t) {
            const long long row = m0 + 32 * TA * wm + 32 * t + il;
            const float mx = fmaxf(rowmx[t], __shfl_xor(rowmx[t], 32, 64));
            if (h == 0 && row < p.M) atomicMax(p.rowmax_out + row, __builtin_bit_cast(unsigned, mx));
        }
    }
#endif
    if (EPI == 2) {
        // a row's squared error over this column tile: the lane's 32 columns (above, fixed order), + the other half-wave's,
        // then the four column waves in order through LDS (free: the loop's last barrier has been passed)
        float *red = reinterpret_cast<float *>(sbuf);          // [WN][ROWS]
#pragma unroll
        for (int t = 0; t < TA; ++t) {
            const float both = rowsq[t] + __shfl_xor(rowsq[t], 32, 64);
            if (h == 0 && tiler) red[wn * ROWS + 32 * TA * wm + 32 * t + il] = both;
        }
        __syncthreads();
        for (int r = (int)threadIdx.x; r < ROWS; r += kGsThreads) {
            if (m0 + r >= p.M) continue;
            float sum = red[r];
#pragma unroll
            for (int w = 1; w < WN; ++w) sum = sum + red[w * ROWS + r];
            p.rowsum[(size_t)(n0 / kGsCols) * p.M + m0 + r] = sum;
        }
        // (the persistent loop's barrier at its top keeps the next tile's staging off `red`)
    }
}

template <int EPI, int COLS>
__global__ __launch_bounds__(kGsThreads) void gemm_split_kernel(const GemmSplitParams p) {
    constexpr int kSmallRows = (8 / (COLS / 64)) * 32;   // 64 (COLS = 256) or 128 (COLS = 128)
    constexpr int kBigRows = (GS_PC && COLS == 256) ? 192 : 256;   // (GS_PC: 96 accumulators per lane leave room for a stage's operands in 170 registers)
    extern __shared__ __attribute__((aligned(16))) char gs_smem[];
    unsigned *sbuf = reinterpret_cast<unsigned *>(gs_smem);
    __shared__ unsigned s_tile;
    const int tid = threadIdx.x;
    for (;;) {
        __syncthreads();                       // (the previous tile's LDS reads are done; s_tile may be rewritten)
        if (tid == 0) s_tile = atomicAdd(p.counter, 1u);
        __syncthreads();
        const unsigned tile = s_tile;
        if (tile >= p.n_tiles) {
            // the last workgroup to leave re-arms the dispenser for the next launch (nobody takes a ticket after it)
            if (tid == 0 && atomicAdd(p.counter + 1, 1u) == gridDim.x - 1) {
                p.counter[0] = 0u;
                p.counter[1] = 0u;
            }
            break;
        }
        // column tile fastest: the workgroups that share a row tile's A strip run at the same time (L2).  Whole rounds of
        // the chip take 256-row tiles (fewest LDS reads per matrix instruction); what is left over after the last whole
        // round is cut into 64-row tiles so that it spreads over all CUs instead of giving a few of them a fourth big
        // tile (100 000 x 512: 782 big tiles on 256 CUs were 4 tile times for 3.05 rounds of work).
        if (tile < p.n_big) {
            int ct = (int)(tile % (unsigned)p.n_col_tiles), rt = (int)(tile / (unsigned)p.n_col_tiles);
#if GS_XCD_GROUP
            // tickets t, t + 8, .. (the same XCD while tickets are taken in workgroup order) share one row tile's A strip
            const unsigned grp = 8u * (unsigned)p.n_col_tiles, g = tile / grp, in = tile % grp;
            if ((g + 1) * grp <= p.n_big) { rt = (int)(g * 8u + (in & 7u)); ct = (int)(in >> 3); }
#endif
            gs_tile<EPI, kBigRows / kSmallRows, COLS>(p, sbuf, (long long)rt * kBigRows, ct * COLS);
        } else {
            const unsigned st = tile - p.n_big;
            const int ct = (int)(st % (unsigned)p.n_col_tiles), rt = (int)(st / (unsigned)p.n_col_tiles);
            gs_tile<EPI, 1, COLS>(p, sbuf, (long long)p.rt_big * kBigRows + (long long)rt * kSmallRows, ct * COLS);
        }
    }
}

// reconstruction loss of a row = its column tiles' sums in order
__global__ __launch_bounds__(256) void recon_rows_finish_kernel(const float *__restrict__ rowsum, int nct, long long M,
                                                                float *__restrict__ out) {
    const long long m = (long long)blockIdx.x * 256 + threadIdx.x;
    if (m >= M) return;
    float s = rowsum[m];
    for (int c = 1; c < nct; ++c) s = s + rowsum[(size_t)c * M + m];
    out[m] = s;
}

}  // namespace rqhip

using namespace rqhip;

static int gs_cols(int Nc) { return Nc % 256 == 0 ? 256 : 128; }   // tile width
extern "C" int rqhip_gemm_split_supported(int Nc, int R) { return (Nc > 0 && R > 0 && Nc % 128 == 0 && R % kGsK == 0) ? 1 : 0; }

extern "C" size_t rqhip_weight_planes_bytes(int Nc, int R) {
    if (!rqhip_gemm_split_supported(Nc, R)) return 0;
    return (size_t)(R / kGsK) * 2 * kGsNP * Nc * 16 + 64 + (GS_F16 ? (size_t)Nc * sizeof(int) : 0);    // + the tile counter (+ GS_F16: the weight rows' exponents) behind the image
}

extern "C" int rqhip_weight_planes(const float *w, int rows, int cols, int transpose, void *planes, size_t planes_bytes,
                                   rqhip_stream_t stream) {
    // w is [rows, cols] row-major.  transpose == 0: B = w (Nc = rows, R = cols); transpose == 1: B = w^T (Nc = cols, R = rows)
    const int Nc = transpose ? cols : rows, R = transpose ? rows : cols;
    if (!w || !planes || !rqhip_gemm_split_supported(Nc, R) || planes_bytes < rqhip_weight_planes_bytes(Nc, R)) {
        set_error("weight_planes: bad arguments or unsupported shape (Nc = %d must be a multiple of 128, R = %d of 16)", Nc, R);
        return RQHIP_EARG;
    }
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const long long total = (long long)Nc * (R / 2);
#if GS_F16
    hipLaunchKernelGGL(weight_exps_kernel, dim3(Nc), dim3(64), 0, s, w, Nc, R, transpose,
                       reinterpret_cast<int *>(reinterpret_cast<unsigned *>(planes) + (size_t)(R / kGsK) * 2 * kGsNP * Nc * 4 + 16));
    RQ_CHECK_LAUNCH("weight_exps_kernel");
#endif
    hipLaunchKernelGGL(weight_planes_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, w, Nc, R, transpose,
                       reinterpret_cast<unsigned *>(planes));
    RQ_CHECK_LAUNCH("weight_planes_kernel");
    return RQHIP_OK;
}

static int gemm_split_launch(const float *A, int64_t M, int R, const void *planes, int Nc, int epi, float *C, int flags_tile,
                             const float *X, float row_scale, float *rowsum, rqhip_stream_t stream);

#if GS_F16
static const int *g_gs_a_exp = nullptr;   // (prototype plumbing: the row exponents of the next gemm_split_launch)
static int g_gs_a_is_max = 0;
static unsigned *g_gs_rowmax_out = nullptr;
// exponents of the rows of A for rqhip_gemm_split_f16; A rows must be 16-byte aligned (R % 4 == 0)
extern "C" int rqhip_row_exponents(const float *A, int64_t M, int R, int *exps, rqhip_stream_t stream) {
    if (M < 0 || R <= 0 || (R % 4) != 0 || (M > 0 && (!A || !exps))) {
        set_error("row_exponents: bad arguments");
        return RQHIP_EARG;
    }
    if (M == 0) return RQHIP_OK;
    hipLaunchKernelGGL(row_exps_kernel, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), A,
                       (long long)M, R, exps);
    RQ_CHECK_LAUNCH("row_exps_kernel");
    return RQHIP_OK;
}
// rqhip_gemm_split with the rows of A scaled by 2^-a_exp[row] before the fp16 split (a_exp from rqhip_row_exponents)
extern "C" int rqhip_gemm_split_f16(const float *A, const int *a_exp, int64_t M, int R, const void *planes, int Nc, int relu,
                                    float *C, rqhip_stream_t stream) {
    g_gs_a_exp = a_exp;
    const int rc = gemm_split_launch(A, M, R, planes, Nc, relu & 1, C, 0, nullptr, 0.0f, nullptr, stream);
    g_gs_a_exp = nullptr;
    return rc;
}
// the chained form: a_max = the row maxima another GEMM left in its rowmax_out (bit patterns; used in place of exponents), and
// rowmax_out (zeroed by the caller, or nullptr) receives this GEMM's for the next one.  NOT validated on the GPU yet.
extern "C" int rqhip_gemm_split_f16_chain(const float *A, const unsigned *a_max, int64_t M, int R, const void *planes, int Nc,
                                          int relu, float *C, unsigned *rowmax_out, rqhip_stream_t stream) {
    g_gs_a_exp = reinterpret_cast<const int *>(a_max);
    g_gs_a_is_max = a_max ? 1 : 0;
    g_gs_rowmax_out = rowmax_out;
    const int rc = gemm_split_launch(A, M, R, planes, Nc, relu & 1, C, 0, nullptr, 0.0f, nullptr, stream);
    g_gs_a_exp = nullptr;
    g_gs_a_is_max = 0;
    g_gs_rowmax_out = nullptr;
    return rc;
}
#endif

extern "C" int rqhip_gemm_split(const float *A, int64_t M, int R, const void *planes, int Nc, int relu, float *C,
                                rqhip_stream_t stream) {
    return gemm_split_launch(A, M, R, planes, Nc, relu & 1, C, (relu >> 8) & 0xfff, nullptr, 0.0f, nullptr,
                             stream);   // (bits 8.. of `relu`: tile rows, A/B)
}

extern "C" size_t rqhip_gemm_split_recon_workspace_bytes(int64_t M, int Nc) {
    return (M > 0 && Nc > 0 && Nc % 256 == 0) ? (size_t)(Nc / 256) * (size_t)M * sizeof(float) : 0;
}

extern "C" int rqhip_gemm_split_recon(const float *A, int64_t M, int R, const void *planes, int Nc, const float *X,
                                      float row_scale, float *G, float *loss_rows, void *workspace, size_t workspace_bytes,
                                      rqhip_stream_t stream) {
    if (M > 0 && (!X || !G || !loss_rows || !workspace || Nc % 256 != 0 ||
                  workspace_bytes < rqhip_gemm_split_recon_workspace_bytes(M, Nc) || (reinterpret_cast<uintptr_t>(X) & 15u) != 0)) {
        set_error("gemm_split_recon: bad arguments (X, G, loss_rows, workspace of rqhip_gemm_split_recon_workspace_bytes)");
        return RQHIP_EARG;
    }
    const int rc = gemm_split_launch(A, M, R, planes, Nc, 2, G, 0, X, row_scale, reinterpret_cast<float *>(workspace), stream);
    if (rc != RQHIP_OK || M == 0) return rc;
    hipLaunchKernelGGL(recon_rows_finish_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), reinterpret_cast<const float *>(workspace), Nc / 256,
                       (long long)M, loss_rows);
    RQ_CHECK_LAUNCH("recon_rows_finish_kernel");
    return RQHIP_OK;
}

static int gemm_split_launch(const float *A, int64_t M, int R, const void *planes, int Nc, int epi, float *C, int flags_tile,
                             const float *X, float row_scale, float *rowsum, rqhip_stream_t stream) {
    if (M < 0 || !planes || (M > 0 && (!A || !C)) || !rqhip_gemm_split_supported(Nc, R)) {
        set_error("gemm_split: bad arguments or unsupported shape (Nc = %d, R = %d)", Nc, R);
        return RQHIP_EARG;
    }
    auto al16 = [](const void *q) { return (reinterpret_cast<uintptr_t>(q) & 15u) == 0; };
    if (!al16(A) || !al16(C) || !al16(planes) || (R % 4) != 0) {
        set_error("gemm_split: pointers must be 16-byte aligned");
        return RQHIP_EARG;
    }
    if (M == 0) return RQHIP_OK;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    GemmSplitParams p;
    p.A = A; p.planes = reinterpret_cast<const unsigned *>(planes); p.C = C; p.M = M; p.R = R; p.Nc = Nc;
    p.X = X; p.rowsum = rowsum; p.row_scale = row_scale;
#if GS_F16
    p.a_exp = g_gs_a_exp;    // (set by rqhip_gemm_split_f16 around this call; nullptr otherwise)
    p.a_is_max = g_gs_a_is_max;
    p.rowmax_out = g_gs_rowmax_out;
    p.b_exp = reinterpret_cast<const int *>(reinterpret_cast<const unsigned *>(planes) + (size_t)(R / kGsK) * 2 * kGsNP * Nc * 4 + 16);
#endif
    const int cus = cu_count();
    const int cols = gs_cols(Nc), small_rows = cols == 256 ? 64 : 128;
    p.n_col_tiles = Nc / cols;
    // whole rounds of 256-row tiles, the remainder as 64-row (128-row for the 128-column tile) tiles (see the kernel);
    // flags_tile (tools only): 256 = big tiles for every row, 64 = small tiles for every row
    const int big_rows = (GS_PC && cols == 256) ? 192 : 256;
    const long long rt256 = (M + big_rows - 1) / big_rows;
    long long rt_big = ((rt256 * p.n_col_tiles) / cus) * cus / p.n_col_tiles;   // row tiles of the whole rounds
    if (rt_big * big_rows > M) rt_big = M / big_rows;
    // (measured at 100 000 rows: worth it when the leftover is a small part of a round -- Nc = 512: 14 of 256 slots, 517 ->
    // 456 us; a leftover of half a round runs as fast in big tiles -- Nc = 256 / 768: 135 / 149 slots)
    if ((rt256 * p.n_col_tiles) % cus > (3 * cus) / 10 && rt256 * p.n_col_tiles >= cus) rt_big = rt256;
    if (flags_tile == 256) rt_big = rt256;
    if (flags_tile == 64) rt_big = 0;
    const long long rem_rows = M - rt_big * big_rows > 0 ? M - rt_big * big_rows : 0;
    const long long rt_small = (rem_rows + small_rows - 1) / small_rows;
    p.rt_big = (int)rt_big;
    p.n_big = (unsigned)(rt_big * p.n_col_tiles);
    p.n_tiles = p.n_big + (unsigned)(rt_small * p.n_col_tiles);
    // the tile dispenser lives behind the weight image (zeroed by rqhip_weight_planes, re-armed by every launch): one
    // GEMM at a time per image, i.e. launches on one stream
    p.counter = const_cast<unsigned *>(p.planes) + (size_t)(R / kGsK) * 2 * kGsNP * Nc * 4;
    const size_t lds = (size_t)2 * (kGsNP * 2 * (256 + cols) * 16);
    const long long tiles = (long long)p.n_tiles;
    const long long slots = (long long)cus;                // one workgroup per CU (240 VGPRs x 512 threads)
    const int grid = (int)(tiles < slots ? tiles : slots);
    auto go = [&](auto kern) -> int {
        static LdsGrant grant;
        RQ_RETURN_IF_HIP(grant.ensure(reinterpret_cast<const void *>(kern), (int)lds));   // (+ 4 bytes of static LDS)
        hipLaunchKernelGGL(kern, dim3(grid), dim3(kGsThreads), lds, s, p);
        RQ_CHECK_LAUNCH("gemm_split_kernel");
        return 0;
    };
    if (cols == 128) {
        if (epi == 2) {
            set_error("gemm_split: the reconstruction-loss epilogue needs Nc %% 256 == 0 (Nc = %d)", Nc);
            return RQHIP_EARG;
        }
        return epi == 1 ? go(gemm_split_kernel<1, 128>) : go(gemm_split_kernel<0, 128>);
    }
    return epi == 2 ? go(gemm_split_kernel<2, 256>) : epi == 1 ? go(gemm_split_kernel<1, 256>) : go(gemm_split_kernel<0, 256>);
}

#ifdef GS_TIMING
extern "C" int rqhip_gs_debug_read(unsigned long long *out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(rqhip::gs_dbg), sizeof(unsigned long long) * 8 * 64);
}
#endif
