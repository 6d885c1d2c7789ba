// gemm_split.hip -- the activation GEMMs of the encoder / decoder MLPs on the 16-bit matrix cores without narrowing the
// arithmetic below fp32's (gfx950).  SURVEY.md section 8 row f2; reference modules/encoder.py:25-38 (`relu(x W^T)` forward) and
// its autograd (`g W` data gradient, ReLU backward), modules/rqvae.py:146,152 + modules/loss.py:5-10 (the last decoder layer with
// the reconstruction loss).
//
//   C[M, Nc] = A[M, R] . B[Nc, R]^T      A: fp32 activations (x, or a gradient), streamed from HBM
//                                         B: a weight matrix (W for the forward, W^T for the data gradient), small
//   epilogues: store, ReLU, reconstruction loss (EPI 2), ReLU backward of the layer below (EPI 3: C = (A.B^T) where Y > 0).
//
// Two arithmetics behind one kernel template (`NP` = pieces per operand):
//   NP = 2  RQHIP_SPLIT_F16X2 (the product path, round 4): every row of A and every row of B is scaled by an EXACT power of two
//           (2^-e, e = exponent of the row's largest |value| - 14: the scaled row has its maximum in [2^14, 2^15), the top of fp16's range), every scaled value is
//           split into two fp16 pieces h = RN16(v), m = RN16(v - h) (11 + 11 significant bits and a sign: v - h - m is 0 or
//           +-2^-23 of the row maximum's binade), the product is hh + hm + mh -- three v_mfma_f32_32x32x16_f16 (products of two
//           fp16 values are exact in fp32, accumulation in fp32; dropped: mm <= 2^-22 of a product) -- and the epilogue
//           multiplies back by 2^(e_row + e_column), also exact.  The row maxima of A come from the kernel that wrote A
//           (every epilogue here can emit the row and column maxima of what it stores) or from rqhip_maxima.
//   NP = 3  RQHIP_SPLIT_BF16X3 (round 3; kept for A/B): three exact bf16 pieces h + m + l, six products, no scaling.
// Both are held to the same gate (tests/test_gpu_gemm_split.py): max error against fp64 <= the library fp32 GEMM's on the same
// inputs, on every operand family including the worst-case mantissas of the 11-bit split and cancellation-heavy rows.
//
// Mapping
//   * both operands are consumed along the reduction index as they lie in memory (a lane's operand = 8 consecutive r of
//     one row of A / one row of B): no transposition.  The weight is split once per forward by `weight_images_kernel` (all
//     layers of an MLP in ONE launch) into the stage-major image [R/16][piece][half][Nc] x 16 bytes.
//   * the product kernel (`gemm_f16_kernel`, f16x2, layers of 256 (mod 256) columns): tile = 128 rows x 256 columns, 4 waves
//     side by side (each 128 x 64), TWO persistent workgroups per CU; 16-deep stages: the fp32 rows of A are requested two
//     stages ahead, split by the VALU into a double-buffered LDS image (the only thing in LDS), a wave's B operands go from
//     the L2-resident image straight into registers one stage ahead, the A fragments of row block t + 1 are read while the
//     matrix instructions of block t run; whole rounds of the chip in 128-row tiles, the leftover in 64-row tiles; tiles
//     are handed out by atomic counters.
//   * `gemm_split_kernel` (both operands staged through LDS: round 3's loop) remains for the bf16x3 A/B arm (256 x 256 tiles,
//     8 waves, one workgroup per CU) and for 128-column tiles.
//   * results do not depend on which workgroup computes a tile, nor on the kernel (same products in the same order per
//     accumulator): bit-reproducible run to run and across the variants.
// Where the time goes (tools/gemm_probe2.py, GS_PROBE builds; 768 -> 512 at 100 000 rows, 340 us): tile prologue + epilogue +
// loop skeleton alone 105 us, matrix instructions alone + that 171 us, everything but the matrix instructions 251 us; the
// split arithmetic costs 77 us, the B loads 50, the LDS reads 40, the A loads 35, the stage barriers nothing; the same
// stores in a contiguous pattern would save 27 us (62 on the reconstruction epilogue).  DESIGN.md section 4.3d.
// The round-3 / round-4 schedule experiments live in tools/experiments/gemm_split_r0{3,4}_variants.hip.
#include "rqhip_common.h"

namespace rqhip {

typedef float gs_f32x16 __attribute__((ext_vector_type(16)));
typedef float gs_f32x4 __attribute__((ext_vector_type(4)));
typedef float gs_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 gs_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 gs_bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 gs_f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 gs_f16x2 __attribute__((ext_vector_type(2)));
typedef unsigned gs_u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned gs_u32x4 __attribute__((ext_vector_type(4)));
typedef int gs_i32x4 __attribute__((ext_vector_type(4)));

constexpr int kGsUB = 2;                    // a wave's tile is (32 TA) x (32 UB)
constexpr int kGsK = 16;                    // reduction depth of a stage = one K step of the matrix instruction
constexpr int kGsColmaxLds = 1024;          // column maxima are pre-reduced in LDS for Nc up to this (else straight to memory)
constexpr int kGsTailWords = kWeightImageTailWords;   // words behind a weight image that hold the tile dispensers (zero between launches)

// (a, b) -> packed bf16 pieces {piece(a), piece(b)}; a = h + m + l exactly (likewise b)
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
// (a, b) -> packed fp16 pieces: h = RN16(v), m = RN16(v - h)
__device__ __forceinline__ void gs_split2_f16(float a, float b, unsigned &h, unsigned &m) {
    const gs_f16x2 hh = __builtin_convertvector(gs_f32x2{a, b}, gs_f16x2);
    h = __builtin_bit_cast(unsigned, hh);
    const gs_f32x2 hf = __builtin_convertvector(hh, gs_f32x2);
    const gs_f16x2 mm = __builtin_convertvector(gs_f32x2{a - hf.x, b - hf.y}, gs_f16x2);
    m = __builtin_bit_cast(unsigned, mm);
}
// The scale exponent of a row / column whose largest |value| has these bits: the row is multiplied by 2^-e, which puts its maximum
// into [2^14, 2^15) -- the top of fp16's range (largest finite value 65504) -- so that the low piece m = RN16(v - h) of every entry
// down to 2^-16 of the row maximum is still a NORMAL-precision fp16 number (spacing of fp16 subnormals: 2^-24; with the maximum
// in [1, 2), round 4's first form, entries below a quarter of the maximum already lost bits of their low piece).
// 0 for 0 / inf / nan (such rows are not scaled).
constexpr int kGsF16Top = 14;
__device__ __forceinline__ int gs_exp_of_bits(unsigned b) {
    b &= 0x7fffffffu;
    const int e = (int)(b >> 23);
    return (b == 0u || e == 255) ? 0 : (e == 0 ? -126 : e - 127) - kGsF16Top;
}
__device__ __forceinline__ unsigned gs_abs_bits(float v) { return __builtin_bit_cast(unsigned, v) & 0x7fffffffu; }
// max of the bit patterns of |values| == bit pattern of the largest |value| for everything that is not a NaN; a NaN wins
// (its pattern is above +inf's), which marks the row / column "do not scale" exactly as an infinity does
__device__ __forceinline__ unsigned gs_umax(unsigned a, unsigned b) { return a > b ? a : b; }
// maximum over lanes 0 .. 31 (result in lane 31) and over lanes 32 .. 63 (result in lane 63): four row_shr steps inside the
// 16-lane DPP rows (zeros shifted in: the identity of an unsigned maximum), then lane 15 of rows 0 / 2 broadcast into rows 1 / 3
__device__ __forceinline__ unsigned gs_half_wave_umax(unsigned v) {
    v = gs_umax(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true));   // row_shr:1
    v = gs_umax(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true));   // row_shr:2
    v = gs_umax(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true));   // row_shr:4
    v = gs_umax(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true));   // row_shr:8
    v = gs_umax(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, true));   // row_bcast:15 into rows 1 and 3
    return v;
}

// ---- weight images: every layer of an MLP in one launch -----------------------------------------------------------------------
// image[s][piece][half][n] (16 bytes: r = 16 s + 8 half + j, j < 8) of src[n][r] (transpose == 0, src is [Nc, R]) or of
// src[r][n] (transpose == 1, src is [R, Nc]: the data gradient multiplies by W, i.e. B = W^T).  Behind the image: the tile
// dispensers (kGsTailWords words), and for NP = 2 the exponents of the Nc rows of B.
constexpr int kImgCols = 4;         // rows of B per block (one per wave)
constexpr int kImgMaxJobs = 16;
struct ImageJob {
    const float *src;
    unsigned *image;
    int Nc, R, transpose, np;
    int block0;                     // first block of this job in the launch
};
struct ImageJobs {
    ImageJob j[kImgMaxJobs];
    int n;
};

__global__ __launch_bounds__(256) void weight_images_kernel(const ImageJobs jobs) {
    int ji = 0;
#pragma unroll 1
    for (int i = 1; i < jobs.n; ++i)
        if ((int)blockIdx.x >= jobs.j[i].block0) ji = i;
    const ImageJob job = jobs.j[ji];
    const int blk = (int)blockIdx.x - job.block0;
    const int Nc = job.Nc, R = job.R, np = job.np, n0 = blk * kImgCols;
    unsigned *tail = job.image + (size_t)(R / kGsK) * 2 * np * Nc * 4;
    if (blk == 0 && threadIdx.x < kGsTailWords) tail[threadIdx.x] = 0u;   // the tile dispensers
    __shared__ unsigned s_max[kImgCols];
    if (threadIdx.x < kImgCols) s_max[threadIdx.x] = 0u;
    __syncthreads();
    const int pairs = R / 2, total = kImgCols * pairs;
    // thread -> (row of B, r pair): the pair index runs fastest along the contiguous direction of src
    auto locate = [&](int idx, int &nl, int &rp) {
        if (job.transpose) { nl = idx % kImgCols; rp = idx / kImgCols; }
        else { rp = idx % pairs; nl = idx / pairs; }
    };
    auto fetch = [&](int nl, int rp, float &a, float &b) {
        const int n = n0 + nl, r = 2 * rp;
        a = job.transpose ? job.src[(size_t)r * Nc + n] : job.src[(size_t)n * R + r];
        b = job.transpose ? job.src[(size_t)(r + 1) * Nc + n] : job.src[(size_t)n * R + r + 1];
    };
    if (np == 2) {   // the exponent of every row of B first
        if (job.transpose) {   // src[r][n]: thread = (row of B, 1 of 64 r phases), partial maxima met in LDS
            __shared__ unsigned s_part[256];
            const int nl = threadIdx.x % kImgCols;
            unsigned m = 0u;
            for (int r = threadIdx.x / kImgCols; r < R; r += 256 / kImgCols) m = gs_umax(m, gs_abs_bits(job.src[(size_t)r * Nc + n0 + nl]));
            s_part[threadIdx.x] = m;
            __syncthreads();
            if (threadIdx.x < kImgCols) {
                for (int i = threadIdx.x + kImgCols; i < 256; i += kImgCols) m = gs_umax(m, s_part[i]);
                s_max[threadIdx.x] = m;
            }
        } else {               // src[n][r]: a wave per row of B
            const int lane = threadIdx.x & 63, nl = threadIdx.x >> 6;
            unsigned m = 0u;
            for (int rp = lane; rp < pairs; rp += 64) {
                const gs_f32x2 v = *reinterpret_cast<const gs_f32x2 *>(job.src + (size_t)(n0 + nl) * R + 2 * rp);
                m = gs_umax(m, gs_umax(gs_abs_bits(v.x), gs_abs_bits(v.y)));
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) m = gs_umax(m, (unsigned)__shfl_xor((int)m, o, 64));
            if (lane == 0) s_max[nl] = m;
        }
        __syncthreads();
        if (threadIdx.x < kImgCols) reinterpret_cast<int *>(tail + kGsTailWords)[n0 + threadIdx.x] = gs_exp_of_bits(s_max[threadIdx.x]);
    }
    for (int idx = threadIdx.x; idx < total; idx += 256) {
        int nl, rp;
        float a, b;
        locate(idx, nl, rp);
        fetch(nl, rp, a, b);
        const int n = n0 + nl, r = 2 * rp;
        unsigned h, m, l = 0u;
        if (np == 2) {
            const int e = gs_exp_of_bits(s_max[nl]);
            gs_split2_f16(ldexpf(a, -e), ldexpf(b, -e), h, m);
        } else {
            gs_split2(a, b, h, m, l);
        }
        const int s = r >> 4, half = (r >> 3) & 1, j2 = (r & 7) >> 1;      // dword j2 of the 16-byte element
        const size_t base = ((size_t)(s * np) * 2 + half) * Nc + n;
        job.image[(base + 0 * 2 * (size_t)Nc) * 4 + j2] = h;
        job.image[(base + 1 * 2 * (size_t)Nc) * 4 + j2] = m;
        if (np == 3) job.image[(base + 2 * 2 * (size_t)Nc) * 4 + j2] = l;
    }
}

// ---- maxima of a matrix in one pass ---------------------------------------------------------------------------------------------
// row_max[m] / col_max[c]: bit patterns of the largest |value| of row m / column c of A [M, R] (optionally of A masked by Y > 0,
// which is then also written to `out`: the ReLU backward as its own pass, for callers that do not get it from an epilogue).
// 256 threads = 4 waves, a wave takes rows; col_max is maxed into atomically (zeroed by the caller).
__global__ __launch_bounds__(256) void maxima_kernel(const float *__restrict__ A, const float *__restrict__ Y, float *__restrict__ out,
                                                     long long M, int R, unsigned *__restrict__ row_max, unsigned *__restrict__ col_max) {
    extern __shared__ unsigned mx_smem[];            // [R] column maxima of this block (when col_max)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (col_max) {
        for (int c = threadIdx.x; c < R; c += 256) mx_smem[c] = 0u;
        __syncthreads();
    }
    constexpr int kQ = 4;                             // float4s per lane held for the column maxima of one 1024-column chunk
    // columns in chunks of 1024 (any R): the same wave takes the same rows in every chunk, so a row's maximum is carried from
    // chunk to chunk through row_max itself (written by lane 0, read back by lane 0)
    for (int c0 = 0; c0 < R; c0 += 256 * kQ) {
    const int R4 = (R - c0 < 256 * kQ ? R - c0 : 256 * kQ) / 4;
    unsigned cm[kQ][4];
#pragma unroll
    for (int q = 0; q < kQ; ++q)
#pragma unroll
        for (int j = 0; j < 4; ++j) cm[q][j] = 0u;
    for (long long row = (long long)blockIdx.x * 4 + wave; row < M; row += (long long)gridDim.x * 4) {
        const gs_f32x4 *src = reinterpret_cast<const gs_f32x4 *>(A + (size_t)row * R + c0);
        const gs_f32x4 *ys = Y ? reinterpret_cast<const gs_f32x4 *>(Y + (size_t)row * R + c0) : nullptr;
        gs_f32x4 *dst = out ? reinterpret_cast<gs_f32x4 *>(out + (size_t)row * R + c0) : nullptr;
        unsigned rm = 0u;
#pragma unroll
        for (int q = 0; q < kQ; ++q) {
            const int i = lane + 64 * q;
            if (i >= R4) break;
            gs_f32x4 v = src[i];
            if (ys) {   // threshold_backward(g, y, 0): 0 where y <= 0
                const gs_f32x4 y4 = ys[i];
                v.x = y4.x <= 0.0f ? 0.0f : v.x; v.y = y4.y <= 0.0f ? 0.0f : v.y;
                v.z = y4.z <= 0.0f ? 0.0f : v.z; v.w = y4.w <= 0.0f ? 0.0f : v.w;
                if (dst) dst[i] = v;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const unsigned b = gs_abs_bits(v[j]);
                rm = gs_umax(rm, b);
                cm[q][j] = gs_umax(cm[q][j], b);
            }
        }
        if (row_max) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) rm = gs_umax(rm, (unsigned)__shfl_xor((int)rm, o, 64));
            if (lane == 0) row_max[row] = c0 == 0 ? rm : gs_umax(rm, row_max[row]);
        }
    }
    if (col_max) {
#pragma unroll
        for (int q = 0; q < kQ; ++q) {
            const int i = lane + 64 * q;
            if (i >= R4) break;
#pragma unroll
            for (int j = 0; j < 4; ++j) atomicMax(&mx_smem[c0 + 4 * i + j], cm[q][j]);
        }
    }
    }   // column chunks
    if (col_max) {
        __syncthreads();
        for (int c = threadIdx.x; c < R; c += 256)
            if (mx_smem[c]) atomicMax(col_max + c, mx_smem[c]);
    }
}

struct GemmSplitParams {
    const float *A;          // [M, R]
    const unsigned *planes;  // weight image, see weight_images_kernel
    float *C;                // [M, Nc]
    long long M;
    int R, Nc;
    int n_col_tiles;
    // tiles 0 .. n_big - 1 are 256 rows high (rows [0, 256 rt_big)), the rest 64 rows high (from row 256 rt_big on)
    unsigned n_big, n_tiles;
    int rt_big;
    unsigned *counter;       // the tile dispensers behind the weight image (kGsTailWords words, zero between launches)
    int n_queues;            // gemm_f16_kernel: 1 (chip-wide dispenser) or 8 (one per XCD)
    int rt_fastest;          // gemm_f16_kernel, one queue: tiles in row-tile-fastest order (A/B)
    int static_tiles;        // gemm_f16_kernel: workgroup b takes tiles b, b + grid, ... (no dispenser: see the kernel)
    // EPI == 2 (the last decoder layer fused with the reconstruction loss): C receives (2 (A.B^T - X)) * row_scale, and
    // rowsum[ct][m] the squared error of row m over column tile ct.  EPI == 3: X is Y, the activation whose ReLU is undone
    const float *X;
    float *rowsum;
    float row_scale;
    // NP == 2: a_max [a_parts][M] bit patterns whose maximum over the parts is row m's largest |value| (an epilogue's c_rowmax, or
    // rqhip_maxima's row_max with one part); b_exp [Nc] lives behind the image
    const unsigned *a_max;
    int a_parts;
    const int *b_exp;
    // optional outputs of every epilogue: c_rowmax [column tiles][M] (part = column tile; plain stores), c_colmax [Nc]
    // (atomic maxima: zeroed by the caller) -- the scales of the kernels that read C next
    unsigned *c_rowmax, *c_colmax;
};

// developer-only phase-skipping probes of gs_tile2 (tools/ab_build.sh <name> gemm_split.hip -DGS_PROBE=<bits>; results are WRONG
// with any bit set): 1 no stage barriers, 2 no split / LDS writes, 4 no A loads, 8 no B loads, 16 no matrix instructions,
// 32 no LDS reads of A, 64 A loads in a contiguous pattern (same bytes, wrong values).  (Round 4 also had 128: epilogue loads / stores in
// a contiguous pattern -- what the transposing epilogue below was built on.)
#ifndef GS_PROBE
#define GS_PROBE 0
#endif
#ifndef GS_PROLOGUE_ORDER
#define GS_PROLOGUE_ORDER 1
#endif
#ifndef GS_FAST_EPILOGUE
#define GS_FAST_EPILOGUE 1
#endif
// pad between the (piece, half) regions of gs_tile2's A stage, dwords (developer A/B: tools/ab_build.sh pad16 gemm_split.hip -DGS_REGION_PAD=16)
#ifndef GS_REGION_PAD
#define GS_REGION_PAD 32
#endif

// DPP row (16 lanes) reductions: after four row_shr steps lane 15 of every 16-lane row holds the row's result (zeros shifted in)
__device__ __forceinline__ unsigned gs_row16_umax(unsigned v) {
    v = gs_umax(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true));
    v = gs_umax(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true));
    v = gs_umax(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true));
    v = gs_umax(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true));
    return v;
}
__device__ __forceinline__ float gs_row16_sum(float v) {   // fixed order: ((v + shr1) + shr2) + shr4) + shr8
    v = v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x111, 0xf, 0xf, true));
    v = v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x112, 0xf, 0xf, true));
    v = v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x114, 0xf, 0xf, true));
    v = v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x118, 0xf, 0xf, true));
    return v;
}

constexpr int kGsTS = 68;     // floats per row of a wave's transposition block (64 + 4: b128 writes of 16 rows hit 64 distinct banks)
// dynamic LDS the epilogue needs: a [32][kGsTS] transposition block per wave + the two [WN][ROWS] reduction arrays
__host__ __device__ constexpr size_t gs_epilogue_lds(int waves, int rows, int wn) { return (size_t)waves * 32 * kGsTS * 4 + (size_t)2 * wn * rows * 4; }

// The epilogue of a tile (shared by the tile loops below): undo the scales, apply EPI, store, emit maxima / row losses.
// The matrix instruction leaves a lane with ONE output row and 4-column pieces of it 32 bytes apart: stored as they lie, every
// store instruction touched 32 rows x 32 bytes (and the aux loads of EPI 2 / 3 likewise) -- a quarter of every 128-byte line per
// request; the same bytes in a contiguous pattern were measured 27 us (of 340) faster on 768 -> 512 and 62 us (of 417) on the
// reconstruction epilogue (tools/gemm_probe2.py).  So every 32-row block of a wave's 128 x 64 tile is transposed through a
// wave-private LDS block first: written as the accumulators lie, read back with 16 lanes along a row, and all the element-wise
// work (scales, ReLU / loss / mask, maxima) happens on that side -- a load / store instruction covers 4 rows x 256 contiguous
// bytes.  Row statistics (squared error, maximum) are reduced over the 16 lanes of a row by DPP, in a fixed order.
template <int EPI, int TA, int COLS, int NP, int WAVES>
__device__ __forceinline__ void gs_epilogue(const GemmSplitParams &p, gs_f32x16 (&acc)[TA][kGsUB], unsigned *sbuf, const int *s_aexp,
                                            unsigned *s_colmax, long long m0, int n0) {
    constexpr int kGsThreads = 64 * WAVES;
    constexpr int UB = kGsUB, WN = COLS / (32 * UB), WM = WAVES / WN, ROWS = WM * 32 * TA;
    static_assert(UB == 2, "a wave's tile is 64 columns wide");
    // (the thread number is laundered through an empty asm: everything below derives from it, so the compiler cannot form the
    // epilogue's addresses before the main loop and carry them through it -- the loop runs at the register limit)
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63, wave = tid >> 6;
    const int il = lane & 31, h = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;
    const int cl = lane & 15, rl = lane >> 4;                 // after the transposition: column quad cl of the wave's 64, row 4 k + rl
    float *tb = reinterpret_cast<float *>(sbuf) + (size_t)wave * 32 * kGsTS;
    float *red = reinterpret_cast<float *>(sbuf) + (size_t)WAVES * 32 * kGsTS;       // [WN][ROWS] squared errors
    unsigned *mred = reinterpret_cast<unsigned *>(red) + WN * ROWS;                   // [WN][ROWS] row maxima
    const bool want_rowmax = p.c_rowmax != nullptr, want_colmax = p.c_colmax != nullptr;
    const int colw = n0 + 64 * wn + 4 * cl;                   // this lane's four columns
    gs_i32x4 ec = {0, 0, 0, 0};
    if (NP == 2) ec = *reinterpret_cast<const gs_i32x4 *>(p.b_exp + colw);
    unsigned cmx[4] = {0u, 0u, 0u, 0u};
#if GS_FAST_EPILOGUE
    // ---- the straight-line form (round 6): a FULL tile (no row past M) whose launch wants both maxima -- every launch of a training step but
    // the last row tile's.  The general loop below is 3 basic blocks per row quad (the row test around the store, the uniform tests of the two
    // maxima pointers, the one-lane LDS store of a row statistic), each starting with its own LDS read and ending in a drain
    // (`s_waitcnt vmcnt(0) lgkmcnt(0)`): ~400 cycles per quad, 32 quads per 128-row tile, 8 % of a 768-deep tile and a third of a 128-deep
    // one.  Here a 32-row block is ONE basic block: its 8 + 8 LDS reads and (EPI 2 / 3) the NEXT block's 8 aux loads are issued up front,
    // stores leave without a wait, the one-lane LDS stores of the row statistics become 64-lane stores whose other lanes hit the
    // transposition block's padding columns, result addresses are one per-lane base + uniform offsets.  Same arithmetic per element, same
    // (order-free) maxima, same fixed-order row sums: identical result bits.
    if (NP == 2 && want_rowmax && want_colmax && m0 + ROWS <= p.M) {
        const size_t row0 = (size_t)(m0 + 32 * TA * wm + rl);
        float *cbase = p.C + row0 * p.Nc + colw;
        const float *xbase = EPI >= 2 ? p.X + row0 * p.Nc + colw : nullptr;
        unsigned *pad = reinterpret_cast<unsigned *>(tb) + il * kGsTS + 64 + h;       // this lane's own padding word of the block
        gs_f32x4 xn[8];
        if (EPI >= 2) {
#pragma unroll
            for (int k = 0; k < 8; ++k) xn[k] = *reinterpret_cast<const gs_f32x4 *>(xbase + (size_t)(4 * k) * p.Nc);
        }
#pragma unroll
        for (int t = 0; t < TA; ++t) {
#pragma unroll
            for (int u = 0; u < UB; ++u)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    *reinterpret_cast<gs_f32x4 *>(tb + il * kGsTS + 32 * u + 8 * g + 4 * h) =
                        gs_f32x4{acc[t][u][4 * g], acc[t][u][4 * g + 1], acc[t][u][4 * g + 2], acc[t][u][4 * g + 3]};
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            gs_f32x4 v8[8], x8[8];
            int er8[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                v8[k] = *reinterpret_cast<const gs_f32x4 *>(tb + (4 * k + rl) * kGsTS + 4 * cl);
                er8[k] = s_aexp[32 * TA * wm + 32 * t + 4 * k + rl];
            }
            if (EPI >= 2) {
#pragma unroll
                for (int k = 0; k < 8; ++k) x8[k] = xn[k];
                if (t + 1 < TA) {
#pragma unroll
                    for (int k = 0; k < 8; ++k) xn[k] = *reinterpret_cast<const gs_f32x4 *>(xbase + (size_t)(32 * (t + 1) + 4 * k) * p.Nc);
                }
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int rloc = 32 * TA * wm + 32 * t + 4 * k + rl;
                gs_f32x4 v = v8[k];
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = ldexpf(v[j], er8[k] + ec[j]);
                const gs_f32x4 x4 = EPI >= 2 ? x8[k] : gs_f32x4{0.f, 0.f, 0.f, 0.f};
                if (EPI == 1) {
                    v.x = v.x < 0.0f ? 0.0f : v.x; v.y = v.y < 0.0f ? 0.0f : v.y;
                    v.z = v.z < 0.0f ? 0.0f : v.z; v.w = v.w < 0.0f ? 0.0f : v.w;
                }
                float sq = 0.0f;
                if (EPI == 2) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float d = v[j] - x4[j];
                        sq = sq + d * d;
                        v[j] = (2.0f * d) * p.row_scale;
                    }
                }
                if (EPI == 3) {
                    v.x = x4.x <= 0.0f ? 0.0f : v.x; v.y = x4.y <= 0.0f ? 0.0f : v.y;
                    v.z = x4.z <= 0.0f ? 0.0f : v.z; v.w = x4.w <= 0.0f ? 0.0f : v.w;
                }
                unsigned rmx = 0u;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const unsigned bb = gs_abs_bits(v[j]);
                    rmx = gs_umax(rmx, bb);
                    cmx[j] = gs_umax(cmx[j], bb);
                }
                *reinterpret_cast<gs_f32x4 *>(cbase + (size_t)(32 * t + 4 * k) * p.Nc) = v;
                if (EPI == 2) {
                    const float s16 = gs_row16_sum(sq);
                    float *dst = cl == 15 ? red + wn * ROWS + rloc : reinterpret_cast<float *>(pad);
                    *dst = s16;
                }
                const unsigned m16 = gs_row16_umax(rmx);
                unsigned *dstm = cl == 15 ? mred + wn * ROWS + rloc : pad;
                *dstm = m16;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    } else
#endif
    {
    // aux values (X of EPI 2, Y of EPI 3) are requested kAux row quads ahead of their use
    constexpr int kAux = 4, NQ = 8 * TA;
    gs_f32x4 xa[kAux];
    auto aux_of = [&](int q) {
        long long r = m0 + 32 * TA * wm + 4 * q + rl;        // (q = 8 t + k: the row quads of the wave's tile in order)
        r = r < p.M ? r : p.M - 1;
        return *reinterpret_cast<const gs_f32x4 *>(p.X + (size_t)r * p.Nc + colw);
    };
    if (EPI >= 2) {
#pragma unroll
        for (int q = 0; q < kAux; ++q) xa[q] = aux_of(q);
    }
#pragma unroll
    for (int t = 0; t < TA; ++t) {
        // acc[t][u][r]: row 32 t + il of the wave's tile, column 32 u + 8 (r >> 2) + 4 h + (r & 3)
#pragma unroll
        for (int u = 0; u < UB; ++u)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *reinterpret_cast<gs_f32x4 *>(tb + il * kGsTS + 32 * u + 8 * g + 4 * h) =
                    gs_f32x4{acc[t][u][4 * g], acc[t][u][4 * g + 1], acc[t][u][4 * g + 2], acc[t][u][4 * g + 3]};
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int q = 8 * t + k;
            const int rloc = 32 * TA * wm + 32 * t + 4 * k + rl;     // row inside the workgroup's tile
            const long long grow = m0 + rloc;
            gs_f32x4 v = *reinterpret_cast<const gs_f32x4 *>(tb + (4 * k + rl) * kGsTS + 4 * cl);
            if (NP == 2) {   // undo the row and column scales (exact)
                const int er = s_aexp[rloc];
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = ldexpf(v[j], er + ec[j]);
            }
            gs_f32x4 x4 = {0.f, 0.f, 0.f, 0.f};
            if (EPI >= 2) {
                x4 = xa[q % kAux];
                if (q + kAux < NQ) xa[q % kAux] = aux_of(q + kAux);
            }
            if (EPI == 1) {   // (a NaN stays a NaN, as torch.relu)
                v.x = v.x < 0.0f ? 0.0f : v.x; v.y = v.y < 0.0f ? 0.0f : v.y;
                v.z = v.z < 0.0f ? 0.0f : v.z; v.w = v.w < 0.0f ? 0.0f : v.w;
            }
            float sq = 0.0f;
            if (EPI == 2) {   // as csrc/recon_loss.hip: d = x_hat - x, loss += d d, gradient (2 d) row_scale
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float d = v[j] - x4[j];
                    sq = sq + d * d;
                    v[j] = (2.0f * d) * p.row_scale;
                }
            }
            if (EPI == 3) {   // threshold_backward(g, y, 0): 0 where y <= 0
                v.x = x4.x <= 0.0f ? 0.0f : v.x; v.y = x4.y <= 0.0f ? 0.0f : v.y;
                v.z = x4.z <= 0.0f ? 0.0f : v.z; v.w = x4.w <= 0.0f ? 0.0f : v.w;
            }
            unsigned rmx = 0u;
            if (grow < p.M) {
                if (want_rowmax | want_colmax) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const unsigned bb = gs_abs_bits(v[j]);
                        rmx = gs_umax(rmx, bb);
                        cmx[j] = gs_umax(cmx[j], bb);
                    }
                }
                *reinterpret_cast<gs_f32x4 *>(p.C + (size_t)grow * p.Nc + colw) = v;   // (4 rows x 256 contiguous bytes per instruction)
            }
            // the row's 64 columns of this wave lie in the 16 lanes of a DPP row: lane 15 of it ends up with the whole
            if (EPI == 2) {
                const float s16 = gs_row16_sum(sq);
                if (cl == 15) red[wn * ROWS + rloc] = s16;
            }
            if (want_rowmax) {
                const unsigned m16 = gs_row16_umax(rmx);
                if (cl == 15) mred[wn * ROWS + rloc] = m16;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();            // (the block is rewritten by the next t)
    }
    }
    if (want_colmax) {   // a lane's four columns over all its rows; the four lanes that share them (rl = 0 .. 3), then one atomic per column
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            unsigned m = gs_umax(cmx[j], (unsigned)__shfl_xor((int)cmx[j], 16, 64));
            m = gs_umax(m, (unsigned)__shfl_xor((int)m, 32, 64));
            if (rl == 0 && m != 0u) {
                if (p.Nc <= kGsColmaxLds) atomicMax(&s_colmax[colw + j], m);
                else atomicMax(p.c_colmax + colw + j, m);
            }
        }
    }
    if (want_rowmax || EPI == 2) {
        // the WN column waves of a row meet in LDS (red / mred lie behind the transposition blocks), in wave order
        __syncthreads();
        if (want_rowmax) {
            // (parts are counted per 256 columns by the callers: a 512-column tile writes its maximum into both of its parts)
            unsigned *dst = p.c_rowmax + (size_t)(n0 / (COLS > 256 ? 256 : COLS)) * p.M;
            for (int r = tid; r < ROWS; r += kGsThreads) {
                if (m0 + r >= p.M) continue;
                unsigned mx = mred[r];
#pragma unroll
                for (int w = 1; w < WN; ++w) mx = gs_umax(mx, mred[w * ROWS + r]);
                dst[m0 + r] = mx;
                if (COLS > 256) dst[p.M + m0 + r] = mx;
            }
        }
        if (EPI == 2) {
            for (int r = tid; r < ROWS; r += kGsThreads) {
                if (m0 + r >= p.M) continue;
                float sum = red[r];
#pragma unroll
                for (int w = 1; w < WN; ++w) sum = sum + red[w * ROWS + r];
                p.rowsum[(size_t)(n0 / COLS) * p.M + m0 + r] = sum;
            }
        }
        // (the persistent loop's barrier at its top keeps the next tile's staging off these arrays)
    }
}

// one output tile of ROWS x COLS: 8 waves of (32 TA) x 64, WN = COLS / 64 of them side by side
// EPI: 0 = store, 1 = ReLU, 2 = reconstruction loss, 3 = masked by Y > 0 (see GemmSplitParams)
template <int EPI, int TA, int COLS, int NP, int WAVES>
__device__ __forceinline__ void gs_tile(const GemmSplitParams &p, unsigned *sbuf, int *s_aexp, unsigned *s_colmax, long long m0, int n0) {
    constexpr int kGsWaves = WAVES, kGsThreads = 64 * WAVES;
    constexpr int UB = kGsUB, WN = COLS / (32 * UB), WM = kGsWaves / WN;
    constexpr int ROWS = WM * 32 * TA, AQ = (ROWS * 4 + kGsThreads - 1) / kGsThreads;   // float4s of A per thread and stage
    constexpr int PA = NP * 2 * ROWS * 4, PB = NP * 2 * COLS * 4;                       // dwords per stage image
    constexpr int BQ = (2 * NP * COLS + kGsThreads - 1) / kGsThreads;                   // 16-byte elements of B per thread and stage
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int il = lane & 31, h = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;
    const int n_stage = p.R / kGsK;
    constexpr int APASS = kGsThreads / 4;                  // rows one staging pass of the workgroup covers
    // staging roles: A -- thread (row = tid >> 2 (+ APASS q), kq = tid & 3) owns 4 consecutive r of one row; B -- BQ
    // 16-byte elements of the stage's weight image per thread
    const int arow = tid >> 2, akq = tid & 3;
    constexpr bool kAllLive = (ROWS * 4) % kGsThreads == 0;
    bool a_live[AQ];
    const float *asrc[AQ];
    int a_e[AQ];                                           // NP == 2: exponents of this thread's rows of A
#pragma unroll
    for (int q = 0; q < AQ; ++q) {
        a_live[q] = kAllLive || arow + APASS * q < ROWS;
        // rows past M (and the slots of a short tile) re-read row M - 1 and are never stored: every load of the stage loop is
        // unconditional (as gs_tile2), the loop body one basic block without a drain of the vector-memory counter
        long long arow_g = m0 + arow + APASS * q;
        arow_g = arow_g < p.M ? arow_g : p.M - 1;
        asrc[q] = p.A + (size_t)arow_g * p.R + 4 * akq;
        a_e[q] = 0;
        if (NP == 2) {
            // (four parts per round, their loads independent of one another: a dependent load per part put a_parts memory
            // latencies in front of every tile -- 12 parts x 2 rows x ~0.7 us on a 768-column producer)
            unsigned mx = 0u;
            const unsigned *am = p.a_max + arow_g;
            for (int part = 0; part < p.a_parts; part += 4) {
                unsigned v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = am[(size_t)(part + j < p.a_parts ? part + j : part) * p.M];
                mx = gs_umax(gs_umax(mx, gs_umax(v[0], v[1])), gs_umax(v[2], v[3]));
            }
            a_e[q] = gs_exp_of_bits(mx);
            if (a_live[q] && akq == 0) s_aexp[arow + APASS * q] = a_e[q];   // (read by the epilogue, many barriers later)
        }
    }

    gs_f32x16 acc[TA][UB];
#pragma unroll
    for (int t = 0; t < TA; ++t)
#pragma unroll
        for (int u = 0; u < UB; ++u)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][u][r] = 0.0f;

    gs_f32x4 ra0[AQ], ra1[AQ];   // rows of A: two stages in flight (requested two iterations before they are split)
    gs_u32x4 rb[BQ];
    // (both fetches are UNCONDITIONAL: past the last stage they re-read it.  A load inside `if (stage < n_stage)` makes the
    // compiler's s_waitcnt insertion assume it may not have been issued, and the wait for the weight image then drains it)
    auto fetchA = [&](int stage, gs_f32x4 *dst) {
        stage = stage < n_stage ? stage : n_stage - 1;
#pragma unroll
        for (int q = 0; q < AQ; ++q) dst[q] = *reinterpret_cast<const gs_f32x4 *>(asrc[q] + stage * kGsK);   // (non-temporal loads of A: +3 ... +4 %)
    };
    auto fetchB = [&](int stage) {
        stage = stage < n_stage ? stage : n_stage - 1;
        // stage image: [piece][half][Nc] 16-byte elements; this tile's part is columns n0 .. n0 + COLS - 1 of each of the
        // 2 NP (piece, half) rows: element e = tid + 512 q  ->  (ph = e / COLS, col = e % COLS)
        const gs_u32x4 *img = reinterpret_cast<const gs_u32x4 *>(p.planes) + (size_t)stage * 2 * NP * p.Nc + n0;
#pragma unroll
        for (int q = 0; q < BQ; ++q) {
            const int e = tid + kGsThreads * q;
            if ((2 * NP * COLS) % kGsThreads != 0 && e >= 2 * NP * COLS) continue;
            rb[q] = img[(size_t)(e / COLS) * p.Nc + (e % COLS)];
        }
    };
    auto stash = [&](int buf, const gs_f32x4 *ra) {
        unsigned *dA = sbuf + buf * (PA + PB), *dB = dA + PA;
#pragma unroll
        for (int q = 0; q < AQ; ++q) {
            if (!kAllLive && !a_live[q]) continue;
            unsigned h01, m01, l01 = 0u, h23, m23, l23 = 0u;
            if (NP == 2) {
                gs_split2_f16(ldexpf(ra[q].x, -a_e[q]), ldexpf(ra[q].y, -a_e[q]), h01, m01);
                gs_split2_f16(ldexpf(ra[q].z, -a_e[q]), ldexpf(ra[q].w, -a_e[q]), h23, m23);
            } else {
                gs_split2(ra[q].x, ra[q].y, h01, m01, l01);
                gs_split2(ra[q].z, ra[q].w, h23, m23, l23);
            }
            // element [piece][half = akq >> 1][row] is 16 bytes = r 8 half .. 8 half + 7; this thread fills its half (akq & 1)
            unsigned *d = dA + (((akq >> 1) * ROWS) + arow + APASS * q) * 4 + 2 * (akq & 1);
            *reinterpret_cast<gs_u32x2 *>(d + 0 * 2 * ROWS * 4) = gs_u32x2{h01, h23};
            *reinterpret_cast<gs_u32x2 *>(d + 1 * 2 * ROWS * 4) = gs_u32x2{m01, m23};
            if (NP == 3) *reinterpret_cast<gs_u32x2 *>(d + 2 * 2 * ROWS * 4) = gs_u32x2{l01, l23};
        }
#pragma unroll
        for (int q = 0; q < BQ; ++q) {
            const int e = tid + kGsThreads * q;
            if ((2 * NP * COLS) % kGsThreads != 0 && e >= 2 * NP * COLS) continue;
            *reinterpret_cast<gs_u32x4 *>(dB + (size_t)e * 4) = rb[q];   // [ph][col] order == the image's
        }
    };
    auto multiply = [&](int buf) {
        const gs_bf16x8 *aA = reinterpret_cast<const gs_bf16x8 *>(sbuf + buf * (PA + PB));
        const gs_bf16x8 *bB = reinterpret_cast<const gs_bf16x8 *>(sbuf + buf * (PA + PB) + PA);
        // the weight columns take the instruction's ROW role: the accumulator is the tile transposed, a lane holds four
        // consecutive columns of one output row per register quad -> 16-byte result stores
        if constexpr (NP == 2) {
            // both column halves' operands first (16 registers), then every row block once: 2 TA + 4 LDS reads per 6 TA matrix
            // instructions.  (With three matrix instructions per block the u-outer order of the bf16 path -- 2 (2 + 2 TA) reads
            // -- puts the LDS read time of eight waves next to the matrix time of a stage.)
            gs_f16x8 fb[UB][2];
#pragma unroll
            for (int u = 0; u < UB; ++u)
#pragma unroll
                for (int pc = 0; pc < 2; ++pc)
                    fb[u][pc] = __builtin_bit_cast(gs_f16x8, bB[(pc * 2 + h) * COLS + wn * 32 * UB + 32 * u + il]);
#pragma unroll
            for (int t = 0; t < TA; ++t) {
                gs_f16x8 fa[2];
#pragma unroll
                for (int pc = 0; pc < 2; ++pc)
                    fa[pc] = __builtin_bit_cast(gs_f16x8, aA[(pc * 2 + h) * ROWS + wm * 32 * TA + 32 * t + il]);
#pragma unroll
                for (int u = 0; u < UB; ++u) {
                    gs_f32x16 c16 = acc[t][u];
                    c16 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb[u][0], fa[1], c16, 0, 0, 0);   // m h (smallest first)
                    c16 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb[u][1], fa[0], c16, 0, 0, 0);   // h m
                    c16 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb[u][0], fa[0], c16, 0, 0, 0);   // h h
                    acc[t][u] = c16;
                }
            }
        } else {
            // column half outer, row block inner: 12 + 12 operand registers live (the row blocks are read once per column half:
            // 30 instead of 18 LDS reads per stage, which the LDS has room for at six matrix instructions per block)
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                gs_bf16x8 b[3];
#pragma unroll
                for (int pc = 0; pc < 3; ++pc) b[pc] = bB[(pc * 2 + h) * COLS + wn * 32 * UB + 32 * u + il];
#pragma unroll
                for (int t = 0; t < TA; ++t) {
                    gs_bf16x8 a[3];
#pragma unroll
                    for (int pc = 0; pc < 3; ++pc) a[pc] = aA[(pc * 2 + h) * ROWS + wm * 32 * TA + 32 * t + il];
                    gs_f32x16 c16 = acc[t][u];
                    c16 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[1], a[1], c16, 0, 0, 0);   // m m (smallest first)
                    c16 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[0], a[2], c16, 0, 0, 0);   // l h
                    c16 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[2], a[0], c16, 0, 0, 0);   // h l
                    c16 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[0], a[1], c16, 0, 0, 0);   // m h
                    c16 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[1], a[0], c16, 0, 0, 0);   // h m
                    c16 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[0], a[0], c16, 0, 0, 0);   // h h
                    acc[t][u] = c16;
                }
            }
        }
    };

    // Order of the requests inside an iteration: the weight image FIRST, then the A rows.  Loads complete in order
    // (vmcnt): the next iteration waits for the image it stages, and with the A rows requested before it that wait also
    // drained the A rows of the stage after -- their latency had ONE iteration to hide in, not two.
    // A rows are requested TWO iterations before they are split (scattered 64-byte pieces of 256 rows: their latency is
    // longer than one iteration's matrix work), the weight image (L2-resident) one iteration before.
#if GS_PROLOGUE_ORDER
    // (the requests in the order of one loop iteration -- B, A1 | B, A0 -- behind stage 0's: see gs_tile2)
    gs_f32x4 rp[AQ];
    fetchA(0, rp);
    fetchB(0);
    __builtin_amdgcn_sched_barrier(0);
    fetchA(1, ra1);
    __builtin_amdgcn_sched_barrier(0);
    stash(0, rp);
    __builtin_amdgcn_sched_barrier(0);
    fetchB(1);
    __builtin_amdgcn_sched_barrier(0);
    fetchA(2, ra0);
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
#else
    fetchA(0, ra0);
    fetchB(0);
    fetchA(1, ra1);
    stash(0, ra0);
    fetchB(1);
    fetchA(2, ra0);
    __syncthreads();
#endif
    // Every wave stages the next stage first, then multiplies the current one.
    const int n_pair = n_stage & ~1;
    for (int c = 0; c < n_pair; c += 2) {
        stash(1, ra1);                             // stage c + 1
        fetchB(c + 2);
        fetchA(c + 3, ra1);
        multiply(0);
        __syncthreads();
        stash(0, ra0);                             // stage c + 2 (past the end: the last stage again, never multiplied)
        fetchB(c + 3);
        fetchA(c + 4, ra0);
        multiply(1);
        __syncthreads();
    }
    if (n_stage & 1) {                             // the last stage of an odd count lies in buffer 0
        multiply(0);
        __syncthreads();
    }

    gs_epilogue<EPI, TA, COLS, NP, WAVES>(p, acc, sbuf, s_aexp, s_colmax, m0, n0);
}

// ---- the product tile loop (round 4): f16x2, 256-column tiles, 4 waves side by side (each 32 TA rows x 64 columns) -------------
// What differs from gs_tile above:
//   * a wave's weight columns are its own (no other wave of the workgroup multiplies by them), so its B operands go from the
//     (L2-resident) weight image straight into the matrix instruction's registers, one stage ahead: no LDS round trip for B
//     (a third of the LDS reads and two thirds of the LDS write bytes of a stage), the LDS holds the split A stage only;
//   * the A fragments of row block t + 1 are read while the matrix instructions of block t run (two register sets);
//   * every load is unconditional (rows past M re-read row M - 1 and are never stored): the loop body is ONE basic block,
//     which lets the scheduler overlap the split arithmetic of the next stage with the matrix instructions of this one.
// Same products in the same order per accumulator as gs_tile<.., NP = 2, ..>: identical result bits.
template <int EPI, int TA, int COLS = 256>
__device__ __forceinline__ void gs_tile2(const GemmSplitParams &p, unsigned *sbuf, int *s_aexp, unsigned *s_colmax, long long m0, int n0) {
    // COLS = 256: 4 waves side by side (the product); COLS = 512: 8 waves side by side, ONE workgroup per CU -- the full output width of the
    // 512-column layers, so that a strip of A is fetched and split once per row tile (A/B arm, tile_rows = -5)
    constexpr int WAVES = COLS / 64, UB = kGsUB, kThreads = 64 * WAVES;
    constexpr int ROWS = 32 * TA;
    constexpr int AQ = (ROWS * 4 + kThreads - 1) / kThreads;         // float4s of A per thread and stage (2 for 128 rows, 1 for 32)
    constexpr bool kAllLive = (ROWS * 4) % kThreads == 0;
    constexpr int kRegion = ROWS * 4 + GS_REGION_PAD;                // dwords per [piece][half] region of the A stage (+ a pad so that
                                                                     // the two regions a 16-lane ds_write group fills land in different banks)
    constexpr int PA = 4 * kRegion;                                  // dwords per A stage
    const int tid = threadIdx.x, lane = tid & 63, wn = tid >> 6;
    const int il = lane & 31, h = lane >> 5;
    const int n_stage = p.R / kGsK;
    constexpr int APASS = kThreads / 4;
    const int arow = tid >> 2, akq = tid & 3;
    const float *asrc[AQ];
    int a_e[AQ];
    bool a_live[AQ];
#pragma unroll
    for (int q = 0; q < AQ; ++q) {
        a_live[q] = kAllLive || arow + APASS * q < ROWS;
        long long arow_g = m0 + arow + APASS * q;
        arow_g = arow_g < p.M ? arow_g : p.M - 1;
        asrc[q] = p.A + (size_t)arow_g * p.R + 4 * akq;
        unsigned mx = 0u;
        const unsigned *am = p.a_max + arow_g;
        for (int part = 0; part < p.a_parts; part += 4) {
            unsigned v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = am[(size_t)(part + j < p.a_parts ? part + j : part) * p.M];
            mx = gs_umax(gs_umax(mx, gs_umax(v[0], v[1])), gs_umax(v[2], v[3]));
        }
        a_e[q] = gs_exp_of_bits(mx);
        if (a_live[q] && akq == 0) s_aexp[arow + APASS * q] = a_e[q];   // (read by the epilogue, many barriers later)
    }

    gs_f32x16 acc[TA][UB];
#pragma unroll
    for (int t = 0; t < TA; ++t)
#pragma unroll
        for (int u = 0; u < UB; ++u)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][u][r] = 0.0f;

    // this lane's B elements of a stage: image[stage][pc][h][n0 + 64 wn + 32 u + il]
    const gs_u32x4 *bsrc = reinterpret_cast<const gs_u32x4 *>(p.planes) + (size_t)h * p.Nc + n0 + 64 * wn + il;
    const size_t b_stage = (size_t)4 * p.Nc, b_piece = (size_t)2 * p.Nc;
    gs_f32x4 ra0[AQ], ra1[AQ];
    gs_u32x4 fb0[UB][2], fb1[UB][2];
    auto fetchA = [&](int stage, gs_f32x4 *dst) {
        stage = stage < n_stage ? stage : n_stage - 1;
#pragma unroll
        for (int q = 0; q < AQ; ++q) {
            if (GS_PROBE & 4) dst[q] = gs_f32x4{1.f + stage, 2.f, 3.f, 4.f};
            else if (GS_PROBE & 64) {   // the strip of ROWS whole rows is one contiguous block: stage s takes its s-th ROWS x 64 bytes
                size_t o = (size_t)m0 * p.R + (size_t)stage * ROWS * kGsK + (size_t)(tid + kThreads * q) * 4;
                if (o + 4 > (size_t)p.M * p.R) o = 0;
                dst[q] = *reinterpret_cast<const gs_f32x4 *>(p.A + o);
            }
            else dst[q] = *reinterpret_cast<const gs_f32x4 *>(asrc[q] + stage * kGsK);
        }
    };
    auto fetchB = [&](int stage, gs_u32x4 (*dst)[2]) {
        stage = stage < n_stage ? stage : n_stage - 1;
        const gs_u32x4 *img = bsrc + (size_t)stage * b_stage;
#pragma unroll
        for (int u = 0; u < UB; ++u)
#pragma unroll
            for (int pc = 0; pc < 2; ++pc) {
                if (GS_PROBE & 8) dst[u][pc] = gs_u32x4{(unsigned)stage, 1u, 2u, 3u};
                else dst[u][pc] = img[(size_t)pc * b_piece + 32 * u];
            }
    };
    auto stash = [&](int buf, const gs_f32x4 *ra) {
        unsigned *dA = sbuf + buf * PA;
        if (GS_PROBE & 2) return;
#pragma unroll
        for (int q = 0; q < AQ; ++q) {
            if (!kAllLive && !a_live[q]) continue;
            unsigned h01, m01, h23, m23;
            gs_split2_f16(ldexpf(ra[q].x, -a_e[q]), ldexpf(ra[q].y, -a_e[q]), h01, m01);
            gs_split2_f16(ldexpf(ra[q].z, -a_e[q]), ldexpf(ra[q].w, -a_e[q]), h23, m23);
            // region [piece][half = akq >> 1], element [row] = 16 bytes (r 8 half .. 8 half + 7); this thread fills its 8 bytes
            unsigned *d = dA + (akq >> 1) * kRegion + (arow + APASS * q) * 4 + 2 * (akq & 1);
            *reinterpret_cast<gs_u32x2 *>(d) = gs_u32x2{h01, h23};
            *reinterpret_cast<gs_u32x2 *>(d + 2 * kRegion) = gs_u32x2{m01, m23};
        }
    };
    auto loadA = [&](int buf, int t, gs_f16x8 *fa) {
        const gs_u32x4 *aA = reinterpret_cast<const gs_u32x4 *>(sbuf + buf * PA);
#pragma unroll
        for (int pc = 0; pc < 2; ++pc) {
            if (GS_PROBE & 32) fa[pc] = gs_f16x8{(_Float16)(float)(t + buf), 1, 2, 3, 4, 5, 6, 7};
            else fa[pc] = __builtin_bit_cast(gs_f16x8, aA[(pc * 2 + h) * (kRegion / 4) + 32 * t + il]);
        }
    };
    auto multiply = [&](int buf, const gs_u32x4 (*fbr)[2]) {
        gs_f16x8 fa[2][2];
        loadA(buf, 0, fa[0]);
#pragma unroll
        for (int t = 0; t < TA; ++t) {
            if (t + 1 < TA) loadA(buf, t + 1, fa[(t + 1) & 1]);
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                const gs_f16x8 bh = __builtin_bit_cast(gs_f16x8, fbr[u][0]), bm = __builtin_bit_cast(gs_f16x8, fbr[u][1]);
                gs_f32x16 c16 = acc[t][u];
                if (GS_PROBE & 16) {   // (keeps every operand alive without the matrix pipe)
                    c16[0] += (float)bh[0] + (float)bm[1] + (float)fa[t & 1][0][2] + (float)fa[t & 1][1][3];
                } else {
                    c16 = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh, fa[t & 1][1], c16, 0, 0, 0);   // m h (smallest first)
                    c16 = __builtin_amdgcn_mfma_f32_32x32x16_f16(bm, fa[t & 1][0], c16, 0, 0, 0);   // h m
                    c16 = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh, fa[t & 1][0], c16, 0, 0, 0);   // h h
                }
                acc[t][u] = c16;
            }
        }
    };

#if GS_PROLOGUE_ORDER
    // The prologue's requests in the ORDER OF ONE LOOP ITERATION (A1, B0, A0, B1), behind the rows of stage 0: vector-memory results arrive in
    // request order and the compiler's s_waitcnt on a loop header is the minimum over the edges into it -- with the prologue's own order
    // (A0 B0 A1 | B1 A0) the header got vmcnt(5/4) and the middle of the first half vmcnt(3/2), i.e. every half iteration waited for the rows
    // requested ONE half iteration earlier (HBM latency) and for image blocks requested a few hundred cycles earlier (L2 latency), where
    // the steady state allows vmcnt(11/10) and (7/6): two stages for the rows, one for the image.
    gs_f32x4 rp[AQ];
    fetchA(0, rp);
    __builtin_amdgcn_sched_barrier(0);
    fetchA(1, ra1);
    __builtin_amdgcn_sched_barrier(0);
    fetchB(0, fb0);
    __builtin_amdgcn_sched_barrier(0);
    fetchA(2, ra0);
    __builtin_amdgcn_sched_barrier(0);
    fetchB(1, fb1);
    __builtin_amdgcn_sched_barrier(0);
    stash(0, rp);
    __syncthreads();
#else
    fetchA(0, ra0);
    fetchB(0, fb0);
    fetchA(1, ra1);
    stash(0, ra0);
    fetchB(1, fb1);
    fetchA(2, ra0);
    __syncthreads();
#endif
    const int n_pair = n_stage & ~1;
    for (int c = 0; c < n_pair; c += 2) {
        stash(1, ra1);                             // stage c + 1
        fetchA(c + 3, ra1);
        multiply(0, fb0);
        fetchB(c + 2, fb0);
        if (!(GS_PROBE & 1)) __syncthreads();
        stash(0, ra0);                             // stage c + 2 (past the end: the last stage again, never multiplied)
        fetchA(c + 4, ra0);
        multiply(1, fb1);
        fetchB(c + 3, fb1);
        if (!(GS_PROBE & 1)) __syncthreads();
    }
    if (n_stage & 1) {                             // the last stage of an odd count lies in buffer 0
        multiply(0, fb0);
        __syncthreads();
    }
    gs_epilogue<EPI, TA, COLS, 2, WAVES>(p, acc, sbuf, s_aexp, s_colmax, m0, n0);
}

// The staged-B loop (gs_tile): instantiated for the two shapes the product kernel does not take -- the three-piece bf16
// arithmetic (A/B arm `bench.py --mlp split6`: 8 waves, one workgroup per CU, 256-row tiles) and 128-column tiles of the
// f16x2 arithmetic (layers of 128 (mod 256) columns: 4 waves, two workgroups per CU, 128-row tiles).
// BIGROWS (0 = 32 WAVES): rows of the big tile.  <.., 256, 2, 8, 128> is round 6's occupancy A/B arm (tile_rows = -6): 8 waves of
// 64 x 64 (64 accumulators, <= 128 registers), TWO workgroups per CU = four waves per SIMD, both operands through LDS.
template <int EPI, int COLS, int NP, int WAVES, int BIGROWS = 0>
__global__ __launch_bounds__(64 * WAVES, BIGROWS ? 4 : 2) void gemm_split_kernel(const GemmSplitParams p) {   // (second argument: waves per SIMD)
    constexpr int kGsThreads = 64 * WAVES;
    constexpr int kSmallRows = (WAVES / (COLS / 64)) * 32;   // WAVES = 8: 64 (COLS = 256) or 128 (COLS = 128); WAVES = 4: half
    constexpr int kBigRows = BIGROWS ? BIGROWS : 32 * WAVES;
    extern __shared__ __attribute__((aligned(16))) char gs_smem[];
    unsigned *sbuf = reinterpret_cast<unsigned *>(gs_smem);
    __shared__ unsigned s_tile;
    __shared__ int s_aexp[kBigRows];
    __shared__ unsigned s_colmax[kGsColmaxLds];
    const int tid = threadIdx.x;
    const bool lds_colmax = p.c_colmax != nullptr && p.Nc <= kGsColmaxLds;
    if (lds_colmax)
        for (int c = tid; c < p.Nc; c += kGsThreads) s_colmax[c] = 0u;
    // static schedule (round 6, as gemm_f16_kernel; p.static_tiles == 0: the dispenser of rounds 3-5, A/B)
    unsigned static_next = blockIdx.x;
    for (;;) {
        __syncthreads();                       // (the previous tile's LDS reads are done; s_tile may be rewritten)
        unsigned tile;
        if (p.static_tiles) {
            tile = static_next;
            static_next += gridDim.x;
            if (tile >= p.n_tiles) break;
        } else {
            if (tid == 0) s_tile = atomicAdd(p.counter, 1u);
            __syncthreads();
            tile = s_tile;
            if (tile >= p.n_tiles) {
                // the last workgroup to leave re-arms the dispenser for the next launch (nobody takes a ticket after it)
                if (tid == 0 && atomicAdd(p.counter + 1, 1u) == gridDim.x - 1) {
                    p.counter[0] = 0u;
                    p.counter[1] = 0u;
                }
                break;
            }
        }
        // column tile fastest: the workgroups that share a row tile's A strip run at the same time (L2).  Whole rounds of
        // the chip take big tiles (fewest LDS reads per matrix instruction); what is left over after the last whole
        // round is cut into small tiles so that it spreads over all CUs instead of giving a few of them one more big
        // tile (100 000 x 512: 782 big tiles on 256 CUs were 4 tile times for 3.05 rounds of work).
        if (tile < p.n_big) {
            const int ct = (int)(tile % (unsigned)p.n_col_tiles), rt = (int)(tile / (unsigned)p.n_col_tiles);
            gs_tile<EPI, kBigRows / kSmallRows, COLS, NP, WAVES>(p, sbuf, s_aexp, s_colmax, (long long)rt * kBigRows, ct * COLS);
        } else {
            const unsigned st = tile - p.n_big;
            const int ct = (int)(st % (unsigned)p.n_col_tiles), rt = (int)(st / (unsigned)p.n_col_tiles);
            gs_tile<EPI, 1, COLS, NP, WAVES>(p, sbuf, s_aexp, s_colmax, (long long)p.rt_big * kBigRows + (long long)rt * kSmallRows, ct * COLS);
        }
    }
    if (lds_colmax) {   // (every wave of the workgroup passed the loop's barriers after its last LDS maximum)
        __syncthreads();
        for (int c = tid; c < p.Nc; c += kGsThreads)
            if (s_colmax[c]) atomicMax(p.c_colmax + c, s_colmax[c]);
    }
}

// The product kernel: gs_tile2 tiles of 128 rows (64-row tiles for what is left over after the whole rounds), 4 waves, TWO
// persistent workgroups per CU: one's tile prologue, barriers and epilogue fall into the other's matrix instructions.
// Tile dispensers: n_queues of them.  With 8 (round 5's experiment) workgroup b -- which runs on XCD b % 8,
// workgroups are dealt to the XCDs round robin -- takes from queue b % 8, which holds the row tiles rt = q (mod 8) with their column
// tiles back to back: the column tiles of one row tile, which read the same strip of A, then run on ONE XCD and share it in that
// XCD's L2.  Measured (profiles/r05_gemm_xcd_dispenser_ab.txt): the HBM traffic falls from 1.40-1.55 x to 1.08-1.15 x the
// algorithmic bytes, bits unchanged -- and the kernels get 35-40 % SLOWER (768 -> 512: 247 -> ~340 us in the step): the column
// tiles of a row tile now stream the same lines of A in lockstep through one L2.  The kernels are not HBM-bound (2-3 TB/s), so the
// product uses ONE queue (n_queues = 1: the chip-wide dispenser of round 4); the per-XCD form stays selectable (tile_rows = -8).
// A workgroup whose own queue is empty takes from the next one.  counter[0..7] big-tile queues, [8..15] small-tile queues,
// [16] workgroups that have left (the last one re-arms all of them).  (Round 4 also tried starting the second workgroup of
// every CU half a tile late: neutral to -7 %, tools/experiments/gemm_split_r04_variants.hip.)
constexpr int kGsQueues = 8;
// SMALL_TA: height of the leftover tiles in 32-row blocks -- 2 (64 rows) or 1 (32 rows: round 6).  Whole rounds of the chip run 128-row
// tiles; what is left is a fraction f of a round, and costs one more 128-row tile time as big tiles, ceil(2 f) x 64 as 64-row tiles,
// ceil(4 f) x 32 as 32-row tiles.  At 100 000 rows (781.25 row tiles on 512 slots) the one-column-tile layers leave f = 0.53 (two rounds
// of 128 became 128 + 3 x 32), the two-column-tile layers f = 0.05 (3 x 128 + 64 became 3 x 128 + 32), the three-column-tile layer f = 0.58.
template <int EPI, int COLS = 256, int SMALL_TA = 2>
__global__ __launch_bounds__(COLS, COLS == 256 ? 2 : 1) void gemm_f16_kernel(const GemmSplitParams p) {
    constexpr int kThreads = COLS, kBigRows = 128, kSmallRows = 32 * SMALL_TA;
    constexpr unsigned kSmallBit = 0x80000000u, kNone = 0xffffffffu;
    extern __shared__ __attribute__((aligned(16))) char gs_smem[];
    unsigned *sbuf = reinterpret_cast<unsigned *>(gs_smem);
    __shared__ unsigned s_tile;
    __shared__ int s_aexp[kBigRows];
    __shared__ unsigned s_colmax[kGsColmaxLds];
    const int tid = threadIdx.x;
    const bool lds_colmax = p.c_colmax != nullptr && p.Nc <= kGsColmaxLds;
    if (lds_colmax)
        for (int c = tid; c < p.Nc; c += kThreads) s_colmax[c] = 0u;
    const unsigned nct = (unsigned)p.n_col_tiles;
    const unsigned rt_big = p.n_big / nct, rt_small = (p.n_tiles - p.n_big) / nct;
    // p.n_queues: 8 = one queue per XCD; 1 (the default, see rqhip_gemm_split_ex) = a single queue for the chip
    const int nq = p.n_queues;
    const int xcd = (int)(blockIdx.x % (unsigned)nq);
    // Static schedule (round 6, the default): workgroup b takes tiles b, b + grid, b + 2 grid, ... -- the whole rounds of big tiles give
    // every workgroup the same number of them, the leftover tiles go one each to the first workgroups, which is what the dispenser
    // hands out when all workgroups run alike; no device-scope atomic (three per workgroup and launch: first ticket, the ticket
    // that says "none left", the leave counter that re-arms the dispenser -- serialised on one address, 512 workgroups at once at
    // the start of every launch), no barrier pair around a ticket in LDS, and the tile's row and column are scalar registers.
    if (p.static_tiles) {
        for (unsigned tile = blockIdx.x; tile < p.n_tiles; tile += gridDim.x) {
            __syncthreads();                   // (the previous tile's LDS reads are done)
            if (tile < p.n_big) {
                const int ct = (int)(tile % nct), rt = (int)(tile / nct);
                gs_tile2<EPI, kBigRows / 32, COLS>(p, sbuf, s_aexp, s_colmax, (long long)rt * kBigRows, ct * COLS);
            } else {
                const unsigned st = tile - p.n_big;
                const int ct = (int)(st % nct), rt = (int)(st / nct);
                gs_tile2<EPI, kSmallRows / 32, COLS>(p, sbuf, s_aexp, s_colmax, (long long)p.rt_big * kBigRows + (long long)rt * kSmallRows, ct * COLS);
            }
        }
        if (lds_colmax) {
            __syncthreads();
            for (int c = tid; c < p.Nc; c += kThreads)
                if (s_colmax[c]) atomicMax(p.c_colmax + c, s_colmax[c]);
        }
        return;
    }
    int big_skip = 0, small_skip = 0;          // (thread 0) queues found empty so far, in this workgroup's visiting order
    for (;;) {
        __syncthreads();                       // (the previous tile's LDS reads are done; s_tile may be rewritten)
        if (tid == 0) {
            unsigned t = kNone;
            for (; big_skip < nq; ++big_skip) {
                const unsigned q = (unsigned)((xcd + big_skip) % nq);
                const unsigned j = atomicAdd(p.counter + q, 1u);
                const unsigned rt = (j / nct) * (unsigned)nq + q;
                if (rt < rt_big) { t = rt * nct + j % nct; break; }
            }
            if (t == kNone) {
                for (; small_skip < nq; ++small_skip) {
                    const unsigned q = (unsigned)((xcd + small_skip) % nq);
                    const unsigned j = atomicAdd(p.counter + kGsQueues + q, 1u);
                    const unsigned rt = (j / nct) * (unsigned)nq + q;
                    if (rt < rt_small) { t = (rt * nct + j % nct) | kSmallBit; break; }
                }
            }
            s_tile = t;
        }
        __syncthreads();
        const unsigned tile = s_tile;
        if (tile == kNone) {
            if (tid == 0 && atomicAdd(p.counter + 2 * kGsQueues, 1u) == gridDim.x - 1) {
                for (int i = 0; i <= 2 * kGsQueues; ++i) p.counter[i] = 0u;
            }
            break;
        }
        // p.rt_fastest (tools, tile_rows = -2): row tile fastest -- all workgroups work through ONE column tile of the weight at a time
        // (the two workgroups of a CU read the same B stages), at the price of fetching every A strip once per column tile
        if (!(tile & kSmallBit)) {
            const int ct = p.rt_fastest ? (int)(tile / rt_big) : (int)(tile % nct), rt = p.rt_fastest ? (int)(tile % rt_big) : (int)(tile / nct);
            gs_tile2<EPI, kBigRows / 32, COLS>(p, sbuf, s_aexp, s_colmax, (long long)rt * kBigRows, ct * COLS);
        } else {
            const unsigned st = tile & ~kSmallBit;
            const int ct = p.rt_fastest ? (int)(st / rt_small) : (int)(st % nct), rt = p.rt_fastest ? (int)(st % rt_small) : (int)(st / nct);
            gs_tile2<EPI, kSmallRows / 32, COLS>(p, sbuf, s_aexp, s_colmax, (long long)p.rt_big * kBigRows + (long long)rt * kSmallRows, ct * COLS);
        }
    }
    if (lds_colmax) {   // (every wave of the workgroup passed the loop's barriers after its last LDS maximum)
        for (int c = tid; c < p.Nc; c += kThreads)
            if (s_colmax[c]) atomicMax(p.c_colmax + c, s_colmax[c]);
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
static int gs_np(int arith) { return arith == RQHIP_SPLIT_F16X2 ? 2 : arith == RQHIP_SPLIT_BF16X3 ? 3 : 0; }
extern "C" int rqhip_gemm_split_supported(int Nc, int R) { return (Nc > 0 && R > 0 && Nc % 128 == 0 && R % kGsK == 0) ? 1 : 0; }

extern "C" size_t rqhip_weight_image_bytes(int Nc, int R, int arith) {
    const int np = gs_np(arith);
    if (!np || !rqhip_gemm_split_supported(Nc, R)) return 0;
    return (size_t)(R / kGsK) * 2 * np * Nc * 16 + kGsTailWords * 4 + (np == 2 ? (size_t)Nc * sizeof(int) : 0);   // + the tile counter (+ exponents)
}
extern "C" size_t rqhip_weight_planes_bytes(int Nc, int R) { return rqhip_weight_image_bytes(Nc, R, RQHIP_SPLIT_BF16X3); }

extern "C" int rqhip_weight_images(const rqhip_image_job *jobs, int n_jobs, rqhip_stream_t stream) {
    if (n_jobs < 0 || (n_jobs > 0 && !jobs)) {
        set_error("weight_images: bad job list");
        return RQHIP_EARG;
    }
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    for (int first = 0; first < n_jobs; first += kImgMaxJobs) {
        ImageJobs batch;
        batch.n = n_jobs - first < kImgMaxJobs ? n_jobs - first : kImgMaxJobs;
        int blocks = 0;
        for (int i = 0; i < batch.n; ++i) {
            const rqhip_image_job &j = jobs[first + i];
            // w is [rows, cols] row-major.  transpose == 0: B = w (Nc = rows, R = cols); transpose == 1: B = w^T (Nc = cols, R = rows)
            const int Nc = j.transpose ? j.cols : j.rows, R = j.transpose ? j.rows : j.cols;
            const size_t need = rqhip_weight_image_bytes(Nc, R, j.arith);
            if (!j.w || !j.image || need == 0 || j.image_bytes < need || (reinterpret_cast<uintptr_t>(j.image) & 15u) != 0) {
                set_error("weight_images: job %d: bad arguments or unsupported shape (Nc = %d must be a multiple of 128, R = %d of 16, "
                          "image of rqhip_weight_image_bytes, 16-byte aligned)", first + i, Nc, R);
                return RQHIP_EARG;
            }
            ImageJob &d = batch.j[i];
            d.src = j.w; d.image = reinterpret_cast<unsigned *>(j.image); d.Nc = Nc; d.R = R; d.transpose = j.transpose ? 1 : 0;
            d.np = gs_np(j.arith); d.block0 = blocks;
            blocks += Nc / kImgCols;
        }
        hipLaunchKernelGGL(weight_images_kernel, dim3((unsigned)blocks), dim3(256), 0, s, batch);
        RQ_CHECK_LAUNCH("weight_images_kernel");
    }
    return RQHIP_OK;
}

extern "C" int rqhip_weight_planes(const float *w, int rows, int cols, int transpose, void *planes, size_t planes_bytes,
                                   rqhip_stream_t stream) {
    rqhip_image_job j;
    j.w = w; j.rows = rows; j.cols = cols; j.transpose = transpose; j.arith = RQHIP_SPLIT_BF16X3; j.image = planes; j.image_bytes = planes_bytes;
    return rqhip_weight_images(&j, 1, stream);
}

extern "C" int rqhip_maxima(const float *A, const float *Y, float *masked_out, int64_t M, int R, unsigned *row_max, unsigned *col_max,
                            rqhip_stream_t stream) {
    auto al16 = [](const void *q) { return (reinterpret_cast<uintptr_t>(q) & 15u) == 0; };
    if (M < 0 || R <= 0 || (R % 4) != 0 || R > 16384 || (M > 0 && !A) || (masked_out && !Y) || !al16(A) || !al16(Y) || !al16(masked_out)) {
        set_error("maxima: bad arguments (R = %d must be a multiple of 4, at most 16384; 16-byte aligned rows; masked_out needs Y)", R);
        return RQHIP_EARG;
    }
    if (M == 0 || (!row_max && !col_max && !masked_out)) return RQHIP_OK;
    long long blocks = (M + 3) / 4;
    const long long cap = (long long)cu_count() * 8;
    if (blocks > cap) blocks = cap;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    profile_begin(s, RQHIP_PROF_MAXIMA, 0.0, (double)M * R * 4 * ((Y ? 2 : 1) + (masked_out ? 1 : 0)));
    hipLaunchKernelGGL(maxima_kernel, dim3((unsigned)blocks), dim3(256), col_max ? (size_t)R * 4 : 0, s, A, Y, masked_out, (long long)M, R,
                       row_max, col_max);
    profile_end(s);
    RQ_CHECK_LAUNCH("maxima_kernel");
    return RQHIP_OK;
}

extern "C" size_t rqhip_gemm_split_recon_workspace_bytes(int64_t M, int Nc) {
    return (M > 0 && Nc > 0 && Nc % 256 == 0) ? (size_t)(Nc / 256) * (size_t)M * sizeof(float) : 0;
}

extern "C" int rqhip_gemm_split_ex(const rqhip_gemm_args *a, rqhip_stream_t stream) {
    if (!a) {
        set_error("gemm_split: null argument block");
        return RQHIP_EARG;
    }
    const int np = gs_np(a->arith), epi = a->epilogue;
    const int64_t M = a->M;
    const int R = a->R, Nc = a->Nc;
    if (!np || M < 0 || !a->image || (M > 0 && (!a->A || !a->C)) || !rqhip_gemm_split_supported(Nc, R) || epi < 0 || epi > 3) {
        set_error("gemm_split: bad arguments or unsupported shape (Nc = %d, R = %d, arithmetic %d, epilogue %d)", Nc, R, a->arith, epi);
        return RQHIP_EARG;
    }
    auto al16 = [](const void *q) { return (reinterpret_cast<uintptr_t>(q) & 15u) == 0; };
    if (!al16(a->A) || !al16(a->C) || !al16(a->image) || !al16(a->aux) || (R % 4) != 0) {
        set_error("gemm_split: pointers must be 16-byte aligned");
        return RQHIP_EARG;
    }
    if (M > 0 && np == 2 && (!a->a_row_max || a->a_row_parts < 1)) {
        set_error("gemm_split: RQHIP_SPLIT_F16X2 needs the row maxima of A (a_row_max from rqhip_maxima or from the epilogue that wrote A)");
        return RQHIP_EARG;
    }
    if (M > 0 && epi >= RQHIP_EPI_RECON && !a->aux) {
        set_error("gemm_split: epilogue %d needs aux (X / Y, [M, Nc])", epi);
        return RQHIP_EARG;
    }
    const int cols = gs_cols(Nc);
    if (epi == RQHIP_EPI_RECON && M > 0 && (cols != 256 || !a->loss_rows || !a->workspace ||
                                           a->workspace_bytes < rqhip_gemm_split_recon_workspace_bytes(M, Nc))) {
        set_error("gemm_split: the reconstruction-loss epilogue needs Nc %% 256 == 0 (Nc = %d), loss_rows and a workspace of "
                  "rqhip_gemm_split_recon_workspace_bytes", Nc);
        return RQHIP_EARG;
    }
    if (M == 0) return RQHIP_OK;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    GemmSplitParams p;
    p.A = a->A; p.planes = reinterpret_cast<const unsigned *>(a->image); p.C = a->C; p.M = M; p.R = R; p.Nc = Nc;
    p.X = a->aux; p.rowsum = reinterpret_cast<float *>(a->workspace); p.row_scale = a->row_scale;
    p.a_max = a->a_row_max; p.a_parts = a->a_row_parts;
    p.c_rowmax = a->c_row_max; p.c_colmax = a->c_col_max;
    p.n_queues = a->tile_rows == -8 ? kGsQueues : 1;   // (tools: tile_rows = -8 selects the per-XCD dispensers, A/B)
    p.rt_fastest = a->tile_rows == -2 ? 1 : 0;         // (tools: tile_rows = -2: row tile fastest, A/B)
    p.static_tiles = (a->tile_rows == -12 || a->tile_rows == -8 || a->tile_rows == -2) ? 0 : 1;   // (-12: the dispenser of rounds 4-5, A/B)
    // the tile dispenser lives behind the weight image (zeroed by rqhip_weight_images, re-armed by every launch): one
    // GEMM at a time per image, i.e. launches on one stream
    p.counter = const_cast<unsigned *>(p.planes) + (size_t)(R / kGsK) * 2 * np * Nc * 4;
    p.b_exp = reinterpret_cast<const int *>(p.counter + kGsTailWords);
    const int cus = cu_count();
    // the product kernel (gemm_f16_kernel: 4 waves, two workgroups per CU, B operands straight from the image) takes the f16x2
    // arithmetic at 256-column tiles; the staged loop the rest: bf16x3 with 8 waves / one workgroup per CU, f16x2 at 128-column
    // tiles with 4 waves / two per CU.  tile_rows (tools only): force big (256 / 128) or small (64 / 32) tiles in the staged loop.
    const bool tile2 = np == 2 && cols == 256;
    const bool wide = tile2 && a->tile_rows == -5 && Nc % 512 == 0 && epi != RQHIP_EPI_RECON;   // (A/B arm: 128 x 512 tiles, 8 waves, one workgroup per CU)
    const bool occ4 = tile2 && a->tile_rows == -6 && epi != RQHIP_EPI_RECON;                    // (A/B arm: 128 x 256 tiles, 8 waves of 64 x 64, two workgroups per CU)
    const int waves = (np == 3 || wide || occ4) ? 8 : 4;
    const int big_rows = (wide || occ4) ? 128 : 32 * waves;
    int small_rows = tile2 ? 64 : (waves / (cols / 64)) * 32;
    p.n_col_tiles = Nc / (wide ? 512 : cols);
    const long long slots = (long long)cus * ((waves == 4 || occ4) ? 2 : 1);
    // whole rounds of big tiles, the remainder as small tiles (see the kernels)
    const long long rt_all = (M + big_rows - 1) / big_rows;
    long long rt_big = ((rt_all * p.n_col_tiles) / slots) * slots / p.n_col_tiles;   // row tiles of the whole rounds
    if (rt_big * big_rows > M) rt_big = M / big_rows;
    // (measured at 100 000 rows: cutting the leftover into small tiles is worth it when it is a small part of a round -- Nc = 512:
    // 14 of 256 slots, 517 -> 456 us; a leftover of half a round runs as fast in big tiles -- Nc = 256 / 768: 135 / 149 slots)
    bool all_big = (rt_all * p.n_col_tiles) % slots > (3 * slots) / 10 && rt_all * p.n_col_tiles >= slots;
    bool tiny = false;
    if (tile2 && !wide && a->tile_rows == -9) {   // (A/B arm, neutral in the step: profiles/r06_step_ab.txt)
        // the product kernel also has 32-row leftover tiles (gemm_f16_kernel<EPI, 256, 1>).  A round of tiles does not get shorter in
        // proportion to its height -- a stage has a floor of barrier and memory latency whatever its matrix work, prologue and epilogue
        // do not shrink: measured (tools/tile_time.py, profiles/r06_tile_time.txt) one round of 64-row tiles takes 0.65-0.9 of a round of
        // 128-row tiles, one of 32-row tiles 0.5-0.78 (deep to shallow reductions).  Priced at 1 : 0.72 : 0.56.
        const long long left_rows = M - rt_big * big_rows;
        if (left_rows > 0) {
            auto rounds = [&](int h) { return ((left_rows + h - 1) / h * p.n_col_tiles + slots - 1) / slots; };
            const double c_big = (double)rounds(128) * 1.0, c_small = (double)rounds(64) * 0.72, c_tiny = (double)rounds(32) * 0.56;
            tiny = c_tiny < c_small && c_tiny < c_big;
            all_big = !tiny && c_big <= c_small && rt_all * p.n_col_tiles >= slots;
            if (tiny) small_rows = 32;
        }
    }
    if (all_big) rt_big = rt_all;
    if (tile2 && !wide && (a->tile_rows == -10 || a->tile_rows == -11)) {   // (tools: the whole matrix in 64-row / 32-row tiles -- tile-time calibration)
        rt_big = 0;
        tiny = a->tile_rows == -11;
        small_rows = tiny ? 32 : 64;
    }
    if (!tile2 && (a->tile_rows == 256 || a->tile_rows == 128)) rt_big = rt_all;
    if (!tile2 && (a->tile_rows == 64 || a->tile_rows == 32)) rt_big = 0;
    const long long rem_rows = M - rt_big * big_rows > 0 ? M - rt_big * big_rows : 0;
    const long long rt_small = (rem_rows + small_rows - 1) / small_rows;
    p.rt_big = (int)rt_big;
    p.n_big = (unsigned)(rt_big * p.n_col_tiles);
    p.n_tiles = p.n_big + (unsigned)(rt_small * p.n_col_tiles);
    size_t lds = (tile2 && !occ4) ? (size_t)2 * 4 * (128 * 4 + 32) * 4 : (size_t)2 * (np * 2 * (big_rows + cols) * 16);   // the stage buffers
    const size_t lds_epi = gs_epilogue_lds(waves, big_rows, wide ? 8 : cols / 64);                              // re-used by the epilogue
    if (lds < lds_epi) lds = lds_epi;
    const long long tiles = (long long)p.n_tiles;
    const int grid = (int)(tiles < slots ? tiles : slots);
    auto go = [&](auto kern) -> int {
        static LdsGrant grant;
        RQ_RETURN_IF_HIP(grant.ensure(reinterpret_cast<const void *>(kern), (int)lds));   // (+ the static LDS)
        // algorithmic work of the launch: 2 M Nc R FLOP; bytes: A once, C once (+ the aux matrix)
        profile_begin(s, RQHIP_PROF_GEMM_SPLIT, 2.0 * (double)M * Nc * R, 4.0 * (double)M * (R + Nc * (epi >= 2 ? 2 : 1)));
        hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * waves), lds, s, p);
        profile_end(s);
        RQ_CHECK_LAUNCH("gemm_split_kernel");
        return 0;
    };
    int rc;
    if (occ4) {
        rc = epi == 3 ? go(gemm_split_kernel<3, 256, 2, 8, 128>) : epi == 1 ? go(gemm_split_kernel<1, 256, 2, 8, 128>) : go(gemm_split_kernel<0, 256, 2, 8, 128>);
    } else if (wide) {
        rc = epi == 3 ? go(gemm_f16_kernel<3, 512>) : epi == 1 ? go(gemm_f16_kernel<1, 512>) : go(gemm_f16_kernel<0, 512>);
    } else if (tile2 && tiny) {
        rc = epi == 3 ? go(gemm_f16_kernel<3, 256, 1>) : epi == 2 ? go(gemm_f16_kernel<2, 256, 1>) : epi == 1 ? go(gemm_f16_kernel<1, 256, 1>)
                                                                                                          : go(gemm_f16_kernel<0, 256, 1>);
    } else if (tile2) {
        rc = epi == 3 ? go(gemm_f16_kernel<3>) : epi == 2 ? go(gemm_f16_kernel<2>) : epi == 1 ? go(gemm_f16_kernel<1>) : go(gemm_f16_kernel<0>);
    } else if (np == 2) {
        rc = epi == 3 ? go(gemm_split_kernel<3, 128, 2, 4>) : epi == 1 ? go(gemm_split_kernel<1, 128, 2, 4>) : go(gemm_split_kernel<0, 128, 2, 4>);
    } else {
        if (epi == 3) {
            set_error("gemm_split: the masked epilogue exists for RQHIP_SPLIT_F16X2 only");
            return RQHIP_EUNSUPPORTED;
        }
        if (cols == 128) rc = epi == 1 ? go(gemm_split_kernel<1, 128, 3, 8>) : go(gemm_split_kernel<0, 128, 3, 8>);
        else rc = epi == 2 ? go(gemm_split_kernel<2, 256, 3, 8>) : epi == 1 ? go(gemm_split_kernel<1, 256, 3, 8>) : go(gemm_split_kernel<0, 256, 3, 8>);
    }
    if (rc) return rc;
    if (epi == RQHIP_EPI_RECON) {
        hipLaunchKernelGGL(recon_rows_finish_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, s,
                           reinterpret_cast<const float *>(a->workspace), Nc / 256, (long long)M, a->loss_rows);
        RQ_CHECK_LAUNCH("recon_rows_finish_kernel");
    }
    return RQHIP_OK;
}

// ---- the round-3 entry points: RQHIP_SPLIT_BF16X3 --------------------------------------------------------------------------------
extern "C" int rqhip_gemm_split(const float *A, int64_t M, int R, const void *planes, int Nc, int relu, float *C,
                                rqhip_stream_t stream) {
    rqhip_gemm_args a = {};
    a.A = A; a.M = M; a.R = R; a.image = planes; a.Nc = Nc; a.arith = RQHIP_SPLIT_BF16X3; a.epilogue = relu & 1; a.C = C;
    a.tile_rows = (relu >> 8) & 0xfff;   // (bits 8.. of `relu`: tile rows, A/B)
    return rqhip_gemm_split_ex(&a, stream);
}

extern "C" int rqhip_gemm_split_recon(const float *A, int64_t M, int R, const void *planes, int Nc, const float *X,
                                      float row_scale, float *G, float *loss_rows, void *workspace, size_t workspace_bytes,
                                      rqhip_stream_t stream) {
    rqhip_gemm_args a = {};
    a.A = A; a.M = M; a.R = R; a.image = planes; a.Nc = Nc; a.arith = RQHIP_SPLIT_BF16X3; a.epilogue = RQHIP_EPI_RECON; a.C = G;
    a.aux = X; a.row_scale = row_scale; a.loss_rows = loss_rows; a.workspace = workspace; a.workspace_bytes = workspace_bytes;
    if (M > 0 && (!X || !G || !loss_rows || !workspace)) {
        set_error("gemm_split_recon: bad arguments (X, G, loss_rows, workspace of rqhip_gemm_split_recon_workspace_bytes)");
        return RQHIP_EARG;
    }
    return rqhip_gemm_split_ex(&a, stream);
}
