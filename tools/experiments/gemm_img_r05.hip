// gemm_img.hip -- round 5: the MLP GEMMs and weight gradients on OPERAND IMAGES (gfx950).  SURVEY.md section 8 row f2;
// reference modules/encoder.py:25-38 (`relu(x W^T)` forward and its autograd: `g W` data gradient, ReLU backward, `g^T x`
// weight gradient), modules/rqvae.py:146,152 + modules/loss.py:5-10 (the last decoder layer with the reconstruction loss).
//
// Same arithmetic as csrc/gemm_split.hip's product path (RQHIP_SPLIT_F16X2: a value under an exact power-of-two scale is two fp16
// pieces h = RN16(v), m = RN16(v - h); a product is hh + hm + mh on v_mfma_f32_32x32x16_f16, fp32 accumulation) -- but the
// fp32 -> fp16 x 2 split is no longer done by the kernels that CONSUME an activation (once per column tile of every GEMM and once
// per slab of every weight gradient: 77 of 340 us of an activation GEMM, 175 of 350 us of a weight gradient in round 4's
// probes).  The kernel that PRODUCES an activation (or a gradient) writes it already split, in the layouts its consumers stream:
//
//   an IMAGE of X [M, N] (N % 16 == 0), columns in segments of `seg` (= the producing tile's width, 256 or 128):
//     E  int32 [N / seg][M]       the exponent of (row, segment): the segment of the row is stored times 2^-E, its largest |value| in
//                                 [2^14, 2^15) (the top of fp16's range, so the low piece of every entry down to 2^-16 of it keeps a
//                                 normal-precision low piece); kGiEZero for an all-zero segment, 0 for one holding inf / nan (not scaled)
//     R  [ceil(M/64)][N/16][piece][half][64 rows] x 16 bytes   (8 consecutive columns 16 ks + 8 half .. + 7 of one row, one piece):
//                                 the A operand of a GEMM that reduces over the columns -- a 16-deep stage of a 64-row block is 4 KB
//                                 contiguous, already in the LDS image's order: staging is 16-byte copies, no VALU work
//     T  [ceil(M/32) * 2][piece][octet][N] x 16 bytes          (8 rows of one column, one piece): the operands of the weight
//                                 gradient, which reduces over the ROWS.  Inside every 32-row block the rows are stored in the order
//                                 the producing epilogue holds them (position 8 rl + k <-> row 4 k + rl): a reduction does not care
//                                 about the order of its terms as long as both operands use the same one.  Rows >= M hold zeros.
//   The exponent belongs to the tile that produced the segment (its row maximum over the tile's columns), so no pass over the
//   finished matrix is needed (round 4: rqhip_maxima over the input batch and over the library layers' outputs, row / column
//   maxima emitted by every epilogue, atomics).  The consumers pay for per-segment exponents with a rescale of their fp32
//   accumulators by an exact power of two at each segment boundary of the reduction (gemm: 32 TA multiplies per lane, once per
//   256 columns) or with one packed fp16 multiply per staged 16 bytes of ONE operand (weight gradient: the per-row factor
//   2^(e_g + e_x - e_ref) on g; x goes to LDS untouched).
//
// Kernels
//   img_pack_kernel      fp32 [M, N] (optionally masked by Y > 0) -> image: the input batch, the outputs of the library layers
//   gemm_img_kernel<EPI, COLS>   C = epilogue(A . B^T), A an image (R + E), B a weight image (rqhip_weight_images); writes any of
//                                fp32 C, the image of C (R, T, E), the reconstruction loss rows.  Tile loop, B operands and matrix
//                                instruction order as gemm_f16_kernel (csrc/gemm_split.hip); per-XCD tile dispensers.
//   wgrad_img_kernel<...>        dW[n, k] = sum_m g[m, n] x[m, k] from the T images of g and x; partial sums per row range, reduced by
//                                wgrad.hip's tree (rqhip_linear_wgrad_img there owns the plan).
// Error model (tests/test_split_bound.py restates it on the CPU and proves the bound): every operand element is represented with
// |v - (h + m) 2^E| <= 2^-22 |v| + 2^-25 2^E, the three products are exact, the dropped m m term is <= 2^-22 |a b|.
#include "rqhip_common.h"

namespace rqhip {

typedef float gi_f32x16 __attribute__((ext_vector_type(16)));
typedef float gi_f32x4 __attribute__((ext_vector_type(4)));
typedef float gi_f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 gi_f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 gi_f16x2 __attribute__((ext_vector_type(2)));
typedef unsigned gi_u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned gi_u32x4 __attribute__((ext_vector_type(4)));
typedef int gi_i32x4 __attribute__((ext_vector_type(4)));

constexpr int kGiK = 16;          // reduction depth of a stage = one K step of the matrix instruction
constexpr int kGiEZero = -200;    // exponent of an all-zero segment (never wins a maximum, any power of two times zero is zero)
constexpr int kGiLive = -100;     // exponents above this belong to segments that hold something
constexpr int kGiSpread = 64;     // segments of a row more than 2^64 below its largest one are dropped (see gi_tile)
constexpr int kGiMaxSeg = 8;      // segments of the A image a GEMM tile keeps exponents for
constexpr int kGiTS = 68;         // floats per row of a wave's transposition block (csrc/gemm_split.hip:kGsTS)

__device__ __forceinline__ unsigned gi_abs_bits(float v) { return __builtin_bit_cast(unsigned, v) & 0x7fffffffu; }
__device__ __forceinline__ unsigned gi_umax(unsigned a, unsigned b) { return a > b ? a : b; }
// exponent of a segment whose largest |value| has these bits: its values times 2^-e have their maximum in [2^14, 2^15)
__device__ __forceinline__ int gi_exp_of_bits(unsigned b) {
    b &= 0x7fffffffu;
    const int e = (int)(b >> 23);
    if (b == 0u) return kGiEZero;
    if (e == 255) return 0;
    return (e == 0 ? -126 : e - 127) - 14;
}
// (a, b) -> packed fp16 pieces: h = RN16(v), m = RN16(v - h)
__device__ __forceinline__ void gi_split2(float a, float b, unsigned &h, unsigned &m) {
    const gi_f16x2 hh = __builtin_convertvector(gi_f32x2{a, b}, gi_f16x2);
    h = __builtin_bit_cast(unsigned, hh);
    const gi_f32x2 hf = __builtin_convertvector(hh, gi_f32x2);
    const gi_f16x2 mm = __builtin_convertvector(gi_f32x2{a - hf.x, b - hf.y}, gi_f16x2);
    m = __builtin_bit_cast(unsigned, mm);
}
// DPP row (16 lanes) reductions: lane 15 of every 16-lane row ends up with the row's result (zeros shifted in)
__device__ __forceinline__ unsigned gi_row16_umax(unsigned v) {
    v = gi_umax(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true));
    v = gi_umax(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true));
    v = gi_umax(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true));
    v = gi_umax(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true));
    return v;
}
__device__ __forceinline__ float gi_row16_sum(float v) {   // fixed order: (((v + shr1) + shr2) + shr4) + shr8
    v = v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x111, 0xf, 0xf, true));
    v = v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x112, 0xf, 0xf, true));
    v = v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x114, 0xf, 0xf, true));
    v = v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x118, 0xf, 0xf, true));
    return v;
}

// 16-byte element index of (row, stage ks, ph = piece * 2 + half) in an R image of KS = N / 16 stages
__device__ __forceinline__ size_t gi_r_index(long long row, int ks, int ph, int KS) {
    return (((size_t)(row >> 6) * KS + ks) * 4 + ph) * 64 + (size_t)(row & 63);
}
// 16-byte element index of (32-row block b, row residue rl = row % 4, piece, column) in a T image of N columns: the element holds
// rows 32 b + 4 k + rl, k = 0 .. 7; it belongs to stage 2 b + (rl >> 1), octet rl & 1
// ... at POSITION gi_t_pos(col) of its (stage, piece, octet) run: inside every 64-column group the columns are stored in the order
// the producing lanes hold them (position 16 j + c <-> column 4 c + j), so that a store instruction's 16 lanes write 16 consecutive
// elements (256 contiguous bytes) -- in column order they wrote every fourth element, four times the cache-line requests (measured:
// + 50 us on a 100 000 x 512 result).  The weight gradient multiplies positions and maps its result back with gi_t_col.
__device__ __forceinline__ int gi_t_pos(int col) { return (col & ~63) + 16 * (col & 3) + ((col & 63) >> 2); }
__device__ __forceinline__ int gi_t_col(int pos) { return (pos & ~63) + 4 * (pos & 15) + ((pos & 63) >> 4); }
__device__ __forceinline__ size_t gi_t_index(long long b, int rl, int piece, int col, int N) {
    return ((((size_t)(2 * b + (rl >> 1)) * 2 + piece) * 2 + (rl & 1)) * N) + gi_t_pos(col);
}

// where a lane's block of results goes
struct GiOut {
    float *C;        // fp32 [M, N] or null
    unsigned *R;     // image planes or null
    unsigned *T;
    int *E;          // [N / seg][M] (required with R or T)
    long long M;
    int N;           // columns of the whole matrix
};

// One 32-row block of a lane's results -> the image.  The lane holds, for k = 0 .. 7, the four consecutive columns colw .. colw + 3 of
// row row0 + 4 k + rl (v[k]) and that row's exponent e[k] (of the segment the columns lie in).  R: 8 bytes per piece and row (the
// lane's half of a 16-byte element); T: per column and piece one 16-byte element = the lane's eight rows, gathered over k.
__device__ __forceinline__ void gi_emit_block(const GiOut &o, const gi_f32x4 (&v)[8], const int (&e)[8], long long row0, int colw, int rl) {
    const int KS = o.N / kGiK;
    const int ks = colw >> 4, half = (colw >> 3) & 1, sub = (colw >> 2) & 1;
    unsigned th[4][4], tm[4][4];        // [column][k pair]: {piece(k), piece(k + 1)} of that column
#pragma unroll
    for (int k2 = 0; k2 < 4; ++k2) {
        unsigned h01[2], h23[2], m01[2], m23[2];
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int k = 2 * k2 + kk;
            const long long row = row0 + 4 * k + rl;
            const bool live = row < o.M;
            // (a zero segment has e = kGiEZero: ldexp(0, 200) = 0; rows past M are stored as zeros: the T image's reduction runs over them)
            const float a0 = live ? ldexpf(v[k].x, -e[k]) : 0.0f, a1 = live ? ldexpf(v[k].y, -e[k]) : 0.0f;
            const float a2 = live ? ldexpf(v[k].z, -e[k]) : 0.0f, a3 = live ? ldexpf(v[k].w, -e[k]) : 0.0f;
            gi_split2(a0, a1, h01[kk], m01[kk]);
            gi_split2(a2, a3, h23[kk], m23[kk]);
            if (o.R && live) {
                unsigned *d = o.R + gi_r_index(row, ks, half, KS) * 4 + 2 * sub;
                *reinterpret_cast<gi_u32x2 *>(d) = gi_u32x2{h01[kk], h23[kk]};
                *reinterpret_cast<gi_u32x2 *>(d + 2 * 64 * 4) = gi_u32x2{m01[kk], m23[kk]};      // piece 1: two (piece, half) runs further
            }
        }
        // {x(k), x(k + 1)} of one column from the row-packed pairs: low halves -> column j, high halves -> column j + 1
        th[0][k2] = __builtin_amdgcn_perm(h01[1], h01[0], 0x05040100u); th[1][k2] = __builtin_amdgcn_perm(h01[1], h01[0], 0x07060302u);
        th[2][k2] = __builtin_amdgcn_perm(h23[1], h23[0], 0x05040100u); th[3][k2] = __builtin_amdgcn_perm(h23[1], h23[0], 0x07060302u);
        tm[0][k2] = __builtin_amdgcn_perm(m01[1], m01[0], 0x05040100u); tm[1][k2] = __builtin_amdgcn_perm(m01[1], m01[0], 0x07060302u);
        tm[2][k2] = __builtin_amdgcn_perm(m23[1], m23[0], 0x05040100u); tm[3][k2] = __builtin_amdgcn_perm(m23[1], m23[0], 0x07060302u);
    }
    if (o.T && row0 < ((o.M + 31) & ~31LL)) {
        const long long b = row0 >> 5;
        gi_u32x4 *dh = reinterpret_cast<gi_u32x4 *>(o.T) + gi_t_index(b, rl, 0, colw, o.N);      // column colw + j: 16 j elements further
        gi_u32x4 *dm = reinterpret_cast<gi_u32x4 *>(o.T) + gi_t_index(b, rl, 1, colw, o.N);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            dh[16 * j] = gi_u32x4{th[j][0], th[j][1], th[j][2], th[j][3]};
            dm[16 * j] = gi_u32x4{tm[j][0], tm[j][1], tm[j][2], tm[j][3]};
        }
    }
}

// ---- fp32 -> image ----------------------------------------------------------------------------------------------------------------
// A workgroup of WAVES waves takes 32 rows x (64 WAVES) columns = one segment of one 32-row block: lane (cl = lane & 15, rl = lane >> 4)
// of wave w holds columns c0 + 64 w + 4 cl .. + 3 of rows 4 k + rl -- the arrangement of a GEMM epilogue after its transposition.
// With Y: A masked by Y > 0 (the ReLU backward), also written to masked_out when that is given.
struct ImgPackParams {
    const float *A, *Y;
    float *masked_out;
    GiOut o;
    int seg;
    const int *run_flag;     // optional: the kernel does nothing unless *run_flag != 0
};

template <int WAVES>
__global__ __launch_bounds__(64 * WAVES) void img_pack_kernel(const ImgPackParams p) {
    if (p.run_flag && *p.run_flag == 0) return;
    __shared__ unsigned s_rmax[32];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int cl = lane & 15, rl = lane >> 4;
    const int nseg = p.o.N / p.seg;
    const long long b = blockIdx.x / nseg;
    const int sg = (int)(blockIdx.x % nseg);
    const long long row0 = b * 32;
    const int colw = sg * p.seg + 64 * wave + 4 * cl;
    if (tid < 32) s_rmax[tid] = 0u;
    __syncthreads();
    gi_f32x4 v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const long long row = row0 + 4 * k + rl;
        v[k] = gi_f32x4{0.f, 0.f, 0.f, 0.f};
        if (row < p.o.M) {
            v[k] = *reinterpret_cast<const gi_f32x4 *>(p.A + (size_t)row * p.o.N + colw);
            if (p.Y) {   // threshold_backward(a, y, 0): 0 where y <= 0
                const gi_f32x4 y4 = *reinterpret_cast<const gi_f32x4 *>(p.Y + (size_t)row * p.o.N + colw);
                v[k].x = y4.x <= 0.0f ? 0.0f : v[k].x; v[k].y = y4.y <= 0.0f ? 0.0f : v[k].y;
                v[k].z = y4.z <= 0.0f ? 0.0f : v[k].z; v[k].w = y4.w <= 0.0f ? 0.0f : v[k].w;
                if (p.masked_out) *reinterpret_cast<gi_f32x4 *>(p.masked_out + (size_t)row * p.o.N + colw) = v[k];
            }
        }
        unsigned rmx = gi_umax(gi_umax(gi_abs_bits(v[k].x), gi_abs_bits(v[k].y)), gi_umax(gi_abs_bits(v[k].z), gi_abs_bits(v[k].w)));
        rmx = gi_row16_umax(rmx);
        if (cl == 15 && rmx != 0u) atomicMax(&s_rmax[4 * k + rl], rmx);
    }
    __syncthreads();
    int e[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        e[k] = gi_exp_of_bits(s_rmax[4 * k + rl]);
        const long long row = row0 + 4 * k + rl;
        if (wave == 0 && cl == 0 && row < p.o.M) p.o.E[(size_t)sg * p.o.M + row] = e[k];
    }
    gi_emit_block(p.o, v, e, row0, colw, rl);
}

// image -> fp32 (tests, and the fallback of callers that need the values an image stands for): v = (h + m) 2^E, from the R planes
__global__ __launch_bounds__(256) void img_unpack_kernel(const unsigned *__restrict__ R, const int *__restrict__ E, long long M, int N, int seg,
                                                         float *__restrict__ out) {
    const int KS = N / kGiK;
    const size_t total = (size_t)M * (N / 8);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const long long row = (long long)(i / (N / 8));
        const int c8 = (int)(i % (N / 8)), ks = c8 >> 1, half = c8 & 1;
        const gi_f16x8 h = __builtin_bit_cast(gi_f16x8, reinterpret_cast<const gi_u32x4 *>(R)[gi_r_index(row, ks, half, KS)]);
        const gi_f16x8 m = __builtin_bit_cast(gi_f16x8, reinterpret_cast<const gi_u32x4 *>(R)[gi_r_index(row, ks, 2 + half, KS)]);
        const int e = E[(size_t)((8 * c8) / seg) * M + row];
#pragma unroll
        for (int j = 0; j < 8; ++j) out[(size_t)row * N + 8 * c8 + j] = ldexpf((float)h[j] + (float)m[j], e);
    }
}

// the same from the T planes (tests: both layouts of an image must stand for the same values)
__global__ __launch_bounds__(256) void img_unpack_t_kernel(const unsigned *__restrict__ T, const int *__restrict__ E, long long M, int N, int seg,
                                                           float *__restrict__ out) {
    const size_t total = (size_t)((M + 31) / 32) * 4 * N;           // (block, rl, column)
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int col = (int)(i % N);
        const int rl = (int)((i / N) % 4);
        const long long b = (long long)(i / ((size_t)4 * N));
        const gi_f16x8 h = __builtin_bit_cast(gi_f16x8, reinterpret_cast<const gi_u32x4 *>(T)[gi_t_index(b, rl, 0, col, N)]);   // (positions inside)
        const gi_f16x8 m = __builtin_bit_cast(gi_f16x8, reinterpret_cast<const gi_u32x4 *>(T)[gi_t_index(b, rl, 1, col, N)]);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const long long row = 32 * b + 4 * k + rl;
            if (row < M) out[(size_t)row * N + col] = ldexpf((float)h[k] + (float)m[k], E[(size_t)(col / seg) * M + row]);
        }
    }
}

// ---- the GEMM -----------------------------------------------------------------------------------------------------------------------
__device__ __attribute__((aligned(16))) unsigned g_gi_zero16[4] = {0u, 0u, 0u, 0u};   // what the stages of a dropped segment read

struct GemmImgParams {
    const unsigned *aR;      // R planes of A [M, R]
    const int *aE;           // [a_nseg][M]
    int a_nseg, a_seg_shift; // segments of A's columns; log2(stages per segment) (a_nseg > 1)
    const unsigned *planes;  // weight image (rqhip_weight_images, RQHIP_SPLIT_F16X2)
    const int *b_exp;        // [Nc] behind it
    unsigned *counter;       // the tile dispensers behind it
    long long M;
    int R, Nc, n_col_tiles;
    unsigned n_big, n_tiles;
    int rt_big;
    GiOut o;                 // outputs: fp32 C and / or the image of C (o.N == Nc, segment = the tile width)
    const float *X;          // EPI 2: the target [M, Nc]
    const unsigned *yR;      // EPI 3: R planes of Y [M, Nc] (C is zeroed where the high piece of Y is not positive) ...
    const unsigned *yT;      // ... or, preferred when present, its T planes (coalesced: four 16-byte loads per 32-row block and lane)
    float *rowsum;           // EPI 2: [column tiles][M]
    float row_scale;         // EPI 2: C = (2 (A.B^T - X)) * row_scale ...
    const float *row_scales; // ... or * row_scales[m] when given
    const int *run_flag;     // optional: the kernel does nothing unless *run_flag != 0
};

// One tile of (WM 32 TA) rows x COLS columns; 4 waves as WM x WN, WN = COLS / 64, each 32 TA x 64.
// EPI: 0 store, 1 ReLU, 2 reconstruction loss, 3 masked by Y > 0.
template <int EPI, int TA, int COLS>
__device__ __forceinline__ void gi_tile(const GemmImgParams &p, unsigned *sbuf, int *s_unit, int *s_E, unsigned *s_rmax, float *s_red,
                                        long long m0, int n0) {
    constexpr int WAVES = 4, UB = 2, kThreads = 64 * WAVES;
    constexpr int WN = COLS / 64, WM = WAVES / WN, ROWS = WM * 32 * TA;
    constexpr int NEL = 4 * ROWS;                                    // 16-byte elements of an A stage: [ph][row]
    constexpr int AQ = (NEL + kThreads - 1) / kThreads;
    constexpr int kRegion = ROWS * 4 + 32;                           // dwords per (piece, half) region (+ 128 bytes: see gs_tile2)
    constexpr int PA = 4 * kRegion;
    constexpr int kNone = -(1 << 30);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int il = lane & 31, h = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;
    const int n_stage = p.R / kGiK, KS = n_stage;
    const int nseg = p.a_nseg;

    // exponents of the tile's rows of A, all segments, into LDS (rows past M: row M - 1's, never stored)
    for (int i = tid; i < nseg * ROWS; i += kThreads) {
        const int sg = i / ROWS, r = i % ROWS;
        long long g = m0 + r;
        g = g < p.M ? g : p.M - 1;
        s_E[i] = p.aE[(size_t)sg * p.M + g];
    }
    if (tid < ROWS) s_rmax[tid] = 0u;
    __syncthreads();
    // A row's segments whose exponent lies more than kGiSpread below the row's largest are DROPPED (their stages read zeros): what
    // they could add is below 2^-64 of the row's scale, and keeping them would let the accumulator rescale between two live
    // segments overflow.  bit s of the mask: segment s is dead (zero, or dropped).
    auto dead_mask = [&](int r) -> unsigned {
        int emax = kGiEZero;
        for (int sg = 0; sg < nseg; ++sg) emax = s_E[sg * ROWS + r] > emax ? s_E[sg * ROWS + r] : emax;
        unsigned mask = 0u;
        for (int sg = 0; sg < nseg; ++sg) {
            const int e = s_E[sg * ROWS + r];
            if (e <= kGiLive || e < emax - kGiSpread) mask |= 1u << sg;
        }
        return mask;
    };

    // staging roles: element e = tid + kThreads q of the stage image [ph][row]
    const unsigned *asrc[AQ];
    unsigned akill[AQ];
    int adst[AQ];
    bool a_live[AQ];
#pragma unroll
    for (int q = 0; q < AQ; ++q) {
        const int e = tid + kThreads * q;
        a_live[q] = (NEL % kThreads == 0) || e < NEL;
        const int ph = (a_live[q] ? e : 0) / ROWS, r = (a_live[q] ? e : 0) % ROWS;
        long long g = m0 + r;
        g = g < p.M ? g : p.M - 1;
        asrc[q] = p.aR + gi_r_index(g, 0, ph, KS) * 4;
        adst[q] = ph * kRegion + r * 4;
        akill[q] = nseg > 1 ? dead_mask(r) : 0u;
    }

    gi_f32x16 acc[TA][UB];
#pragma unroll
    for (int t = 0; t < TA; ++t)
#pragma unroll
        for (int u = 0; u < UB; ++u)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][u][r] = 0.0f;
    // the unit of a lane's accumulators of row block t: they hold sums of products times 2^-(unit + e_column)
    int unit[TA];
    unsigned mdead[TA];
#pragma unroll
    for (int t = 0; t < TA; ++t) {
        unit[t] = kNone;
        mdead[t] = nseg > 1 ? dead_mask(wm * 32 * TA + 32 * t + il) : 0u;
    }
    // entering segment sg: a live segment's products come in units of ITS exponent -- bring the accumulators there (exact: a power of two)
    auto enter_segment = [&](int sg) {
#pragma unroll
        for (int t = 0; t < TA; ++t) {
            const int e = s_E[sg * ROWS + wm * 32 * TA + 32 * t + il];
            const bool live = nseg > 1 ? !((mdead[t] >> sg) & 1u) : e > kGiLive;
            if (!live) continue;
            if (unit[t] != kNone && unit[t] != e) {
                const float f = ldexpf(1.0f, unit[t] - e);           // |unit - e| <= kGiSpread between live segments
#pragma unroll
                for (int u = 0; u < UB; ++u)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[t][u][r] = acc[t][u][r] * f;
            }
            unit[t] = e;
        }
    };

    const gi_u32x4 *bsrc = reinterpret_cast<const gi_u32x4 *>(p.planes) + (size_t)h * p.Nc + n0 + 64 * wn + il;
    const size_t b_stage = (size_t)4 * p.Nc, b_piece = (size_t)2 * p.Nc;
    gi_u32x4 ra0[AQ], ra1[AQ];
    gi_u32x4 fb0[UB][2], fb1[UB][2];
    // (every fetch is unconditional: past the last stage it re-reads it -- see gs_tile2)
    auto fetchA = [&](int stage, gi_u32x4 *dst) {
        stage = stage < n_stage ? stage : n_stage - 1;
#pragma unroll
        for (int q = 0; q < AQ; ++q) {
            const unsigned *src = asrc[q] + (size_t)stage * 1024;     // a stage of a 64-row block: 4 x 64 x 16 bytes
            if (akill[q] != 0u && ((akill[q] >> (stage >> p.a_seg_shift)) & 1u)) src = g_gi_zero16;
            dst[q] = *reinterpret_cast<const gi_u32x4 *>(src);
        }
    };
    auto fetchB = [&](int stage, gi_u32x4 (*dst)[2]) {
        stage = stage < n_stage ? stage : n_stage - 1;
        const gi_u32x4 *img = bsrc + (size_t)stage * b_stage;
#pragma unroll
        for (int u = 0; u < UB; ++u)
#pragma unroll
            for (int pc = 0; pc < 2; ++pc) dst[u][pc] = img[(size_t)pc * b_piece + 32 * u];
    };
    auto stash = [&](int buf, const gi_u32x4 *ra) {
        unsigned *dA = sbuf + buf * PA;
#pragma unroll
        for (int q = 0; q < AQ; ++q) {
            if (NEL % kThreads != 0 && !a_live[q]) continue;
            *reinterpret_cast<gi_u32x4 *>(dA + adst[q]) = ra[q];
        }
    };
    auto loadA = [&](int buf, int t, gi_f16x8 *fa) {
        const gi_u32x4 *aA = reinterpret_cast<const gi_u32x4 *>(sbuf + buf * PA);
#pragma unroll
        for (int pc = 0; pc < 2; ++pc) fa[pc] = __builtin_bit_cast(gi_f16x8, aA[(pc * 2 + h) * (kRegion / 4) + wm * 32 * TA + 32 * t + il]);
    };
    auto multiply = [&](int buf, const gi_u32x4 (*fbr)[2]) {
        gi_f16x8 fa[2][2];
        loadA(buf, 0, fa[0]);
#pragma unroll
        for (int t = 0; t < TA; ++t) {
            if (t + 1 < TA) loadA(buf, t + 1, fa[(t + 1) & 1]);
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                const gi_f16x8 bh = __builtin_bit_cast(gi_f16x8, fbr[u][0]), bm = __builtin_bit_cast(gi_f16x8, fbr[u][1]);
                gi_f32x16 c16 = acc[t][u];
                c16 = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh, fa[t & 1][1], c16, 0, 0, 0);   // m h (smallest first)
                c16 = __builtin_amdgcn_mfma_f32_32x32x16_f16(bm, fa[t & 1][0], c16, 0, 0, 0);   // h m
                c16 = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh, fa[t & 1][0], c16, 0, 0, 0);   // h h
                acc[t][u] = c16;
            }
        }
    };

    enter_segment(0);
    fetchA(0, ra0);
    fetchB(0, fb0);
    fetchA(1, ra1);
    stash(0, ra0);
    fetchB(1, fb1);
    fetchA(2, ra0);
    __syncthreads();
    const int n_pair = n_stage & ~1;
    const int seg_mask = (1 << p.a_seg_shift) - 1;
    for (int c = 0; c < n_pair; c += 2) {
        if (nseg > 1 && c != 0 && (c & seg_mask) == 0) enter_segment(c >> p.a_seg_shift);   // (stages per segment: a power of two >= 2)
        stash(1, ra1);                             // stage c + 1
        fetchA(c + 3, ra1);
        multiply(0, fb0);
        fetchB(c + 2, fb0);
        __syncthreads();
        stash(0, ra0);                             // stage c + 2 (past the end: the last stage again, never multiplied)
        fetchA(c + 4, ra0);
        multiply(1, fb1);
        fetchB(c + 3, fb1);
        __syncthreads();
    }
    if (n_stage & 1) {                             // the last stage of an odd count lies in buffer 0 (single-segment images only)
        multiply(0, fb0);
        __syncthreads();
    }
    if (wn == 0 && h == 0) {
#pragma unroll
        for (int t = 0; t < TA; ++t) s_unit[wm * 32 * TA + 32 * t + il] = unit[t] == kNone ? 0 : unit[t];
    }
    __syncthreads();

    // ---- epilogue, phase A: every 32-row block of the wave's tile through the wave-private transposition block; scales undone,
    // EPI applied, fp32 stored (when asked for), the results KEPT in registers (the accumulators' own, dead by then) and the row
    // maxima over the tile's columns met in LDS.  (The thread number is laundered: see gs_epilogue.)
    int ltid = threadIdx.x;
    asm volatile("" : "+v"(ltid));
    const int llane = ltid & 63, lwave = ltid >> 6;
    const int lil = llane & 31, lh = llane >> 5;
    const int lwm = lwave / WN, lwn = lwave % WN;
    const int cl = llane & 15, rl = llane >> 4;
    float *tb = reinterpret_cast<float *>(sbuf) + (size_t)lwave * 32 * kGiTS;
    const int colw = n0 + 64 * lwn + 4 * cl;                  // this lane's four columns
    const gi_i32x4 ec = *reinterpret_cast<const gi_i32x4 *>(p.b_exp + colw);
    const bool want_img = p.o.R != nullptr || p.o.T != nullptr;
    constexpr int kAux = 4, NQ = 8 * TA;
    gi_f32x4 xa[kAux];
    auto aux_of = [&](int q) {
        long long r = m0 + 32 * TA * lwm + 4 * q + rl;
        r = r < p.M ? r : p.M - 1;
        return *reinterpret_cast<const gi_f32x4 *>(p.X + (size_t)r * p.Nc + colw);
    };
    if (EPI == 2) {
#pragma unroll
        for (int q = 0; q < kAux; ++q) xa[q] = aux_of(q);
    }
    gi_f32x4 vv[TA][8];
#pragma unroll
    for (int t = 0; t < TA; ++t) {
        gi_u32x4 ym[4];
        if (EPI == 3 && p.yT) {   // (requested before the transposition: the block's whole mask in four coalesced loads)
            long long yb32 = (m0 + 32 * TA * lwm + 32 * t) >> 5;
            const long long last = (p.M - 1) >> 5;
            yb32 = yb32 < last ? yb32 : last;                         // (a block past the matrix: any valid block, its rows are not stored)
            const gi_u32x4 *ys = reinterpret_cast<const gi_u32x4 *>(p.yT) + gi_t_index(yb32, rl, 0, colw, p.Nc);
#pragma unroll
            for (int j = 0; j < 4; ++j) ym[j] = ys[16 * j];
        }
#pragma unroll
        for (int u = 0; u < UB; ++u)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *reinterpret_cast<gi_f32x4 *>(tb + lil * kGiTS + 32 * u + 8 * g + 4 * lh) =
                    gi_f32x4{acc[t][u][4 * g], acc[t][u][4 * g + 1], acc[t][u][4 * g + 2], acc[t][u][4 * g + 3]};
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int q = 8 * t + k;
            const int rloc = 32 * TA * lwm + 32 * t + 4 * k + rl;     // row inside the workgroup's tile
            const long long grow = m0 + rloc;
            gi_f32x4 v = *reinterpret_cast<const gi_f32x4 *>(tb + (4 * k + rl) * kGiTS + 4 * cl);
            const int er = s_unit[rloc];
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = ldexpf(v[j], er + ec[j]);   // undo the row and column scales (exact)
            if (EPI == 1) {   // (a NaN stays a NaN, as torch.relu)
                v.x = v.x < 0.0f ? 0.0f : v.x; v.y = v.y < 0.0f ? 0.0f : v.y;
                v.z = v.z < 0.0f ? 0.0f : v.z; v.w = v.w < 0.0f ? 0.0f : v.w;
            }
            if (EPI == 2) {   // as csrc/recon_loss.hip: d = x_hat - x, loss += d d, gradient (2 d) row_scale
                const gi_f32x4 x4 = xa[q % kAux];
                if (q + kAux < NQ) xa[q % kAux] = aux_of(q + kAux);
                const float rs = p.row_scales ? p.row_scales[grow < p.M ? grow : p.M - 1] : p.row_scale;
                float sq = 0.0f;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float d = v[j] - x4[j];
                    sq = sq + d * d;
                    v[j] = (2.0f * d) * rs;
                }
                const float s16 = gi_row16_sum(sq);
                if (cl == 15) s_red[lwn * ROWS + rloc] = s16;
            }
            if (EPI == 3) {   // threshold_backward(g, y, 0) from the high piece of y's image: y > 0 <=> its fp16 bits in 0x0001 .. 0x7fff
                unsigned yb[4];                                           // (and above 0x7c00: a NaN passes, as `y <= 0 ? 0 : g` lets it)
                if (p.yT) {   // the T planes: the lane's four columns x eight rows of this block are four 16-byte elements (ym)
#pragma unroll
                    for (int j = 0; j < 4; ++j) yb[j] = (ym[j][k >> 1] >> (16 * (k & 1))) & 0xffffu;
                } else {
                    const long long yr = grow < p.M ? grow : p.M - 1;
                    const gi_u32x2 yh = *reinterpret_cast<const gi_u32x2 *>(p.yR + gi_r_index(yr, colw >> 4, (colw >> 3) & 1, p.Nc / kGiK) * 4
                                                                             + 2 * ((colw >> 2) & 1));
                    yb[0] = yh.x & 0xffffu; yb[1] = yh.x >> 16; yb[2] = yh.y & 0xffffu; yb[3] = yh.y >> 16;
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = ((yb[j] - 1u) < 0x7fffu || (yb[j] & 0x7fffu) > 0x7c00u) ? v[j] : 0.0f;
            }
            if (grow < p.M) {
                if (p.o.C) *reinterpret_cast<gi_f32x4 *>(p.o.C + (size_t)grow * p.Nc + colw) = v;
                if (want_img) {
                    unsigned rmx = gi_umax(gi_umax(gi_abs_bits(v.x), gi_abs_bits(v.y)), gi_umax(gi_abs_bits(v.z), gi_abs_bits(v.w)));
                    rmx = gi_row16_umax(rmx);                 // (every lane of a 16-lane row takes this branch or none: same grow)
                    if (cl == 15 && rmx != 0u) atomicMax(&s_rmax[rloc], rmx);
                }
            }
            vv[t][k] = v;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();            // (the block is rewritten by the next t)
    }
    if (want_img || EPI == 2) __syncthreads();
    // ---- phase B: the exponent of (row, this column tile) is known: split and store
    if (want_img) {
#pragma unroll
        for (int t = 0; t < TA; ++t) {
            int e[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int rloc = 32 * TA * lwm + 32 * t + 4 * k + rl;
                e[k] = gi_exp_of_bits(s_rmax[rloc]);
                if (lwn == 0 && cl == 0 && m0 + rloc < p.M) p.o.E[(size_t)(n0 / COLS) * p.M + m0 + rloc] = e[k];
            }
            gi_emit_block(p.o, vv[t], e, m0 + 32 * TA * lwm + 32 * t, colw, rl);
        }
    }
    if (EPI == 2) {   // the WN column waves of a row, in wave order
        for (int r = ltid; r < ROWS; r += kThreads) {
            if (m0 + r >= p.M) continue;
            float sum = s_red[r];
#pragma unroll
            for (int w = 1; w < WN; ++w) sum = sum + s_red[w * ROWS + r];
            p.rowsum[(size_t)(n0 / COLS) * p.M + m0 + r] = sum;
        }
    }
    // (the persistent loop's barrier at its top keeps the next tile off s_rmax / s_red / the stage buffers)
}

// The persistent tile loop: tiles of kBigRows rows for the whole rounds of the chip, kBigRows / 2 for what is left over, TWO workgroups
// per CU, tile dispensers as gemm_f16_kernel (csrc/gemm_split.hip: n_queues = 1 chip-wide, 8 = one per XCD).
constexpr int kGiQueues = 8;
template <int EPI, int COLS>
__global__ __launch_bounds__(256, 2) void gemm_img_kernel(const GemmImgParams p, const int n_queues) {
    if (p.run_flag && *p.run_flag == 0) return;
    constexpr int TA_BIG = COLS == 256 ? 4 : 2, kBigRows = 128, kSmallRows = 64;
    constexpr unsigned kSmallBit = 0x80000000u, kNone = 0xffffffffu;
    extern __shared__ __attribute__((aligned(16))) char gi_smem[];
    unsigned *sbuf = reinterpret_cast<unsigned *>(gi_smem);
    __shared__ unsigned s_tile;
    __shared__ int s_unit[kBigRows];
    __shared__ int s_E[kGiMaxSeg * kBigRows];
    __shared__ unsigned s_rmax[kBigRows];
    __shared__ float s_red[(COLS / 64) * kBigRows];
    const int tid = threadIdx.x;
    const unsigned nct = (unsigned)p.n_col_tiles;
    const unsigned rt_big = p.n_big / nct, rt_small = (p.n_tiles - p.n_big) / nct;
    const int nq = n_queues;
    const int xcd = (int)(blockIdx.x % (unsigned)nq);
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
                    const unsigned j = atomicAdd(p.counter + kGiQueues + q, 1u);
                    const unsigned rt = (j / nct) * (unsigned)nq + q;
                    if (rt < rt_small) { t = (rt * nct + j % nct) | kSmallBit; break; }
                }
            }
            s_tile = t;
        }
        __syncthreads();
        const unsigned tile = s_tile;
        if (tile == kNone) {
            if (tid == 0 && atomicAdd(p.counter + 2 * kGiQueues, 1u) == gridDim.x - 1) {
                for (int i = 0; i <= 2 * kGiQueues; ++i) p.counter[i] = 0u;
            }
            break;
        }
        if (!(tile & kSmallBit)) {
            const int ct = (int)(tile % nct), rt = (int)(tile / nct);
            gi_tile<EPI, TA_BIG, COLS>(p, sbuf, s_unit, s_E, s_rmax, s_red, (long long)rt * kBigRows, ct * COLS);
        } else {
            const unsigned st = tile & ~kSmallBit;
            const int ct = (int)(st % nct), rt = (int)(st / nct);
            gi_tile<EPI, TA_BIG / 2, COLS>(p, sbuf, s_unit, s_E, s_rmax, s_red, (long long)p.rt_big * kBigRows + (long long)rt * kSmallRows, ct * COLS);
        }
    }
}

// reconstruction loss of a row = its column tiles' sums in order (csrc/gemm_split.hip:recon_rows_finish_kernel's twin with a run flag)
__global__ __launch_bounds__(256) void gi_recon_rows_finish_kernel(const float *__restrict__ rowsum, int nct, long long M, float *__restrict__ out,
                                                                   const int *__restrict__ run_flag) {
    if (run_flag && *run_flag == 0) return;
    const long long m = (long long)blockIdx.x * 256 + threadIdx.x;
    if (m >= M) return;
    float s = rowsum[m];
    for (int c = 1; c < nct; ++c) s = s + rowsum[(size_t)c * M + m];
    out[m] = s;
}

// flag = 1 when some row's upstream gradient is not the announced scale (bit compare), else 0: one workgroup
__global__ __launch_bounds__(1024) void gi_rows_differ_kernel(const float *__restrict__ g_out, long long M, float announced, int *__restrict__ flag) {
    __shared__ int s_any;
    if (threadIdx.x == 0) s_any = 0;
    __syncthreads();
    const unsigned want = __builtin_bit_cast(unsigned, announced);
    int any = 0;
    for (long long m = threadIdx.x; m < M; m += 1024) any |= __builtin_bit_cast(unsigned, g_out[m]) != want;
    if (any) s_any = 1;
    __syncthreads();
    if (threadIdx.x == 0) *flag = s_any;
}

// ---- the weight gradient --------------------------------------------------------------------------------------------------------------
// dW[n, k] = sum_m g[m, n] x[m, k] from the T images.  A workgroup owns an Nt x Kt block of dW (inside ONE segment of g's columns and
// one of x's) and a range of 32-row blocks; per 16-row stage it stages 2 x 2 x (Nt + Kt) 16-byte elements: x's go to LDS as they
// are, g's are multiplied by the stage's row factors f(row) = 2^(e_g(row) + e_x(row) - e_ref) first (one v_pk_mul_f16 per dword:
// exact unless it underflows, which costs low-order bits of rows far below the range's largest -- what round 4's per-column scales
// cost the same rows); e_ref = the largest e_g + e_x of the range; dW = acc 2^e_ref.  Matrix instructions, wave tiling, pipeline
// and the reduction of the ranges' partial blocks as wgrad_split_kernel (csrc/wgrad_split.hip).
struct WgradImgParams {
    const unsigned *gT, *xT;
    const int *gE, *xE;          // [segments][M]
    float *out;
    long long M;
    int N, K;
    int g_seg, x_seg;
    int nslab_n, nslab_k, msplit;
    long long n_blocks;          // ceil(M / 32)
    const int *run_flag;
};

template <int TA, int TB, int WA, int WB>
__global__ __launch_bounds__(64 * WA * WB) void wgrad_img_kernel(const WgradImgParams p) {
    if (p.run_flag && *p.run_flag == 0) return;
    constexpr int Nt = 32 * TA * WA, Kt = 32 * TB * WB, NT = 64 * WA * WB;
    constexpr int NEL = 4 * (Nt + Kt);                 // 16-byte elements per stage: g [piece][octet][Nt], then x [piece][octet][Kt]
    constexpr int UQ = NEL / NT;
    static_assert(NEL % NT == 0 && (4 * Nt) % NT == 0, "a thread's elements are all g's or all x's per q");
    constexpr int PART = NEL * 4;                      // dwords per stage buffer
    extern __shared__ __attribute__((aligned(16))) char wi_smem[];
    unsigned *sbuf = reinterpret_cast<unsigned *>(wi_smem);      // [2][PART]
    __shared__ __attribute__((aligned(16))) unsigned s_f[2][8];  // the stage's 16 row factors as packed fp16 pairs: [octet][k pair]
    __shared__ int s_eref;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int il = lane & 31, h = lane >> 5;
    const int wa = wave / WB, wb = wave % WB;
    const int nslabs = p.nslab_n * p.nslab_k;
    int slab, split;
    if ((p.msplit & 7) == 0) {   // slabs of one row range back to back on one XCD (they share its strips in that L2)
        const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
        slab = j % nslabs;
        split = (j / nslabs) * 8 + xcd;
    } else {
        slab = blockIdx.x % nslabs;
        split = blockIdx.x / nslabs;
    }
    const int slab_n = slab / p.nslab_k, slab_k = slab % p.nslab_k;
    const int n0 = slab_n * Nt, k0 = slab_k * Kt;
    const long long b_begin = p.n_blocks * split / p.msplit, b_end = p.n_blocks * (split + 1) / p.msplit;
    const long long n_stage = 2 * (b_end - b_begin);
    const int *gE = p.gE + (size_t)(n0 / p.g_seg) * p.M, *xE = p.xE + (size_t)(k0 / p.x_seg) * p.M;

    // e_ref of the range
    if (tid == 0) s_eref = kGiEZero * 2;
    __syncthreads();
    {
        int mx = kGiEZero * 2;
        for (long long r = b_begin * 32 + tid; r < b_end * 32 && r < p.M; r += NT) {
            const int a = gE[r], b = xE[r];
            if (a > kGiLive && b > kGiLive) mx = a + b > mx ? a + b : mx;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const int other = __shfl_xor(mx, o, 64);
            mx = other > mx ? other : mx;
        }
        if (lane == 0) atomicMax(&s_eref, mx);
    }
    __syncthreads();
    const int eref = s_eref;

    gi_f32x16 acc[TA][TB];
#pragma unroll
    for (int t = 0; t < TA; ++t)
#pragma unroll
        for (int u = 0; u < TB; ++u)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][u][r] = 0.0f;

    // staging roles: element e = tid + NT q of the stage: e < 4 Nt -> g's [ph][col] (ph = piece * 2 + octet), else x's
    const gi_u32x4 *src[UQ];
    size_t step[UQ];
    int oct[UQ];
    bool isg[UQ];
#pragma unroll
    for (int q = 0; q < UQ; ++q) {
        const int e = tid + NT * q;
        isg[q] = e < 4 * Nt;
        const int idx = isg[q] ? e : e - 4 * Nt, W = isg[q] ? Nt : Kt;
        const int ph = idx / W, col = idx % W;
        oct[q] = ph & 1;
        const int Ntot = isg[q] ? p.N : p.K, c0 = isg[q] ? n0 : k0;
        // T image: stage s, piece, octet, column -> ((s * 2 + piece) * 2 + octet) * Ntot + column = (s * 4 + ph) * Ntot + column
        src[q] = reinterpret_cast<const gi_u32x4 *>(isg[q] ? p.gT : p.xT) + ((size_t)(2 * b_begin) * 4 + ph) * Ntot + c0 + col;
        step[q] = (size_t)4 * Ntot;
    }
    gi_u32x4 rv[UQ];
    auto fetch = [&](long long stage) {
        (void)stage;
#pragma unroll
        for (int q = 0; q < UQ; ++q) {
            rv[q] = *src[q];
            src[q] += step[q];
        }
    };
    // the 16 row factors of stage s (global stage index S = 2 b_begin + s): position (octet o, slot k) <-> row 32 (S / 2) + 4 k + 2 (S % 2) + o
    auto factors = [&](long long stage, int buf) {
        if (tid < 16) {
            const long long S = 2 * b_begin + stage;
            const int o = tid >> 3, k = tid & 7;
            const long long row = 32 * (S >> 1) + 4 * k + 2 * (int)(S & 1) + o;
            unsigned bits = 0u;
            if (row < p.M) {
                const int a = gE[row], b = xE[row];
                if (a > kGiLive && b > kGiLive) {
                    const int d = a + b - eref;                               // <= 0
                    bits = d >= -14 ? (unsigned)(d + 15) << 10 : (d >= -24 ? 1u << (d + 24) : 0u);   // 2^d as fp16 (subnormal below 2^-14)
                }
            }
            reinterpret_cast<unsigned short *>(&s_f[buf][0])[tid] = (unsigned short)bits;
        }
    };
    auto stash = [&](int buf) {
        unsigned *dst = sbuf + buf * PART;
#pragma unroll
        for (int q = 0; q < UQ; ++q) {
            gi_u32x4 v = rv[q];
            if (isg[q]) {
                const gi_u32x4 f = *reinterpret_cast<const gi_u32x4 *>(&s_f[buf][4 * oct[q]]);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const gi_f16x2 prod = __builtin_bit_cast(gi_f16x2, v[j]) * __builtin_bit_cast(gi_f16x2, f[j]);
                    v[j] = __builtin_bit_cast(unsigned, prod);
                }
            }
            *reinterpret_cast<gi_u32x4 *>(dst + (size_t)(tid + NT * q) * 4) = v;
        }
    };
    auto multiply = [&](int buf) {
        const gi_u32x4 *gA = reinterpret_cast<const gi_u32x4 *>(sbuf + buf * PART);
        const gi_u32x4 *xB = gA + 4 * Nt;
        gi_f16x8 a[TA][2], b[TB][2];
#pragma unroll
        for (int t = 0; t < TA; ++t)
#pragma unroll
            for (int pc = 0; pc < 2; ++pc) a[t][pc] = __builtin_bit_cast(gi_f16x8, gA[(pc * 2 + h) * Nt + wa * 32 * TA + 32 * t + il]);
#pragma unroll
        for (int u = 0; u < TB; ++u)
#pragma unroll
            for (int pc = 0; pc < 2; ++pc) b[u][pc] = __builtin_bit_cast(gi_f16x8, xB[(pc * 2 + h) * Kt + wb * 32 * TB + 32 * u + il]);
#pragma unroll
        for (int t = 0; t < TA; ++t)
#pragma unroll
            for (int u = 0; u < TB; ++u) {
                gi_f32x16 c16 = acc[t][u];
                c16 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[t][1], b[u][0], c16, 0, 0, 0);   // m h (smallest first)
                c16 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[t][0], b[u][1], c16, 0, 0, 0);   // h m
                c16 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[t][0], b[u][0], c16, 0, 0, 0);   // h h
                acc[t][u] = c16;
            }
    };

    // Pipeline as wgrad_split_kernel: loads two stages ahead; the two halves of the workgroup's waves stage / multiply out of phase.
    // The factors of stage c + 2 are written (by 16 threads) during iteration c, one barrier before iteration c + 1's stash reads them.
    const bool stage_first = wave < (WA * WB) / 2;
    if (n_stage > 0) {
        factors(0, 0);
        fetch(0);
        __syncthreads();
        stash(0);
        if (n_stage > 1) {
            factors(1, 1);
            fetch(1);
        }
    }
    __syncthreads();
    for (long long c = 0; c < n_stage; ++c) {
        const int buf = (int)(c & 1);
        // s_f[buf] held stage c's factors, last read by the stash of iteration c - 1, i.e. before the barrier that ended it: stage
        // c + 2's go there now and are read by iteration c + 1's stash, after this iteration's barrier
        if (c + 2 < n_stage) factors(c + 2, buf);
        if (stage_first) {
            if (c + 1 < n_stage) stash(buf ^ 1);
            if (c + 2 < n_stage) fetch(c + 2);
        }
        multiply(buf);
        if (!stage_first) {
            if (c + 1 < n_stage) stash(buf ^ 1);
            if (c + 2 < n_stage) fetch(c + 2);
        }
        __syncthreads();
    }

    // partial block -> workspace (or dW itself when there is a single row range).  acc[t][u][r] belongs to the T-image POSITIONS
    // wave's base + 32 t + 8 (r >> 2) + 4 h + (r & 3) of g and wave's base + 32 u + il of x (the slab starts are multiples of 64)
    float *dst = p.out + (size_t)split * p.N * p.K;
#pragma unroll
    for (int t = 0; t < TA; ++t)
#pragma unroll
        for (int u = 0; u < TB; ++u)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = n0 + gi_t_col(wa * 32 * TA + 32 * t + 8 * (r >> 2) + 4 * h + (r & 3));      // positions -> columns
                const int k = k0 + gi_t_col(wb * 32 * TB + 32 * u + il);
                dst[(size_t)n * p.K + k] = eref > kGiEZero ? ldexpf(acc[t][u][r], eref) : 0.0f;
            }
}


// the weight-gradient kernel for wgrad.hip's plan (cfg: 0 = 256 x 256, 1 = 128 x 256, 2 = 256 x 128 blocks of dW)
int launch_wgrad_img(int cfg, const unsigned *gT, const int *gE, int g_seg, const unsigned *xT, const int *xE, int x_seg, long long M, int N, int K,
                     float *out, int nslab_n, int nslab_k, int msplit, const int *run_flag, hipStream_t s) {
    WgradImgParams p;
    p.gT = gT; p.xT = xT; p.gE = gE; p.xE = xE; p.out = out; p.M = M; p.N = N; p.K = K; p.g_seg = g_seg; p.x_seg = x_seg;
    p.nslab_n = nslab_n; p.nslab_k = nslab_k; p.msplit = msplit; p.n_blocks = (M + 31) / 32; p.run_flag = run_flag;
    auto go = [&](auto kern, int Nt, int Kt, int waves) -> int {
        static LdsGrant grant;
        const size_t lds = (size_t)2 * 4 * (Nt + Kt) * 16;
        RQ_RETURN_IF_HIP(grant.ensure(reinterpret_cast<const void *>(kern), (int)lds));      // (+ the kernel's static LDS)
        hipLaunchKernelGGL(kern, dim3(nslab_n * nslab_k * msplit), dim3(64 * waves), lds, s, p);
        RQ_CHECK_LAUNCH("wgrad_img_kernel");
        return 0;
    };
    switch (cfg) {
        case 0: return go(wgrad_img_kernel<4, 2, 2, 4>, 256, 256, 8);    // 8 waves of 128 x 64
        case 1: return go(wgrad_img_kernel<2, 2, 2, 4>, 128, 256, 8);    // 8 waves of 64 x 64
        default: return go(wgrad_img_kernel<2, 2, 4, 2>, 256, 128, 8);
    }
}

}  // namespace rqhip

using namespace rqhip;

static bool gi_al16(const void *q) { return (reinterpret_cast<uintptr_t>(q) & 15u) == 0; }
static int gi_seg_of(int N) { return N % 256 == 0 ? 256 : 128; }   // segment width of an image this library produces for N columns

extern "C" int rqhip_img_supported(int N) { return (N > 0 && N % 128 == 0) ? 1 : 0; }
extern "C" int rqhip_img_seg(int N) { return rqhip_img_supported(N) ? gi_seg_of(N) : 0; }
extern "C" size_t rqhip_img_r_bytes(int64_t M, int N) { return (M > 0 && N > 0 && N % 16 == 0) ? (size_t)((M + 63) / 64) * 64 * (size_t)N * 4 : 0; }
extern "C" size_t rqhip_img_t_bytes(int64_t M, int N) { return (M > 0 && N > 0) ? (size_t)((M + 31) / 32) * 32 * (size_t)N * 4 : 0; }
extern "C" size_t rqhip_img_e_bytes(int64_t M, int N) { return rqhip_img_supported(N) && M > 0 ? (size_t)(N / gi_seg_of(N)) * (size_t)M * sizeof(int) : 0; }

static int gi_check_img(const rqhip_img *im, const char *what, bool need_r, bool need_t) {
    if (!im || im->M < 0 || !rqhip_img_supported(im->N) || im->seg != gi_seg_of(im->N) || (im->M > 0 && !im->E) ||
        (im->M > 0 && need_r && !im->R) || (im->M > 0 && need_t && !im->T) || !gi_al16(im->R) || !gi_al16(im->T) || !gi_al16(im->E)) {
        set_error("%s: bad image (N = %d must be a multiple of 128, seg = rqhip_img_seg(N), E%s%s present, 16-byte aligned buffers)", what,
                  im ? im->N : 0, need_r ? ", R" : "", need_t ? ", T" : "");
        return RQHIP_EARG;
    }
    return RQHIP_OK;
}

extern "C" int rqhip_img_pack(const float *A, const float *Y, float *masked_out, int64_t M, int N, const rqhip_img *out, const int *run_flag,
                              rqhip_stream_t stream) {
    if (int rc = gi_check_img(out, "img_pack", false, false)) return rc;
    if (M < 0 || out->M != M || out->N != N || (M > 0 && !A) || (masked_out && !Y) || !gi_al16(A) || !gi_al16(Y) || !gi_al16(masked_out) ||
        (!out->R && !out->T)) {
        set_error("img_pack: bad arguments (A [M, N] 16-byte aligned, out image of the same M, N with R and / or T; masked_out needs Y)");
        return RQHIP_EARG;
    }
    if (M == 0) return RQHIP_OK;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    ImgPackParams p;
    p.A = A; p.Y = Y; p.masked_out = masked_out; p.seg = out->seg; p.run_flag = run_flag;
    p.o.C = nullptr; p.o.R = reinterpret_cast<unsigned *>(out->R); p.o.T = reinterpret_cast<unsigned *>(out->T); p.o.E = out->E; p.o.M = M; p.o.N = N;
    const unsigned blocks = (unsigned)(((M + 31) / 32) * (N / out->seg));
    profile_begin(s, RQHIP_PROF_MAXIMA, 0.0, (double)M * N * 4 * ((Y ? 2 : 1) + (masked_out ? 1 : 0) + (out->R ? 1 : 0) + (out->T ? 1 : 0)));
    if (out->seg == 256) hipLaunchKernelGGL(img_pack_kernel<4>, dim3(blocks), dim3(256), 0, s, p);
    else hipLaunchKernelGGL(img_pack_kernel<2>, dim3(blocks), dim3(128), 0, s, p);
    profile_end(s);
    RQ_CHECK_LAUNCH("img_pack_kernel");
    return RQHIP_OK;
}

extern "C" int rqhip_img_unpack(const rqhip_img *im, int from_t, float *out, rqhip_stream_t stream) {
    if (int rc = gi_check_img(im, "img_unpack", !from_t, from_t != 0)) return rc;
    if (im->M > 0 && !out) {
        set_error("img_unpack: null output");
        return RQHIP_EARG;
    }
    if (im->M == 0) return RQHIP_OK;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const unsigned blocks = (unsigned)cu_count() * 8;
    if (from_t) hipLaunchKernelGGL(img_unpack_t_kernel, dim3(blocks), dim3(256), 0, s, reinterpret_cast<const unsigned *>(im->T), im->E, (long long)im->M, im->N, im->seg, out);
    else hipLaunchKernelGGL(img_unpack_kernel, dim3(blocks), dim3(256), 0, s, reinterpret_cast<const unsigned *>(im->R), im->E, (long long)im->M, im->N, im->seg, out);
    RQ_CHECK_LAUNCH("img_unpack_kernel");
    return RQHIP_OK;
}

extern "C" int rqhip_rows_differ(const float *g_out, int64_t M, float announced, int *flag, rqhip_stream_t stream) {
    if (M < 0 || !flag || (M > 0 && !g_out)) {
        set_error("rows_differ: bad arguments");
        return RQHIP_EARG;
    }
    hipLaunchKernelGGL(gi_rows_differ_kernel, dim3(1), dim3(1024), 0, reinterpret_cast<hipStream_t>(stream), g_out, (long long)M, announced, flag);
    RQ_CHECK_LAUNCH("gi_rows_differ_kernel");
    return RQHIP_OK;
}

extern "C" int rqhip_gemm_img_supported(int Nc, int R) { return (rqhip_img_supported(Nc) && rqhip_img_supported(R)) ? 1 : 0; }

extern "C" int rqhip_gemm_img(const rqhip_gemm_img_args *a, rqhip_stream_t stream) {
    if (!a) {
        set_error("gemm_img: null argument block");
        return RQHIP_EARG;
    }
    if (int rc = gi_check_img(&a->A, "gemm_img (A)", true, false)) return rc;
    const int64_t M = a->A.M;
    const int R = a->A.N, Nc = a->Nc, epi = a->epilogue;
    const bool want_img = a->out.R || a->out.T;
    if (!a->image || !rqhip_gemm_img_supported(Nc, R) || epi < 0 || epi > 3 || !gi_al16(a->image) || !gi_al16(a->C) || !gi_al16(a->aux) ||
        (M > 0 && !a->C && !want_img && epi != RQHIP_EPI_RECON) || R / a->A.seg > kGiMaxSeg) {
        set_error("gemm_img: bad arguments or unsupported shape (Nc = %d, R = %d: multiples of 128, R <= %d; epilogue %d; an output)", Nc, R,
                  kGiMaxSeg * 256, epi);
        return RQHIP_EARG;
    }
    if (want_img) {
        if (int rc = gi_check_img(&a->out, "gemm_img (out)", false, false)) return rc;
        if (a->out.M != M || a->out.N != Nc) {
            set_error("gemm_img: the output image must be [M, Nc]");
            return RQHIP_EARG;
        }
    }
    const int cols = gi_seg_of(Nc);
    if (M > 0 && epi == RQHIP_EPI_RECON && (cols != 256 || !a->aux || !a->loss_rows || !a->workspace ||
                                           a->workspace_bytes < (size_t)(Nc / 256) * (size_t)M * sizeof(float))) {
        set_error("gemm_img: the reconstruction-loss epilogue needs Nc %% 256 == 0 (Nc = %d), aux = X [M, Nc], loss_rows and a workspace of "
                  "rqhip_gemm_split_recon_workspace_bytes", Nc);
        return RQHIP_EARG;
    }
    if (M > 0 && epi == RQHIP_EPI_MASK) {
        if (int rc = gi_check_img(&a->Y, "gemm_img (Y)", a->Y.T == nullptr, false)) return rc;      // R planes, or T planes
        if (a->Y.M != M || a->Y.N != Nc) {
            set_error("gemm_img: the mask image must be [M, Nc]");
            return RQHIP_EARG;
        }
    }
    if (M == 0) return RQHIP_OK;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    GemmImgParams p;
    p.aR = reinterpret_cast<const unsigned *>(a->A.R); p.aE = a->A.E; p.a_nseg = R / a->A.seg;
    p.a_seg_shift = a->A.seg == 256 ? 4 : 3;
    p.planes = reinterpret_cast<const unsigned *>(a->image);
    p.counter = const_cast<unsigned *>(p.planes) + (size_t)(R / kGiK) * 2 * 2 * Nc * 4;
    p.b_exp = reinterpret_cast<const int *>(p.counter + kWeightImageTailWords);
    p.M = M; p.R = R; p.Nc = Nc; p.n_col_tiles = Nc / cols;
    p.o.C = a->C; p.o.R = reinterpret_cast<unsigned *>(a->out.R); p.o.T = reinterpret_cast<unsigned *>(a->out.T); p.o.E = a->out.E; p.o.M = M; p.o.N = Nc;
    p.X = a->aux; p.yR = reinterpret_cast<const unsigned *>(a->Y.R); p.yT = reinterpret_cast<const unsigned *>(a->Y.T); p.rowsum = reinterpret_cast<float *>(a->workspace);
    p.row_scale = a->row_scale; p.row_scales = a->row_scales; p.run_flag = a->run_flag;
    const int cus = cu_count();
    const long long slots = (long long)cus * 2, big_rows = 128, small_rows = 64;
    // whole rounds of big tiles, the remainder as small tiles (as rqhip_gemm_split_ex)
    const long long rt_all = (M + big_rows - 1) / big_rows;
    long long rt_big = ((rt_all * p.n_col_tiles) / slots) * slots / p.n_col_tiles;
    if (rt_big * big_rows > M) rt_big = M / big_rows;
    if ((rt_all * p.n_col_tiles) % slots > (3 * slots) / 10 && rt_all * p.n_col_tiles >= slots) rt_big = rt_all;
    const long long rem_rows = M - rt_big * big_rows > 0 ? M - rt_big * big_rows : 0;
    const long long rt_small = (rem_rows + small_rows - 1) / small_rows;
    p.rt_big = (int)rt_big;
    p.n_big = (unsigned)(rt_big * p.n_col_tiles);
    p.n_tiles = p.n_big + (unsigned)(rt_small * p.n_col_tiles);
    const int wm = 4 / (cols / 64);
    size_t lds = (size_t)2 * 4 * (128 * 4 + 32) * 4;                           // two A stages of a 128-row tile
    const size_t lds_epi = (size_t)4 * 32 * kGiTS * 4;                          // re-used by the epilogue's transposition blocks
    (void)wm;
    if (lds < lds_epi) lds = lds_epi;
    const int grid = (int)((long long)p.n_tiles < slots ? (long long)p.n_tiles : slots);
    const int nq = a->xcd_queues ? kGiQueues : 1;
    auto go = [&](auto kern) -> int {
        static LdsGrant grant;
        RQ_RETURN_IF_HIP(grant.ensure(reinterpret_cast<const void *>(kern), (int)lds));
        profile_begin(s, RQHIP_PROF_GEMM_SPLIT, 2.0 * (double)M * Nc * R, 4.0 * (double)M * (R + Nc * (epi == 2 ? 2 : 1) * (a->C ? 1 : 0)
                                                                                             + Nc * ((a->out.R ? 1 : 0) + (a->out.T ? 1 : 0))));
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, s, p, nq);
        profile_end(s);
        RQ_CHECK_LAUNCH("gemm_img_kernel");
        return 0;
    };
    int rc;
    if (cols == 256) rc = epi == 3 ? go(gemm_img_kernel<3, 256>) : epi == 2 ? go(gemm_img_kernel<2, 256>) : epi == 1 ? go(gemm_img_kernel<1, 256>) : go(gemm_img_kernel<0, 256>);
    else rc = epi == 3 ? go(gemm_img_kernel<3, 128>) : epi == 1 ? go(gemm_img_kernel<1, 128>) : go(gemm_img_kernel<0, 128>);
    if (rc) return rc;
    if (epi == RQHIP_EPI_RECON) {
        hipLaunchKernelGGL(gi_recon_rows_finish_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, s,
                           reinterpret_cast<const float *>(a->workspace), Nc / 256, (long long)M, a->loss_rows, a->run_flag);
        RQ_CHECK_LAUNCH("gi_recon_rows_finish_kernel");
    }
    return RQHIP_OK;
}
