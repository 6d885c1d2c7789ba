// mlp_small.hip -- the encoder / decoder Linear layers at the batch sizes the reference ships (rows < 4096: configs/rqvae_amazon.gin:7
// batch 640, rqvae_ml32m.gin:7 batch 64), forward and data gradient, in exact fp32 arithmetic on the fp32 matrix instruction (gfx950).
// SURVEY.md section 8 row f2; reference modules/encoder.py:25-38 (`relu(x W^T)` forward; autograd's `g W` with the ReLU backward of the
// layer below).  Replaces the hipBLASLt calls + threshold_backward launches of rounds 1-5 below 4096 rows.
//
// At these sizes a layer is a few hundred 32 x 32 output tiles and a reduction of 32 ... 768 terms; nothing is HBM- or pipe-bound.  What
// the first form of this kernel (32x32x2 instructions, every lane loading 16 bytes of its own row per 8-term block) ran into is the
// per-CU load path: a wave's load instruction touched 32 cache lines and used 32 bytes of each, the other three quarters of a line were
// asked for by LATER instructions -- with eight waves per CU the 32 KB L1 had lost the line by then (deeper prefetch made it slower:
// tools/linear_small_ab.py, profiles/r06_linear_small.txt).  So:
//   * one workgroup per 32 x (32 CB) output tile, its reduction split over KS waves (contiguous ranges of 32-term GROUPS = one 128-byte
//     line of every operand row); the KS partial tiles meet in LDS and are summed in wave order;
//   * v_mfma_f32_16x16x4_f32: lane (i, kq) supplies k-slot kq of row i, and loads the 32 bytes (terms 8 kq .. 8 kq + 7) of its row's
//     line with two back-to-back 16-byte loads: a wave instruction covers 16 rows x 64 bytes, the four quarters of a line are requested
//     within two instructions and the line is dead afterwards.  Instruction e of a group consumes term 8 kq + e of every slot: the
//     chain runs over a group in the order 0 8 16 24 1 9 17 25 ... 7 15 23 31 (restated by oracle/rq_oracle.c:rqo_linear_small);
//   * two register buffers (ping-pong), the next group's loads issued before the current group's 16 (32) matrix instructions per tile row;
//   * the weight in either orientation without a transposed copy: W [N, Kr] (forward: as A) or W [Kr, N] (data gradient g W): lane
//     (j, kq) loads, for each of its 8 terms, the 2 CB consecutive COLUMNS 2 CB j ... of that term's row (8 or 16 bytes; 16 lanes =
//     one or two full lines) and serves column 2 CB j + ct in column tile ct -- a column permutation the epilogue undoes for free;
//   * epilogues: store, ReLU, or "where aux > 0" (the ReLU backward of the layer below: the masked gradient is written once, no
//     threshold_backward launch).
// Arithmetic: out[m][n] = ((p_0 + p_1) + ...) + p_{KS-1}, p_w = one fp32 FMA chain from +0 over wave w's groups.  (KS, CB) follow
// from the shape alone (ls_plan) -- same bits eager and replayed, on any box.
#include "rqhip_common.h"

namespace rqhip {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

struct LinSmall {
    const float *a;     // [M, Kr]
    const float *w;     // WKN ? [Kr, N] : [N, Kr]
    const float *aux;   // [M, N] (epilogue 3)
    float *out;         // [M, N]
    int M, N, Kr, epi;
};

// one group (32 reduction terms) of a wave's operands: a[rt][e] = A[row 16 rt + i][32 g + 8 kq + e]; b[ct][e] likewise for W
template <int CB>
struct LsRegs {
    float a[2][8];
    float b[2 * CB][8];
};

template <int CB, bool WKN>
__device__ __forceinline__ void ls_load(const float *const (&ap)[2], const float *const (&bp)[2 * CB], const float *bq, int N, int g,
                                        LsRegs<CB> &r) {
    constexpr int CT = 2 * CB;
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
        const f32x4 lo = *reinterpret_cast<const f32x4 *>(ap[rt] + 32 * g), hi = *reinterpret_cast<const f32x4 *>(ap[rt] + 32 * g + 4);
        r.a[rt][0] = lo.x; r.a[rt][1] = lo.y; r.a[rt][2] = lo.z; r.a[rt][3] = lo.w;
        r.a[rt][4] = hi.x; r.a[rt][5] = hi.y; r.a[rt][6] = hi.z; r.a[rt][7] = hi.w;
    }
    if constexpr (WKN) {
        const float *q = bq + (size_t)(32 * g) * N;       // term 32 g + 8 kq + e, columns CT j ..
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            if constexpr (CB == 1) {
                const f32x2 v = *reinterpret_cast<const f32x2 *>(q + (size_t)e * N);
                r.b[0][e] = v.x; r.b[1][e] = v.y;
            } else {
                const f32x4 v = *reinterpret_cast<const f32x4 *>(q + (size_t)e * N);
                r.b[0][e] = v.x; r.b[1][e] = v.y; r.b[2][e] = v.z; r.b[3][e] = v.w;
            }
        }
    } else {
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            const f32x4 lo = *reinterpret_cast<const f32x4 *>(bp[ct] + 32 * g), hi = *reinterpret_cast<const f32x4 *>(bp[ct] + 32 * g + 4);
            r.b[ct][0] = lo.x; r.b[ct][1] = lo.y; r.b[ct][2] = lo.z; r.b[ct][3] = lo.w;
            r.b[ct][4] = hi.x; r.b[ct][5] = hi.y; r.b[ct][6] = hi.z; r.b[ct][7] = hi.w;
        }
    }
}

template <int CB>
__device__ __forceinline__ void ls_mma(const LsRegs<CB> &r, f32x4 (&acc)[2][2 * CB]) {
#pragma unroll
    for (int e = 0; e < 8; ++e)
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int ct = 0; ct < 2 * CB; ++ct) acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(r.a[rt][e], r.b[ct][e], acc[rt][ct], 0, 0, 0);
}

// keeps a group's loads together and in front of the previous group's matrix instructions (the scheduler otherwise sinks them between
// the instructions, half a group later: 3 % slower over the 15 launches of a step, tools/linear_small_ab.py)
#ifdef LS_NO_SCHED_FENCE
#define LS_FENCE
#else
#define LS_FENCE __builtin_amdgcn_sched_barrier(0)
#endif

template <int CB, int KS, bool WKN>
__global__ __launch_bounds__(64 * KS) void lin_small_kernel(const LinSmall p) {
    constexpr int NT = 64 * KS, TW = 32 * CB, CT = 2 * CB;
    __shared__ float red[KS * 32 * TW];
    const int t = threadIdx.x, lane = t & 63, i = lane & 15, kq = lane >> 4;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int n0 = blockIdx.x * TW, m0 = blockIdx.y * 32;
    const int ng = p.Kr >> 5;
    const int lo = (int)((long long)w * ng / KS), hi = (int)((long long)(w + 1) * ng / KS);
    f32x4 acc[2][CT];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) acc[rt][ct] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    if (lo < hi) {
        const float *ap[2], *bp[CT];
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)       // rows past the batch: a valid row is read, nothing is stored
            ap[rt] = p.a + (size_t)min(m0 + 16 * rt + i, p.M - 1) * p.Kr + 8 * kq;
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) bp[ct] = p.w + (size_t)(n0 + 16 * ct + i) * p.Kr + 8 * kq;
        const float *bq = p.w + (size_t)(8 * kq) * p.N + n0 + CT * i;
        LsRegs<CB> r0, r1;
        ls_load<CB, WKN>(ap, bp, bq, p.N, lo, r0);
        int g = lo;
        for (; g + 2 <= hi; g += 2) {
            ls_load<CB, WKN>(ap, bp, bq, p.N, g + 1, r1);
            LS_FENCE;
            ls_mma<CB>(r0, acc);
            LS_FENCE;
            ls_load<CB, WKN>(ap, bp, bq, p.N, min(g + 2, hi - 1), r0);
            LS_FENCE;
            ls_mma<CB>(r1, acc);
            LS_FENCE;
        }
        if (g < hi) ls_mma<CB>(r0, acc);
    }
    // acc[rt][ct][r]: row 16 rt + 4 kq + r of the tile; column 16 ct + i (forward) or CT i + ct (data gradient: the lane's own columns)
    float *mine = red + (size_t)w * 32 * TW;
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r) mine[(16 * rt + 4 * kq + r) * TW + (WKN ? CT * i + ct : 16 * ct + i)] = acc[rt][ct][r];
    __syncthreads();
    for (int f = t; f < 8 * TW; f += NT) {          // 32 rows x TW / 4 float4s
        const int row = f / (TW / 4), c4 = f % (TW / 4);
        const float *src = red + row * TW + 4 * c4;
        f32x4 v = *reinterpret_cast<const f32x4 *>(src);
#pragma unroll
        for (int k = 1; k < KS; ++k) {
            const f32x4 q = *reinterpret_cast<const f32x4 *>(src + (size_t)k * 32 * TW);
            v.x = v.x + q.x; v.y = v.y + q.y; v.z = v.z + q.z; v.w = v.w + q.w;
        }
        if (m0 + row >= p.M) continue;
        const size_t at = (size_t)(m0 + row) * p.N + n0 + 4 * c4;
        if (p.epi == RQHIP_EPI_RELU) {          // (a NaN stays a NaN, as torch.relu)
            v.x = v.x < 0.0f ? 0.0f : v.x; v.y = v.y < 0.0f ? 0.0f : v.y;
            v.z = v.z < 0.0f ? 0.0f : v.z; v.w = v.w < 0.0f ? 0.0f : v.w;
        } else if (p.epi == RQHIP_EPI_MASK) {   // threshold_backward(out, aux, 0)
            const f32x4 k = *reinterpret_cast<const f32x4 *>(p.aux + at);
            v.x = k.x <= 0.0f ? 0.0f : v.x; v.y = k.y <= 0.0f ? 0.0f : v.y;
            v.z = k.z <= 0.0f ? 0.0f : v.z; v.w = k.w <= 0.0f ? 0.0f : v.w;
        }
        *reinterpret_cast<f32x4 *>(p.out + at) = v;
    }
}

// (column blocks per tile, waves per tile) of a shape: enough waves to occupy the chip's 1024 SIMDs without a second round of workgroups
static void ls_plan(int64_t M, int N, int Kr, int &cb, int &ks) {
    const int64_t row_tiles = (M + 31) / 32;
    const int64_t wg1 = row_tiles * (N / 32);
    constexpr int cus = 256;   // MI355X; a constant, so the plan -- and with it the summation order -- depends on the shape alone
    cb = (N % 64 == 0 && wg1 > cus) ? 2 : 1;
    const int64_t wgs = row_tiles * (N / (32 * cb));
    (void)wgs; (void)Kr;
    ks = 4;     // measured (tools/linear_small_ab.py): 8 or 16 waves per tile never beat 4, at 640 rows or at 64
}

template <int CB, int KS>
static void ls_launch(const LinSmall &p, int w_kn, dim3 grid, hipStream_t s) {
    if (w_kn)
        hipLaunchKernelGGL((lin_small_kernel<CB, KS, true>), grid, dim3(64 * KS), 0, s, p);
    else
        hipLaunchKernelGGL((lin_small_kernel<CB, KS, false>), grid, dim3(64 * KS), 0, s, p);
}

}  // namespace rqhip

using namespace rqhip;

extern "C" int rqhip_linear_small_supported(int64_t M, int N, int Kr) {
    return (M > 0 && M < (1ll << 22) && N > 0 && Kr > 0 && N % 32 == 0 && Kr % 32 == 0) ? 1 : 0;
}

extern "C" int rqhip_linear_small_plan(int64_t M, int N, int Kr, int *col_blocks, int *waves) {
    if (!rqhip_linear_small_supported(M, N, Kr) || !col_blocks || !waves) {
        set_error("linear_small_plan: unsupported shape M=%lld N=%d Kr=%d (N and Kr must be multiples of 32)", (long long)M, N, Kr);
        return RQHIP_EARG;
    }
    ls_plan(M, N, Kr, *col_blocks, *waves);
    return RQHIP_OK;
}

extern "C" int rqhip_linear_small(const float *a, const float *w, int w_kn, float *out, int64_t M, int N, int Kr, int epilogue,
                                  const float *aux, int col_blocks, int waves, rqhip_stream_t stream) {
    if (M == 0) return RQHIP_OK;
    if (!rqhip_linear_small_supported(M, N, Kr) || !a || !w || !out) {
        set_error("linear_small: M=%lld N=%d Kr=%d: N and Kr must be positive multiples of 32, pointers non-null", (long long)M, N, Kr);
        return RQHIP_EARG;
    }
    if ((epilogue != RQHIP_EPI_STORE && epilogue != RQHIP_EPI_RELU && epilogue != RQHIP_EPI_MASK) || (epilogue == RQHIP_EPI_MASK && !aux)) {
        set_error("linear_small: epilogue %d (store 0 / relu 1 / mask 3 with aux) is not one of this kernel's", epilogue);
        return RQHIP_EARG;
    }
    if (((uintptr_t)a | (uintptr_t)w | (uintptr_t)out | (uintptr_t)aux) & 15u) {
        set_error("linear_small: a, w, out and aux must be 16-byte aligned");
        return RQHIP_EARG;
    }
    int cb, ks;
    ls_plan(M, N, Kr, cb, ks);
    if (col_blocks > 0) cb = col_blocks;
    if (waves > 0) ks = waves;
    if (!((cb == 1 || (cb == 2 && N % 64 == 0)) && (ks == 4 || ks == 8 || (ks == 16 && cb == 1)))) {
        set_error("linear_small: (col_blocks, waves) = (%d, %d) is not one of (1|2, 4|8) / (1, 16)", cb, ks);
        return RQHIP_EARG;
    }
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    LinSmall p;
    p.a = a; p.w = w; p.aux = aux; p.out = out;
    p.M = (int)M; p.N = N; p.Kr = Kr; p.epi = epilogue;
    const dim3 grid((unsigned)(N / (32 * cb)), (unsigned)((M + 31) / 32));
    profile_begin(s, RQHIP_PROF_LINEAR_SMALL, 2.0 * (double)M * N * Kr, 4.0 * ((double)M * (N + Kr) + (double)N * Kr));
    if (cb == 1 && ks == 4) ls_launch<1, 4>(p, w_kn, grid, s);
    else if (cb == 1 && ks == 8) ls_launch<1, 8>(p, w_kn, grid, s);
    else if (cb == 1) ls_launch<1, 16>(p, w_kn, grid, s);
    else if (ks == 4) ls_launch<2, 4>(p, w_kn, grid, s);
    else ls_launch<2, 8>(p, w_kn, grid, s);
    RQ_CHECK_LAUNCH("lin_small_kernel");
    profile_end(s);
    return RQHIP_OK;
}
