// wgrad.hip -- weight gradient of a bias-free Linear(+ReLU) layer with the ReLU backward fused in (gfx950).
//
// SURVEY.md section 8 row f2 (reference modules/encoder.py:7-38: the encoder/decoder MLPs around the quantiser).
// For y = relu(x W^T) autograd runs, per layer,   g_pre = gy * (y > 0)   (one elementwise pass over [M, N]),
//   dW = g_pre^T x   (a GEMM whose reduction runs over the M = 100 000 batch rows)   and   gx = g_pre W.
// The library does the dW shapes of this MLP badly -- tiny outputs (128 x 256 ... 512 x 768) with a huge reduction
// dimension: 46-97 TFLOP/s on the small layers -- and the mask passes cost 0.32 ms of a 5.9 ms step.  This kernel
// computes dW with the mask applied while the gradient rows are staged, and writes g_pre back once for the gx GEMM
// that follows, so the separate pass disappears.
//
// Mapping
//   * dW[n,k] = sum_m g_pre[m,n] x[m,k]:  v_mfma_f32_32x32x2_f32 with A = g_pre^T (32 n x 2 m), B = x (2 m x 32 k):
//     the reduction index m is the ROW index of both operands, so both are consumed exactly as they lie in memory;
//     no transposition anywhere.
//   * a workgroup owns an Nt x Kt block of dW ("slab") and a contiguous range of rows; it streams 32-row chunks of
//     gy / y / x through a double-buffered LDS stage (global_load_dwordx4 -> mask in registers -> ds_write_b128;
//     the loads of chunk c+1 are in flight while chunk c is multiplied).
//   * a wave owns (32 TA) x (32 TB) of the block.  Tile t of its TA n-tiles takes the features n0 + TA*i + t
//     (i = lane & 31), so ONE ds_read_b128 / b64 / b32 of TA consecutive floats is the A operand of all TA tiles;
//     likewise for k.  Per pair of rows: one A read, one B read, TA*TB MFMAs.  All LDS accesses are conflict free.
//   * row ranges are reduced in a fixed order: every workgroup writes its partial block to the workspace and a second
//     kernel adds the partials as a balanced binary tree over the range index (adjacent pairs first) -- bit-reproducible,
//     no atomics, and parallel: a first version summed the 42..256 partials of an element sequentially in one thread
//     and that chain of dependent loads, not the GEMM, was most of the small layers' time.
//   * measured dead end: an LDS-free variant in which every lane loads its own operands straight into a register
//     ring (16 row pairs ahead) is 30 % SLOWER on the big layers (each strip is fetched by 4 / 2 waves through L1).
//   * the slabs of one row range run back to back on ONE XCD (workgroup b lands on XCD b % 8), so the gy / y / x strips
//     they share are fetched from HBM once and hit in that XCD's L2 for the others.
#include "rqhip_common.h"

namespace rqhip {

typedef float wg_f32x16 __attribute__((ext_vector_type(16)));
typedef float wg_f32x4 __attribute__((ext_vector_type(4)));
typedef float wg_f32x2 __attribute__((ext_vector_type(2)));

constexpr int kWgChunk = 32;  // granularity of the row ranges (part of the summation-order contract, see the oracle)

struct WgradParams {
    const float *g;   // [M,N] upstream gradient wrt the layer output
    const float *y;   // [M,N] layer output (ReLU applied) or nullptr: no mask
    const float *x;   // [M,K] layer input
    float *gm;        // [M,N] masked gradient written back (may alias g) or nullptr
    float *out;       // partial blocks [msplit][N][K] (or dW itself when msplit == 1)
    long long M;
    int N, K;
    int nslab_n, nslab_k, msplit;
    long long n_chunks;  // ceil(M / 32)
    int pow2;            // msplit rounded up to a power of two (reduction tree)
};

template <int V>
struct VecOf;
template <>
struct VecOf<1> { typedef float type; };
template <>
struct VecOf<2> { typedef wg_f32x2 type; };
template <>
struct VecOf<4> { typedef wg_f32x4 type; };

template <int V>
__device__ __forceinline__ float vec_get(const typename VecOf<V>::type &v, int i) {
    if constexpr (V == 1) return v;
    else return v[i];
}

// TA x TB tiles per wave, WA x WB waves per workgroup; MASK: y given; MC: rows per LDS stage
template <int TA, int TB, int WA, int WB, bool MASK, int MC>
__global__ __launch_bounds__(64 * WA * WB) void wgrad_kernel(const WgradParams p) {
    constexpr int Nt = 32 * TA * WA, Kt = 32 * TB * WB, NT = 64 * WA * WB;
    constexpr int G4 = MC * Nt / 4, X4 = MC * Kt / 4;           // float4s per stage
    constexpr int GQ = (G4 + NT - 1) / NT, XQ = (X4 + NT - 1) / NT;  // per thread
    extern __shared__ __attribute__((aligned(16))) char wg_smem[];
    float *sG = reinterpret_cast<float *>(wg_smem);            // [2][MC][Nt]
    float *sX = sG + 2 * MC * Nt;                              // [2][MC][Kt]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int il = lane & 31, h = lane >> 5;
    const int wa = wave / WB, wb = wave % WB;
    // The slabs of ONE row range share its gy / y / x strips, and they only share them in cache if they run on the same
    // XCD (each has its own L2; workgroup b lands on XCD b % 8): XCD x takes the row ranges x, x + 8, ... and walks
    // their slabs back to back.  (With consecutive workgroups as the slabs of a range, the PMC counters showed every
    // strip fetched once per slab: 1.76 GB for the 512 x 768 layer against 0.72 GB of operands.  Measured effect of the
    // placement on time: none -- the kernel is not traffic-bound -- so the number of row ranges is NOT rounded to a
    // multiple of 8 to make it applicable: 42 ranges x 6 slabs on 252 CUs beat 40 x 6 on 240.)
    const int nslabs = p.nslab_n * p.nslab_k;
    int slab, split;
    if ((p.msplit & 7) == 0) {
        const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
        slab = j % nslabs;
        split = (j / nslabs) * 8 + xcd;
    } else {
        slab = blockIdx.x % nslabs;
        split = blockIdx.x / nslabs;
    }
    const int slab_n = slab / p.nslab_k, slab_k = slab % p.nslab_k;
    const int n0 = slab_n * Nt, k0 = slab_k * Kt;
    // row range of this workgroup: granules [G s / msplit, G (s+1) / msplit) of 32 rows; walked in stages of MC rows
    const long long r_begin = (p.n_chunks * split / p.msplit) * kWgChunk;
    long long r_end = (p.n_chunks * (split + 1) / p.msplit) * kWgChunk;
    if (r_end > p.M) r_end = p.M;
    const long long c_begin = 0, c_end = (r_end - r_begin + MC - 1) / MC;   // stages
    const bool write_back = MASK && p.gm != nullptr && slab_k == 0;

    wg_f32x16 acc[TA][TB];
#pragma unroll
    for (int t = 0; t < TA; ++t)
#pragma unroll
        for (int u = 0; u < TB; ++u)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][u][r] = 0.0f;

    wg_f32x4 rg[GQ], ry[MASK ? GQ : 1], rx[XQ];
    auto fetch = [&](long long chunk) {
        const long long row0 = r_begin + chunk * MC;
#pragma unroll
        for (int q = 0; q < GQ; ++q) {
            const int f = tid + q * NT;
            const int r = f / (Nt / 4), c4 = f % (Nt / 4);
            const long long row = row0 + r;
            const bool ok = (G4 % NT == 0 || f < G4) && row < r_end;
            const size_t off = (size_t)(ok ? row : 0) * p.N + n0 + 4 * c4;
            rg[q] = ok ? *reinterpret_cast<const wg_f32x4 *>(p.g + off) : wg_f32x4{0.f, 0.f, 0.f, 0.f};
            if (MASK) ry[q] = ok ? *reinterpret_cast<const wg_f32x4 *>(p.y + off) : wg_f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int q = 0; q < XQ; ++q) {
            const int f = tid + q * NT;
            const int r = f / (Kt / 4), c4 = f % (Kt / 4);
            const long long row = row0 + r;
            const bool ok = (X4 % NT == 0 || f < X4) && row < r_end;
            rx[q] = ok ? *reinterpret_cast<const wg_f32x4 *>(p.x + (size_t)row * p.K + k0 + 4 * c4)
                       : wg_f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    auto stash = [&](long long chunk, int buf) {
        const long long row0 = r_begin + chunk * MC;
        float *dG = sG + buf * MC * Nt, *dX = sX + buf * MC * Kt;
#pragma unroll
        for (int q = 0; q < GQ; ++q) {
            const int f = tid + q * NT;
            if (G4 % NT != 0 && f >= G4) continue;
            wg_f32x4 v = rg[q];
            if (MASK) {  // threshold_backward(gy, y, 0): 0 where y <= 0
                v.x = ry[q].x <= 0.0f ? 0.0f : v.x;
                v.y = ry[q].y <= 0.0f ? 0.0f : v.y;
                v.z = ry[q].z <= 0.0f ? 0.0f : v.z;
                v.w = ry[q].w <= 0.0f ? 0.0f : v.w;
            }
            *reinterpret_cast<wg_f32x4 *>(dG + 4 * f) = v;
            if (write_back) {
                const int r = f / (Nt / 4), c4 = f % (Nt / 4);
                const long long row = row0 + r;
                if (row < r_end) *reinterpret_cast<wg_f32x4 *>(p.gm + (size_t)row * p.N + n0 + 4 * c4) = v;
            }
        }
#pragma unroll
        for (int q = 0; q < XQ; ++q) {
            const int f = tid + q * NT;
            if (X4 % NT != 0 && f >= X4) continue;
            *reinterpret_cast<wg_f32x4 *>(dX + 4 * f) = rx[q];
        }
    };

    if (c_begin < c_end) {
        fetch(c_begin);
        stash(c_begin, 0);
    }
    __syncthreads();
    for (long long c = c_begin; c < c_end; ++c) {
        const int buf = (int)((c - c_begin) & 1);
        const bool more = c + 1 < c_end;
        if (more) fetch(c + 1);
        const float *gA = sG + buf * MC * Nt + wa * 32 * TA + TA * il;
        const float *xB = sX + buf * MC * Kt + wb * 32 * TB + TB * il;
        typename VecOf<TA>::type a_cur = *reinterpret_cast<const typename VecOf<TA>::type *>(gA + h * Nt);
        typename VecOf<TB>::type b_cur = *reinterpret_cast<const typename VecOf<TB>::type *>(xB + h * Kt);
#pragma unroll
        for (int s = 0; s < MC / 2; ++s) {
            typename VecOf<TA>::type a_nxt = a_cur;
            typename VecOf<TB>::type b_nxt = b_cur;
            if (s + 1 < MC / 2) {
                a_nxt = *reinterpret_cast<const typename VecOf<TA>::type *>(gA + (2 * (s + 1) + h) * Nt);
                b_nxt = *reinterpret_cast<const typename VecOf<TB>::type *>(xB + (2 * (s + 1) + h) * Kt);
            }
#pragma unroll
            for (int t = 0; t < TA; ++t)
#pragma unroll
                for (int u = 0; u < TB; ++u)
                    acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x2f32(vec_get<TA>(a_cur, t), vec_get<TB>(b_cur, u),
                                                                     acc[t][u], 0, 0, 0);
            a_cur = a_nxt;
            b_cur = b_nxt;
        }
        if (more) stash(c + 1, buf ^ 1);
        __syncthreads();
    }

    // partial block -> workspace (or dW itself when there is a single row range).  acc[t][u][r]: n = n0 + wave's
    // 32 TA base + TA * (8 (r >> 2) + 4 h + (r & 3)) + t,  k = k0 + wave's 32 TB base + TB * il + u
    float *dst = p.out + (size_t)split * p.N * p.K;
#pragma unroll
    for (int t = 0; t < TA; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int n = n0 + wa * 32 * TA + TA * (8 * (r >> 2) + 4 * h + (r & 3)) + t;
            float *row = dst + (size_t)n * p.K + k0 + wb * 32 * TB + TB * il;
            if constexpr (TB == 1) {
                row[0] = acc[t][0][r];
            } else if constexpr (TB == 2) {
                *reinterpret_cast<wg_f32x2 *>(row) = wg_f32x2{acc[t][0][r], acc[t][1][r]};
            } else {
                *reinterpret_cast<wg_f32x4 *>(row) = wg_f32x4{acc[t][0][r], acc[t][1][r], acc[t][2][r], acc[t][3][r]};
            }
        }
}

// dW = balanced-tree sum of the row ranges' partial blocks: leaves padded with zeros to a power of two P, then
// a[j] += a[j + s] for every j that is a multiple of 2s, s = 1, 2, 4, ... (adjacent pairs first; fixed order -- the oracle's
// rqo_linear_wgrad restates it).
//
// A workgroup of 256 threads owns E = 256 / T float4 elements of dW; thread (q, e) sums the subtree of leaves
// [q P/T, (q+1) P/T) of element e in registers -- eight leaves fetched at a time, each fetch a run of E consecutive float4s --
// and the T subtree sums meet in LDS for the upper levels of the same tree.  T is chosen per launch (wgrad_reduce_threads) so
// that every layer has >= 256 K threads in flight: the partial blocks are 4-64 MB per layer and the kernel is a pure stream.
// (Rounds 2-4 staged P x 4 float4s per workgroup in LDS and walked the whole tree there, 4 float4s of dW per workgroup of 256
// threads: 2.5 TB/s over the step's 320 MB of partial blocks, 126 us of a 2.63 ms step, profiles/r05_bench_kernel_stats_summary.txt.)
__device__ __forceinline__ wg_f32x4 wg_add4(wg_f32x4 l, const wg_f32x4 r) {
    l.x = l.x + r.x; l.y = l.y + r.y; l.z = l.z + r.z; l.w = l.w + r.w;
    return l;
}

__device__ __forceinline__ void wgrad_reduce_body(const float *__restrict__ part, int msplit, int pow2, int tlog, size_t nk4,
                                                  float *__restrict__ dw, unsigned block) {
    __shared__ wg_f32x4 a[256];   // [q][e]
    const int T = 1 << tlog, E = 256 >> tlog;
    const int e = threadIdx.x & (E - 1), q = threadIdx.x >> (8 - tlog);
    const size_t elem = (size_t)block * E + e;
    const bool live = elem < nk4;
    const int lpt = pow2 >> tlog;                 // leaves per thread (a power of two >= 1)
    const wg_f32x4 *src = reinterpret_cast<const wg_f32x4 *>(part) + elem;
    const wg_f32x4 zero = wg_f32x4{0.f, 0.f, 0.f, 0.f};
    constexpr int kLevels = 10;                   // subtrees of up to 8 << 10 leaves per thread
    wg_f32x4 st[kLevels];
#pragma unroll
    for (int j = 0; j < kLevels; ++j) st[j] = zero;
    wg_f32x4 sum = zero;
    const int groups = lpt >= 8 ? lpt >> 3 : 1;
    for (int g = 0; g < groups; ++g) {
        const int k0 = q * lpt + g * 8;
        wg_f32x4 v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = (live && i < lpt && k0 + i < msplit) ? src[(size_t)(k0 + i) * nk4] : zero;
        // the levels of the tree inside this group (only those the subtree really has: no additions of padding beyond P)
        if (lpt >= 2) {
#pragma unroll
            for (int i = 0; i < 8; i += 2) v[i] = wg_add4(v[i], v[i + 1]);
        }
        if (lpt >= 4) {
            v[0] = wg_add4(v[0], v[2]);
            v[4] = wg_add4(v[4], v[6]);
        }
        if (lpt >= 8) v[0] = wg_add4(v[0], v[4]);
        // groups combine like a binary counter: group g closes every level whose bit of g is set (left + right)
        wg_f32x4 c = v[0];
        bool placed = false;
#pragma unroll
        for (int j = 0; j < kLevels; ++j) {
            if (!placed) {
                if ((g >> j) & 1) {
                    c = wg_add4(st[j], c);
                } else {
                    st[j] = c;
                    placed = true;
                }
            }
        }
        sum = c;   // (after the last group -- all ones -- c has climbed to the subtree's root)
    }
    a[threadIdx.x] = sum;
    __syncthreads();
    for (int s = 1; s < T; s <<= 1) {
        if ((q & (2 * s - 1)) == 0) a[threadIdx.x] = wg_add4(a[threadIdx.x], a[(q + s) * E + e]);
        __syncthreads();
    }
    if (q == 0 && live) reinterpret_cast<wg_f32x4 *>(dw)[elem] = a[e];
}

__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float *__restrict__ part, int msplit, int pow2, int tlog,
                                                           size_t nk4, float *__restrict__ dw) {
    wgrad_reduce_body(part, msplit, pow2, tlog, nk4, dw, blockIdx.x);
}

// the reductions of a batched weight-gradient launch (rqhip_linear_wgrad_f16_batch) as ONE launch: same tree per job
struct WgradReduceJobs {
    const float *part[4];
    float *dw[4];
    size_t nk4[4];
    int tlog[4];
    unsigned first[4];
    int n, msplit, pow2;
};
__global__ __launch_bounds__(256) void wgrad_reduce_jobs_kernel(const WgradReduceJobs jobs) {
    const float *part = jobs.part[0];
    float *dw = jobs.dw[0];
    size_t nk4 = jobs.nk4[0];
    int tlog = jobs.tlog[0];
    unsigned first = 0u;
#pragma unroll
    for (int j = 1; j < 4; ++j)
        if (j < jobs.n && blockIdx.x >= jobs.first[j]) {
            part = jobs.part[j]; dw = jobs.dw[j]; nk4 = jobs.nk4[j]; tlog = jobs.tlog[j]; first = jobs.first[j];
        }
    wgrad_reduce_body(part, jobs.msplit, jobs.pow2, tlog, nk4, dw, blockIdx.x - first);
}

// log2 of the subtrees per element: enough threads for a streaming launch, never more subtrees than leaves
static int wgrad_reduce_tlog(size_t nk4, int pow2) {
    int tlog = 2;
    while (tlog < 6 && (nk4 << tlog) < (size_t)256 * 1024) tlog += 2;
    while ((1 << tlog) > pow2) --tlog;
    return tlog;
}

struct WgradPlan {
    int cfg;            // 0: 256x256, 1: 128x256, 2: 256x128, 3: 32x128, 4: 128x32; -1: unsupported
    int Nt, Kt, mc;     // block of dW per workgroup, rows per LDS stage
    int nslab_n, nslab_k, msplit, pow2;
    size_t lds;
};

static WgradPlan wgrad_plan(long long M, int N, int K) {
    WgradPlan pl;
    pl.cfg = -1;
    if (N % 256 == 0 && K % 256 == 0) { pl.cfg = 0; pl.Nt = 256; pl.Kt = 256; pl.mc = 32; }
    else if (N % 128 == 0 && K % 256 == 0) { pl.cfg = 1; pl.Nt = 128; pl.Kt = 256; pl.mc = 32; }
    else if (N % 256 == 0 && K % 128 == 0) { pl.cfg = 2; pl.Nt = 256; pl.Kt = 128; pl.mc = 32; }
    else if (N % 32 == 0 && K % 128 == 0) { pl.cfg = 3; pl.Nt = 32; pl.Kt = 128; pl.mc = 64; }
    else if (N % 128 == 0 && K % 32 == 0) { pl.cfg = 4; pl.Nt = 128; pl.Kt = 32; pl.mc = 64; }
    if (pl.cfg < 0) return pl;
    pl.nslab_n = N / pl.Nt;
    pl.nslab_k = K / pl.Kt;
    const long long chunks = (M + kWgChunk - 1) / kWgChunk;
    const int slabs = pl.nslab_n * pl.nslab_k;
    long long ms = cu_count() / slabs;      // one workgroup per CU (the LDS stage is 40-128 KiB)
    if (ms < 1) ms = 1;
    // A multiple of 8 row ranges whenever there are that many: the kernels then put the slabs of one row range on ONE XCD
    // (`(msplit & 7) == 0` branch), where they share the range's rows in that L2.  With 6 slabs on 256 CUs (dW [512, 768]) the
    // count was 42: the slabs of a range ran on six different XCDs and every one fetched its operand strips from beyond the L2 --
    // PMC 2 * FETCH_SIZE + WRITE_SIZE = 1 266 MB per launch against 512 MB algorithmic (2.5 x), 5.1 TB/s: the kernel was running
    // at the fabric's limit on re-reads (profiles/r04_pmc_traffic_c2.json).  40 ranges leave 16 of 256 CUs idle and win.
    if (ms >= 8) ms &= ~7LL;
    // up to 128 rows: ONE range (no partial blocks, no reduction launch -- at batch 64 the eight reductions were a fifth of the step's
    // launches).  (Capping larger batches at >= 128 rows per range was measured slower: batch 640, hipGraph step 0.358 -> 0.406 ms --
    // the kernel is latency-bound per 16-row stage there and the longer ranges cost more than the smaller reduction saves.)
    if (chunks <= 4) ms = 1;
    if (ms > chunks) ms = chunks > 0 ? chunks : 1;
    pl.msplit = (int)ms;
    pl.pow2 = 1;
    while (pl.pow2 < pl.msplit) pl.pow2 <<= 1;
    pl.lds = (size_t)2 * pl.mc * (pl.Nt + pl.Kt) * sizeof(float);
    return pl;
}

int wgrad_split_cfg(int N, int K);   // wgrad_split.hip
int launch_wgrad_split_jobs(int n, const float *const *g, const float *const *x, long long M, const int *N, const int *K, float *const *out,
                            int msplit, const unsigned *const *g_max, const unsigned *const *x_max, int half, hipStream_t s);
int launch_wgrad_split(int cfg, const float *g, const float *y, const float *x, long long M, int N, int K, float *gm, float *out,
                       int nslab_n, int nslab_k, int msplit, const unsigned *g_max, const unsigned *x_max, hipStream_t s);

}  // namespace rqhip

using namespace rqhip;

extern "C" size_t rqhip_linear_wgrad_workspace_bytes(int64_t M, int N, int K) {
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    const WgradPlan pl = wgrad_plan(M, N, K);
    if (pl.cfg < 0 || pl.msplit <= 1) return 16;
    return (size_t)pl.msplit * N * K * sizeof(float);
}

extern "C" int rqhip_linear_wgrad_supported(int N, int K) { return wgrad_plan(1 << 20, N, K).cfg >= 0 ? 1 : 0; }

extern "C" int rqhip_linear_wgrad_plan(int64_t M, int N, int K, int *msplit) {
    const WgradPlan pl = wgrad_plan(M, N, K);
    if (msplit) *msplit = pl.cfg < 0 ? 0 : pl.msplit;
    return pl.cfg;
}

template <int TA, int TB, int WA, int WB, int MC>
static int wgrad_launch(const WgradParams &p, const WgradPlan &pl, bool mask, hipStream_t s) {
    auto go = [&](auto kern) -> int {
        static LdsGrant grant;
        RQ_RETURN_IF_HIP(grant.ensure(reinterpret_cast<const void *>(kern), 160 * 1024));
        hipLaunchKernelGGL(kern, dim3(pl.nslab_n * pl.nslab_k * pl.msplit), dim3(64 * WA * WB), pl.lds, s, p);
        RQ_CHECK_LAUNCH("wgrad_kernel");
        return 0;
    };
    return mask ? go(wgrad_kernel<TA, TB, WA, WB, true, MC>) : go(wgrad_kernel<TA, TB, WA, WB, false, MC>);
}

extern "C" int rqhip_linear_wgrad(const float *g, const float *y, const float *x, int64_t M, int N, int K,
                                  float *g_masked, float *dW, void *workspace, size_t workspace_bytes,
                                  rqhip_stream_t stream) {
    return rqhip_linear_wgrad_ex(g, y, x, M, N, K, g_masked, dW, workspace, workspace_bytes, 0u, stream);
}

static int linear_wgrad_impl(const float *g, const float *y, const float *x, int64_t M, int N, int K, float *g_masked, float *dW,
                             void *workspace, size_t workspace_bytes, unsigned flags, const unsigned *g_col_max,
                             const unsigned *x_col_max, rqhip_stream_t stream);

extern "C" int rqhip_linear_wgrad_ex(const float *g, const float *y, const float *x, int64_t M, int N, int K,
                                     float *g_masked, float *dW, void *workspace, size_t workspace_bytes,
                                     unsigned flags, rqhip_stream_t stream) {
    return linear_wgrad_impl(g, y, x, M, N, K, g_masked, dW, workspace, workspace_bytes, flags, nullptr, nullptr, stream);
}

extern "C" int rqhip_linear_wgrad_f16(const float *g, const float *y, const float *x, int64_t M, int N, int K,
                                      const unsigned *g_col_max, const unsigned *x_col_max, float *g_masked, float *dW,
                                      void *workspace, size_t workspace_bytes, rqhip_stream_t stream) {
    if (M > 0 && (!g_col_max || !x_col_max)) {
        set_error("linear_wgrad_f16: the column maxima of g and x are required (rqhip_maxima or an epilogue's c_col_max)");
        return RQHIP_EARG;
    }
    return linear_wgrad_impl(g, y, x, M, N, K, g_masked, dW, workspace, workspace_bytes, 0u, g_col_max, x_col_max, stream);
}

static int linear_wgrad_impl(const float *g, const float *y, const float *x, int64_t M, int N, int K, float *g_masked, float *dW,
                             void *workspace, size_t workspace_bytes, unsigned flags, const unsigned *g_col_max,
                             const unsigned *x_col_max, rqhip_stream_t stream) {
    if (flags & ~RQHIP_WGRAD_FP32) {
        set_error("linear_wgrad: unknown flags 0x%x", flags);
        return RQHIP_EARG;
    }
    if (M < 0 || N <= 0 || K <= 0 || !dW || (M > 0 && (!g || !x))) {
        set_error("linear_wgrad: null pointer or bad size");
        return RQHIP_EARG;
    }
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (M == 0) {
        if (int rc = fill_words(dW, 0u, (size_t)N * K * sizeof(float), s)) return rc;
        return RQHIP_OK;
    }
    const WgradPlan pl = wgrad_plan(M, N, K);
    if (pl.cfg < 0) {
        set_error("linear_wgrad: unsupported layer shape N=%d K=%d (see rqhip_linear_wgrad_supported)", N, K);
        return RQHIP_EUNSUPPORTED;
    }
    auto al16 = [](const void *q) { return (reinterpret_cast<uintptr_t>(q) & 15u) == 0; };
    if (!al16(g) || !al16(y) || !al16(x) || !al16(g_masked) || !al16(dW) || !al16(workspace)) {
        set_error("linear_wgrad: pointers must be 16-byte aligned");
        return RQHIP_EARG;
    }
    if (pl.msplit > 1 && (!workspace || workspace_bytes < rqhip_linear_wgrad_workspace_bytes(M, N, K))) {
        set_error("linear_wgrad: workspace too small");
        return RQHIP_EWORKSPACE;
    }
    WgradParams p;
    p.g = g; p.y = y; p.x = x; p.gm = g_masked;
    p.out = pl.msplit > 1 ? reinterpret_cast<float *>(workspace) : dW;
    p.M = M; p.N = N; p.K = K;
    p.nslab_n = pl.nslab_n; p.nslab_k = pl.nslab_k; p.msplit = pl.msplit;
    p.n_chunks = (M + kWgChunk - 1) / kWgChunk;
    p.pow2 = pl.pow2;
    const bool mask = y != nullptr;
    int rc = 0;
    // large layers: the six-term bf16-split kernel (wgrad_split.hip) unless the oracle-exact fp32 kernel is asked for
    const int scfg = (flags & RQHIP_WGRAD_FP32) ? -1 : wgrad_split_cfg(N, K);
    // (bench only: one profile record for the kernel and its partial-sum reduction; algorithmic work 2 M N K FLOP, bytes: g, x
    // (and y) once, g_pre once when it is written back)
    profile_begin(s, RQHIP_PROF_WGRAD, 2.0 * (double)M * N * K, 4.0 * (double)M * (N * (1 + (y ? 1 : 0) + (g_masked && y ? 1 : 0)) + K));
    if (scfg >= 0 && scfg == pl.cfg) {
        rc = launch_wgrad_split(scfg, g, y, x, M, N, K, g_masked, p.out, pl.nslab_n, pl.nslab_k, pl.msplit, g_col_max, x_col_max, s);
    } else
    switch (pl.cfg) {
        case 0: rc = wgrad_launch<4, 2, 2, 4, 32>(p, pl, mask, s); break;
        case 1: rc = wgrad_launch<4, 1, 1, 8, 32>(p, pl, mask, s); break;
        case 2: rc = wgrad_launch<2, 2, 4, 2, 32>(p, pl, mask, s); break;
        case 3: rc = wgrad_launch<1, 1, 1, 4, 64>(p, pl, mask, s); break;
        default: rc = wgrad_launch<1, 1, 4, 1, 64>(p, pl, mask, s); break;
    }
    if (rc) { profile_end(s); return rc; }
    if (pl.msplit > 1) {
        const size_t nk4 = (size_t)N * K / 4;
        const int tlog = wgrad_reduce_tlog(nk4, pl.pow2);
        const size_t per_wg = (size_t)256 >> tlog;
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((nk4 + per_wg - 1) / per_wg)), dim3(256), 0, s,
                           reinterpret_cast<const float *>(workspace), pl.msplit, pl.pow2, tlog, nk4, dW);
        RQ_CHECK_LAUNCH("wgrad_reduce_kernel");
    }
    profile_end(s);
    return RQHIP_OK;
}

// ---- several layers' weight gradients in one launch (rqhip_linear_wgrad_f16_batch; csrc/wgrad_split.hip:wgrad_split_jobs_kernel) ----
// The common number of row ranges of a batch: the CU count over the tiles of all jobs, a multiple of 8 when there are that many (the
// kernel then keeps the slabs of one range on one XCD), never more than the 32-row granules of the batch.  0: not batchable (a job
// that is not tiled 256 x 256, fewer than two or more than four jobs, a batch of at most 128 rows, more tiles than CUs).
// A batch is either all 256 x 256-tiled (wgrad_split_cfg 0) or all half-size tiles (cfg 1 / 2, freely mixed): *half says which.
static int wgrad_batch_ranges(long long M, const int *N, const int *K, int n, int *half = nullptr) {
    if (n < 2 || n > 4 || M <= 128 || !N || !K) return 0;
    long long tiles = 0;
    int n_full = 0;
    for (int j = 0; j < n; ++j) {
        if (N[j] <= 0 || K[j] <= 0) return 0;
        const int cfg = wgrad_split_cfg(N[j], K[j]);
        if (cfg < 0) return 0;
        n_full += cfg == 0;
        tiles += (long long)(N[j] / (cfg == 1 ? 128 : 256)) * (K[j] / (cfg == 2 ? 128 : 256));
    }
    if (n_full != 0 && n_full != n) return 0;
    if (half) *half = n_full == 0;
    long long ms = cu_count() / tiles;
    if (ms < 1) return 0;
    if (ms >= 8) ms &= ~7LL;
    const long long chunks = (M + kWgChunk - 1) / kWgChunk;
    if (ms > chunks) ms = chunks;
    return (int)ms;
}

extern "C" int rqhip_linear_wgrad_f16_batch_plan(int64_t M, const int *N, const int *K, int n) { return wgrad_batch_ranges(M, N, K, n); }

extern "C" size_t rqhip_linear_wgrad_f16_batch_workspace_bytes(int64_t M, const int *N, const int *K, int n) {
    const int ms = wgrad_batch_ranges(M, N, K, n);
    if (ms <= 1) return 16;
    size_t total = 0;
    for (int j = 0; j < n; ++j) total += (size_t)ms * N[j] * K[j] * sizeof(float);
    return total;
}

extern "C" int rqhip_linear_wgrad_f16_batch(const rqhip_wgrad_job *jobs, int n, int64_t M, void *workspace, size_t workspace_bytes,
                                            rqhip_stream_t stream) {
    if (!jobs || n < 2 || n > 4) {
        set_error("linear_wgrad_f16_batch: 2 to 4 jobs (got %d)", n);
        return RQHIP_EARG;
    }
    int N[4], K[4];
    const float *g[4], *x[4];
    const unsigned *gm[4], *xm[4];
    float *out[4];
    auto al16 = [](const void *q) { return (reinterpret_cast<uintptr_t>(q) & 15u) == 0; };
    for (int j = 0; j < n; ++j) {
        const rqhip_wgrad_job &b = jobs[j];
        if (!b.g || !b.x || !b.dW || !b.g_col_max || !b.x_col_max || !al16(b.g) || !al16(b.x) || !al16(b.dW)) {
            set_error("linear_wgrad_f16_batch: job %d: null or misaligned pointer (g, x, dW 16-byte aligned; both column maxima)", j);
            return RQHIP_EARG;
        }
        N[j] = b.N; K[j] = b.K; g[j] = b.g; x[j] = b.x; gm[j] = b.g_col_max; xm[j] = b.x_col_max;
    }
    int half = 0;
    const int ms = wgrad_batch_ranges(M, N, K, n, &half);
    if (ms < 1) {
        set_error("linear_wgrad_f16_batch: not batchable (every dW tiled 256 x 256, or every dW tiled 128 x 256 / 256 x 128; more than 128 rows; "
                  "tiles <= CUs: rqhip_linear_wgrad_f16_batch_plan)");
        return RQHIP_EUNSUPPORTED;
    }
    if (ms > 1 && (!workspace || !al16(workspace) || workspace_bytes < rqhip_linear_wgrad_f16_batch_workspace_bytes(M, N, K, n))) {
        set_error("linear_wgrad_f16_batch: workspace too small (rqhip_linear_wgrad_f16_batch_workspace_bytes)");
        return RQHIP_EWORKSPACE;
    }
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    float *part = reinterpret_cast<float *>(workspace);
    double flops = 0.0, bytes = 0.0;
    for (int j = 0; j < n; ++j) {
        out[j] = ms > 1 ? part : jobs[j].dW;
        part += (size_t)ms * N[j] * K[j];
        flops += 2.0 * (double)M * N[j] * K[j];
        bytes += 4.0 * (double)M * (N[j] + K[j]);
    }
    profile_begin(s, RQHIP_PROF_WGRAD, flops, bytes);
    int rc = launch_wgrad_split_jobs(n, g, x, M, N, K, out, ms, gm, xm, half, s);
    if (rc) { profile_end(s); return rc; }
    if (ms > 1) {
        int pow2 = 1;
        while (pow2 < ms) pow2 <<= 1;
        WgradReduceJobs rj;                 // the same balanced tree over a job's partial blocks as rqhip_linear_wgrad_f16's, all jobs in one launch
        rj.n = n; rj.msplit = ms; rj.pow2 = pow2;
        unsigned blocks = 0;
        for (int j = 0; j < 4; ++j) {
            const int i = j < n ? j : 0;
            const size_t nk4 = (size_t)N[i] * K[i] / 4;
            rj.part[j] = out[i]; rj.dw[j] = jobs[i].dW; rj.nk4[j] = nk4; rj.tlog[j] = wgrad_reduce_tlog(nk4, pow2);
            rj.first[j] = blocks;
            const size_t per_wg = (size_t)256 >> rj.tlog[j];
            if (j < n) blocks += (unsigned)((nk4 + per_wg - 1) / per_wg);
        }
        hipLaunchKernelGGL(wgrad_reduce_jobs_kernel, dim3(blocks), dim3(256), 0, s, rj);
        RQ_CHECK_LAUNCH("wgrad_reduce_jobs_kernel");
    }
    profile_end(s);
    return RQHIP_OK;
}
