// wgrad_jobs.hip -- the weight gradients of a whole Linear stack at the batch sizes the reference trains with, ONE launch (gfx950).
// SURVEY.md section 8 row f2; reference modules/encoder.py:25-38 (autograd of relu(x W^T) for every layer of the MLP),
// configs/rqvae_amazon.gin (batch 640), configs/rqvae_ml32m.gin (batch 64).
//
// dW_i[n,k] = sum_m g_i[m,n] x_i[m,k] for the layers i of one MLP, g_i already masked by the layer's ReLU.  At M = 640 the kernels
// of wgrad_split.hip / wgrad.hip are all latency: eight launches of 18-20 us plus eight reductions of 5 us per training step --
// 16 row ranges per 256 x 256 block so that 96 workgroups exist at all, 25 MB of partial blocks for 0.5 GFLOP -- 190 of the
// step's 410 us of kernel time (profiles/r05_small_batch_kernels.txt).  Here:
//   * a JOB TABLE: every (layer, 64 x 64 block of dW) of the stack is one workgroup of the same launch -- 138 workgroups for
//     768-512-256-128-32; a block is reduced over ALL M rows by its workgroup: no row ranges, no partial blocks, no second kernel,
//     a fixed summation order;
//   * a workgroup is 16 waves: thread (column c of the block's g / x columns, row octet o) fetches EIGHT CONSECUTIVE ROWS of its
//     column for the 64-row stage two ahead (eight coalesced dword loads), splits them into the three bf16 pieces once for the
//     whole workgroup and writes each piece as one 16-byte LDS element [piece][octet][column] -- exactly a lane's matrix operand;
//     wave (k, s) then multiplies K step k of the stage for sub-block s (32 x 32) with six ds_read_b128 and six matrix
//     instructions; one barrier per 64 rows; the four K-step partial sums of a sub-block meet in LDS at the end (fixed order).
//     Per stage and SIMD that is ~800 cycles of VALU (the split) beside ~770 of matrix time.  (A first version kept raw fp32 rows in
//     LDS, 4 waves, every wave splitting its own operands: 22.9 us per launch at M = 640, one wave per SIMD and all latency.)
//   * arithmetic as wgrad_split.hip's three-piece path: v = h + m + l exactly in bf16, the six piece products that matter on
//     v_mfma_f32_32x32x16_bf16, fp32 accumulation (dropped terms <= 2^-23 of a product) -- no column maxima needed, which at these
//     sizes would be one more launch per layer.  An entry's error is within (sqrt(M) + 8) 2^-24 of the sum of its terms' magnitudes
//     and about the library fp32 GEMM's against fp64 (tests/test_gpu_wgrad.py).
// Layers 32 wide (the latent side) take 32 x 64 / 64 x 32 blocks of the same code (two of the four wave columns multiply).
#include "rqhip_common.h"

namespace rqhip {

typedef float wj_f32x16 __attribute__((ext_vector_type(16)));
typedef float wj_f32x4 __attribute__((ext_vector_type(4)));
typedef float wj_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 wj_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 wj_bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned wj_u32x4 __attribute__((ext_vector_type(4)));

constexpr int kWjMaxJobs = 8;
constexpr int kWjRows = 64;          // rows per LDS stage: four K steps of the matrix instruction, one per wave group
constexpr int kWjThreads = 1024;     // 16 waves: 4 K steps x 4 sub-blocks
constexpr int kWjMaxCols = 128;      // staged columns: the block's columns of g + of x (at most 64 + 64)
constexpr int kWjItems = (kWjMaxCols * (kWjRows / 8) + kWjThreads - 1) / kWjThreads;   // (column, octet) items per thread: 1
constexpr size_t kWjLds = (size_t)2 * 3 * (kWjRows / 8) * kWjMaxCols * 16;             // [buffer][piece][octet][column] x 16 B = 96 KB

struct WgradJob {
    const float *g, *x;
    float *dw;
    int N, K;
    int first;      // first workgroup of this job
    int blocks_k;   // blocks along K
    int shape;      // bit 0: 64 (else 32) rows of dW per block, bit 1: 64 (else 32) columns -- sub-blocks of 32 x 32, one per wave
};
struct WgradJobs {
    WgradJob job[kWjMaxJobs];
    int n_jobs;
    long long M;
};

// (a, b) -> the packed bf16 pieces {piece(a), piece(b)} of a = h + m + l (exact), likewise b  (as wgrad_split.hip:ws_split2)
__device__ __forceinline__ void wj_split2(float a, float b, unsigned &h, unsigned &m, unsigned &l) {
    const wj_bf16x2 hh = __builtin_convertvector(wj_f32x2{a, b}, wj_bf16x2);
    h = __builtin_bit_cast(unsigned, hh);
    const float ra = a - __builtin_bit_cast(float, h << 16), rb = b - __builtin_bit_cast(float, h & 0xffff0000u);
    const wj_bf16x2 mm = __builtin_convertvector(wj_f32x2{ra, rb}, wj_bf16x2);
    m = __builtin_bit_cast(unsigned, mm);
    const float sa = ra - __builtin_bit_cast(float, m << 16), sb = rb - __builtin_bit_cast(float, m & 0xffff0000u);
    const wj_bf16x2 ll = __builtin_convertvector(wj_f32x2{sa, sb}, wj_bf16x2);
    l = __builtin_bit_cast(unsigned, ll);
}

// eight consecutive rows of one column -> the three bf16x8 operands
__device__ __forceinline__ void wj_split8(const float (&v)[8], wj_bf16x8 (&pc)[3]) {
    wj_u32x4 h, m, l;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        unsigned hj, mj, lj;
        wj_split2(v[2 * j], v[2 * j + 1], hj, mj, lj);
        h[j] = hj; m[j] = mj; l[j] = lj;
    }
    pc[0] = __builtin_bit_cast(wj_bf16x8, h);
    pc[1] = __builtin_bit_cast(wj_bf16x8, m);
    pc[2] = __builtin_bit_cast(wj_bf16x8, l);
}

// The pipelined loads of the kernel are issued and awaited BY HAND.  The compiler's own s_waitcnt placement waited for every load
// in flight before each split (vmcnt(0): its count of the younger register set's loads does not survive the loop's back edge), which
// makes the two-stage prefetch no prefetch.  An asm load is invisible to that pass; wj_await<N> is the wait -- "at most N loads still
// in flight", i.e. everything but the N youngest has arrived -- and ties the eight registers to itself so that no use moves above it.
__device__ __forceinline__ float wj_load(const float *p) {
    float v;
    asm volatile("global_load_dword %0, %1, off" : "=v"(v) : "v"(p));
    return v;
}
template <int N>
__device__ __forceinline__ void wj_await(float (&r)[8]) {
    asm volatile("s_waitcnt vmcnt(%8)"
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7])
                 : "n"(N)
                 : "memory");
}

__global__ __launch_bounds__(kWjThreads) void wgrad_jobs_kernel(const WgradJobs p) {
    extern __shared__ __attribute__((aligned(16))) char wj_smem[];
    wj_bf16x8 *planes = reinterpret_cast<wj_bf16x8 *>(wj_smem);   // [2][3][8][cols]

    // this workgroup's job: the table is by value (scalar registers); fields are picked with constant indices
    int ji = 0;
#pragma unroll
    for (int i = 1; i < kWjMaxJobs; ++i)
        if (i < p.n_jobs && (int)blockIdx.x >= p.job[i].first) ji = i;
    const float *g = p.job[0].g, *x = p.job[0].x;
    float *dw = p.job[0].dw;
    int N = p.job[0].N, K = p.job[0].K, first = p.job[0].first, blocks_k = p.job[0].blocks_k, shape = p.job[0].shape;
#pragma unroll
    for (int i = 1; i < kWjMaxJobs; ++i)
        if (ji == i) {
            g = p.job[i].g; x = p.job[i].x; dw = p.job[i].dw;
            N = p.job[i].N; K = p.job[i].K; first = p.job[i].first; blocks_k = p.job[i].blocks_k; shape = p.job[i].shape;
        }
    const int local = (int)blockIdx.x - first;
    const int bn = local / blocks_k, bk = local % blocks_k;
    const int TN = (shape & 1) ? 64 : 32, TK = (shape & 2) ? 64 : 32;
    const int WA = TN / 32, WB = TK / 32;         // sub-blocks (32 x 32) along N and K
    const int n0 = bn * TN, k0 = bk * TK;
    const int cols = TN + TK;                     // staged columns: g's, then x's
    const int buf_elems = 3 * (kWjRows / 8) * cols;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int il = lane & 31, h = lane >> 5;
    const int kgroup = wave >> 2, sub = wave & 3;   // K step of every stage / 32 x 32 sub-block this wave multiplies
    const int wa = sub / WB, wb = sub % WB;
    const bool multiplies = sub < WA * WB;          // (blocks of the 32-wide layers have fewer sub-blocks than wave columns)
    const long long M = p.M;

    // staging items: item e -> column e % cols, row octet e / cols.  Formed once: the pointer to the item's first row (advanced by
    // 64 rows per stage fetched -- stages are fetched in order), the first row (for the tail test), the LDS element.  Blocks with
    // fewer than 1024 items (the 32-wide layers) give the spare threads a second copy of the first items: they fetch, split and
    // store the same values to the same place -- no thread-dependent branch around the split, which would make every refill of the
    // registers wait for all loads in flight.
    const float *src[kWjItems];
    unsigned ld[kWjItems];
    int row[kWjItems], lds_at[kWjItems];
#pragma unroll
    for (int i = 0; i < kWjItems; ++i) {
        const int e = (tid + kWjThreads * i) % (cols * (kWjRows / 8));
        const int c = e % cols, o = e / cols;
        const bool isg = c < TN;
        row[i] = 8 * o;
        lds_at[i] = o * cols + c;
        ld[i] = (unsigned)(isg ? N : K);
        src[i] = isg ? g + (size_t)(8 * o) * N + n0 + c : x + (size_t)(8 * o) * K + k0 + (c - TN);
    }
    const float *src_tail = src[0] + (size_t)(M / kWjRows) * kWjRows * ld[0];   // the item's first row of the partial stage
    // Whole stages (every row below M) are fetched with plain loads in the pipelined loop; the last, partial stage is fetched ONCE,
    // under row tests, and multiplied after the loop.  (With the row test on every load of the loop each load sat in its
    // own branch and the compiler made every split wait for ALL outstanding loads: the two-stage prefetch was none.)  
    const int n_full = (int)(M / kWjRows);
    const bool has_tail = (M % kWjRows) != 0;
    static_assert(kWjItems == 1, "the hand-placed waits count eight loads per register set");
    auto fetch = [&](float (&r)[kWjItems][8]) {          // compiler-managed loads: prologue of short jobs, last stages
#pragma unroll
        for (int j = 0; j < 8; ++j) r[0][j] = src[0][(size_t)j * ld[0]];
        src[0] += (size_t)kWjRows * ld[0];
    };
    auto fetch_by_hand = [&](float (&r)[kWjItems][8]) {  // awaited with wj_await (only in straight-line code, see there)
#pragma unroll
        for (int j = 0; j < 8; ++j) r[0][j] = wj_load(src[0] + (size_t)j * ld[0]);
        src[0] += (size_t)kWjRows * ld[0];
    };
    auto stash = [&](int buf, const float (&r)[kWjItems][8]) {
#pragma unroll
        for (int i = 0; i < kWjItems; ++i) {
            wj_bf16x8 pc[3];
            wj_split8(r[i], pc);
#pragma unroll
            for (int q = 0; q < 3; ++q) planes[buf * buf_elems + q * (kWjRows / 8) * cols + lds_at[i]] = pc[q];
        }
    };

    wj_f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    const int a_at = (2 * kgroup + h) * cols + wa * 32 + il, b_at = (2 * kgroup + h) * cols + TN + wb * 32 + il;
    auto multiply = [&](int buf) {
        if (!multiplies) return;
        const wj_bf16x8 *pl = planes + buf * buf_elems;
        wj_bf16x8 a[3], b[3];
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            a[q] = pl[q * (kWjRows / 8) * cols + a_at];
            b[q] = pl[q * (kWjRows / 8) * cols + b_at];
        }
        // smallest products first (their sum is formed before it meets the large ones)
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], acc, 0, 0, 0);   // m m
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], acc, 0, 0, 0);   // l h
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], acc, 0, 0, 0);   // h l
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], acc, 0, 0, 0);   // m h
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], acc, 0, 0, 0);   // h m
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], acc, 0, 0, 0);   // h h
    };

    // Pipeline: stage c is multiplied out of LDS buffer c & 1 while stage c + 1 is split from registers into the other buffer and
    // the loads of stages c + 2 / c + 3 are in flight in the two register sets (a load has two barrier intervals to arrive).
    float ra[kWjItems][8], rb[kWjItems][8];
    int c = 0;
    if (n_full >= 5) {
        // Steady state, loads and waits by hand: both fetches of a pair of stages exist, so at every split exactly the other set's
        // eight loads are younger.  No branch between a load and its wait: a value that changed registers on the way (a copy on an
        // edge) would be copied before it has arrived.
        fetch_by_hand(ra);               // stage 0
        fetch_by_hand(rb);               // stage 1
        wj_await<8>(ra[0]);
        stash(0, ra);
        fetch_by_hand(ra);               // stage 2
        __syncthreads();
        for (; c + 4 < n_full; c += 2) {
            multiply(0);
            wj_await<8>(rb[0]);
            stash(1, rb);
            fetch_by_hand(rb);           // stage c + 3
            __syncthreads();
            multiply(1);
            wj_await<8>(ra[0]);
            stash(0, ra);
            fetch_by_hand(ra);           // stage c + 4
            __syncthreads();
        }
        wj_await<0>(rb[0]);              // everything has arrived: from here on the values are ordinary
        wj_await<0>(ra[0]);
    } else {
        // a short job (at most four whole stages): no overlap, every fetch awaited where it is issued.  (By hand as well: with
        // compiler-managed loads on this path the compiler made the loads above wait for them -- same registers, other branch.)
        if (n_full > 0) {
            fetch_by_hand(ra);           // stage 0
            wj_await<0>(ra[0]);
            stash(0, ra);
        }
        if (n_full > 1) {
            fetch_by_hand(rb);           // stage 1
            wj_await<0>(rb[0]);
        }
        if (n_full > 2) {
            fetch_by_hand(ra);           // stage 2
            wj_await<0>(ra[0]);
        }
        __syncthreads();
    }
    // the partial stage's rows, under row tests, while the last whole stages are multiplied (compiler-managed loads: issued here,
    // after the hand-awaited ones, they do not make the compiler wait inside the pipelined part)
    float rt[kWjItems][8];
    if (has_tail) {
        const float *tsrc = src_tail;
#pragma unroll
        for (int j = 0; j < 8; ++j) rt[0][j] = ((long long)n_full * kWjRows + row[0] + j < M) ? tsrc[(size_t)j * ld[0]] : 0.0f;
    }
    for (; c < n_full; c += 2) {         // the last stages (all of a short job): the same steps, each under its test
        multiply(0);
        if (c + 1 < n_full) stash(1, rb);
        if (c + 3 < n_full) fetch(rb);   // stage c + 3
        __syncthreads();
        if (c + 1 >= n_full) break;
        multiply(1);
        if (c + 2 < n_full) stash(0, ra);
        if (c + 4 < n_full) fetch(ra);   // stage c + 4
        __syncthreads();
    }
    if (has_tail) {                      // (every wave is past the loop's last barrier: both buffers are free)
        stash(0, rt);
        __syncthreads();
        multiply(0);
        __syncthreads();
    }

    // the four K-step partial sums of a sub-block: ((k0 + k1) + k2) + k3, through LDS (every wave is past its last operand read)
    float *red = reinterpret_cast<float *>(wj_smem);             // [kgroup 1..3][sub][r][lane]
    if (kgroup > 0 && multiplies) {
#pragma unroll
        for (int r = 0; r < 16; ++r) red[(((kgroup - 1) * 4 + sub) * 16 + r) * 64 + lane] = acc[r];
    }
    __syncthreads();
    if (kgroup == 0 && multiplies) {
#pragma unroll
        for (int kg = 1; kg < 4; ++kg)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = acc[r] + red[(((kg - 1) * 4 + sub) * 16 + r) * 64 + lane];
        // acc[r]: n = n0 + 32 wa + 8 (r >> 2) + 4 h + (r & 3), k = k0 + 32 wb + il
        float *dst = dw + (size_t)(n0 + wa * 32 + 4 * h) * K + k0 + wb * 32 + il;
#pragma unroll
        for (int r = 0; r < 16; ++r) dst[(size_t)(8 * (r >> 2) + (r & 3)) * K] = acc[r];
    }
}

static int wgrad_jobs_shape(int N, int K) {
    if (N % 32 != 0 || K % 32 != 0) return -1;
    return (N % 64 == 0 ? 1 : 0) | (K % 64 == 0 ? 2 : 0);
}

}  // namespace rqhip

using namespace rqhip;

extern "C" int rqhip_linear_wgrad_jobs_supported(int N, int K) { return (N > 0 && K > 0 && wgrad_jobs_shape(N, K) >= 0) ? 1 : 0; }

extern "C" int rqhip_linear_wgrad_jobs(const float *const *g, const float *const *x, float *const *dW, const int *N, const int *K,
                                       int n_jobs, int64_t M, rqhip_stream_t stream) {
    if (n_jobs < 0 || n_jobs > kWjMaxJobs || M < 0 || (n_jobs > 0 && (!g || !x || !dW || !N || !K))) {
        set_error("linear_wgrad_jobs: 0 <= n_jobs <= %d and M >= 0 are required (got n_jobs=%d, M=%lld)", kWjMaxJobs, n_jobs,
                  (long long)M);
        return RQHIP_EARG;
    }
    if (n_jobs == 0) return RQHIP_OK;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    WgradJobs p;
    p.n_jobs = n_jobs;
    p.M = M;
    int blocks = 0;
    double flops = 0.0, bytes = 0.0;
    for (int i = 0; i < kWjMaxJobs; ++i) {
        WgradJob &j = p.job[i];
        if (i >= n_jobs) {
            j = p.job[0];
            j.first = 0x7fffffff;
            continue;
        }
        const int shape = (N[i] > 0 && K[i] > 0) ? wgrad_jobs_shape(N[i], K[i]) : -1;
        if (shape < 0 || !g[i] || !x[i] || !dW[i]) {
            set_error("linear_wgrad_jobs: job %d: dW [%d, %d] is not tiled (both dimensions multiples of 32) or a pointer is null", i, N[i],
                      K[i]);
            return RQHIP_EARG;
        }
        j.g = g[i]; j.x = x[i]; j.dw = dW[i];
        j.N = N[i]; j.K = K[i];
        j.shape = shape;
        const int TN = (shape & 1) ? 64 : 32, TK = (shape & 2) ? 64 : 32;
        j.blocks_k = K[i] / TK;
        j.first = blocks;
        blocks += (N[i] / TN) * j.blocks_k;
        flops += 2.0 * (double)M * N[i] * K[i];
        bytes += 4.0 * ((double)M * (N[i] + K[i]) + (double)N[i] * K[i]);
    }
    profile_begin(s, RQHIP_PROF_WGRAD, flops, bytes);
    static LdsGrant grant;
    RQ_RETURN_IF_HIP(grant.ensure(reinterpret_cast<const void *>(wgrad_jobs_kernel), (int)kWjLds));
    hipLaunchKernelGGL(wgrad_jobs_kernel, dim3((unsigned)blocks), dim3(kWjThreads), kWjLds, s, p);
    RQ_CHECK_LAUNCH("wgrad_jobs_kernel");
    profile_end(s);
    return RQHIP_OK;
}
