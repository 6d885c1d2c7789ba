// sid_match.hip -- integer tuple matching on semantic ids for the decoder-side consumers (gfx950).
//
//   prefix index : "does this h-tuple occur as the first h columns of some corpus row?"
//                  reference: EncoderDecoderRetrievalModel._check_valid_prefix (modules/model.py:169-182), which
//                  materialises an [N, P, h] equality tensor at EVERY beam step of generate() (model.py:349,364).
//   top-k match  : position of the first generated tuple equal to the target tuple
//                  reference: TopKAccumulator.accumulate (evaluate/metrics.py:16-25).
//
// Design.  The corpus is fixed for a whole evaluation, so the set of its prefixes is built ONCE: H open-addressing
// hash sets (one per prefix length), each slot holding the index of the first corpus row that inserted the prefix.
// Membership is decided by comparing the query with that row's ids, never by the hash alone, so answers are exact
// and independent of insertion order.  A lookup is one hash, ~1 probe and one h-tuple compare per query: P*h*8
// bytes of query traffic plus one random 4-byte slot read and one random row read per probe -- HBM/L2-latency
// bound integer work, nothing to put on the matrix cores.  The index costs 8 bytes per corpus row and prefix
// length (load factor <= 1/2), i.e. 320 MB for 10 M items x 4 levels: resident in HBM for the run.
#include "rqhip_common.h"

namespace rqhip {

__device__ __forceinline__ unsigned sid_mix(unsigned h, unsigned v) {
    h ^= v + 0x9e3779b9u + (h << 6) + (h >> 2);
    h *= 0x85ebca6bu;
    h ^= h >> 13;
    return h;
}

__device__ __forceinline__ unsigned sid_hash_step(unsigned h, int64_t v) {
    h = sid_mix(h, (unsigned)(unsigned long long)v);
    const unsigned hi = (unsigned)((unsigned long long)v >> 32);
    return hi ? sid_mix(h, hi ^ 0x5bd1e995u) : h;  // ids are small non-negative numbers: the high word is 0
}

__device__ __forceinline__ unsigned sid_hash_final(unsigned h) {
    h ^= h >> 16;
    h *= 0xc2b2ae35u;
    h ^= h >> 15;
    return h;
}

static unsigned long long slots_for(long long N) {
    unsigned long long p = 64;
    const unsigned long long want = 2ull * (unsigned long long)(N > 0 ? N : 1);
    while (p < want) p <<= 1;
    return p;
}

// one thread per corpus row; inserts its H prefixes (hash extended incrementally)
__global__ __launch_bounds__(256) void prefix_build_kernel(const int64_t *__restrict__ corpus, long long N, int H,
                                                           long long ld, int *__restrict__ table, unsigned mask) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const int64_t *mine = corpus + (size_t)i * ld;
    const size_t slots = (size_t)mask + 1;
    unsigned hsh = 0x9747b28cu;
    for (int h = 1; h <= H; ++h) {
        hsh = sid_hash_step(hsh, mine[h - 1]);
        int *tab = table + (size_t)(h - 1) * slots;
        unsigned s = sid_hash_final(hsh) & mask;
        for (;;) {
            // plain load first: short prefixes are shared by thousands of rows, which would otherwise all
            // serialise on the same atomic
            int prev = __builtin_nontemporal_load(&tab[s]);
            if (prev == -1) prev = atomicCAS(&tab[s], -1, (int)i);
            if (prev == -1) break;  // inserted
            const int64_t *other = corpus + (size_t)prev * ld;
            bool same = true;
            for (int c = 0; c < h && same; ++c) same = other[c] == mine[c];
            if (same) break;  // already present
            s = (s + 1) & mask;
        }
    }
}

// one thread per query prefix
__global__ __launch_bounds__(256) void prefix_lookup_kernel(const int *__restrict__ table, unsigned mask,
                                                            const int64_t *__restrict__ corpus, long long ld,
                                                            const int64_t *__restrict__ prefix, long long P, int h,
                                                            long long ldp, uint8_t *__restrict__ valid) {
    const long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= P) return;
    const int64_t *mine = prefix + (size_t)q * ldp;
    unsigned hsh = 0x9747b28cu;
    for (int c = 0; c < h; ++c) hsh = sid_hash_step(hsh, mine[c]);
    const int *tab = table + (size_t)(h - 1) * ((size_t)mask + 1);
    unsigned s = sid_hash_final(hsh) & mask;
    bool found = false;
    for (;;) {
        const int prev = tab[s];
        if (prev == -1) break;
        const int64_t *other = corpus + (size_t)prev * ld;
        bool same = true;
        for (int c = 0; c < h && same; ++c) same = other[c] == mine[c];
        if (same) {
            found = true;
            break;
        }
        s = (s + 1) & mask;
    }
    valid[q] = found ? 1 : 0;
}

__global__ void fill_u8_kernel(uint8_t *__restrict__ v, long long n, uint8_t x) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) v[i] = x;
}

// one thread per (row, candidate): consecutive threads read consecutive D-tuples of top_k (coalesced); the target
// row is shared by the K threads of a row (cache hits).  rank[row] was preset to -1 == UINT64_MAX: an unsigned
// atomic minimum leaves the FIRST matching position, or -1 when nothing matches.
__global__ __launch_bounds__(256) void topk_match_kernel(const int64_t *__restrict__ actual,
                                                         const int64_t *__restrict__ top_k, long long B, int K, int D,
                                                         unsigned long long *__restrict__ rank) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= B * K) return;
    const long long row = t / K;
    const int k = (int)(t - row * K);
    const int64_t *a = actual + (size_t)row * D;
    const int64_t *c = top_k + (size_t)t * D;
    bool same = true;
    for (int d = 0; d < D && same; ++d) same = a[d] == c[d];
    if (same) atomicMin(&rank[row], (unsigned long long)k);
}

}  // namespace rqhip

using namespace rqhip;

extern "C" size_t rqhip_prefix_index_bytes(int64_t N, int H) {
    if (N < 0 || H < 1) return 0;
    return (size_t)slots_for(N) * (size_t)H * sizeof(int);
}

static int check_corpus(const char *who, const int64_t *corpus, int64_t N, int H, int64_t ld) {
    if (N < 0 || H < 1 || H > RQHIP_MAX_PREFIX_LEN || ld < H || (N > 0 && !corpus)) {
        set_error("%s: bad corpus (N=%lld, H=%d, ld=%lld; 1 <= H <= %d, ld >= H)", who, (long long)N, H,
                  (long long)ld, RQHIP_MAX_PREFIX_LEN);
        return RQHIP_EARG;
    }
    if (N >= (1ll << 30)) {
        set_error("%s: N=%lld exceeds the 2^30 rows this implementation indexes with 32 bits", who, (long long)N);
        return RQHIP_EUNSUPPORTED;
    }
    return RQHIP_OK;
}

extern "C" int rqhip_prefix_index_build(const int64_t *corpus, int64_t N, int H, int64_t ld, void *index,
                                        size_t index_bytes, rqhip_stream_t stream) {
    if (int rc = check_corpus("prefix_index_build", corpus, N, H, ld)) return rc;
    const size_t need = rqhip_prefix_index_bytes(N, H);
    if (!index || index_bytes < need) {
        set_error("prefix_index_build: index buffer too small (%zu < %zu)", index_bytes, need);
        return RQHIP_EWORKSPACE;
    }
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (int rc = fill_words(index, 0xffffffffu, need, s)) return rc;
    if (N == 0) return RQHIP_OK;
    const int tb = 256;
    hipLaunchKernelGGL(prefix_build_kernel, dim3((unsigned)((N + tb - 1) / tb)), dim3(tb), 0, s, corpus, (long long)N,
                       H, (long long)ld, reinterpret_cast<int *>(index), (unsigned)(slots_for(N) - 1));
    RQ_CHECK_LAUNCH("prefix_build_kernel");
    return RQHIP_OK;
}

extern "C" int rqhip_prefix_lookup(const void *index, size_t index_bytes, const int64_t *corpus, int64_t N, int H,
                                   int64_t ld, const int64_t *prefix, int64_t P, int h, int64_t ldp, uint8_t *valid,
                                   rqhip_stream_t stream) {
    if (int rc = check_corpus("prefix_lookup", corpus, N, H, ld)) return rc;
    if (P < 0 || h < 0 || h > H || ldp < h || (P > 0 && (!valid || (h > 0 && !prefix)))) {
        set_error("prefix_lookup: bad query (P=%lld, h=%d, ldp=%lld; 0 <= h <= H=%d, ldp >= h)", (long long)P, h,
                  (long long)ldp, H);
        return RQHIP_EARG;
    }
    if (!index || index_bytes < rqhip_prefix_index_bytes(N, H)) {
        set_error("prefix_lookup: index buffer too small for N=%lld, H=%d", (long long)N, H);
        return RQHIP_EWORKSPACE;
    }
    if (P == 0) return RQHIP_OK;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int tb = 256;
    const unsigned gb = (unsigned)((P + tb - 1) / tb);
    if (h == 0 || N == 0) {
        // the empty prefix matches every corpus row (all() over no columns), so it is valid iff the corpus is
        // not empty; nothing is valid against an empty corpus (any() over no rows)
        hipLaunchKernelGGL(fill_u8_kernel, dim3(gb), dim3(tb), 0, s, valid, (long long)P, (uint8_t)(N > 0 ? 1 : 0));
        RQ_CHECK_LAUNCH("fill_u8_kernel");
        return RQHIP_OK;
    }
    hipLaunchKernelGGL(prefix_lookup_kernel, dim3(gb), dim3(tb), 0, s, reinterpret_cast<const int *>(index),
                       (unsigned)(slots_for(N) - 1), corpus, (long long)ld, prefix, (long long)P, h, (long long)ldp,
                       valid);
    RQ_CHECK_LAUNCH("prefix_lookup_kernel");
    return RQHIP_OK;
}

extern "C" int rqhip_topk_first_match(const int64_t *actual, const int64_t *top_k, int64_t B, int K, int D,
                                      int64_t *rank, rqhip_stream_t stream) {
    if (B < 0 || K < 0 || D < 0 || (B > 0 && (!rank || (K > 0 && D > 0 && (!actual || !top_k))))) {
        set_error("topk_first_match: bad arguments (B=%lld, K=%d, D=%d)", (long long)B, K, D);
        return RQHIP_EARG;
    }
    if (B == 0) return RQHIP_OK;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (int rc = fill_words(rank, 0xffffffffu, (size_t)B * sizeof(int64_t), s)) return rc;
    if (K == 0) return RQHIP_OK;
    const long long total = (long long)B * K;
    if (total >= (1ll << 40)) {
        set_error("topk_first_match: B*K=%lld is too large", total);
        return RQHIP_EUNSUPPORTED;
    }
    const int tb = 256;
    hipLaunchKernelGGL(topk_match_kernel, dim3((unsigned)((total + tb - 1) / tb)), dim3(tb), 0, s, actual, top_k,
                       (long long)B, K, D, reinterpret_cast<unsigned long long *>(rank));
    RQ_CHECK_LAUNCH("topk_match_kernel");
    return RQHIP_OK;
}
