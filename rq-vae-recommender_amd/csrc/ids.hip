// ids.hip -- semantic-id statistics on the device (gfx950): duplicate detection over L-tuples of ids.
//
//   n_distinct : number of rows with no LATER identical row == number of distinct tuples
//                (reference modules/rqvae.py:159-167 builds a B x B x L boolean tensor for this).
//   rank[i]    : number of EARLIER rows with the same tuple -- the dedup column of
//                SemanticIdTokenizer.precompute_corpus_ids (modules/tokenizer/semids.py:92-108, O(N^2 L)).
//
// Phase 1 (group): every row inserts itself into an open-addressing hash table whose slots hold the row
//   index of the first inserter; later rows compare their tuple with that row's (exact, no reliance on the
//   hash being collision free).  group[i] = slot index; n_distinct = number of successful inserts.
// Phase 2 (rank, only when requested): stable LSD radix sort of (group, row) pairs, 8 bits per pass; inside a
//   run of equal groups rows are then in ascending order and rank = position - run start.
// Integer work only: results are exact and deterministic.
#include "rqhip_common.h"

namespace rqhip {

constexpr int kSortBlock = 2048;  // elements per wave in the radix passes

__device__ __forceinline__ unsigned mix32(unsigned h, unsigned v) {
    h ^= v + 0x9e3779b9u + (h << 6) + (h >> 2);
    h *= 0x85ebca6bu;
    h ^= h >> 13;
    return h;
}

// n_distinct / ticket: two counters behind the table, cleared with it by ONE fill to all-ones (count = -1, ticket = -1): the leaders of a
// wave add their number to `count`, the last workgroup to take a ticket writes count + 1 to `out` -- the statistic of a training step is two
// launches (fill, this kernel) instead of four (two fills, this kernel, a one-thread store).
__global__ void ids_group_kernel(const int64_t *__restrict__ ids, long long B, int L, int *__restrict__ table,
                                 unsigned mask, unsigned *__restrict__ group,
                                 unsigned long long *__restrict__ n_distinct, unsigned *__restrict__ ticket, int64_t *__restrict__ out,
                                 float *__restrict__ frac) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    bool leader = false;
    if (i < B) {
        unsigned h = 0x12345u;
        for (int l = 0; l < L; ++l) h = mix32(h, (unsigned)ids[(size_t)l * B + i]);
        unsigned s = h & mask;
        for (;;) {
            int prev = atomicCAS(&table[s], -1, (int)i);
            if (prev == -1) {
                leader = true;
                break;
            }
            bool same = true;
            for (int l = 0; l < L && same; ++l) same = ids[(size_t)l * B + prev] == ids[(size_t)l * B + i];
            if (same) break;
            s = (s + 1) & mask;
        }
        if (group) group[i] = s;
    }
    const unsigned long long m = __ballot(leader);
    if ((threadIdx.x & 63) == 0 && m) atomicAdd(n_distinct, (unsigned long long)__builtin_popcountll(m));
    if (!out && !frac) return;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // this wave's atomics are PERFORMED (acknowledged), not merely issued ...
    __syncthreads();                                         // ... in every wave of the workgroup, before thread 0 takes the ticket
    if (threadIdx.x == 0) {
        __threadfence();                                     // ONE device-scope release per workgroup, not one per wave
        const unsigned old = atomicAdd(ticket, 1u);          // -1, 0, 1, ...: the last of gridDim.x workgroups sees gridDim.x - 2
        if (old + 1u == gridDim.x - 1u) {
            __threadfence();
            const unsigned long long n = atomicAdd(n_distinct, 0ull) + 1ull;
            if (out) *out = (int64_t)n;
            if (frac) *frac = (float)(long long)n / (float)B;      // int64 tensor / int in torch: both to fp32, IEEE divide
        }
    }
}

__global__ void ids_store_count_kernel(const unsigned long long *__restrict__ n, int64_t *__restrict__ out) {
    *out = (int64_t)*n;
}

// ---- stable LSD radix sort of (key, value) pairs, one wave per kSortBlock elements ------------------------
__global__ __launch_bounds__(64) void radix_hist_kernel(const unsigned *__restrict__ keys, long long B, int shift,
                                                        unsigned *__restrict__ hist /*[256][nblocks]*/,
                                                        int nblocks) {
    __shared__ unsigned h[256];
    const int lane = threadIdx.x;
    for (int d = lane; d < 256; d += 64) h[d] = 0;
    __syncthreads();
    const long long base = (long long)blockIdx.x * kSortBlock;
    for (int e = lane; e < kSortBlock; e += 64) {
        const long long i = base + e;
        if (i < B) atomicAdd(&h[(keys[i] >> shift) & 255u], 1u);
    }
    __syncthreads();
    for (int d = lane; d < 256; d += 64) hist[(size_t)d * nblocks + blockIdx.x] = h[d];
}

// exclusive scan of hist in (digit-major, block-minor) order; single workgroup
__global__ __launch_bounds__(1024) void radix_scan_kernel(unsigned *__restrict__ hist, long long n) {
    __shared__ unsigned long long part[1024];
    const int t = threadIdx.x;
    const long long per = (n + 1023) / 1024;
    const long long lo = (long long)t * per, hi = (lo + per < n) ? lo + per : n;
    unsigned long long s = 0;
    for (long long i = lo; i < hi; ++i) s += hist[i];
    part[t] = s;
    __syncthreads();
    if (t == 0) {
        unsigned long long run = 0;
        for (int k = 0; k < 1024; ++k) {
            const unsigned long long v = part[k];
            part[k] = run;
            run += v;
        }
    }
    __syncthreads();
    unsigned long long run = part[t];
    for (long long i = lo; i < hi; ++i) {
        const unsigned v = hist[i];
        hist[i] = (unsigned)run;
        run += v;
    }
}

__global__ __launch_bounds__(64) void radix_scatter_kernel(const unsigned *__restrict__ keys,
                                                           const unsigned *__restrict__ vals, long long B, int shift,
                                                           const unsigned *__restrict__ offs, int nblocks,
                                                           unsigned *__restrict__ keys_out,
                                                           unsigned *__restrict__ vals_out) {
    __shared__ unsigned cur[256];
    const int lane = threadIdx.x;
    for (int d = lane; d < 256; d += 64) cur[d] = offs[(size_t)d * nblocks + blockIdx.x];
    __syncthreads();
    const long long base = (long long)blockIdx.x * kSortBlock;
    const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    for (int e0 = 0; e0 < kSortBlock; e0 += 64) {
        const long long i = base + e0 + lane;
        const bool ok = i < B;
        const unsigned key = ok ? keys[i] : 0u;
        const unsigned val = ok ? vals[i] : 0u;
        const unsigned dig = (key >> shift) & 255u;
        unsigned long long todo = __ballot(ok);
        unsigned pos = 0;
        while (todo) {  // one iteration per distinct digit in this 64-element chunk, ascending lane order kept
            const int first = __builtin_ctzll(todo);
            const unsigned dsel = __shfl(dig, first, 64);
            const unsigned long long same = __ballot(ok && dig == dsel);
            if (ok && dig == dsel) pos = cur[dsel] + (unsigned)__builtin_popcountll(same & lt);
            __syncthreads();
            if (lane == first) cur[dsel] += (unsigned)__builtin_popcountll(same);
            __syncthreads();
            todo &= ~same;
        }
        if (ok) {
            keys_out[pos] = key;
            vals_out[pos] = val;
        }
    }
}

__global__ void iota_kernel(unsigned *__restrict__ v, long long B) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < B) v[i] = (unsigned)i;
}

// sorted by group, rows ascending inside a run: rank = position - run start.  Run starts are marked, then
// propagated with an inclusive max-scan (positions are monotone), so collapsed codebooks (one huge run)
// cost the same as healthy ones.
__global__ void run_start_kernel(const unsigned *__restrict__ keys, long long B, unsigned *__restrict__ start) {
    const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= B) return;
    start[p] = (p == 0 || keys[p - 1] != keys[p]) ? (unsigned)p : 0u;
}

// inclusive max-scan of `start` (monotone: the latest run start at or before p); single workgroup
__global__ __launch_bounds__(1024) void max_scan_kernel(unsigned *__restrict__ v, long long n) {
    __shared__ unsigned part[1024];
    const int t = threadIdx.x;
    const long long per = (n + 1023) / 1024;
    const long long lo = (long long)t * per, hi = (lo + per < n) ? lo + per : n;
    unsigned m = 0;
    for (long long i = lo; i < hi; ++i) m = max(m, v[i]);
    part[t] = m;
    __syncthreads();
    if (t == 0) {
        unsigned run = 0;
        for (int k = 0; k < 1024; ++k) {
            const unsigned x = part[k];
            part[k] = run;
            run = max(run, x);
        }
    }
    __syncthreads();
    unsigned run = part[t];
    for (long long i = lo; i < hi; ++i) {
        run = max(run, v[i]);
        v[i] = run;
    }
}

__global__ void rank_write_kernel(const unsigned *__restrict__ start, const unsigned *__restrict__ rows, long long B,
                                  int64_t *__restrict__ rank) {
    const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p < B) rank[rows[p]] = (int64_t)(p - (long long)start[p]);
}

static unsigned pow2_at_least(unsigned long long n) {
    unsigned long long p = 64;
    while (p < n) p <<= 1;
    return (unsigned)p;
}

struct DedupLayout {
    size_t table_slots, off_table, off_count, off_keys0, off_vals0, off_keys1, off_vals1, off_hist, total;
    int nblocks;
};

static DedupLayout dedup_layout(long long B) {
    DedupLayout d;
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    d.table_slots = pow2_at_least((unsigned long long)(B > 0 ? B : 1) * 2);
    d.nblocks = (int)((B + kSortBlock - 1) / kSortBlock);
    if (d.nblocks < 1) d.nblocks = 1;
    size_t o = 0;
    d.off_table = o; o = al(o + d.table_slots * 4);
    d.off_count = o; o = al(o + 16);     // count (8 bytes), ticket (4)
    d.off_keys0 = o; o = al(o + (size_t)(B > 0 ? B : 1) * 4);
    d.off_vals0 = o; o = al(o + (size_t)(B > 0 ? B : 1) * 4);
    d.off_keys1 = o; o = al(o + (size_t)(B > 0 ? B : 1) * 4);
    d.off_vals1 = o; o = al(o + (size_t)(B > 0 ? B : 1) * 4);
    d.off_hist = o; o = al(o + (size_t)256 * d.nblocks * 4);
    d.total = o;
    return d;
}

}  // namespace rqhip

using namespace rqhip;

extern "C" size_t rqhip_dedup_workspace_bytes(int64_t B) { return dedup_layout(B).total; }

extern "C" int rqhip_unique_fraction(const int64_t *ids, int64_t B, int L, float *p_unique, void *workspace, size_t workspace_bytes,
                                     rqhip_stream_t stream) {
    if (B < 1 || B >= (1ll << 30) || L < 1 || !ids || !p_unique) {
        set_error("unique_fraction: bad arguments (B=%lld, L=%d): 1 <= B < 2^30 rows, non-null pointers", (long long)B, L);
        return RQHIP_EARG;
    }
    const DedupLayout lay = dedup_layout(B);
    if (!workspace || workspace_bytes < lay.total) {
        set_error("unique_fraction: workspace too small (%zu < %zu: rqhip_dedup_workspace_bytes)", workspace_bytes, lay.total);
        return RQHIP_EWORKSPACE;
    }
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    char *ws = reinterpret_cast<char *>(workspace);
    int *table = reinterpret_cast<int *>(ws + lay.off_table);
    unsigned long long *count = reinterpret_cast<unsigned long long *>(ws + lay.off_count);
    if (int rc = fill_words(table, 0xffffffffu, lay.off_count + 16 - lay.off_table, s)) return rc;
    const int gb = (int)((B + 255) / 256);
    hipLaunchKernelGGL(ids_group_kernel, dim3(gb), dim3(256), 0, s, ids, (long long)B, L, table, (unsigned)(lay.table_slots - 1),
                       (unsigned *)nullptr, count, reinterpret_cast<unsigned *>(count + 1), (int64_t *)nullptr, p_unique);
    RQ_CHECK_LAUNCH("ids_group_kernel");
    return RQHIP_OK;
}

extern "C" int rqhip_dedup_rank(const int64_t *ids, int64_t B, int L, int K, int64_t *rank, int64_t *n_distinct,
                                void *workspace, size_t workspace_bytes, rqhip_stream_t stream) {
    (void)K;
    if (B < 0 || L < 1 || (B > 0 && !ids)) {
        set_error("dedup_rank: bad arguments (B=%lld, L=%d)", (long long)B, L);
        return RQHIP_EARG;
    }
    if (B >= (1ll << 30)) {
        set_error("dedup_rank: B=%lld exceeds the 2^30 rows this implementation indexes with 32 bits", (long long)B);
        return RQHIP_EUNSUPPORTED;
    }
    const DedupLayout lay = dedup_layout(B);
    if (!workspace || workspace_bytes < lay.total) {
        set_error("dedup_rank: workspace too small (%zu < %zu)", workspace_bytes, lay.total);
        return RQHIP_EWORKSPACE;
    }
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    char *ws = reinterpret_cast<char *>(workspace);
    int *table = reinterpret_cast<int *>(ws + lay.off_table);
    unsigned long long *count = reinterpret_cast<unsigned long long *>(ws + lay.off_count);
    unsigned *keys0 = reinterpret_cast<unsigned *>(ws + lay.off_keys0);
    unsigned *vals0 = reinterpret_cast<unsigned *>(ws + lay.off_vals0);
    unsigned *keys1 = reinterpret_cast<unsigned *>(ws + lay.off_keys1);
    unsigned *vals1 = reinterpret_cast<unsigned *>(ws + lay.off_vals1);
    unsigned *hist = reinterpret_cast<unsigned *>(ws + lay.off_hist);

    if (B == 0) {
        if (int rc = fill_words(count, 0u, 8, s)) return rc;
        if (n_distinct) {
            hipLaunchKernelGGL(ids_store_count_kernel, dim3(1), dim3(1), 0, s, count, n_distinct);
            RQ_CHECK_LAUNCH("ids_store_count_kernel");
        }
        return RQHIP_OK;
    }
    // the table (empty = -1) and, right behind it, the two counters (count - 1, ticket - 1): one fill
    unsigned *ticket = reinterpret_cast<unsigned *>(count + 1);
    if (int rc = fill_words(table, 0xffffffffu, lay.off_count + 16 - lay.off_table, s)) return rc;
    const int tb = 256;
    const int gb = (int)((B + tb - 1) / tb);
    hipLaunchKernelGGL(ids_group_kernel, dim3(gb), dim3(tb), 0, s, ids, (long long)B, L, table,
                       (unsigned)(lay.table_slots - 1), rank ? keys0 : nullptr, count, ticket, n_distinct, (float *)nullptr);
    RQ_CHECK_LAUNCH("ids_group_kernel");
    if (!rank) return RQHIP_OK;

    hipLaunchKernelGGL(iota_kernel, dim3(gb), dim3(tb), 0, s, vals0, (long long)B);
    RQ_CHECK_LAUNCH("iota_kernel");
    // group ids are < table_slots: sort only the bits that can be set
    int bits = 0;
    while ((1ull << bits) < lay.table_slots) ++bits;
    unsigned *kin = keys0, *vin = vals0, *kout = keys1, *vout = vals1;
    for (int shift = 0; shift < bits; shift += 8) {
        hipLaunchKernelGGL(radix_hist_kernel, dim3(lay.nblocks), dim3(64), 0, s, kin, (long long)B, shift, hist,
                           lay.nblocks);
        RQ_CHECK_LAUNCH("radix_hist_kernel");
        hipLaunchKernelGGL(radix_scan_kernel, dim3(1), dim3(1024), 0, s, hist, (long long)256 * lay.nblocks);
        RQ_CHECK_LAUNCH("radix_scan_kernel");
        hipLaunchKernelGGL(radix_scatter_kernel, dim3(lay.nblocks), dim3(64), 0, s, kin, vin, (long long)B, shift,
                           hist, lay.nblocks, kout, vout);
        RQ_CHECK_LAUNCH("radix_scatter_kernel");
        unsigned *t = kin; kin = kout; kout = t;
        t = vin; vin = vout; vout = t;
    }
    // kin/vin hold the sorted pairs; kout is free: run starts + max-scan there
    hipLaunchKernelGGL(run_start_kernel, dim3(gb), dim3(tb), 0, s, kin, (long long)B, kout);
    RQ_CHECK_LAUNCH("run_start_kernel");
    hipLaunchKernelGGL(max_scan_kernel, dim3(1), dim3(1024), 0, s, kout, (long long)B);
    RQ_CHECK_LAUNCH("max_scan_kernel");
    hipLaunchKernelGGL(rank_write_kernel, dim3(gb), dim3(tb), 0, s, kout, vin, (long long)B, rank);
    RQ_CHECK_LAUNCH("rank_write_kernel");
    return RQHIP_OK;
}
