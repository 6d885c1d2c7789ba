// rq_backward.hip -- backward of the fused residual-quantisation stack (gfx950).
//
// Closed form of what torch.autograd computes through the reference's level loop (modules/rqvae.py:
// 125-132), the STE / rotation-trick / eval branches of Quantize.forward (modules/quantize.py:137-161),
// the embedding lookup (:101-102) and QuantizeLoss (modules/loss.py:38-41); the recursion is written out
// in oracle/rq_oracle.c:rqo_rq_backward.  HBM-bound: per row it reads res0, the L ids and the upstream
// gradients once and writes g_res0 once; the codeword rows come from L2.
//
// Kernel 1 (rows): one wave owns 32 rows in the pair layout of rq_rowmath.h.  Pass 1 replays the residual
//   chain (bit-identical to the forward) and parks res_l, l >= 1, in the workspace; pass 2 walks the levels
//   backwards carrying G = dL/d res_l in registers, writes g_res0 (exact) and leaves each row's codeword-
//   gradient vector V_l (what the embedding backward would index_add) in the workspace slot of level l.
// Kernel 2 (scatter): the embedding backward proper.  Each workgroup owns a contiguous range of rows and
//   accumulates V_l into an LDS-private [L,K,D+1] table with ds_add_f32 (row stride D+1 spreads codes over
//   the banks), then stores the table as one partial; a direct global atomicAdd version of this (9.6 M
//   atomics on 24 576 addresses at B = 100 k) measured 6.1 ms on MI355X, 60x the whole forward.
// Kernel 3 (reduce): g_codebooks[j] = sum over workgroups of partial[g][j], fixed order.
// When one level's table exceeds LDS (K (D+1) 4 B > 150 KiB) kernel 1 falls back to global atomics.
#include <stdlib.h>
#include "rq_rowmath.h"

namespace rqhip {

struct RqBwdParams {
    const float *res0, *cb;
    const int64_t *ids;
    const float *g_embs, *g_embsum, *g_resid, *g_loss;
    float *g_res0, *g_cb;
    float *ws;  // [L, B, D]: slot l holds res_l (l >= 1) during pass 1, then V_l
    int atomic_scatter;  // 1: scatter codeword gradients with global atomics (tables do not fit LDS)
    long long B, n_tiles;
    int D, L, K;
    float beta;
    // fused kernel only: this launch scatters the codeword gradients of levels [l_begin, l_end) (the ones whose tables
    // fit LDS together) and writes g_res0 iff write_rows
    int l_begin, l_end, write_rows;
#ifdef RQ_BWD_PROBE
    int probe;  // developer A/B build only (tools/probe_backward.sh): bit mask of phases to skip, from $RQ_BWD_PROBE
#endif
};
#ifdef RQ_BWD_PROBE
#define RQ_PROBE(bit) (p.probe & (bit))
#else
#define RQ_PROBE(bit) 0
#endif

// VEC: D == 2 * KSTEPS and every row pointer 16-byte aligned -> rows move as float4 half-rows + v_permlane32_swap (rq_rowmath.h)
// instead of one predicated dword per feature.  Same values in the same registers, so the same result bits.  (At batch 64, D = 64,
// rotation trick -- the reference's rqvae_ml32m.gin -- the dword form was 58 us for TWO waves of work: ~800 predicated loads per lane,
// each in its own branch, profiles/r05_small_batch_kernels.txt.)
template <int KSTEPS, int MODE, bool VEC>
__global__ __launch_bounds__(256) void rq_backward_kernel(const RqBwdParams p) {
    const int lane = threadIdx.x & 63;
    const int il = lane & 31, h = lane >> 5;
    const int D = p.D, L = p.L, K = p.K;
    const long long waves = (long long)gridDim.x * (blockDim.x >> 6);
    const long long gw = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    auto load_row = [&](const float *base, float(&v)[KSTEPS]) {
        if constexpr (VEC) load_pair_row_vec<KSTEPS>(base, h, v);
        else load_pair_row<KSTEPS>(base, D, h, v);
    };
    auto store_row = [&](float *base, const float(&v)[KSTEPS]) {   // (callers test the row)
        if constexpr (VEC) {
            store_pair_row<KSTEPS>(base, h, v);
        } else {
#pragma unroll
            for (int kk = 0; kk < KSTEPS; ++kk) {
                const int d = 2 * kk + h;
                if (d < D) base[d] = v[kk];
            }
        }
    };

    for (long long tile = gw; tile < p.n_tiles; tile += waves) {
        const long long row = tile * 32 + il;
        const bool ok = row < p.B;
        const long long rc = ok ? row : p.B - 1;

        float r[KSTEPS], e[KSTEPS], o[KSTEPS];
        load_row(p.res0 + (size_t)rc * D, r);
        // pass 1: replay res_1 .. res_{L-1}
        for (int l = 0; l + 1 < L; ++l) {
            const long long id = p.ids[(size_t)l * p.B + rc];
            load_row(p.cb + ((size_t)l * K + id) * D, e);
            const float xsq = (MODE == RQHIP_MODE_ROTATION) ? pair_sumsq<KSTEPS>(r) : 0.0f;
            level_output<KSTEPS, MODE>(r, e, xsq, o);
            float *dst = p.ws + ((size_t)(l + 1) * p.B + rc) * D;
#pragma unroll
            for (int kk = 0; kk < KSTEPS; ++kk) r[kk] = r[kk] - o[kk];
            if (ok) store_row(dst, r);
        }
        // r now holds res_{L-1}
        const float gl = p.g_loss ? p.g_loss[rc] : 0.0f;
        float G[KSTEPS];
#pragma unroll
        for (int kk = 0; kk < KSTEPS; ++kk) G[kk] = 0.0f;

        for (int l = L - 1; l >= 0; --l) {
            if (l != L - 1) {
                const float *src = (l == 0) ? p.res0 + (size_t)rc * D : p.ws + ((size_t)l * p.B + rc) * D;
                load_row(src, r);
            }
            const long long id = p.ids[(size_t)l * p.B + rc];
            load_row(p.cb + ((size_t)l * K + id) * D, e);
            const size_t lrow = ((size_t)l * p.B + rc) * D;
            float A[KSTEPS], gr[KSTEPS];
            if constexpr (VEC) {
#pragma unroll
                for (int kk = 0; kk < KSTEPS; ++kk) A[kk] = gr[kk] = 0.0f;
                if (p.g_embs) load_row(p.g_embs + lrow, A);
                if (p.g_embsum) {
                    load_row(p.g_embsum + (size_t)rc * D, gr);   // (gr as the temporary: it is loaded for real below)
#pragma unroll
                    for (int kk = 0; kk < KSTEPS; ++kk) A[kk] = A[kk] + gr[kk];
                }
#pragma unroll
                for (int kk = 0; kk < KSTEPS; ++kk) {
                    A[kk] = A[kk] - G[kk];
                    gr[kk] = 0.0f;
                }
                if (p.g_resid) load_row(p.g_resid + lrow, gr);
            } else {
#pragma unroll
                for (int kk = 0; kk < KSTEPS; ++kk) {
                    const int d = 2 * kk + h;
                    float a = 0.0f;
                    if (p.g_embs && d < D) a = p.g_embs[lrow + d];
                    if (p.g_embsum) a = a + ((d < D) ? p.g_embsum[(size_t)rc * D + d] : 0.0f);
                    A[kk] = a - G[kk];
                    gr[kk] = (p.g_resid && d < D) ? p.g_resid[lrow + d] : 0.0f;
                }
            }
            float *dE = (p.g_cb && p.atomic_scatter) ? p.g_cb + ((size_t)l * K + id) * D : nullptr;
            float *V = (p.g_cb && !p.atomic_scatter) ? p.ws + lrow : nullptr;
            float vv[VEC ? KSTEPS : 1];   // VEC: this level's codeword-gradient vector of the row, stored as a row below
            auto emit = [&](int kk, float v) {
                if constexpr (VEC) {
                    vv[kk] = v;
                } else {
                    const int d = 2 * kk + h;
                    if (ok && d < D) {
                        if (dE) atomicAdd(dE + d, v);
                        if (V) V[d] = v;
                    }
                }
            };
            if (MODE == RQHIP_MODE_ROTATION) {
                float w[KSTEPS], u[KSTEPS], q[KSTEPS], scale;
                const float xsq = pair_sumsq<KSTEPS>(r);
                rotation_lane<KSTEPS>(r, e, xsq, o, w, u, q, scale);
                const float aw = pair_dot<KSTEPS>(A, w), aq = pair_dot<KSTEPS>(A, q);
#pragma unroll
                for (int kk = 0; kk < KSTEPS; ++kk) {
                    const float lin = ((A[kk] - 2.0f * (aw * w[kk])) + 2.0f * (aq * u[kk])) * scale;
                    const float commit = (2.0f * p.beta) * (r[kk] - e[kk]) * gl;
                    const float embg = (2.0f * (e[kk] - r[kk])) * gl;
                    G[kk] = ((gr[kk] + G[kk]) + lin) + commit;
                    emit(kk, embg);
                }
            } else {
#pragma unroll
                for (int kk = 0; kk < KSTEPS; ++kk) {
                    const float commit = (2.0f * p.beta) * (r[kk] - e[kk]) * gl;
                    const float embg = (2.0f * (e[kk] - r[kk])) * gl;
                    if (MODE == RQHIP_MODE_EVAL) {
                        const float contrib = A[kk] + embg;
                        G[kk] = (gr[kk] + G[kk]) + commit;
                        emit(kk, contrib);
                    } else {
                        G[kk] = ((gr[kk] + G[kk]) + A[kk]) + commit;
                        emit(kk, embg);
                    }
                }
            }
            if constexpr (VEC) {
                if (ok && dE) {
#pragma unroll
                    for (int kk = 0; kk < KSTEPS; ++kk) atomicAdd(dE + 2 * kk + h, vv[kk]);
                }
                if (ok && V) store_row(V, vv);
            }
        }
        if (ok && p.g_res0) store_row(p.g_res0 + (size_t)row * D, G);
    }
}



// ---- fused variant for L <= kFusedMaxL: rows + embedding backward in ONE kernel -------------------------------
// Same arithmetic as rq_backward_kernel (g_res0 bit-identical), but the residual chain of a row stays in
// registers (L * KSTEPS values per lane) and each row's codeword-gradient vectors go straight into the
// workgroup's LDS tables: no [L,B,D] round trip through HBM.  Per row: reads res0, ids, upstream gradients
// (12D + 8L bytes), writes g_res0 (4D) -- the algorithmic traffic of SURVEY.md 8d.
constexpr int kFusedMaxL = 4;
constexpr int kFusedThreads = 512;
constexpr int kFusedWaves = kFusedThreads / 64;

// Embedding backward inside the fused kernel WITHOUT atomics ("owner computes"): the workgroup keeps one
// [levels, K, D] table in LDS; after a level's per-row vectors V are known, the waves park them (and the rows' ids) in
// an LDS stage, and every code is then accumulated by exactly ONE wave -- wave (id mod 8) -- which walks the staged
// rows in order (ballot of "mine", lowest set bit first) and adds V to its table row with plain LDS read / add / write.
// One owner per address and in-order LDS execution make the sum order fixed: round, wave, row ascending -- restated by
// oracle/rq_oracle.c:rqo_rq_backward_ordered, so the codebook gradient is bit-reproducible.  (The first version let
// all waves ds_add_f32 into the table: 9.6 M LDS float atomics = 45 of the kernel's 76 us, and an unordered sum.)
// `sg` waves are staged at a time (8, or 4 when a K = 1024 table leaves less LDS).
__device__ __forceinline__ void cb_accumulate(float *__restrict__ tab_l, const float *__restrict__ stage,
                                              const int *__restrict__ ids_s, int rows, int D, int wave, int lane) {
    for (int base = 0; base < rows; base += 64) {
        const int myid = (base + lane < rows) ? ids_s[base + lane] : -1;
        const bool mine = myid >= 0 && (myid & (kFusedWaves - 1)) == wave;
        unsigned long long m = __ballot(mine);
        while (m) {
            const int j = __builtin_ctzll(m);
            m &= m - 1;
            const int id = __builtin_amdgcn_readlane(myid, j);
            // the next row of this wave too, when it belongs to ANOTHER code: two independent read-add-write chains in
            // flight instead of one (same code: strictly one after the other -- the order of the sum is the contract)
            int j2 = -1, id2 = -1;
            if (m) {
                j2 = __builtin_ctzll(m);
                id2 = __builtin_amdgcn_readlane(myid, j2);
                if (id2 != id) m &= m - 1; else j2 = -1;
            }
            if (lane < D) {
                float *t = tab_l + (size_t)id * D + lane;
                if (j2 >= 0) {
                    float *t2 = tab_l + (size_t)id2 * D + lane;
                    const float a = *t, b = *t2;
                    const float va = stage[(size_t)(base + j) * (D + 1) + lane];
                    const float vb = stage[(size_t)(base + j2) * (D + 1) + lane];
                    *t = a + va;
                    *t2 = b + vb;
                } else {
                    *t = *t + stage[(size_t)(base + j) * (D + 1) + lane];
                }
            }
        }
    }
}

// VEC: D == 2*KSTEPS and every row pointer 16-byte aligned -> rows and codewords move as float4 half-rows +
// v_permlane32_swap (rq_rowmath.h) instead of 4 bytes per lane and instruction.
template <int KSTEPS, int MODE, bool VEC>
__global__ __launch_bounds__(kFusedThreads) void rq_backward_fused_kernel(const RqBwdParams p, float *__restrict__ partial,
                                                                         int LKD_total, int sg) {
    extern __shared__ __attribute__((aligned(16))) float acc[];
    const int D = p.D, L = p.L, K = p.K;
    const int tbl = (p.l_end - p.l_begin) * K * D;           // [levels of this launch][K][D]
    float *stage = acc + tbl;                               // [sg * 32][D + 1]
    int *ids_s = reinterpret_cast<int *>(stage + sg * 32 * (D + 1));  // [sg * 32]
    if (p.g_cb)
        for (int e = threadIdx.x; e < tbl; e += kFusedThreads) acc[e] = 0.0f;
    __syncthreads();

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int il = lane & 31, h = lane >> 5;
    constexpr int kWaves = kFusedWaves;
    const long long waves = (long long)gridDim.x * kWaves;
    const long long gw = (long long)wave * gridDim.x + blockIdx.x;
    // every wave of the workgroup runs the same number of rounds (the accumulation below has barriers)
    const long long n_rounds = (p.n_tiles + waves - 1) / waves;

    for (long long it = 0; it < n_rounds; ++it) {
        const long long tile = it * waves + gw;
        const long long row = tile * 32 + il;
        const bool ok = tile < p.n_tiles && row < p.B;
        const long long rc = ok ? row : p.B - 1;

        float rl[kFusedMaxL][KSTEPS];
        float el[kFusedMaxL][KSTEPS];  // the codeword of every level, gathered once
        float o[KSTEPS];
        int idl[kFusedMaxL];
        auto fetch = [&](const float *row_base, float(&v)[KSTEPS]) {
            if (VEC) load_pair_row_vec<KSTEPS>(row_base, h, v);
            else load_pair_row<KSTEPS>(row_base, D, h, v);
        };
        fetch(p.res0 + (size_t)rc * D, rl[0]);
#pragma unroll
        for (int l = 0; l < kFusedMaxL; ++l) {
            if (l < L) {
                idl[l] = (int)p.ids[(size_t)l * p.B + rc];
                fetch(p.cb + ((size_t)l * K + idl[l]) * D, el[l]);
                if (l + 1 < L) {
                    const float xsq = (MODE == RQHIP_MODE_ROTATION) ? pair_sumsq<KSTEPS>(rl[l]) : 0.0f;
                    level_output<KSTEPS, MODE>(rl[l], el[l], xsq, o);
#pragma unroll
                    for (int kk = 0; kk < KSTEPS; ++kk)
                        if (l + 1 < kFusedMaxL) rl[(l + 1 < kFusedMaxL) ? l + 1 : 0][kk] = rl[l][kk] - o[kk];
                }
            }
        }
        const float gl = p.g_loss ? p.g_loss[rc] : 0.0f;
        float G[KSTEPS], gs[KSTEPS];
#pragma unroll
        for (int kk = 0; kk < KSTEPS; ++kk) {
            G[kk] = 0.0f;
            gs[kk] = 0.0f;
        }
        if (p.g_embsum) fetch(p.g_embsum + (size_t)rc * D, gs);
#pragma unroll
        for (int l = kFusedMaxL - 1; l >= 0; --l) {
            if (l < L) {
                const float(&r)[KSTEPS] = rl[l];
                const float(&e)[KSTEPS] = el[l];
                const size_t lrow = ((size_t)l * p.B + rc) * D;
                float A[KSTEPS], gr[KSTEPS];
#pragma unroll
                for (int kk = 0; kk < KSTEPS; ++kk) {
                    A[kk] = 0.0f;
                    gr[kk] = 0.0f;
                }
                if (p.g_embs) fetch(p.g_embs + lrow, A);
                if (p.g_resid) fetch(p.g_resid + lrow, gr);
#pragma unroll
                for (int kk = 0; kk < KSTEPS; ++kk) {
                    float a = A[kk];
                    if (p.g_embsum) a = a + gs[kk];
                    A[kk] = a - G[kk];
                }
                float cbv[KSTEPS];  // this row's contribution to dE_l[id] (features of this lane's parity)
                if (MODE == RQHIP_MODE_ROTATION) {
                    float w[KSTEPS], u[KSTEPS], q[KSTEPS], scale;
                    const float xsq = pair_sumsq<KSTEPS>(r);
                    rotation_lane<KSTEPS>(r, e, xsq, o, w, u, q, scale);
                    const float aw = pair_dot<KSTEPS>(A, w), aq = pair_dot<KSTEPS>(A, q);
#pragma unroll
                    for (int kk = 0; kk < KSTEPS; ++kk) {
                        const float lin = ((A[kk] - 2.0f * (aw * w[kk])) + 2.0f * (aq * u[kk])) * scale;
                        const float commit = (2.0f * p.beta) * (r[kk] - e[kk]) * gl;
                        const float embg = (2.0f * (e[kk] - r[kk])) * gl;
                        G[kk] = ((gr[kk] + G[kk]) + lin) + commit;
                        cbv[kk] = embg;
                    }
                } else {
#pragma unroll
                    for (int kk = 0; kk < KSTEPS; ++kk) {
                        const float commit = (2.0f * p.beta) * (r[kk] - e[kk]) * gl;
                        const float embg = (2.0f * (e[kk] - r[kk])) * gl;
                        if (MODE == RQHIP_MODE_EVAL) {
                            const float contrib = A[kk] + embg;
                            G[kk] = (gr[kk] + G[kk]) + commit;
                            cbv[kk] = contrib;
                        } else {
                            G[kk] = ((gr[kk] + G[kk]) + A[kk]) + commit;
                            cbv[kk] = embg;
                        }
                    }
                }
                if (p.g_cb && l >= p.l_begin && l < p.l_end) {  // uniform: this launch owns level l's table
                    float *tab_l = acc + (size_t)(l - p.l_begin) * K * D;
                    for (int g0 = 0; g0 < kWaves; g0 += sg) {
                        if (wave >= g0 && wave < g0 + sg) {
                            float *st = stage + (size_t)((wave - g0) * 32 + il) * (D + 1) + h;
#pragma unroll
                            for (int kk = 0; kk < KSTEPS; ++kk)
                                if (2 * kk + h < D) st[2 * kk] = cbv[kk];
                            if (h == 0) ids_s[(wave - g0) * 32 + il] = ok ? idl[l] : -1;
                        }
                        __syncthreads();
                        cb_accumulate(tab_l, stage, ids_s, sg * 32, D, wave, lane);
                        __syncthreads();
                    }
                }
            }
        }
        if (ok && p.g_res0 && p.write_rows) {
            if (VEC) {
                store_pair_row<KSTEPS>(p.g_res0 + (size_t)row * D, h, G);
            } else {
#pragma unroll
                for (int kk = 0; kk < KSTEPS; ++kk) {
                    const int d = 2 * kk + h;
                    if (d < D) p.g_res0[(size_t)row * D + d] = G[kk];
                }
            }
        }
    }

    if (p.g_cb) {
        __syncthreads();
        float *out = partial + (size_t)blockIdx.x * LKD_total;
        for (int e2 = threadIdx.x; e2 < (p.l_end - p.l_begin) * K * D; e2 += kFusedThreads) {
            out[e2] = acc[e2];
        }
    }
}

// ---- flat variant for EVAL / STE, D % 4 == 0, D <= 64: the backward of these two modes is purely elementwise ----------
// Nothing in the EVAL / STE recursion crosses features (no dot products -- those belong to the rotation trick), so the
// pair layout of the matrix kernels buys nothing here and costs L * KSTEPS registers per lane (247 VGPRs, two waves per
// SIMD, two half-filled rounds at B = 100 000).  This kernel maps one lane to FOUR consecutive features of one row and
// splits the workgroup's 16 waves into two roles that run concurrently (one workgroup per CU is all the LDS allows, so
// the overlap of the HBM-bound and the LDS-bound half of the work has to happen INSIDE the workgroup):
//   * waves 0-7, "rows": D/4 lanes per row, R = 512 / (D/4) rows per step; every global access is one 16-byte load /
//     store and a wave touches whole consecutive rows.  The row data of step h+1 (res0, upstream gradients, gathered
//     codewords) and the ids of step h+2 are in flight while step h is computed (software pipeline).  Each lane parks
//     its 4 codeword-gradient values per level in the LDS stage [h & 1][levels][R][D] and the table row
//     key = level * K + id in keys[h & 1][][];
//   * waves 8-15, "owners": the embedding backward of the rows staged in step h-1, no atomics.  Every row of the
//     workgroup's LDS table [levels][K][D] is updated by exactly one owner -- half-wave key & 15 when D <= 32, wave
//     key & 7 otherwise -- which adds the staged vectors of its keys in ascending row order with plain LDS read / add /
//     write;
//   * one barrier per step hands the stage buffer over.
// The sum order of a code is therefore: workgroup b of G, step h ascending (rows [(h G + b) R, (h G + b) R + R)), row
// ascending -- oracle/rq_oracle.c:rqo_rq_backward_ordered with unit_rows = R, nw = 1 -- followed by the same 4-segment
// reduce over workgroups.
constexpr int kFlatThreads = 1024;
constexpr int kFlatRowThreads = 512;                    // waves 0-7
constexpr int kFlatOwnerWaves = 8;                      // waves 8-15
constexpr int kFlatMaxD = 64;
// per-owner lists of the staged rows an owner has to add, (key << 12 | staged row): kListCap entries per pass (+2 so the
// prefetch of the next pair never leaves the allocation); more rows of one owner in a step take another pass
constexpr int kListCap = 32, kListStride = kListCap + 2;
constexpr size_t kFlatListBytes = 2 * kFlatOwnerWaves * kListStride * sizeof(unsigned);

// NL: number of levels when known at compile time (3, 4), 0 = p.L.  TRAIN: the upstream gradients are the training
// step's -- g_embsum and g_loss given, g_embs and g_resid absent -- so their loads and tests are compiled out.  (The rows
// role is bound by instruction issue, not by HBM: ~400 VALU instructions per lane and step in the first version, 64-bit
// address chains, per-load predication, tests of L and of four optional pointers.)
// MMP / MMB (round 5; 0 = the ordered owners above): the codebook gradient as a ONE-HOT MATRIX PRODUCT.  dE_l[k] = sum over the rows
// with id_l = k of their staged vector is One_l^T . V_l; the owner waves form it on the bf16 matrix cores instead of walking an LDS
// table: per staged step every 16 rows x level ("pair") are transposed ONCE into the matrix instruction's B operand as three EXACT
// bf16 pieces of the fp32 values (v = h + m + l, 24 bits), owner wave w holds the accumulators of code blocks 32 (w + 8 b), b < MMB,
// of the launch's MMP levels in registers (16 per block and level), builds its one-hot A operand from the 16 keys in registers and
// issues 3 matrix instructions per pair and block.  No LDS read-modify-write, no per-CU table in LDS (its 96 KB hold the operand
// images instead), the per-workgroup partial tables are flushed from registers; the products 1.0 x piece are exact, the sum's ORDER is
// the matrix pipe's: not restatable by the oracle, so rqhip_rq_backward keeps the ordered form and rqhip_rq_backward_ex selects this
// one (RQHIP_BWD_CBGRAD_MATRIX; tests/test_gpu_parity.py: g_res0 identical bits, codebook gradient no further from fp64 than the
// ordered kernel's).  D = 32 only (R = 64 rows per step, 4 pairs per level).
// MEASURED (profiles/r05_cbgrad_matrix_ab.txt): correct on the first GPU run and SLOWER than the ordered owners -- 38.2 vs 31.7 us at
// 100 000 x 3 x 256, 242 vs 184 us at 1 M rows, 247 vs 149 us at 262 144 x 4 x 1024: every owner wave re-reads all B-operand images of a
// step (8 x 36 KB of LDS reads) and builds 12 one-hot operands (20 VALU instructions each) beside the rows role on the same SIMDs, 3.9 us
// per 64-row step against the ordered form's 2.8; and at 100 000 rows 13 us of the call are the per-CU table flush + reduce either way.
// The product (rqhip/ops.py:use_cbgrad) therefore keeps the ordered form; what meets the <= 15 us target is a design WITHOUT per-CU tables
// (rows pre-sorted by code once per batch: DESIGN.md section 8).
typedef __bf16 bw_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bw_bf16x2 __attribute__((ext_vector_type(2)));
typedef float bw_f32x2 __attribute__((ext_vector_type(2)));
typedef float bw_f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned bw_u32x4 __attribute__((ext_vector_type(4)));
typedef int bw_i32x4 __attribute__((ext_vector_type(4)));
// (a, b) -> three dwords, each the packed bf16 pieces {piece(a), piece(b)}; a = h + m + l exactly (likewise b)
__device__ __forceinline__ void bw_split2(float a, float b, unsigned &h, unsigned &m, unsigned &l) {
    const bw_bf16x2 hh = __builtin_convertvector(bw_f32x2{a, b}, bw_bf16x2);
    h = __builtin_bit_cast(unsigned, hh);
    const float ra = a - __builtin_bit_cast(float, h << 16), rb = b - __builtin_bit_cast(float, h & 0xffff0000u);
    const bw_bf16x2 mm = __builtin_convertvector(bw_f32x2{ra, rb}, bw_bf16x2);
    m = __builtin_bit_cast(unsigned, mm);
    const float sa = ra - __builtin_bit_cast(float, m << 16), sb = rb - __builtin_bit_cast(float, m & 0xffff0000u);
    const bw_bf16x2 ll = __builtin_convertvector(bw_f32x2{sa, sb}, bw_bf16x2);
    l = __builtin_bit_cast(unsigned, ll);
}
constexpr int kMmPairOps = 3 * 2 * 32;      // 16-byte elements of a pair's B operand: [piece][octet][feature]

template <int MODE, bool PAIR, int NL, bool TRAIN, int MMP = 0, int MMB = 0>
__global__ __launch_bounds__(kFlatThreads) void rq_backward_flat_kernel(const RqBwdParams p, float *__restrict__ partial,
                                                                       int LKD_total, int R, int LPR) {
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    extern __shared__ __attribute__((aligned(16))) float acc[];
    constexpr int LM = NL ? NL : kFusedMaxL;
    constexpr bool MM = MMP > 0;
    const int D = p.D, L = NL ? NL : p.L, K = p.K;
    const int nl = p.l_end - p.l_begin;
    const int tbl = (p.g_cb && !MM) ? nl * K * D : 0;        // [levels of this launch][K][D]
    const int items = nl * R;                               // staged rows x levels per step
    float *stage0 = acc + tbl;                              // [2][nl][R][D]
    int *keys0 = reinterpret_cast<int *>(stage0 + 2 * (size_t)items * D);   // [2][nl][R]
    unsigned *lists = reinterpret_cast<unsigned *>(keys0 + 2 * items);
    // MM: instead of the lists, [2][pairs] B-operand images and [2][pairs][16] keys (pairs = MMP levels x 4 groups of 16 rows)
    constexpr int kPairs = MM ? MMP * 4 : 1;
    bw_u32x4 *bops = reinterpret_cast<bw_u32x4 *>(lists);
    int *bkeys = reinterpret_cast<int *>(bops + 2 * kPairs * kMmPairOps);
    bw_f32x16 cacc[MM ? MMP : 1][MM ? MMB : 1];
    if constexpr (MM) {
#pragma unroll
        for (int a = 0; a < MMP; ++a)
#pragma unroll
            for (int b = 0; b < MMB; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) cacc[a][b][r] = 0.0f;
    }
    const f32x4 zero4 = {0.0f, 0.0f, 0.0f, 0.0f};
    for (int e = threadIdx.x * 4; e < tbl; e += kFlatThreads * 4) *reinterpret_cast<f32x4 *>(acc + e) = zero4;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool row_role = tid < kFlatRowThreads;             // wave-uniform
    const long long blocks = (p.B + R - 1) / R;
    const long long n_steps = (blocks + gridDim.x - 1) / gridDim.x;   // same for every workgroup (barriers below)

    // ---- state of the "rows" role ---------------------------------------------------------------------------------------
    // Addresses are a uniform base (scalar registers) + a 32-bit byte offset per lane (the host checks B * D * 4 < 2^32).
    // Lanes without a row (tail of the last step, idle slots) read row B-1 like everybody else and do not store.
    const int rl = (tid & (kFlatRowThreads - 1)) / LPR, ch = (tid & (kFlatRowThreads - 1)) - rl * LPR;
    const bool slot = rl < R;                                // row slot inside the step, 16-byte chunk inside the row
    const unsigned Bm1 = (unsigned)(p.B - 1), chb = (unsigned)ch * 16u, rowb = (unsigned)D * 4u;
    auto at = [](const void *base, unsigned byte_off) { return reinterpret_cast<const char *>(base) + byte_off; };
    auto ld4 = [&](const float *base, unsigned byte_off) { return *reinterpret_cast<const f32x4 *>(at(base, byte_off)); };
    auto row_of = [&](long long h, bool &ok) -> unsigned {  // this lane's (clamped) row in step h; ok: it owns that row
        const long long blk = h * gridDim.x + blockIdx.x;
        const bool live = h < n_steps && blk < blocks;       // uniform
        const unsigned row = (live ? (unsigned)(blk * R) : 0u) + (unsigned)rl;
        ok = live && slot && row <= Bm1;
        return row < Bm1 ? row : Bm1;
    };
    auto load_ids = [&](unsigned row, int(&id)[LM]) {
#pragma unroll
        for (int l = 0; l < LM; ++l)
            if (l < L) id[l] = *reinterpret_cast<const int *>(at(p.ids + (size_t)l * p.B, row * 8u));   // low dword
    };
    auto gather = [&](const int(&id)[LM], f32x4(&e)[LM]) {
#pragma unroll
        for (int l = 0; l < LM; ++l)
            if (l < L) e[l] = ld4(p.cb + (size_t)l * K * D, (unsigned)id[l] * rowb + chb);
    };
    bool ok_c = false, ok_n = false, ok_nn = false;
    unsigned row_c = 0, row_n = 0, row_nn = 0;
    int id_c[LM], id_n[LM], id_nn[LM];
    f32x4 e_c[LM], e_n[LM], r0_c = zero4, r0_n = zero4, gs_c = zero4, gs_n = zero4;
    float gl_c = 0.0f, gl_n = 0.0f;
#pragma unroll
    for (int l = 0; l < LM; ++l) { id_c[l] = id_n[l] = id_nn[l] = 0; e_c[l] = e_n[l] = zero4; }
    if (row_role) {
        row_c = row_of(0, ok_c);
        row_n = row_of(1, ok_n);
        load_ids(row_c, id_c);
        r0_c = ld4(p.res0, row_c * rowb + chb);
        if (TRAIN || p.g_embsum) gs_c = ld4(p.g_embsum, row_c * rowb + chb);
        if (TRAIN || p.g_loss) gl_c = *reinterpret_cast<const float *>(at(p.g_loss, row_c * 4u));
        load_ids(row_n, id_n);
        gather(id_c, e_c);
    }

    // ---- state of the "owners" role -------------------------------------------------------------------------------------
    const int ow = wave - (kFlatThreads / 64 - kFlatOwnerWaves);            // owner wave 0..7 (negative: a rows wave)
    const int f = PAIR ? (lane & 31) : lane, half = PAIR ? (lane >> 5) : 0;
    const int nown = PAIR ? 2 * kFlatOwnerWaves : kFlatOwnerWaves;
    const int own0 = PAIR ? 2 * ow : ow;                                    // this wave's first (or only) owner

    // step h: the rows waves stage block h, the owner waves add block h-1; the last step only drains
    // (MM: a staged block is converted in the step after it was staged and multiplied in the step after that)
    const long long last = p.g_cb ? (MM ? n_steps + 1 : n_steps) : n_steps - 1;
    // The two roles run SEPARATE loops with the same number of barriers (a workgroup barrier counts arriving waves, whatever code path
    // they come from): written as one loop with an if / else inside, every loop-carried value of one role is live through the other role's
    // branch and the register allocator adds the two roles' needs -- with the matrix form's 48 accumulator registers that spilled.
    if (row_role) {
      for (long long h = 0; h <= last; ++h) {
        {
            if (h < n_steps) {
                const bool ok = ok_c;
                float *stage = stage0 + (size_t)(h & 1) * items * D;
                int *keys = keys0 + (h & 1) * items;
                const unsigned off_c = row_c * rowb + chb;
                // ---- this step's rows: residual chain forward, then the levels backwards carrying G = dL/d res_l ---------
                f32x4 r[LM];
                r[0] = r0_c;
#pragma unroll
                for (int l = 0; l + 1 < LM; ++l) {
                    if (l + 1 < L) {
                        const f32x4 o = (MODE == RQHIP_MODE_EVAL) ? e_c[l] : r[l] + (e_c[l] - r[l]);   // level_output<MODE>
                        r[l + 1] = r[l] - o;
                    }
                }
                f32x4 G = zero4;
#pragma unroll
                for (int l = LM - 1; l >= 0; --l) {
                    if (l < L) {
                        f32x4 A = zero4, gr = zero4;
                        if (!TRAIN && p.g_embs) A = ld4(p.g_embs + (size_t)l * p.B * D, off_c);
                        if (!TRAIN && p.g_resid) gr = ld4(p.g_resid + (size_t)l * p.B * D, off_c);
                        if (TRAIN || p.g_embsum) A = A + gs_c;
                        A = A - G;
                        const f32x4 commit = ((2.0f * p.beta) * (r[l] - e_c[l])) * gl_c;
                        const f32x4 embg = (2.0f * (e_c[l] - r[l])) * gl_c;
                        f32x4 cbv;
                        if (MODE == RQHIP_MODE_EVAL) {
                            cbv = A + embg;
                            G = (gr + G) + commit;
                        } else {
                            G = ((gr + G) + A) + commit;
                            cbv = embg;
                        }
                        if (p.g_cb && l >= p.l_begin && l < p.l_end && slot && !RQ_PROBE(4)) {   // (level test is uniform)
                            const int li = l - p.l_begin;
                            *reinterpret_cast<f32x4 *>(stage + ((size_t)li * R + rl) * D + 4 * ch) = cbv;
                            if (ch == 0) keys[li * R + rl] = ok ? li * K + id_c[l] : -1;
                        }
                    }
                }
                if (ok && p.g_res0 && p.write_rows)
                    *reinterpret_cast<f32x4 *>(reinterpret_cast<char *>(p.g_res0) + off_c) = G;

                // ---- next step's row data and the ids after that ----------------------------------------------------------
                row_nn = row_of(h + 2, ok_nn);
                const unsigned off_n = row_n * rowb + chb;
                r0_n = ld4(p.res0, off_n);
                if (TRAIN || p.g_embsum) gs_n = ld4(p.g_embsum, off_n);
                if (TRAIN || p.g_loss) gl_n = *reinterpret_cast<const float *>(at(p.g_loss, row_n * 4u));
                gather(id_n, e_n);
                load_ids(row_nn, id_nn);
                row_c = row_n; row_n = row_nn;
                ok_c = ok_n; ok_n = ok_nn;
                r0_c = r0_n; gs_c = gs_n; gl_c = gl_n;
#pragma unroll
                for (int l = 0; l < LM; ++l) { id_c[l] = id_n[l]; id_n[l] = id_nn[l]; e_c[l] = e_n[l]; }
            }
        }
        if (p.g_cb) __syncthreads();
      }
    } else {
      for (long long h = 0; h <= last; ++h) {
        if (MM) {
            if constexpr (MM) {
                if (p.g_cb && ow >= 0) {
                    // ---- convert the block staged in step h - 1 into B-operand images (each owner wave: pairs ow, ow + 8, ...)
                    if (h >= 1 && h - 1 < n_steps) {
                        const float *stage = stage0 + (size_t)((h - 1) & 1) * items * D;
                        const int *keys = keys0 + ((h - 1) & 1) * items;
                        const int cbuf = (int)((h - 1) & 1);
                        const int n = lane & 31, o = lane >> 5;
                        for (int pr = ow; pr < kPairs; pr += kFlatOwnerWaves) {
                            const int li = pr >> 2, grp = pr & 3;
                            const float *src = stage + ((size_t)li * 64 + 16 * grp + 8 * o) * 32 + n;
                            float v[8];
#pragma unroll
                            for (int j = 0; j < 8; ++j) v[j] = src[j * 32];
                            bw_u32x4 ph, pm, pl;
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                unsigned hh, mm, ll;
                                bw_split2(v[2 * j], v[2 * j + 1], hh, mm, ll);
                                ph[j] = hh; pm[j] = mm; pl[j] = ll;
                            }
                            bw_u32x4 *dst = bops + (size_t)(cbuf * kPairs + pr) * kMmPairOps + o * 32 + n;
                            dst[0 * 64] = ph;
                            dst[1 * 64] = pm;
                            dst[2 * 64] = pl;
                            if (lane < 16) bkeys[(cbuf * kPairs + pr) * 16 + lane] = keys[li * 64 + 16 * grp + lane];
                        }
                    }
                    // ---- multiply the images converted in step h - 1 (the block staged in step h - 2)
                    if (h >= 2 && h - 2 < n_steps) {
                        const int mbuf = (int)(h & 1);
                        const int il = lane & 31, o = lane >> 5;
#pragma unroll
                        for (int li = 0; li < MMP; ++li) {
#pragma unroll
                            for (int grp = 0; grp < 4; ++grp) {
                                const int pr = li * 4 + grp;
                                const bw_i32x4 *kp = reinterpret_cast<const bw_i32x4 *>(bkeys + (mbuf * kPairs + pr) * 16 + 8 * o);
                                const bw_i32x4 k0 = kp[0], k1 = kp[1];
                                const bw_u32x4 *src = bops + (size_t)(mbuf * kPairs + pr) * kMmPairOps + o * 32 + il;
                                const bw_bf16x8 bh = __builtin_bit_cast(bw_bf16x8, src[0 * 64]);
                                const bw_bf16x8 bm = __builtin_bit_cast(bw_bf16x8, src[1 * 64]);
                                const bw_bf16x8 bl = __builtin_bit_cast(bw_bf16x8, src[2 * 64]);
#pragma unroll
                                for (int nb = 0; nb < MMB; ++nb) {
                                    const int code = li * K + 32 * (ow + kFlatOwnerWaves * nb) + il;     // keys are level * K + id
                                    bw_u32x4 a;     // bf16 1.0 = 0x3f80
                                    a[0] = (k0[0] == code ? 0x3f80u : 0u) | (k0[1] == code ? 0x3f800000u : 0u);
                                    a[1] = (k0[2] == code ? 0x3f80u : 0u) | (k0[3] == code ? 0x3f800000u : 0u);
                                    a[2] = (k1[0] == code ? 0x3f80u : 0u) | (k1[1] == code ? 0x3f800000u : 0u);
                                    a[3] = (k1[2] == code ? 0x3f80u : 0u) | (k1[3] == code ? 0x3f800000u : 0u);
                                    const bw_bf16x8 av = __builtin_bit_cast(bw_bf16x8, a);
                                    bw_f32x16 c16 = cacc[li][nb];
                                    c16 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bl, c16, 0, 0, 0);   // smallest pieces first
                                    c16 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bm, c16, 0, 0, 0);
                                    c16 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bh, c16, 0, 0, 0);
                                    cacc[li][nb] = c16;
                                }
                            }
                        }
                    }
                }
            }
        } else if (h >= 1 && p.g_cb && !RQ_PROBE(4) && ow >= 0 && (PAIR || ow < nown)) {
            // A first version picked the rows with scalar ballot / readlane logic and was bound by the CU's one scalar
            // unit (4.2 us per 128 rows); a second walked per-batch bit masks and paid three dependent LDS latencies per
            // batch of 64 rows (2.8 us).  Now each wave first lists its owners' rows of the whole step (one pass over
            // the keys), then adds them from the list with the next pair prefetched.
            const float *stage = stage0 + (size_t)((h - 1) & 1) * items * D;
            const int *keys = keys0 + ((h - 1) & 1) * items;
            const unsigned *mylist = lists + (own0 + half) * kListStride;
            const int n_items = RQ_PROBE(2) ? 0 : items;
            for (int lo = 0;; lo += kListCap) {
                // build: every lane looks at one staged row per batch; rows of this wave's owners get their rank
                // (rows before them in the step with the same owner) and go to slot rank - lo of the owner's list
                int cnt0 = 0, cnt1 = 0;
                for (int base = 0; base < n_items; base += 64) {
                    const int item = base + lane;
                    const int mykey = item < n_items ? keys[item] : -1;
                    const int own = mykey & (nown - 1);
                    const bool mine0 = mykey >= 0 && own == own0;
                    const bool mine1 = PAIR && mykey >= 0 && own == own0 + 1;
                    const unsigned long long m0 = __ballot(mine0), m1 = PAIR ? __ballot(mine1) : 0ull;
                    const unsigned long long mm = mine1 ? m1 : m0;
                    const int before = __builtin_amdgcn_mbcnt_hi((unsigned)(mm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mm, 0u));
                    const int slot_i = (mine1 ? cnt1 : cnt0) + before - lo;
                    if ((mine0 || mine1) && slot_i >= 0 && slot_i < kListCap && !RQ_PROBE(1))
                        lists[(own0 + (mine1 ? 1 : 0)) * kListStride + slot_i] = ((unsigned)mykey << 12) | (unsigned)item;
                    cnt0 += __popcll(m0);
                    cnt1 += __popcll(m1);
                }
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");   // the lists are written and read by this wave only
                // add: two rows per step, the pair after them already on its way; both table values are read before
                // either is written, and when the two rows hit the same code the first sum is forwarded in registers,
                // so the adds of one code stay strictly in row order
                int n = (half ? cnt1 : cnt0) - lo;
                n = n < 0 ? 0 : (n > kListCap ? kListCap : n);
                if (RQ_PROBE(1)) n = 0;
                unsigned e0 = mylist[0], e1 = mylist[1];
                for (int i = 0; i < n; i += 2) {
                    const unsigned ne0 = mylist[i + 2], ne1 = mylist[i + 3];
                    const bool two = i + 1 < n;
                    const int ka = (int)(e0 >> 12), ja = (int)(e0 & 4095u);
                    const int kb = two ? (int)(e1 >> 12) : ka, jb = two ? (int)(e1 & 4095u) : ja;
                    if (f < D) {
                        float *ta = acc + (size_t)ka * D + f, *tb = acc + (size_t)kb * D + f;
                        const float t_a = *ta, s_a = stage[(size_t)ja * D + f];
                        const float t_b = *tb, s_b = stage[(size_t)jb * D + f];
                        const float va = t_a + s_a;
                        const float vb = ((kb == ka) ? va : t_b) + s_b;
                        *ta = va;
                        if (two) *tb = vb;
                    }
                    e0 = ne0;
                    e1 = ne1;
                }
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                if ((cnt0 > cnt1 ? cnt0 : cnt1) <= lo + kListCap) break;
            }
        }
        if (p.g_cb) __syncthreads();
      }
    }

    if (p.g_cb) {
        float *out = partial + (size_t)blockIdx.x * LKD_total;
        if constexpr (MM) {
            // cacc[li][nb][r]: code 32 (ow + 8 nb) + 8 (r >> 2) + 4 (lane >> 5) + (r & 3), feature lane & 31
            if (ow >= 0) {
                const int il = lane & 31, o = lane >> 5;
#pragma unroll
                for (int li = 0; li < MMP; ++li)
#pragma unroll
                    for (int nb = 0; nb < MMB; ++nb)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int code = 32 * (ow + kFlatOwnerWaves * nb) + 8 * (r >> 2) + 4 * o + (r & 3);
                            if (code < K) out[((size_t)li * K + code) * 32 + il] = cacc[li][nb][r];
                        }
            }
        } else {
            for (int e = threadIdx.x * 4; e < (RQ_PROBE(8) ? 0 : tbl); e += kFlatThreads * 4)
                *reinterpret_cast<f32x4 *>(out + e) = *reinterpret_cast<const f32x4 *>(acc + e);
        }
    }
}

// ---- kernel 2: LDS-private scatter of V into per-workgroup codebook-gradient tables -----------------------
// thread (rs, d): rs = row slot inside the workgroup's step, d = feature.  DR = D rounded up to a power of 2.
__global__ __launch_bounds__(256) void rq_cbgrad_scatter_kernel(const float *__restrict__ V,
                                                                const int64_t *__restrict__ ids, long long B, int D,
                                                                int DR, int K, int l0, int nl, long long rows_per_wg,
                                                                float *__restrict__ partial, int LKD_total) {
    extern __shared__ __attribute__((aligned(16))) float acc[];
    const int stride = D + 1;
    const int tbl = nl * K * stride;
    for (int e = threadIdx.x; e < tbl; e += 256) acc[e] = 0.0f;
    __syncthreads();
    const int d = threadIdx.x % DR, rs = threadIdx.x / DR, rstep = 256 / DR;
    const long long r0 = (long long)blockIdx.x * rows_per_wg;
    const long long r1 = (r0 + rows_per_wg < B) ? r0 + rows_per_wg : B;
    if (d < D) {
        for (long long row = r0 + rs; row < r1; row += rstep) {
            for (int l = 0; l < nl; ++l) {
                const int id = (int)ids[(size_t)(l0 + l) * B + row];
                const float v = V[((size_t)(l0 + l) * B + row) * D + d];
                atomicAdd(&acc[(l * K + id) * stride + d], v);  // ds_add_f32
            }
        }
    }
    __syncthreads();
    float *out = partial + (size_t)blockIdx.x * LKD_total + (size_t)l0 * K * D;
    for (int e = threadIdx.x; e < nl * K * D; e += 256) {
        const int kd = e / D, dd = e - kd * D;
        out[e] = acc[kd * stride + dd];
    }
}

// ---- kernel 2s: the same sum for a SMALL batch, one workgroup per level, no partial tables ------------------------------
// The scatter kernel above zeroes, fills and writes out a [levels, K, D + 1] LDS table per workgroup and a third kernel adds the
// workgroups' tables: for the 64 rows of the reference's rqvae_ml32m.gin that is 2 x 18 + 5 us of moving zeros
// (profiles/r05_small_batch_kernels.txt).  Here workgroup l keeps level l's [K, D] table, walks the batch in blocks of 256 rows and
// adds every row's vector to its code's table row with the fused kernel's owner-computes scheme (cb_accumulate: wave id mod 8
// owns the code, rows in ascending order -- a fixed summation order, no atomics), then writes the table to g_cb[l] itself.
constexpr size_t kScatterLdsBudgetSmall = 150 * 1024;
constexpr int kSmallCbRows = 256;
constexpr long long kSmallCbMaxB = 2048;
static size_t small_cb_lds(int D, int K) {
    return ((size_t)K * D + (size_t)kSmallCbRows * (D + 1)) * sizeof(float) + (size_t)kSmallCbRows * sizeof(int);
}
static bool small_cb_fits(long long B, int D, int K) { return B <= kSmallCbMaxB && D <= 64 && small_cb_lds(D, K) <= kScatterLdsBudgetSmall; }

__global__ __launch_bounds__(kFusedThreads) void rq_cbgrad_small_kernel(const float *__restrict__ V, const int64_t *__restrict__ ids,
                                                                        long long B, int D, int K, float *__restrict__ g_cb) {
    extern __shared__ __attribute__((aligned(16))) float small_smem[];
    float *tab = small_smem;                                   // [K][D]
    float *stage = tab + (size_t)K * D;                        // [kSmallCbRows][D + 1]
    int *ids_s = reinterpret_cast<int *>(stage + (size_t)kSmallCbRows * (D + 1));
    const int l = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // (K * D is a multiple of 4 and the table 16-byte aligned whenever D is a multiple of 4; else one float at a time)
    const bool by4 = (D & 3) == 0;
    if (by4) {
        for (int e = threadIdx.x; e < K * D / 4; e += kFusedThreads) reinterpret_cast<rq_f32x4 *>(tab)[e] = rq_f32x4{0.f, 0.f, 0.f, 0.f};
    } else {
        for (int e = threadIdx.x; e < K * D; e += kFusedThreads) tab[e] = 0.0f;
    }
    for (long long base = 0; base < B; base += kSmallCbRows) {
        const int rows = (int)((B - base < kSmallCbRows) ? B - base : kSmallCbRows);
        __syncthreads();   // the table is zeroed / the previous block's rows are consumed
        for (int e = threadIdx.x; e < rows * D; e += kFusedThreads) {
            const int r = e / D, d = e - r * D;
            stage[(size_t)r * (D + 1) + d] = V[((size_t)l * B + base + r) * D + d];
        }
        for (int r = threadIdx.x; r < rows; r += kFusedThreads) ids_s[r] = (int)ids[(size_t)l * B + base + r];
        __syncthreads();
        cb_accumulate(tab, stage, ids_s, rows, D, wave, lane);
    }
    __syncthreads();
    float *out = g_cb + (size_t)l * K * D;
    if (by4 && (reinterpret_cast<uintptr_t>(out) & 15u) == 0) {
        for (int e = threadIdx.x; e < K * D / 4; e += kFusedThreads)
            reinterpret_cast<rq_f32x4 *>(out)[e] = reinterpret_cast<const rq_f32x4 *>(tab)[e];
    } else {
        for (int e = threadIdx.x; e < K * D; e += kFusedThreads) out[e] = tab[e];
    }
}

// ---- kernel 3: fixed-order sum of the per-workgroup partials ------------------------------------------------
// 64 outputs per 256-thread block; thread (seg, j) sums partials g = seg, seg+4, ... in ascending order, the four
// segment sums are combined as ((s0 + s1) + (s2 + s3)).
__global__ __launch_bounds__(256) void rq_cbgrad_reduce_kernel(const float *__restrict__ partial, int G, int n,
                                                               float *__restrict__ out) {
    __shared__ float seg_sum[4][64];
    const int jl = threadIdx.x & 63, seg = threadIdx.x >> 6;
    const int j = blockIdx.x * 64 + jl;
    float s = 0.0f;
    if (j < n) {
#pragma unroll 8
        for (int g = seg; g < G; g += 4) s = s + partial[(size_t)g * n + j];
    }
    seg_sum[seg][jl] = s;
    __syncthreads();
    if (seg == 0 && j < n) out[j] = (seg_sum[0][jl] + seg_sum[1][jl]) + (seg_sum[2][jl] + seg_sum[3][jl]);
}

constexpr size_t kScatterLdsBudget = 150 * 1024;
constexpr int kMaxScatterWgs = 128;

static int fused_wgs(long long B) {
    long long g = ((B + 31) / 32 + 7) / 8;
    const long long cap = cu_count();
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}

// fused kernel LDS: [levels of the launch][K][D] table + a stage of sg waves x 32 rows x (D+1) floats + their ids
constexpr size_t kFusedLdsBudget = 160 * 1024;
static size_t fused_stage_bytes(int D, int sg) { return (size_t)sg * 32 * (D + 1) * sizeof(float) + (size_t)sg * 32 * sizeof(int); }
// register budget: L * KSTEPS residual values per lane -> D <= 32 (KSTEPS <= 16) only
static bool fused_fits(int D, int K, int L) {
    return D <= 32 && L <= kFusedMaxL && (size_t)K * D * sizeof(float) + fused_stage_bytes(D, 1) <= kFusedLdsBudget;
}
// waves staged at a time: as many as fit next to ONE level's table (8, 4, 2 or 1)
static int fused_stage_waves(int D, int K) {
    int sg = kFusedWaves;
    while (sg > 1 && (size_t)K * D * sizeof(float) + fused_stage_bytes(D, sg) > kFusedLdsBudget) sg >>= 1;
    return sg;
}
// levels whose LDS tables fit together: the fused kernel runs once per such group
static int fused_levels_per_pass(int D, int K, int L) {
    const size_t left = kFusedLdsBudget - fused_stage_bytes(D, fused_stage_waves(D, K));
    int n = (int)(left / ((size_t)K * D * sizeof(float)));
    return n < 1 ? 1 : (n > L ? L : n);
}

// flat kernel (EVAL / STE): rows per workgroup and round, LDS per level of the launch, launch geometry
static bool flat_shape_ok(int D, int L) { return D % 4 == 0 && D <= kFlatMaxD && L <= kFusedMaxL; }
// (the flat kernel addresses rows with 32-bit byte offsets)
static bool flat_offsets_ok(long long B, int D) {
    return (unsigned long long)B * 8ull <= 0xffffffffull && (unsigned long long)B * D * 4ull <= 0xffffffffull;
}
static int flat_rows(int D) { return kFlatRowThreads / (D / 4); }
static size_t flat_level_bytes(int D, int K) {   // one level's table + its share of the two stage buffers and key arrays
    const size_t R = flat_rows(D);
    return (size_t)K * D * sizeof(float) + 2 * (R * D * sizeof(float) + R * sizeof(int));
}
static bool flat_fits(int D, int K, int L) {
    return flat_shape_ok(D, L) && flat_level_bytes(D, K) + kFlatListBytes <= kFusedLdsBudget;
}
static int flat_levels_per_pass(int D, int K, int L) {
    const int n = (int)((kFusedLdsBudget - kFlatListBytes) / flat_level_bytes(D, K));
    return n < 1 ? 1 : (n > L ? L : n);
}
static int flat_wgs(long long B, int D) {
    const long long R = flat_rows(D);
    long long g = (B + R - 1) / R;
    const long long cap = cu_count();
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}

static int scatter_wgs(long long B) {
    long long g = (B + 255) / 256;
    if (g > kMaxScatterWgs) g = kMaxScatterWgs;
    if (g < 1) g = 1;
    return (int)g;
}

static bool scatter_fits_lds(int D, int K) { return (size_t)K * (D + 1) * sizeof(float) <= kScatterLdsBudget; }

template <int KSTEPS, bool VEC>
static int launch_bwd_v(const RqBwdParams &p, int mode, int grid, hipStream_t s) {
    switch (mode) {
        case RQHIP_MODE_EVAL:
            hipLaunchKernelGGL((rq_backward_kernel<KSTEPS, RQHIP_MODE_EVAL, VEC>), dim3(grid), dim3(256), 0, s, p);
            break;
        case RQHIP_MODE_STE:
            hipLaunchKernelGGL((rq_backward_kernel<KSTEPS, RQHIP_MODE_STE, VEC>), dim3(grid), dim3(256), 0, s, p);
            break;
        case RQHIP_MODE_ROTATION:
            hipLaunchKernelGGL((rq_backward_kernel<KSTEPS, RQHIP_MODE_ROTATION, VEC>), dim3(grid), dim3(256), 0, s, p);
            break;
        default:
            set_error("rq_backward: unsupported mode %d", mode);
            return RQHIP_EARG;
    }
    RQ_CHECK_LAUNCH("rq_backward_kernel");
    return 0;
}
// vec: D == 2 * KSTEPS and every row pointer of the call 16-byte aligned
template <int KSTEPS>
static int launch_bwd(const RqBwdParams &p, int mode, int grid, bool vec, hipStream_t s) {
    return vec ? launch_bwd_v<KSTEPS, true>(p, mode, grid, s) : launch_bwd_v<KSTEPS, false>(p, mode, grid, s);
}

}  // namespace rqhip

using namespace rqhip;

// which path rqhip_rq_backward takes for 16-byte aligned tensors: returns 1 and the geometry that fixes the summation
// order of the codebook gradient (workgroups, row units per workgroup and round, rows per unit) when it is accumulated
// in the order restated by the oracle, 0 for the three-kernel path
extern "C" int rqhip_rq_backward_plan(int64_t B, int D, int L, int K, int mode, int *n_wg, int *units_per_wg,
                                      int *unit_rows) {
    const bool ok = B > 0 && D >= 1 && K >= 1 && L >= 1;
    const bool flat = ok && mode != RQHIP_MODE_ROTATION && flat_fits(D, K, L) && flat_offsets_ok(B, D);
    const bool fused = ok && !flat && fused_fits(D, K, L);
    if (n_wg) *n_wg = flat ? flat_wgs(B, D) : fused ? fused_wgs(B) : 0;
    if (units_per_wg) *units_per_wg = flat ? 1 : fused ? kFusedWaves : 0;
    if (unit_rows) *unit_rows = flat ? flat_rows(D) : fused ? 32 : 0;
    return (flat || fused) ? 1 : 0;
}

// layout: [L,B,D] row scratch | [G, L*K*D] per-workgroup partial tables (LDS scatter path only)
extern "C" size_t rqhip_rq_backward_workspace_bytes(int64_t B, int D, int L, int K) {
    if (B <= 0 || D <= 0 || L <= 0 || K <= 0) return 16;
    const size_t rows = (size_t)L * (size_t)B * (size_t)D * sizeof(float);
    size_t g = 0;   // per-workgroup partial tables: the most any path of this shape launches
    if (fused_fits(D, K, L)) g = (size_t)fused_wgs(B);
    if (flat_fits(D, K, L) && (size_t)flat_wgs(B, D) > g) g = (size_t)flat_wgs(B, D);
    if (scatter_fits_lds(D, K) && (size_t)scatter_wgs(B) > g) g = (size_t)scatter_wgs(B);
    return rows + g * (size_t)L * K * D * sizeof(float);
}

static int rq_backward_impl(const float *res0, int64_t B, int D, const float *codebooks, int L, int K,
                            int mode, float beta, const int64_t *ids, const float *g_embs,
                            const float *g_embsum, const float *g_resid, const float *g_loss,
                            float *g_res0, float *g_codebooks, void *workspace, size_t workspace_bytes,
                            unsigned flags, rqhip_stream_t stream);

extern "C" int rqhip_rq_backward(const float *res0, int64_t B, int D, const float *codebooks, int L, int K,
                                 int mode, float beta, const int64_t *ids, const float *g_embs,
                                 const float *g_embsum, const float *g_resid, const float *g_loss,
                                 float *g_res0, float *g_codebooks, void *workspace, size_t workspace_bytes,
                                 rqhip_stream_t stream) {
    // (bench only: one profile record for all kernels of the call; algorithmic bytes 12 D + 8 L per row)
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    profile_begin(s, RQHIP_PROF_RQ_BACKWARD, (double)B * L * 3.0 * D, (double)B * (12.0 * D + 8.0 * L));
    const int rc = rq_backward_impl(res0, B, D, codebooks, L, K, mode, beta, ids, g_embs, g_embsum, g_resid, g_loss, g_res0,
                                    g_codebooks, workspace, workspace_bytes, 0u, stream);
    profile_end(s);
    return rc;
}

// which (levels per launch, code blocks per owner wave) the matrix-form codebook gradient runs a shape with; 0: not at all
static int mm_levels_per_pass(int D, int K, int L, int mode, bool train) {
    if (D != 32 || mode != RQHIP_MODE_STE || !train || K % 32 != 0) return 0;
    if (K <= 256 && L == 3) return 3;                 // 3 x 256: all levels in one launch
    if (K == 1024 && (L == 3 || L == 4)) return 1;    // 1024 codes: four blocks per owner wave, one level per launch
    return 0;
}
extern "C" int rqhip_rq_backward_matrix_form(int D, int K, int L, int mode) { return mm_levels_per_pass(D, K, L, mode, true) ? 1 : 0; }

extern "C" int rqhip_rq_backward_ex(const float *res0, int64_t B, int D, const float *codebooks, int L, int K,
                                    int mode, float beta, const int64_t *ids, const float *g_embs,
                                    const float *g_embsum, const float *g_resid, const float *g_loss,
                                    float *g_res0, float *g_codebooks, void *workspace, size_t workspace_bytes,
                                    unsigned flags, rqhip_stream_t stream) {
    if (flags & ~RQHIP_BWD_CBGRAD_MATRIX) {
        set_error("rq_backward_ex: unknown flags 0x%x", flags);
        return RQHIP_EARG;
    }
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    profile_begin(s, RQHIP_PROF_RQ_BACKWARD, (double)B * L * 3.0 * D, (double)B * (12.0 * D + 8.0 * L));
    const int rc = rq_backward_impl(res0, B, D, codebooks, L, K, mode, beta, ids, g_embs, g_embsum, g_resid, g_loss, g_res0,
                                    g_codebooks, workspace, workspace_bytes, flags, stream);
    profile_end(s);
    return rc;
}

static int rq_backward_impl(const float *res0, int64_t B, int D, const float *codebooks, int L, int K,
                            int mode, float beta, const int64_t *ids, const float *g_embs,
                            const float *g_embsum, const float *g_resid, const float *g_loss,
                            float *g_res0, float *g_codebooks, void *workspace, size_t workspace_bytes,
                            unsigned flags, rqhip_stream_t stream) {
    if (B < 0 || !codebooks || (B > 0 && (!res0 || !ids))) {
        set_error("rq_backward: null pointer or negative B");
        return RQHIP_EARG;
    }
    if (D < 1 || D > 128 || K < 1 || K > 65536 || L < 1 || L > 16) {
        set_error("rq_backward: unsupported shape D=%d K=%d L=%d", D, K, L);
        return RQHIP_EUNSUPPORTED;
    }
    if (mode != RQHIP_MODE_EVAL && mode != RQHIP_MODE_STE && mode != RQHIP_MODE_ROTATION) {
        set_error("rq_backward: mode %d is not EVAL/STE/ROTATION", mode);
        return RQHIP_EARG;
    }
    if (B > 0 && (!workspace || workspace_bytes < rqhip_rq_backward_workspace_bytes(B, D, L, K))) {
        set_error("rq_backward: workspace too small");
        return RQHIP_EWORKSPACE;
    }
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const bool lds_path = scatter_fits_lds(D, K);
    if (g_codebooks && (B == 0 || !lds_path))
        if (int rc = fill_words(g_codebooks, 0u, sizeof(float) * (size_t)L * K * D, s)) return rc;
    if (B == 0) return RQHIP_OK;
    RqBwdParams p;
    p.res0 = res0; p.cb = codebooks; p.ids = ids; p.g_embs = g_embs; p.g_embsum = g_embsum;
    p.g_resid = g_resid; p.g_loss = g_loss; p.g_res0 = g_res0; p.g_cb = g_codebooks;
    p.ws = reinterpret_cast<float *>(workspace);
    p.B = B; p.n_tiles = (B + 31) / 32; p.D = D; p.L = L; p.K = K; p.beta = beta;
    p.atomic_scatter = lds_path ? 0 : 1;
#ifdef RQ_BWD_PROBE
    p.probe = getenv("RQ_BWD_PROBE") ? atoi(getenv("RQ_BWD_PROBE")) : 0;
#endif
    auto al16 = [](const void *q) { return (reinterpret_cast<uintptr_t>(q) & 15u) == 0; };
    if (mode != RQHIP_MODE_ROTATION && flat_fits(D, K, L) && flat_offsets_ok(B, D) && al16(res0) && al16(codebooks) &&
        al16(g_embs) && al16(g_embsum) && al16(g_resid) && al16(g_res0) && al16(workspace)) {
        const int R = flat_rows(D), LPR = D / 4;
        const int G = flat_wgs(B, D);
        float *partial = p.ws + (size_t)L * (size_t)B * (size_t)D;
        const bool train_shape = g_embsum && g_loss && !g_embs && !g_resid;
        // the matrix form of the codebook gradient (RQHIP_BWD_CBGRAD_MATRIX; see the kernel): levels per launch, 0 = the ordered form
        const int mmp = (g_codebooks && (flags & RQHIP_BWD_CBGRAD_MATRIX)) ? mm_levels_per_pass(D, K, L, mode, train_shape) : 0;
        const int per_pass = mmp ? mmp : (g_codebooks ? flat_levels_per_pass(D, K, L) : L);
        for (int l0 = 0; l0 < L; l0 += per_pass) {
            p.l_begin = l0;
            p.l_end = (l0 + per_pass < L) ? l0 + per_pass : L;
            p.write_rows = (l0 == 0);
            const int nl = p.l_end - p.l_begin;
            const int LKD = nl * K * D;
            size_t lds = g_codebooks ? (size_t)nl * flat_level_bytes(D, K) + kFlatListBytes : 0;
            if (mmp)   // two stage buffers + keys, two sets of B-operand images + keys (no table)
                lds = (size_t)2 * nl * (R * D * sizeof(float) + R * sizeof(int)) + (size_t)2 * nl * 4 * (kMmPairOps * 16 + 16 * sizeof(int));
            auto go = [&](auto kern) -> int {
                static LdsGrant grant;
                RQ_RETURN_IF_HIP(grant.ensure(reinterpret_cast<const void *>(kern), (int)kFusedLdsBudget));
                hipLaunchKernelGGL(kern, dim3(G), dim3(kFlatThreads), lds, s, p, partial, LKD, R, LPR);
                RQ_CHECK_LAUNCH("rq_backward_flat_kernel");
                return 0;
            };
            // the training step's shape of upstream gradients and the two level counts of the named configurations get
            // their own instantiations; everything else runs the generic one
            const bool train = train_shape;
            const int nlv = (train && (L == 3 || L == 4)) ? L : 0;
            int rcf;
            if (mmp) {
                rcf = (mmp == 3) ? go(rq_backward_flat_kernel<RQHIP_MODE_STE, true, 3, true, 3, 1>)
                    : (L == 4)   ? go(rq_backward_flat_kernel<RQHIP_MODE_STE, true, 4, true, 1, 4>)
                                 : go(rq_backward_flat_kernel<RQHIP_MODE_STE, true, 3, true, 1, 4>);
            } else
#define RQ_FLAT_GO(M, P)                                                                                              \
    (nlv == 3 ? go(rq_backward_flat_kernel<M, P, 3, true>)                                                            \
              : nlv == 4 ? go(rq_backward_flat_kernel<M, P, 4, true>) : go(rq_backward_flat_kernel<M, P, 0, false>))
            if (mode == RQHIP_MODE_EVAL)
                rcf = D <= 32 ? RQ_FLAT_GO(RQHIP_MODE_EVAL, true) : RQ_FLAT_GO(RQHIP_MODE_EVAL, false);
            else
                rcf = D <= 32 ? RQ_FLAT_GO(RQHIP_MODE_STE, true) : RQ_FLAT_GO(RQHIP_MODE_STE, false);
#undef RQ_FLAT_GO
            if (rcf) return rcf;
            if (!g_codebooks) break;  // nothing to scatter: the first launch has written g_res0
            hipLaunchKernelGGL(rq_cbgrad_reduce_kernel, dim3((LKD + 63) / 64), dim3(256), 0, s, partial, G, LKD,
                               g_codebooks + (size_t)l0 * K * D);
            RQ_CHECK_LAUNCH("rq_cbgrad_reduce_kernel");
        }
        return RQHIP_OK;
    }
    if (fused_fits(D, K, L)) {
        const int G = fused_wgs(B);
        float *partial = p.ws + (size_t)L * (size_t)B * (size_t)D;
        // (D = 128 with the rotation trick keeps the dword form: the row form spills more there)
    const bool vec = !(D > 64 && mode == RQHIP_MODE_ROTATION) && D == 2 * ksteps_for(D) && al16(res0) && al16(codebooks) && al16(g_embs) && al16(g_embsum) &&
                         al16(g_resid) && al16(g_res0);
        // one launch per group of levels whose tables fit LDS together (all of them for 3 x 256 x 32; one level at a
        // time for K = 1024): every launch replays the cheap register chain, the first one writes g_res0
        const int per_pass = g_codebooks ? fused_levels_per_pass(D, K, L) : L;
        for (int l0 = 0; l0 < L; l0 += per_pass) {
            p.l_begin = l0;
            p.l_end = (l0 + per_pass < L) ? l0 + per_pass : L;
            p.write_rows = (l0 == 0);
            const int nl = p.l_end - p.l_begin;
            const int LKD = nl * K * D;
            const int sg = fused_stage_waves(D, K);
            const size_t lds = g_codebooks ? (size_t)nl * K * D * sizeof(float) + fused_stage_bytes(D, sg) : 0;
            auto go = [&](auto kern) -> int {
                static LdsGrant grant;
                RQ_RETURN_IF_HIP(grant.ensure(reinterpret_cast<const void *>(kern), (int)kFusedLdsBudget));
                hipLaunchKernelGGL(kern, dim3(G), dim3(kFusedThreads), lds, s, p, partial, LKD, sg);
                RQ_CHECK_LAUNCH("rq_backward_fused_kernel");
                return 0;
            };
            int rcf = RQHIP_EARG;
#define RQ_FUSED_MODES(KS)                                                                                            \
    switch (mode) {                                                                                                   \
        case RQHIP_MODE_EVAL:                                                                                         \
            rcf = vec ? go(rq_backward_fused_kernel<KS, RQHIP_MODE_EVAL, true>)                                       \
                      : go(rq_backward_fused_kernel<KS, RQHIP_MODE_EVAL, false>);                                     \
            break;                                                                                                    \
        case RQHIP_MODE_STE:                                                                                          \
            rcf = vec ? go(rq_backward_fused_kernel<KS, RQHIP_MODE_STE, true>)                                        \
                      : go(rq_backward_fused_kernel<KS, RQHIP_MODE_STE, false>);                                      \
            break;                                                                                                    \
        default:                                                                                                      \
            rcf = vec ? go(rq_backward_fused_kernel<KS, RQHIP_MODE_ROTATION, true>)                                   \
                      : go(rq_backward_fused_kernel<KS, RQHIP_MODE_ROTATION, false>);                                 \
            break;                                                                                                    \
    }
            switch (ksteps_for(D)) {
                case 4: RQ_FUSED_MODES(4) break;
                case 8: RQ_FUSED_MODES(8) break;
                default: RQ_FUSED_MODES(16) break;
            }
#undef RQ_FUSED_MODES
            if (rcf) return rcf;
            if (!g_codebooks) break;  // nothing to scatter: the first launch has written g_res0
            hipLaunchKernelGGL(rq_cbgrad_reduce_kernel, dim3((LKD + 63) / 64), dim3(256), 0, s, partial, G, LKD,
                               g_codebooks + (size_t)l0 * K * D);
            RQ_CHECK_LAUNCH("rq_cbgrad_reduce_kernel");
        }
        return RQHIP_OK;
    }

    long long want = (p.n_tiles + 3) / 4;
    long long cap = (long long)cu_count() * 8;
    const int grid = (int)(want < cap ? want : cap);
    int rc;
    const bool vec = D == 2 * ksteps_for(D) && al16(res0) && al16(codebooks) && al16(g_embs) && al16(g_embsum) && al16(g_resid) &&
                     al16(g_res0) && al16(workspace);
    switch (ksteps_for(D)) {
        case 4: rc = launch_bwd<4>(p, mode, grid, vec, s); break;
        case 8: rc = launch_bwd<8>(p, mode, grid, vec, s); break;
        case 16: rc = launch_bwd<16>(p, mode, grid, vec, s); break;
        case 32: rc = launch_bwd<32>(p, mode, grid, vec, s); break;
        default: rc = launch_bwd<64>(p, mode, grid, vec, s); break;
    }
    if (rc || !g_codebooks || !lds_path) return rc;

    if (small_cb_fits(B, D, K)) {   // a small batch: one workgroup per level adds its rows' vectors in row order, no partial tables
        static LdsGrant small_grant;
        RQ_RETURN_IF_HIP(small_grant.ensure(reinterpret_cast<const void *>(rq_cbgrad_small_kernel), (int)kScatterLdsBudgetSmall));
        hipLaunchKernelGGL(rq_cbgrad_small_kernel, dim3(L), dim3(kFusedThreads), small_cb_lds(D, K), s, p.ws, ids, (long long)B, D, K,
                           g_codebooks);
        RQ_CHECK_LAUNCH("rq_cbgrad_small_kernel");
        return RQHIP_OK;
    }
    // embedding backward: LDS-private scatter in groups of whole levels, then a fixed-order reduce
    const int G = scatter_wgs(B);
    const long long rows_per_wg = (B + G - 1) / G;
    const int LKD = L * K * D;
    float *partial = p.ws + (size_t)L * (size_t)B * (size_t)D;
    int DR = 1;
    while (DR < D) DR <<= 1;
    const size_t level_bytes = (size_t)K * (D + 1) * sizeof(float);
    int per_pass = (int)(kScatterLdsBudget / level_bytes);
    if (per_pass < 1) per_pass = 1;
    static LdsGrant scatter_grant;
    RQ_RETURN_IF_HIP(scatter_grant.ensure(reinterpret_cast<const void *>(rq_cbgrad_scatter_kernel), (int)kScatterLdsBudget));
    for (int l0 = 0; l0 < L; l0 += per_pass) {
        const int nl = (L - l0 < per_pass) ? L - l0 : per_pass;
        hipLaunchKernelGGL(rq_cbgrad_scatter_kernel, dim3(G), dim3(256), nl * level_bytes, s, p.ws, ids, (long long)B,
                           D, DR, K, l0, nl, rows_per_wg, partial, LKD);
        RQ_CHECK_LAUNCH("rq_cbgrad_scatter_kernel");
    }
    hipLaunchKernelGGL(rq_cbgrad_reduce_kernel, dim3((LKD + 63) / 64), dim3(256), 0, s, partial, G, LKD, g_codebooks);
    RQ_CHECK_LAUNCH("rq_cbgrad_reduce_kernel");
    return RQHIP_OK;
}
