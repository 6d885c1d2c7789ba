/*
 * rq_oracle.c -- CPU restatement of the reference's residual-quantisation hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and only as
 * the checker (or as the timed CPU baseline), never as a compute fallback.
 *
 * What it restates (all citations relative to /root/reference):
 *   modules/quantize.py:104-163   Quantize.forward   (L2 distance, argmin, STE / rotation / eval / gumbel)
 *   modules/quantize.py:34-50     efficient_rotation_trick_transform
 *   modules/loss.py:33-41         QuantizeLoss
 *   modules/rqvae.py:118-139      RqVae.get_semantic_ids (the level loop)
 *   modules/rqvae.py:156-167      embs_norm, p_unique_ids
 *   distributions/gumbel.py:8-20  sample_gumbel / gumbel_softmax_sample
 *   init/kmeans.py:33-72          Kmeans (assign, update, stop test)
 *   modules/tokenizer/semids.py:92-108  dedup column of precompute_corpus_ids
 *   modules/model.py:169-182      _check_valid_prefix (valid-prefix mask of the beam search)
 *   evaluate/metrics.py:16-25     TopKAccumulator.accumulate (first-match rank)
 *
 * The reference computes these with PyTorch CPU ops whose floating-point reduction order is
 * not specified (MKL sgemm, vectorised sums).  This file FIXES an order for every reduction --
 * the same order the HIP kernels use -- so that GPU output can be compared to it bit for bit:
 *
 *   dot(x,c)    : one fp32 FMA chain over d = 0..D-1 starting from 0 (what v_mfma_f32_32x32x2_f32
 *                 computes); the reference's factor 2 (`2 * x @ codebook.T`) multiplies the finished dot.
 *   sumsq(v)    : two accumulators by the parity of d, separately rounded multiply and add
 *                 (the reference's `(v**2).sum()` squares then sums), combined as a0 + a1.
 *   dotp(a,b)   : two FMA accumulators by the parity of d, combined as a0 + a1 (rotation trick).
 *   everything else is elementwise and follows the reference's operator order literally.
 *
 * Parity is pinned (tests/test_oracle_golden.py) against the tests/golden npz fixtures, which were produced by
 * running the reference itself, imported from /root/reference, by oracle/gen_golden.py.
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off -fno-fast-math, no -march flags that
 * would licence other contractions; fmaf() is the only fused operation and is always explicit).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define RQO_OK 0
#define RQO_EARG (-1)
#define RQO_ENOMEM (-2)

enum { RQO_MODE_EVAL = 0, RQO_MODE_STE = 1, RQO_MODE_ROTATION = 2, RQO_MODE_GUMBEL = 3 };

/* ---- fixed-order reductions ------------------------------------------------------------ */

static float sumsq2(const float *v, int D) {
    float a0 = 0.0f, a1 = 0.0f;
    for (int d = 0; d < D; ++d) {
        float p = v[d] * v[d];
        if (d & 1) a1 = a1 + p; else a0 = a0 + p;
    }
    return a0 + a1;
}

static float dotp2(const float *a, const float *b, int D) {
    float a0 = 0.0f, a1 = 0.0f;
    for (int d = 0; d < D; ++d) {
        if (d & 1) a1 = fmaf(a[d], b[d], a1); else a0 = fmaf(a[d], b[d], a0);
    }
    return a0 + a1;
}

/* x . c as one FMA chain over d.  quantize.py:116 computes `2 * x @ codebook.T` == (2x) @ C^T; scaling by 2
 * is exact in binary floating point, so 2 * chain(x, c) is the same number (bar overflow / subnormal partial
 * sums, where the reference's BLAS result is unspecified anyway) -- and lets the matrix pipe consume x as is. */
static float dot_chain(const float *x, const float *c, int D) {
    float acc = 0.0f;
    for (int d = 0; d < D; ++d) acc = fmaf(x[d], c[d], acc);
    return acc;
}

float rqo_sumsq2(const float *v, int D) { return sumsq2(v, D); }

/* torch.min(dim) on CPU: scan ascending, take v when !(v >= best), stop at the first NaN
 * (quantize.py:128; first-index tie-break, a NaN wins). */
static int64_t argmin_torch(const float *v, int n) {
    float best = v[0];
    int64_t idx = 0;
    if (best != best) return 0;
    for (int k = 1; k < n; ++k) {
        float t = v[k];
        if (!(t >= best)) {
            best = t; idx = k;
            if (t != t) break;
        }
    }
    return idx;
}

/* dist[k] = (|x|^2 + |c_k|^2) - (2x).c_k   (quantize.py:112-117, left-to-right) */
static void l2_dist_row(const float *x, const float *cb, const float *csq, int K, int D, float *dist) {
    float xsq = sumsq2(x, D);
    for (int k = 0; k < K; ++k) {
        float t = xsq + csq[k];
        dist[k] = t - 2.0f * dot_chain(x, cb + (size_t)k * D, D);
    }
}

/* rotation trick for one row (quantize.py:34-50,140-153); e = x carries the gradient, the rest is
 * detached.  Writes emb_out; optionally returns w,u(=x-hat),q(=emb-hat) and the scale for backward. */
static void rotation_row(const float *x, const float *emb, int D, float *out,
                         float *w_o, float *u_o, float *q_o, float *scale_o, float *tmp) {
    float nx = sqrtf(sumsq2(x, D));
    float ne = sqrtf(sumsq2(emb, D));
    float du = nx + 1e-8f, dq = ne + 1e-8f;
    float *u = tmp, *q = tmp + D, *w = tmp + 2 * D;
    for (int d = 0; d < D; ++d) { u[d] = x[d] / du; q[d] = emb[d] / dq; w[d] = u[d] + q[d]; }
    float nw = sqrtf(sumsq2(w, D));
    float den = nw > 1e-6f ? nw : 1e-6f;           /* F.normalize(eps=1e-6): v / max(|v|, eps) */
    for (int d = 0; d < D; ++d) w[d] = w[d] / den;
    float ew = dotp2(x, w, D);
    float eu = dotp2(x, u, D);
    float scale = ne / (nx + 1e-6f);
    for (int d = 0; d < D; ++d) {
        float t1 = ew * w[d];
        float t2 = eu * q[d];
        float o = (x[d] - 2.0f * t1) + 2.0f * t2;
        out[d] = o * scale;
    }
    if (w_o) memcpy(w_o, w, sizeof(float) * D);
    if (u_o) memcpy(u_o, u, sizeof(float) * D);
    if (q_o) memcpy(q_o, q, sizeof(float) * D);
    if (scale_o) *scale_o = scale;
}

/* ---- forward: L levels of Quantize.forward chained as RqVae.get_semantic_ids does --------- */
/*
 * res0      [B,D]      encoder output (level-0 input)
 * codebooks [L,K,D]    out_proj(embedding.weight) of each level
 * mode      EVAL / STE / ROTATION  (GUMBEL has its own entry point: it needs the noise)
 * ids       [L,B] int64   (the reference's sem_ids [B,L] has strides (1,B): same memory)
 * embs      [L,B,D] or NULL   quantized.embeddings (emb_out) per level
 * residuals [L,B,D] or NULL   the input of each level
 * emb_sum   [B,D]   or NULL   sum over levels of emb_out, ((e0+e1)+e2)...   (rqvae.py:146)
 * loss      [B]     or NULL   sum over levels of QuantizeLoss               (rqvae.py:128)
 * embs_norm [B,L]   or NULL   ||emb_out_l||_2                               (rqvae.py:158)
 */
/* tie_margin [L,B] or NULL: how decisively each level's argmin was taken (SURVEY.md section 7 "hard parts",
 *   section 8b `tie_margin_flags`).  margin = (d2 - d1) / (|x|^2 + |c_id|^2) with d1 = dist[id] and
 *   d2 = min over k != id of dist[k] (a duplicate of the minimum counts: margin 0); 0 for rows whose
 *   distances may be non-finite (|x|^2 + max_k|c_k|^2 not < 1e38, the kernel's exact-scan rows) and whenever the
 *   quotient is NaN; +Inf when K == 1.  The reference's BLAS sums the same products in another order, so its
 *   distances differ from these by a few ulp of (|x|^2 + |c|^2): rows with margin below ~1e-6 are the ones
 *   whose id may legitimately differ between two correct fp32 evaluations. */
int rqo_rq_forward_ex(const float *res0, int64_t B, int D, const float *codebooks, int L, int K,
                      int mode, float beta, int64_t *ids, float *embs, float *residuals,
                      float *emb_sum, float *loss, float *embs_norm, float *tie_margin);

int rqo_rq_forward(const float *res0, int64_t B, int D, const float *codebooks, int L, int K,
                   int mode, float beta, int64_t *ids, float *embs, float *residuals,
                   float *emb_sum, float *loss, float *embs_norm) {
    return rqo_rq_forward_ex(res0, B, D, codebooks, L, K, mode, beta, ids, embs, residuals, emb_sum, loss,
                             embs_norm, 0);
}

int rqo_rq_forward_ex(const float *res0, int64_t B, int D, const float *codebooks, int L, int K,
                      int mode, float beta, int64_t *ids, float *embs, float *residuals,
                      float *emb_sum, float *loss, float *embs_norm, float *tie_margin) {
    if (B < 0 || D <= 0 || L <= 0 || K <= 0 || !res0 || !codebooks || !ids) return RQO_EARG;
    if (mode != RQO_MODE_EVAL && mode != RQO_MODE_STE && mode != RQO_MODE_ROTATION) return RQO_EARG;
    float *csq = (float *)malloc(sizeof(float) * (size_t)L * K);
    float *dist = (float *)malloc(sizeof(float) * (size_t)K);
    float *row = (float *)malloc(sizeof(float) * (size_t)D * 6);
    if (!csq || !dist || !row) { free(csq); free(dist); free(row); return RQO_ENOMEM; }
    for (int l = 0; l < L; ++l)
        for (int k = 0; k < K; ++k)
            csq[(size_t)l * K + k] = sumsq2(codebooks + ((size_t)l * K + k) * D, D);

    float *res = row, *out = row + D, *tmp = row + 2 * D; /* tmp: 3D */
    float *diff = row + 5 * D;
    for (int64_t i = 0; i < B; ++i) {
        memcpy(res, res0 + (size_t)i * D, sizeof(float) * D);
        float lsum = 0.0f;
        for (int l = 0; l < L; ++l) {
            const float *cb = codebooks + (size_t)l * K * D;
            if (residuals) memcpy(residuals + ((size_t)l * B + i) * D, res, sizeof(float) * D);
            l2_dist_row(res, cb, csq + (size_t)l * K, K, D, dist);
            int64_t id = argmin_torch(dist, K);
            ids[(size_t)l * B + i] = id;
            if (tie_margin) {
                const float *cq = csq + (size_t)l * K;
                float xsq = sumsq2(res, D), cmax = 0.0f, m = 0.0f;
                int nan = 0;
                for (int k = 0; k < K; ++k) { if (cq[k] != cq[k]) nan = 1; else if (cq[k] > cmax) cmax = cq[k]; }
                if (!nan && (xsq + cmax) < 1.0e38f) {
                    float d2 = INFINITY;
                    for (int k = 0; k < K; ++k) if (k != id && dist[k] < d2) d2 = dist[k];
                    m = (d2 - dist[id]) / (xsq + cq[id]);
                    if (m != m) m = 0.0f;
                }
                tie_margin[(size_t)l * B + i] = m;
            }
            const float *emb = cb + (size_t)id * D;
            /* QuantizeLoss (loss.py:38-41): both terms are sum((x-emb)^2), bit-identical */
            for (int d = 0; d < D; ++d) diff[d] = res[d] - emb[d];
            float s = sumsq2(diff, D);
            float lv = s + beta * s;
            if (mode == RQO_MODE_EVAL) {
                for (int d = 0; d < D; ++d) out[d] = emb[d];                    /* quantize.py:160 */
            } else if (mode == RQO_MODE_STE) {
                for (int d = 0; d < D; ++d) out[d] = res[d] + (emb[d] - res[d]); /* quantize.py:139 */
            } else {
                rotation_row(res, emb, D, out, 0, 0, 0, 0, tmp);
            }
            if (l == 0) lsum = lv; else lsum = lsum + lv;                        /* 0 + t == t */
            if (embs) memcpy(embs + ((size_t)l * B + i) * D, out, sizeof(float) * D);
            if (embs_norm) embs_norm[(size_t)i * L + l] = sqrtf(sumsq2(out, D));
            if (emb_sum) {
                float *es = emb_sum + (size_t)i * D;
                for (int d = 0; d < D; ++d) es[d] = (l == 0) ? out[d] : es[d] + out[d];
            }
            for (int d = 0; d < D; ++d) res[d] = res[d] - out[d];               /* rqvae.py:130 */
        }
        if (loss) loss[i] = lsum;
    }
    free(csq); free(dist); free(row);
    return RQO_OK;
}

/* ---- backward of the above (closed forms of what autograd does through the reference) ------
 * Upstream gradients (any may be NULL = zero):
 *   g_embs [L,B,D] wrt embeddings, g_embsum [B,D] wrt emb_sum, g_resid [L,B,D] wrt residuals,
 *   g_loss [B] wrt loss.
 * Outputs: g_res0 [B,D]; g_codebooks [L,K,D] (accumulated over rows in ascending row order,
 * starting from zero).
 * With A_l = g_embs_l + g_embsum - G_{l+1}  (total gradient reaching emb_out_l; G_L = 0) and
 * G_l the total gradient reaching res_l:
 *   EVAL : G_l = g_resid_l + G_{l+1} + 2b(res-emb)gl ;            dE[id] += A_l + 2(emb-res)gl
 *   STE  : G_l = g_resid_l + G_{l+1} + A_l + 2b(res-emb)gl ;      dE[id] += 2(emb-res)gl
 *   ROT  : G_l = g_resid_l + G_{l+1} + s(A_l - 2(A_l.w)w + 2(A_l.q)u) + 2b(res-emb)gl ; dE as STE
 */
static int rq_backward_core(const float *res0, int64_t B, int D, const float *codebooks, int L, int K,
                            int mode, float beta, const int64_t *ids, const float *g_embs,
                            const float *g_embsum, const float *g_resid, const float *g_loss,
                            float *g_res0, float *g_codebooks, float *Vrows);

int rqo_rq_backward(const float *res0, int64_t B, int D, const float *codebooks, int L, int K,
                    int mode, float beta, const int64_t *ids, const float *g_embs,
                    const float *g_embsum, const float *g_resid, const float *g_loss,
                    float *g_res0, float *g_codebooks) {
    return rq_backward_core(res0, B, D, codebooks, L, K, mode, beta, ids, g_embs, g_embsum, g_resid, g_loss, g_res0,
                            g_codebooks, 0);
}

/* Same gradients, but the codebook gradient is accumulated in the FIXED order of csrc/rq_backward.hip's fused
 * kernels, so that the GPU result can be compared bit for bit (the kernels have no atomics: inside a workgroup every
 * code is owned by one wave, which adds the rows in order).  The geometry comes from rqhip_rq_backward_plan:
 *   workgroup b of n_wg, unit j of nw, round it  ->  unit_rows-row unit  (it * nw + j) * n_wg + b
 *   (pair-layout kernel: a unit is one wave's 32-row tile, nw = 8; flat EVAL/STE kernel: a unit is the workgroup's whole
 *   round of 1024 / (D/4) rows, nw = 1);
 *   partial_b[l][id] += V_l(row)   for it ascending, j ascending, rows of the unit ascending   (fp32 adds from 0);
 *   g_codebooks = (s0 + s1) + (s2 + s3),  s_q = 0 + partial_q + partial_{q+4} + ...   (rq_cbgrad_reduce_kernel). */
int rqo_rq_backward_ordered(const float *res0, int64_t B, int D, const float *codebooks, int L, int K,
                            int mode, float beta, const int64_t *ids, const float *g_embs,
                            const float *g_embsum, const float *g_resid, const float *g_loss,
                            float *g_res0, float *g_codebooks, int n_wg, int nw, int unit_rows) {
    if (n_wg <= 0 || nw <= 0 || unit_rows <= 0 || !g_codebooks) return RQO_EARG;
    const size_t LKD = (size_t)L * K * D;
    float *V = (float *)malloc(sizeof(float) * (size_t)L * (size_t)(B > 0 ? B : 1) * D);
    float *part = (float *)calloc((size_t)n_wg * LKD, sizeof(float));
    if (!V || !part) { free(V); free(part); return RQO_ENOMEM; }
    int rc = rq_backward_core(res0, B, D, codebooks, L, K, mode, beta, ids, g_embs, g_embsum, g_resid, g_loss, g_res0,
                              0, V);
    if (rc) { free(V); free(part); return rc; }
    const int64_t n_tiles = (B + unit_rows - 1) / unit_rows, waves = (int64_t)n_wg * nw;
    for (int b = 0; b < n_wg; ++b) {
        float *pb = part + (size_t)b * LKD;
        for (int64_t it = 0; it * waves < n_tiles; ++it)
            for (int j = 0; j < nw; ++j) {
                int64_t tile = it * waves + (int64_t)j * n_wg + b;
                if (tile >= n_tiles) continue;
                for (int64_t i = tile * unit_rows; i < (tile + 1) * unit_rows && i < B; ++i)
                    for (int l = 0; l < L; ++l) {
                        float *dE = pb + ((size_t)l * K + ids[(size_t)l * B + i]) * D;
                        const float *v = V + ((size_t)l * B + i) * D;
                        for (int d = 0; d < D; ++d) dE[d] = dE[d] + v[d];
                    }
            }
    }
    for (size_t e = 0; e < LKD; ++e) {
        float sg[4];
        for (int q = 0; q < 4; ++q) {
            float a = 0.0f;
            for (int g = q; g < n_wg; g += 4) a = a + part[(size_t)g * LKD + e];
            sg[q] = a;
        }
        g_codebooks[e] = (sg[0] + sg[1]) + (sg[2] + sg[3]);
    }
    free(V); free(part);
    return RQO_OK;
}

/* Vrows [L,B,D] (optional): receives, per row and level, the vector the embedding backward adds to dE[id]. */
static int rq_backward_core(const float *res0, int64_t B, int D, const float *codebooks, int L, int K,
                            int mode, float beta, const int64_t *ids, const float *g_embs,
                            const float *g_embsum, const float *g_resid, const float *g_loss,
                            float *g_res0, float *g_codebooks, float *Vrows) {
    if (B < 0 || D <= 0 || L <= 0 || K <= 0 || !res0 || !codebooks || !ids) return RQO_EARG;
    if (mode != RQO_MODE_EVAL && mode != RQO_MODE_STE && mode != RQO_MODE_ROTATION) return RQO_EARG;
    size_t LD = (size_t)L * D;
    float *buf = (float *)malloc(sizeof(float) * (LD * 2 + (size_t)D * 10));
    if (!buf) return RQO_ENOMEM;
    float *resl = buf;            /* [L,D] level inputs */
    float *outl = buf + LD;       /* [L,D] emb_out */
    float *G = buf + 2 * LD;      /* [D] */
    float *A = G + D, *tmp = A + D /* 3D */, *w = tmp + 3 * D, *u = w + D, *q = u + D, *Gn = q + D;
    float *scratch = Gn + D;
    (void)scratch;
    if (g_codebooks) memset(g_codebooks, 0, sizeof(float) * (size_t)L * K * D);
    for (int64_t i = 0; i < B; ++i) {
        /* recompute the forward chain for this row (same arithmetic as rqo_rq_forward) */
        memcpy(resl, res0 + (size_t)i * D, sizeof(float) * D);
        for (int l = 0; l < L; ++l) {
            const float *emb = codebooks + ((size_t)l * K + ids[(size_t)l * B + i]) * D;
            float *r = resl + (size_t)l * D, *o = outl + (size_t)l * D;
            if (mode == RQO_MODE_EVAL) for (int d = 0; d < D; ++d) o[d] = emb[d];
            else if (mode == RQO_MODE_STE) for (int d = 0; d < D; ++d) o[d] = r[d] + (emb[d] - r[d]);
            else rotation_row(r, emb, D, o, 0, 0, 0, 0, tmp);
            if (l + 1 < L) for (int d = 0; d < D; ++d) r[D + d] = r[d] - o[d];
        }
        float gl = g_loss ? g_loss[i] : 0.0f;
        for (int d = 0; d < D; ++d) G[d] = 0.0f;
        for (int l = L - 1; l >= 0; --l) {
            int64_t id = ids[(size_t)l * B + i];
            const float *emb = codebooks + ((size_t)l * K + id) * D;
            const float *r = resl + (size_t)l * D;
            for (int d = 0; d < D; ++d) {
                float a = 0.0f;
                if (g_embs) a = g_embs[((size_t)l * B + i) * D + d];
                if (g_embsum) a = a + g_embsum[(size_t)i * D + d];
                A[d] = a - G[d];
            }
            float *dE = g_codebooks ? g_codebooks + ((size_t)l * K + id) * D : 0;
            if (mode == RQO_MODE_ROTATION) {
                float s;
                float o_unused[1]; (void)o_unused;
                float *o = Gn; /* reuse as scratch for emb_out */
                rotation_row(r, emb, D, o, w, u, q, &s, tmp);
                float aw = dotp2(A, w, D), aq = dotp2(A, q, D);
                for (int d = 0; d < D; ++d) {
                    float lin = ((A[d] - 2.0f * (aw * w[d])) + 2.0f * (aq * u[d])) * s;
                    float gr = g_resid ? g_resid[((size_t)l * B + i) * D + d] : 0.0f;
                    float commit = (2.0f * beta) * (r[d] - emb[d]) * gl;
                    float gnew = ((gr + G[d]) + lin) + commit;
                    if (dE) dE[d] = dE[d] + (2.0f * (emb[d] - r[d])) * gl;
                    if (Vrows) Vrows[((size_t)l * B + i) * D + d] = (2.0f * (emb[d] - r[d])) * gl;
                    Gn[d] = gnew;
                }
                memcpy(G, Gn, sizeof(float) * D);
            } else {
                for (int d = 0; d < D; ++d) {
                    float gr = g_resid ? g_resid[((size_t)l * B + i) * D + d] : 0.0f;
                    float commit = (2.0f * beta) * (r[d] - emb[d]) * gl;
                    float embg = (2.0f * (emb[d] - r[d])) * gl;
                    float gnew;
                    if (mode == RQO_MODE_EVAL) {
                        gnew = (gr + G[d]) + commit;
                        if (dE) dE[d] = dE[d] + (A[d] + embg);
                        if (Vrows) Vrows[((size_t)l * B + i) * D + d] = A[d] + embg;
                    } else {
                        gnew = ((gr + G[d]) + A[d]) + commit;
                        if (dE) dE[d] = dE[d] + embg;
                        if (Vrows) Vrows[((size_t)l * B + i) * D + d] = embg;
                    }
                    G[d] = gnew;
                }
            }
        }
        if (g_res0) memcpy(g_res0 + (size_t)i * D, G, sizeof(float) * D);
    }
    free(buf);
    return RQO_OK;
}

/* ---- Gumbel-softmax level (quantize.py:131-136, gumbel.py:8-20) -----------------------------
 * One level, training mode.  U [B,K] is the uniform noise torch.rand would have drawn.
 * Outputs ids [B], emb [B,D] (= emb_out), loss [B], weights [B,K] (optional, for backward checks).
 */
int rqo_gumbel_forward(const float *x, int64_t B, int D, const float *cb, int K, const float *U,
                       float temperature, float beta, int64_t *ids, float *emb_o, float *loss,
                       float *weights) {
    if (B < 0 || D <= 0 || K <= 0 || !x || !cb || !U || !ids) return RQO_EARG;
    float *csq = (float *)malloc(sizeof(float) * (size_t)K);
    float *dist = (float *)malloc(sizeof(float) * (size_t)K * 2);
    float *e = (float *)malloc(sizeof(float) * (size_t)D * 2);
    if (!csq || !dist || !e) { free(csq); free(dist); free(e); return RQO_ENOMEM; }
    float *wk = dist + K, *diff = e + D;
    for (int k = 0; k < K; ++k) csq[k] = sumsq2(cb + (size_t)k * D, D);
    for (int64_t i = 0; i < B; ++i) {
        const float *xi = x + (size_t)i * D;
        l2_dist_row(xi, cb, csq, K, D, dist);
        ids[i] = argmin_torch(dist, K);
        float m = -INFINITY;
        for (int k = 0; k < K; ++k) {
            float u = U[(size_t)i * K + k];
            float g = -logf(-logf(u + 1e-20f) + 1e-20f);
            float y = ((-dist[k]) + g) / temperature;
            wk[k] = y;
            if (y > m) m = y;
        }
        float Z = 0.0f;
        for (int k = 0; k < K; ++k) { wk[k] = expf(wk[k] - m); Z = Z + wk[k]; }
        for (int k = 0; k < K; ++k) wk[k] = wk[k] / Z;
        for (int d = 0; d < D; ++d) e[d] = 0.0f;
        for (int k = 0; k < K; ++k)
            for (int d = 0; d < D; ++d) e[d] = fmaf(wk[k], cb[(size_t)k * D + d], e[d]);
        for (int d = 0; d < D; ++d) diff[d] = xi[d] - e[d];
        float s = sumsq2(diff, D);
        if (loss) loss[i] = s + beta * s;
        if (emb_o) memcpy(emb_o + (size_t)i * D, e, sizeof(float) * D);
        if (weights) memcpy(weights + (size_t)i * K, wk, sizeof(float) * K);
    }
    free(csq); free(dist); free(e);
    return RQO_OK;
}

/* Backward of one Gumbel level.  g_emb [B,D] wrt emb_out, g_loss [B] wrt loss (either may be NULL).
 * Outputs g_x [B,D], g_cb [K,D] (row-ascending accumulation from zero).
 *   ge   = g_emb + 2(emb-x) gl                      (emb path: emb_out and emb_loss)
 *   dw_k = ge . C_k ;  dC_k += w_k ge
 *   dy_k = w_k (dw_k - sum_j w_j dw_j) / T ;  ddist_k = -dy_k
 *   dist_k = |x|^2 + |C_k|^2 - 2 x.C_k :
 *   dx  += 2x sum_k ddist_k - 2 sum_k ddist_k C_k ;  dC_k += 2 C_k ddist_k - 2 ddist_k x
 *   dx  += 2 beta (x-emb) gl                        (commitment path)
 */
int rqo_gumbel_backward(const float *x, int64_t B, int D, const float *cb, int K, const float *U,
                        float temperature, float beta, const float *g_emb, const float *g_loss,
                        float *g_x, float *g_cb) {
    if (B < 0 || D <= 0 || K <= 0 || !x || !cb || !U) return RQO_EARG;
    float *csq = (float *)malloc(sizeof(float) * (size_t)K);
    float *dist = (float *)malloc(sizeof(float) * (size_t)K * 3);
    float *e = (float *)malloc(sizeof(float) * (size_t)D * 3);
    if (!csq || !dist || !e) { free(csq); free(dist); free(e); return RQO_ENOMEM; }
    float *wk = dist + K, *dd = dist + 2 * K, *ge = e + D, *gx = e + 2 * D;
    for (int k = 0; k < K; ++k) csq[k] = sumsq2(cb + (size_t)k * D, D);
    if (g_cb) memset(g_cb, 0, sizeof(float) * (size_t)K * D);
    for (int64_t i = 0; i < B; ++i) {
        const float *xi = x + (size_t)i * D;
        l2_dist_row(xi, cb, csq, K, D, dist);
        float m = -INFINITY;
        for (int k = 0; k < K; ++k) {
            float u = U[(size_t)i * K + k];
            float g = -logf(-logf(u + 1e-20f) + 1e-20f);
            float y = ((-dist[k]) + g) / temperature;
            wk[k] = y;
            if (y > m) m = y;
        }
        float Z = 0.0f;
        for (int k = 0; k < K; ++k) { wk[k] = expf(wk[k] - m); Z = Z + wk[k]; }
        for (int k = 0; k < K; ++k) wk[k] = wk[k] / Z;
        for (int d = 0; d < D; ++d) e[d] = 0.0f;
        for (int k = 0; k < K; ++k)
            for (int d = 0; d < D; ++d) e[d] = fmaf(wk[k], cb[(size_t)k * D + d], e[d]);
        float gl = g_loss ? g_loss[i] : 0.0f;
        for (int d = 0; d < D; ++d) {
            float ga = g_emb ? g_emb[(size_t)i * D + d] : 0.0f;
            ge[d] = ga + (2.0f * (e[d] - xi[d])) * gl;
        }
        float sw = 0.0f;
        for (int k = 0; k < K; ++k) {
            float dw = 0.0f;
            for (int d = 0; d < D; ++d) dw = fmaf(ge[d], cb[(size_t)k * D + d], dw);
            dd[k] = dw;
            sw = fmaf(wk[k], dw, sw);
        }
        float sdd = 0.0f;
        for (int k = 0; k < K; ++k) {
            float dy = (wk[k] * (dd[k] - sw)) / temperature;
            dd[k] = -dy;
            sdd = sdd + dd[k];
        }
        for (int d = 0; d < D; ++d) gx[d] = (2.0f * xi[d]) * sdd;
        for (int k = 0; k < K; ++k) {
            const float *ck = cb + (size_t)k * D;
            float *gc = g_cb ? g_cb + (size_t)k * D : 0;
            for (int d = 0; d < D; ++d) {
                gx[d] = fmaf(-2.0f * dd[k], ck[d], gx[d]);
                if (gc) {
                    float t = fmaf(wk[k], ge[d], gc[d]);
                    gc[d] = fmaf(2.0f * dd[k], ck[d] - xi[d], t);
                }
            }
        }
        if (g_x)
            for (int d = 0; d < D; ++d)
                g_x[(size_t)i * D + d] = gx[d] + ((2.0f * beta) * (xi[d] - e[d])) * gl;
    }
    free(csq); free(dist); free(e);
    return RQO_OK;
}

/* ---- k-means (init/kmeans.py) --------------------------------------------------------------- */

/* kmeans.py:40-43: squared direct-difference distance, argmin with torch.min semantics.
 * sum over d: two parity accumulators of separately rounded (x-c)^2. */
int rqo_kmeans_assign(const float *x, int64_t B, int D, const float *cent, int K, int64_t *assign) {
    if (B < 0 || D <= 0 || K <= 0 || !x || !cent || !assign) return RQO_EARG;
    float *dist = (float *)malloc(sizeof(float) * (size_t)K);
    float *diff = (float *)malloc(sizeof(float) * (size_t)D);
    if (!dist || !diff) { free(dist); free(diff); return RQO_ENOMEM; }
    for (int64_t i = 0; i < B; ++i) {
        for (int k = 0; k < K; ++k) {
            for (int d = 0; d < D; ++d) diff[d] = x[(size_t)i * D + d] - cent[(size_t)k * D + d];
            dist[k] = sumsq2(diff, D);
        }
        assign[i] = argmin_torch(dist, K);
    }
    free(dist); free(diff);
    return RQO_OK;
}

/* kmeans.py:48-58: centroid k <- mean of its rows (sum in ascending row order, then / count).
 * Empty clusters are left untouched and reported through counts[k] == 0; the caller reseeds them
 * (kmeans.py:50-54 draws torch.randint on the host, one draw per empty cluster, ascending k). */
int rqo_kmeans_update(const float *x, int64_t B, int D, const int64_t *assign, int K, float *cent,
                      int64_t *counts) {
    if (B < 0 || D <= 0 || K <= 0 || !x || !assign || !cent || !counts) return RQO_EARG;
    float *sum = (float *)calloc((size_t)K * D, sizeof(float));
    if (!sum) return RQO_ENOMEM;
    for (int k = 0; k < K; ++k) counts[k] = 0;
    for (int64_t i = 0; i < B; ++i) {
        int64_t k = assign[i];
        if (k < 0 || k >= K) { free(sum); return RQO_EARG; }
        counts[k] += 1;
        for (int d = 0; d < D; ++d) sum[(size_t)k * D + d] = sum[(size_t)k * D + d] + x[(size_t)i * D + d];
    }
    for (int k = 0; k < K; ++k)
        if (counts[k] > 0)
            for (int d = 0; d < D; ++d) cent[(size_t)k * D + d] = sum[(size_t)k * D + d] / (float)counts[k];
    free(sum);
    return RQO_OK;
}

/* kmeans.py:68: torch.norm(c - old, dim=1).max() */
float rqo_kmeans_shift(const float *cent, const float *old, int K, int D) {
    float m = 0.0f;
    float *diff = (float *)malloc(sizeof(float) * (size_t)D);
    if (!diff) return NAN;
    for (int k = 0; k < K; ++k) {
        for (int d = 0; d < D; ++d) diff[d] = cent[(size_t)k * D + d] - old[(size_t)k * D + d];
        float n = sqrtf(sumsq2(diff, D));
        if (n != n) { free(diff); return n; }
        if (n > m) m = n;
    }
    free(diff);
    return m;
}

/* ---- reconstruction loss (modules/loss.py:5-10): out[b] = sum_d (x_hat-x)^2 ------------------------------
 * Fixed order == csrc/recon_loss.hip: 64 lane partials, lane l owning the 16-byte groups l, l+64, ... (or the
 * single elements l, l+64, ... when N is not a multiple of 4), each added in ascending address order; then a
 * 6-round xor butterfly p[l] += p[l ^ m], m = 32..1. */
int rqo_recon_loss(const float *x_hat, const float *x, int64_t B, int N, float *out) {
    if (B < 0 || N <= 0 || !x_hat || !x || !out) return RQO_EARG;
    for (int64_t b = 0; b < B; ++b) {
        const float *a = x_hat + (size_t)b * N, *c = x + (size_t)b * N;
        float p[64], q[64];
        for (int l = 0; l < 64; ++l) {
            float s = 0.0f;
            if ((N & 3) == 0) {
                for (int i = l; i < N / 4; i += 64)
                    for (int j = 0; j < 4; ++j) { float d = a[4 * i + j] - c[4 * i + j]; s = s + d * d; }
            } else {
                for (int i = l; i < N; i += 64) { float d = a[i] - c[i]; s = s + d * d; }
            }
            p[l] = s;
        }
        for (int m = 32; m >= 1; m >>= 1) {
            for (int l = 0; l < 64; ++l) q[l] = p[l] + p[l ^ m];
            memcpy(p, q, sizeof(p));
        }
        out[b] = p[0];
    }
    return RQO_OK;
}

/* ---- id statistics -------------------------------------------------------------------------- */

static int tuple_eq(const int64_t *ids, int64_t B, int L, int64_t a, int64_t b) {
    for (int l = 0; l < L; ++l) if (ids[(size_t)l * B + a] != ids[(size_t)l * B + b]) return 0;
    return 1;
}

/* rank[i] = number of rows j < i with the same L-tuple (semids.py:95-105: in-batch tril hits plus
 * hits against all earlier batches).  ids is [L,B].  O(B^2): small inputs only. */
int rqo_dedup_rank(const int64_t *ids, int64_t B, int L, int64_t *rank) {
    if (B < 0 || L <= 0 || !ids || !rank) return RQO_EARG;
    for (int64_t i = 0; i < B; ++i) {
        int64_t r = 0;
        for (int64_t j = 0; j < i; ++j) r += tuple_eq(ids, B, L, i, j);
        rank[i] = r;
    }
    return RQO_OK;
}

/* rqvae.py:159-167: number of rows with no LATER identical row (== number of distinct tuples);
 * p_unique_ids = that / B. */
int64_t rqo_count_rows_without_later_duplicate(const int64_t *ids, int64_t B, int L) {
    int64_t n = 0;
    for (int64_t i = 0; i < B; ++i) {
        int dup = 0;
        for (int64_t j = i + 1; j < B && !dup; ++j) dup = tuple_eq(ids, B, L, i, j);
        n += !dup;
    }
    return n;
}

/* ---- decoder-side consumers of semantic ids ------------------------------- */

/* modules/model.py:169-182 (_check_valid_prefix): valid[q] = any_n all_c corpus[n,c] == prefix[q,c], c < h.
 * corpus [N,H] dense, prefix [P,h] dense.  Brute force like the reference, O(N P h): small inputs only.
 * h == 0: all() over no columns is true, so valid iff N > 0. */
int rqo_prefix_valid(const int64_t *corpus, int64_t N, int H, const int64_t *prefix, int64_t P, int h,
                     uint8_t *valid) {
    if (N < 0 || P < 0 || H < 0 || h < 0 || h > H || (P > 0 && !valid)) return RQO_EARG;
    for (int64_t q = 0; q < P; ++q) {
        int any = 0;
        for (int64_t n = 0; n < N && !any; ++n) {
            int all = 1;
            for (int c = 0; c < h && all; ++c) all = corpus[n * H + c] == prefix[q * h + c];
            any = all;
        }
        valid[q] = (uint8_t)any;
    }
    return RQO_OK;
}

/* evaluate/metrics.py:16-19: pos_match.all(-1).max(-1) -> (match_found, rank of the first match).
 * rank[b] = first k with top_k[b,k,:] == actual[b,:], or -1.  actual [B,D], top_k [B,K,D]. */
int rqo_topk_first_match(const int64_t *actual, const int64_t *top_k, int64_t B, int K, int D, int64_t *rank) {
    if (B < 0 || K < 0 || D < 0 || (B > 0 && !rank)) return RQO_EARG;
    for (int64_t b = 0; b < B; ++b) {
        rank[b] = -1;
        for (int k = 0; k < K && rank[b] < 0; ++k) {
            int all = 1;
            for (int d = 0; d < D && all; ++d) all = actual[b * D + d] == top_k[(b * K + k) * D + d];
            if (all) rank[b] = k;
        }
    }
    return RQO_OK;
}

/* ---- weight gradient of a bias-free Linear(+ReLU) layer (reference modules/encoder.py:25-38 under autograd) -----
 * dW[n,k] = sum_m g_pre[m,n] x[m,k],  g_pre = g where y > 0 else 0 (aten threshold_backward; g itself when y == NULL).
 * Restates csrc/wgrad.hip's FIXED summation order: the rows are cut into 32-row granules, the granules into `msplit`
 * contiguous ranges (range s = granules [C s / msplit, C (s+1) / msplit)); inside a range one fp32 FMA chain over the
 * rows in ascending order (what the MFMA accumulates); the ranges' sums are then added as a balanced binary tree:
 * padded with zeros to a power of two, a[j] += a[j+s] for j a multiple of 2s, s = 1, 2, 4, ...
 * g_masked [M,N] (optional) receives g_pre. */
int rqo_linear_wgrad(const float *g, const float *y, const float *x, int64_t M, int N, int K, int msplit,
                     float *g_masked, float *dW) {
    if (M < 0 || N <= 0 || K <= 0 || msplit <= 0 || !g || !x || !dW) return RQO_EARG;
    const int64_t C = (M + 31) / 32;
    int P = 1;
    while (P < msplit) P <<= 1;
    float *gp = (float *)malloc(sizeof(float) * (size_t)(M > 0 ? M : 1) * N);
    float *leaf = (float *)malloc(sizeof(float) * (size_t)P);
    if (!gp || !leaf) { free(gp); free(leaf); return RQO_ENOMEM; }
    for (int64_t i = 0; i < M * N; ++i) gp[i] = (y && y[i] <= 0.0f) ? 0.0f : g[i];
    if (g_masked) memcpy(g_masked, gp, sizeof(float) * (size_t)M * N);
    for (int n = 0; n < N; ++n)
        for (int k = 0; k < K; ++k) {
            for (int s = 0; s < P; ++s) {
                float acc = 0.0f;
                if (s < msplit) {
                    int64_t r0 = (C * s / msplit) * 32, r1 = (C * (s + 1) / msplit) * 32;
                    if (r1 > M) r1 = M;
                    for (int64_t m = r0; m < r1; ++m) acc = fmaf(gp[(size_t)m * N + n], x[(size_t)m * K + k], acc);
                }
                leaf[s] = acc;
            }
            for (int s = 1; s < P; s <<= 1)
                for (int j = 0; j < P; j += 2 * s) leaf[j] = leaf[j] + leaf[j + s];
            dW[(size_t)n * K + k] = leaf[0];
        }
    free(gp); free(leaf);
    return RQO_OK;
}

/* ---- the RQ <-> MLP seam's linear layers (csrc/rq_forward.hip: seam_in / seam_out; reference modules/encoder.py:25-38, a bias-free
 * nn.Linear, optionally followed by ReLU) -------------------------------------------------------------------------------------------
 * out[b][n] = epilogue( chain_d x'[b][d] * W(n, d) ): ONE fp32 FMA chain over the input features d = 0 .. n_in - 1 from 0 (what the fp32
 * matrix instruction accumulates, as dot_chain above).
 *   x'     = x where xmask > 0 else 0 (xmask NULL: x) -- aten.threshold_backward(x, xmask, 0), the ReLU backward of a data gradient
 *   W(n,d) = transposed ? w[d * n_out + n] : w[n * n_in + d]
 *   epilogue: 0 store, 1 relu (v < 0 -> 0; NaN stays), 3 mask (omask <= 0 -> 0) */
int rqo_linear_chain(const float *x, const float *xmask, int64_t B, int n_in, const float *w, int n_out, int transposed, int epilogue,
                     const float *omask, float *out) {
    if (B < 0 || n_in < 1 || n_out < 1 || !w || (B > 0 && (!x || !out)) || (epilogue == 3 && !omask)) return -1;
    for (int64_t b = 0; b < B; ++b) {
        for (int n = 0; n < n_out; ++n) {
            float acc = 0.0f;
            for (int d = 0; d < n_in; ++d) {
                float xv = x[(size_t)b * n_in + d];
                if (xmask && xmask[(size_t)b * n_in + d] <= 0.0f) xv = 0.0f;
                acc = fmaf(xv, transposed ? w[(size_t)d * n_out + n] : w[(size_t)n * n_in + d], acc);
            }
            if (epilogue == 1 && acc < 0.0f) acc = 0.0f;
            if (epilogue == 3 && omask[(size_t)b * n_out + n] <= 0.0f) acc = 0.0f;
            out[(size_t)b * n_out + n] = acc;
        }
    }
    return 0;
}

/* ---- the encoder / decoder Linear layers below 4096 rows (csrc/mlp_small.hip: lin_small_kernel; reference modules/encoder.py:25-38 and
 * autograd's data gradient of the same layer) ----------------------------------------------------------------------------------------
 * out[m][n] = epilogue( ((p_0 + p_1) + ...) + p_{waves-1} ),  p_v = ONE fp32 FMA chain from +0 over the 32-term groups
 * [v ng / waves, (v + 1) ng / waves) of the reduction (ng = n_red / 32), a group's terms taken in the order 0 8 16 24 1 9 17 25 ... 7 15 23 31
 * (instruction e of the kernel consumes term 8 kq + e from each of the four k-slots kq of the 16x16x4 fp32 matrix instruction, which
 * accumulates its slots in ascending order).
 *   W(r, n) = w_kn ? w[r * n_out + n] : w[n * n_red + r];  epilogue: 0 store, 1 relu (v < 0 -> 0; NaN stays), 3 mask (aux <= 0 -> 0) */
int rqo_linear_small(const float *a, int64_t M, int n_red, const float *w, int n_out, int w_kn, int waves, int epilogue, const float *aux,
                     float *out) {
    if (M < 0 || n_red < 32 || n_red % 32 || n_out < 1 || waves < 1 || !w || (M > 0 && (!a || !out)) || (epilogue == 3 && !aux)) return -1;
    const int ng = n_red / 32;
    for (int64_t m = 0; m < M; ++m) {
        for (int n = 0; n < n_out; ++n) {
            float tot = 0.0f;
            for (int v = 0; v < waves; ++v) {
                const int lo = (int)((int64_t)v * ng / waves), hi = (int)((int64_t)(v + 1) * ng / waves);
                float part = 0.0f;
                for (int g = lo; g < hi; ++g)
                    for (int e = 0; e < 8; ++e)
                        for (int kq = 0; kq < 4; ++kq) {
                            const int r = 32 * g + 8 * kq + e;
                            part = fmaf(a[(size_t)m * n_red + r], w_kn ? w[(size_t)r * n_out + n] : w[(size_t)n * n_red + r], part);
                        }
                tot = v == 0 ? part : tot + part;
            }
            if (epilogue == 1 && tot < 0.0f) tot = 0.0f;
            if (epilogue == 3 && aux[(size_t)m * n_out + n] <= 0.0f) tot = 0.0f;
            out[(size_t)m * n_out + n] = tot;
        }
    }
    return 0;
}
