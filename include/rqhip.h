/*
 * rqhip.h -- C ABI of librqhip.so: the MI355X (gfx950) residual-quantisation hot path.
 *
 * The reference (EdoardoBotta/RQ-VAE-Recommender) is pure Python: it has no FFI, plugin or operator
 * registry for this path, so the drop-in boundary is its Python module API (modules.quantize.Quantize,
 * modules.rqvae.RqVae, init.kmeans.kmeans_init_, ...; mirrored under rq-vae-recommender_amd/).  This
 * header is the C boundary UNDER that mirror: plain pointers and sizes, no torch types.  Each entry
 * point names the reference code it replaces (paths relative to the reference root); INTEGRATION.md
 * shows the ctypes stub a maintainer of the reference would add to call it from modules/quantize.py.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (hipMalloc'ed / torch CUDA tensor data_ptr), fp32 row-major
 *     contiguous, ids int64; optional outputs/inputs may be NULL where stated.
 *   - `stream` is a hipStream_t passed as void*; kernels are enqueued on it and the call returns
 *     without synchronising.  The library never allocates, frees or retains device memory: scratch
 *     comes from the caller (`workspace`, size from the matching *_workspace_bytes()).
 *   - return value: 0 = success; negative = RQHIP_E* argument/shape error; positive = hipError_t.
 *     rqhip_last_error() returns a thread-local message for the last failure.
 *   - results are bit-exact with oracle/rq_oracle.c (which fixes every floating-point reduction
 *     order) for everything except transcendental functions in the Gumbel path.
 */
#ifndef RQHIP_H
#define RQHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RQHIP_VERSION 462 /* 462: rqhip_linear_wgrad_f16_batch; 461: rqhip_unique_fraction; 460: rqhip_linear_small; 450: rqhip_rq_seam (round 6); major*10000 + minor*100 + patch; 200: rqhip_rq_forward gained tie_margin; 300: rqhip_rq_forward_ex; 301: rqhip_gemm_split_recon;
                            400: RQHIP_SPLIT_F16X2 (rqhip_gemm_split_ex, rqhip_weight_images, rqhip_maxima, rqhip_linear_wgrad_f16), tagged profile records */

#define RQHIP_OK 0
#define RQHIP_EARG (-1)         /* bad pointer / size / mode */
#define RQHIP_EUNSUPPORTED (-2) /* shape outside what the kernels implement (e.g. D > 128) */
#define RQHIP_EWORKSPACE (-3)   /* workspace missing or too small */

/* forward modes == the reference's QuantizeForwardMode (modules/quantize.py:16-20) plus eval */
#define RQHIP_MODE_EVAL 0     /* module.eval(): quantize.py:159-161 */
#define RQHIP_MODE_STE 1      /* QuantizeForwardMode.STE, training: quantize.py:137-139 */
#define RQHIP_MODE_ROTATION 2 /* QuantizeForwardMode.ROTATION_TRICK, training: quantize.py:140-153 */
#define RQHIP_MODE_GUMBEL 3   /* QuantizeForwardMode.GUMBEL_SOFTMAX, training: quantize.py:131-136 */

typedef void *rqhip_stream_t;

int rqhip_version(void);
const char *rqhip_last_error(void);
/* number of compute units of the current device (grid sizing, reported by bench.py) */
int rqhip_device_cu_count(int *cu_count);

/* ------------------------------------------------------------------------------------------------
 * Residual quantisation, forward.  Replaces the level loop of RqVae.get_semantic_ids
 * (modules/rqvae.py:118-139) together with every Quantize.forward it calls (modules/quantize.py:
 * 104-163: L2 distance :112-117, argmin :128, STE :137-139 / rotation trick :140-153 / eval :159-161,
 * QuantizeLoss modules/loss.py:33-41) and the embs.sum / embs.norm consumers (rqvae.py:146,158).
 * L = 1 is a single Quantize.forward.
 *
 *   res0      [B,D]    level-0 input (encoder output)
 *   codebooks [L,K,D]  out_proj(embedding.weight) of each level
 *   mode      RQHIP_MODE_EVAL | _STE | _ROTATION
 *   beta      commitment weight
 *   ids       [L,B] int64 (required)   -- sem_ids[b,l] = ids[l*B + b]
 *   embs      [L,B,D] or NULL          -- quantized.embeddings per level
 *   residuals [L,B,D] or NULL          -- input of each level
 *   emb_sum   [B,D]   or NULL          -- sum over levels of embs, ((e0+e1)+e2)+...
 *   loss      [B]     or NULL          -- sum over levels of the quantize loss
 *   embs_norm [B,L]   or NULL          -- L2 norm of embs per level
 *   tie_margin [L,B]  or NULL          -- how decisively each level's argmin was taken (SURVEY.md section 8b
 *                                         `tie_margin_flags`): (d2 - d1) / (|x|^2 + |c_id|^2), d1 = dist[id],
 *                                         d2 = smallest distance among the OTHER codes (a duplicate of the minimum
 *                                         counts: margin 0); 0 for rows that took the exact non-finite scan and when
 *                                         the quotient is NaN, +Inf when K == 1.  Two correct fp32 evaluations of
 *                                         quantize.py:113-117 (this kernel's FMA chain, the reference's BLAS) differ by
 *                                         a few ulp of |x|^2 + |c|^2, so ids can only differ on rows whose margin is
 *                                         below ~1e-6: a caller that needs the reference's ids bit for bit adjudicates
 *                                         exactly those rows (tests/test_gpu_reference_parity.py does, in fp64).
 *                                         Requesting it selects a kernel variant with more work per code (+6 % kernel time at 100 000 x 3 x 256).
 *   workspace rqhip_rq_forward_workspace_bytes(L,K) bytes of scratch (codebook norms)
 * Limits: 1 <= D <= 128, 1 <= K <= 65536, 1 <= L <= 16.
 */
size_t rqhip_rq_forward_workspace_bytes(int L, int K);
int rqhip_rq_forward(const float *res0, int64_t B, int D, const float *codebooks, int L, int K,
                     int mode, float beta, int64_t *ids, float *embs, float *residuals,
                     float *emb_sum, float *loss, float *embs_norm, float *tie_margin, void *workspace,
                     size_t workspace_bytes, rqhip_stream_t stream);
/* The same call with explicit kernel-selection flags (bench.py's A/B lines and the tests; rqhip_rq_forward passes 0).
 * Every selection returns the same bits -- ids, embeddings, losses -- only the time differs:
 *   RQHIP_FWD_SCAN_FP32    distances on the fp32 matrix pipe (v_mfma_f32_32x32x2_f32, the oracle's FMA chain) even where
 *                          the filtered scan applies (D = 32 / 64, 16-byte aligned rows, no tie_margin): the default
 *                          there scans bf16-split scores on the bf16 matrix cores and re-decides exactly every row whose
 *                          two best scores are closer than the proven error bound (rqhip_filter_bound)
 *   RQHIP_FWD_SCAN_VALU    distances with packed fp32 FMAs on the vector ALU, codes broadcast from LDS, no matrix
 *                          instruction at all (D = 32, 16-byte aligned rows, K <= 1024, no tie_margin; otherwise
 *                          RQHIP_EUNSUPPORTED) -- the LDS/VALU form BASELINE.json's north_star asks the MFMA form to be
 *                          measured against
 *   RQHIP_FWD_NO_COOP_TAIL the partly filled last round of row tiles runs as ordinary tiles (A/B of the cooperative tail)
 */
#define RQHIP_FWD_SCAN_FP32 0x1u
#define RQHIP_FWD_SCAN_VALU 0x2u
#define RQHIP_FWD_NO_COOP_TAIL 0x10u
int rqhip_rq_forward_ex(const float *res0, int64_t B, int D, const float *codebooks, int L, int K,
                        int mode, float beta, int64_t *ids, float *embs, float *residuals,
                        float *emb_sum, float *loss, float *embs_norm, float *tie_margin, void *workspace,
                        size_t workspace_bytes, unsigned flags, rqhip_stream_t stream);
/* The filtered scan's too-close-to-call threshold, distance units: a row is re-decided exactly unless the gap between its
 * two smallest approximate distances exceeds  c1 * |x| * max_k|c_k| + c2 * (|x|^2 + max_k|c_k|^2).  Host-side accessor (no
 * GPU needed): tests/test_filter_bound.py checks the constants against the error bound derived in DESIGN.md section 4.1. */
void rqhip_filter_bound(float *c1, float *c2);             /* D = 32 */
void rqhip_filter_bound_d(int D, float *c1, float *c2);    /* per embedding width: D = 64 uses c1 = 2^-11 */
/* Test hook for that bound: scores[b,k] = the approximate score x_b.c_k - |c_k|^2/2 exactly as the filtered scan's
 * matrix-instruction chain produces it (same staging, same split, same instruction order), D = 32 / 64, x [B,D],
 * codebook [K,D], scores [B,K].  tests/test_gpu_filter_bound.py compares it with the real-arithmetic score on adversarial
 * operands: the hardware's accumulation error must stay inside the share of the bound assigned to it.
 * workspace: rqhip_rq_forward_workspace_bytes(1, K). */
int rqhip_filter_scores(const float *x, int64_t B, int D, const float *codebook, int K, float *scores,
                        void *workspace, size_t workspace_bytes, rqhip_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Residual quantisation, backward: what torch.autograd computes through the code above
 * (embedding backward + STE / rotation / eval branches + QuantizeLoss), in closed form.
 *
 *   upstream gradients (each may be NULL = zeros):
 *     g_embs [L,B,D] wrt embs, g_embsum [B,D] wrt emb_sum, g_resid [L,B,D] wrt residuals,
 *     g_loss [B] wrt loss
 *   outputs: g_res0 [B,D] (may be NULL; exact per-row arithmetic), g_codebooks [L,K,D] (may be NULL; OVERWRITTEN).
 *     For L <= 4 and either D <= 32 or (EVAL / STE, D % 4 == 0, D <= 64) -- rqhip_rq_backward_plan returns 1 -- the rows
 *     of a code are summed in a FIXED order: no atomics, bit-reproducible, restated by the oracle.  Other shapes scatter
 *     with LDS float atomics: the sum order is then not fixed and g_codebooks is reproducible to fp32 rounding only.
 *   workspace: rqhip_rq_backward_workspace_bytes(B,D,L,K) bytes
 */
size_t rqhip_rq_backward_workspace_bytes(int64_t B, int D, int L, int K);
/* 1 when a fixed-order kernel is used (tensors 16-byte aligned, as torch allocates them); then *n_wg, *units_per_wg
 * and *unit_rows give the geometry the summation order is a function of (workgroup b takes the unit_rows-row units
 * (round * units_per_wg + j) * n_wg + b in ascending order: oracle/rq_oracle.c:rqo_rq_backward_ordered) */
int rqhip_rq_backward_plan(int64_t B, int D, int L, int K, int mode, int *n_wg, int *units_per_wg, int *unit_rows);
int rqhip_rq_backward(const float *res0, int64_t B, int D, const float *codebooks, int L, int K,
                      int mode, float beta, const int64_t *ids, const float *g_embs,
                      const float *g_embsum, const float *g_resid, const float *g_loss,
                      float *g_res0, float *g_codebooks, void *workspace, size_t workspace_bytes,
                      rqhip_stream_t stream);
/* The same call with a kernel-selection flag (rqhip_rq_backward passes 0).  RQHIP_BWD_CBGRAD_MATRIX: where rqhip_rq_backward_matrix_form(D,
 * K, L, mode) says so (D = 32, STE, the training step's upstream gradients g_embsum + g_loss only; 3 x <= 256 or 3-4 x 1024 codes) the
 * codebook gradient is accumulated as a one-hot matrix product on the bf16 matrix cores (three exact bf16 pieces of every staged fp32
 * value; the sum's order is the matrix pipe's: reproducible run to run, not restatable by the oracle -- held to "no further from fp64
 * than the ordered kernel"); g_res0 has the same bits either way.  Other shapes ignore the flag. */
#define RQHIP_BWD_CBGRAD_MATRIX 0x1u
int rqhip_rq_backward_matrix_form(int D, int K, int L, int mode);
int rqhip_rq_backward_ex(const float *res0, int64_t B, int D, const float *codebooks, int L, int K, int mode, float beta,
                         const int64_t *ids, const float *g_embs, const float *g_embsum, const float *g_resid, const float *g_loss,
                         float *g_res0, float *g_codebooks, void *workspace, size_t workspace_bytes, unsigned flags,
                         rqhip_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * One Gumbel-softmax level, training mode.  Replaces quantize.py:112-117,128,131-136,157 and
 * distributions/gumbel.py:8-20.  The uniform noise U [B,K] is supplied by the caller (torch.rand on
 * the device), exactly where the reference draws it.
 *   outputs: ids [B] (argmin of the noise-free distance), emb [B,D] (= embeddings), loss [B]
 */
int rqhip_gumbel_forward(const float *x, int64_t B, int D, const float *codebook, int K,
                         const float *U, float temperature, float beta, int64_t *ids, float *emb,
                         float *loss, rqhip_stream_t stream);
/*   g_emb [B,D] / g_loss [B] upstream (may be NULL); outputs g_x [B,D], g_codebook [K,D] (overwritten).
 *   workspace: rqhip_gumbel_backward_workspace_bytes(B,D,K) */
size_t rqhip_gumbel_backward_workspace_bytes(int64_t B, int D, int K);
/* Both Gumbel entry points pick between two implementations with the same results: from `min_rows` rows (default 4096;
 * D = 32, K in {32, 64, 128, 256}, 16-byte aligned rows) the matrix-instruction kernels (32 rows per wave), below it the
 * one-row-per-wave kernels.  set_to > 0 changes the process-wide threshold (tests force the matrix path on small ragged
 * batches with 1); returns the previous value; set_to <= 0 only queries. */
int64_t rqhip_gumbel_matrix_path_min_rows(int64_t set_to);
int rqhip_gumbel_backward(const float *x, int64_t B, int D, const float *codebook, int K,
                          const float *U, float temperature, float beta, const float *g_emb,
                          const float *g_loss, float *g_x, float *g_codebook, void *workspace,
                          size_t workspace_bytes, rqhip_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * k-means codebook initialisation (init/kmeans.py).  The Lloyd loop, the np.random.choice seeding
 * and the torch.randint reseed of empty clusters stay on the host (they consume host RNG streams
 * the reference's results depend on); rqhip_kmeans_assign / _update are the two data-parallel steps of one
 * iteration, rqhip_kmeans_lloyd runs batches of iterations between two host visits.
 *
 * rqhip_kmeans_assign : kmeans.py:40-43  assign[i] = argmin_k sum_d (x[i,d]-c[k,d])^2
 * rqhip_kmeans_update : kmeans.py:44-59  c[k] <- mean of its rows (ascending-row sum / count);
 *                       empty clusters are left untouched, counts[k] == 0 reports them;
 *                       shift_sq_max (device scalar, may be NULL) <- max_k sum_d (c_new-c_old)^2,
 *                       i.e. the square of kmeans.py:68's torch.norm(...).max(), computed BEFORE any
 *                       host reseed (the host adds the reseeded rows' shift itself).
 */
int rqhip_kmeans_assign(const float *x, int64_t B, int D, const float *centroids, int K,
                        int64_t *assign, rqhip_stream_t stream);
int rqhip_kmeans_update(const float *x, int64_t B, int D, const int64_t *assign, int K,
                        float *centroids, int64_t *counts, float *shift_sq_max,
                        rqhip_stream_t stream);
/* rqhip_kmeans_lloyd : a BATCH of up to n_iters Lloyd iterations (kmeans.py:64-70) with no host round trip.
 *   Enqueues n_iters x (assign, update, finalize); each launch first reads state[0] and returns at once when an
 *   earlier iteration of the batch stopped the run.  state: 4 device ints, zero them before the first batch --
 *     state[0] stop flag: 0 running, 1 converged (sqrt of the max squared shift < stop_threshold, kmeans.py:68-69),
 *              2 an empty cluster appeared: its reseed draws from the host's torch RNG (kmeans.py:50-54), so the host
 *              reseeds (counts[k] == 0 names the clusters; their centroids are untouched), clears state[0] and goes on;
 *     state[1] iterations completed so far (the one that raised a flag included);
 *     state[2] fp32 bits of max_k |c_new - c_old|^2 of the last completed iteration (empty clusters excluded).
 *   centroids [K,D] updated in place, assign [B] / counts [K] hold the last completed iteration's values.
 *   The host reads the 16-byte state once per batch. */
int rqhip_kmeans_lloyd(const float *x, int64_t B, int D, float *centroids, int K, int64_t *assign,
                       int64_t *counts, int *state, int n_iters, float stop_threshold, rqhip_stream_t stream);
/* Row-sharded form of one iteration: every rank owns a block of rows,
 *   rqhip_kmeans_partial_sums : assign + this rank's per-cluster sums and counts -> sums [K, D+1] fp32 (count last);
 *   <caller all-reduces `sums` over the ranks: ONE collective of K (D+1) floats per iteration, RCCL on the stream>
 *   rqhip_kmeans_apply_sums   : means, shift, empty / convergence flags from the reduced sums (identical on every rank).
 * Same state words and early-exit rule as rqhip_kmeans_lloyd, so batches of iterations (collectives included) can be
 * enqueued between two host visits. */
int rqhip_kmeans_partial_sums(const float *x, int64_t B, int D, const float *centroids, int K, int64_t *assign,
                              float *sums, int *state, rqhip_stream_t stream);
int rqhip_kmeans_apply_sums(const float *sums, int K, int D, float *centroids, int64_t *counts, int *state,
                            float stop_threshold, rqhip_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Semantic-id statistics.
 * rqhip_dedup_rank: rank[i] = number of rows j < i whose L-tuple equals row i's -- the dedup column
 *   of SemanticIdTokenizer.precompute_corpus_ids (modules/tokenizer/semids.py:92-108).  Also yields
 *   n_distinct = number of rows with no LATER duplicate, i.e. B * p_unique_ids of rqvae.py:159-167.
 *   ids [L,B] int64 with 0 <= id < K; rank [B] int64 or NULL; n_distinct device int64 scalar or NULL.
 *   workspace: rqhip_dedup_workspace_bytes(B)
 */
size_t rqhip_dedup_workspace_bytes(int64_t B);
int rqhip_dedup_rank(const int64_t *ids, int64_t B, int L, int K, int64_t *rank,
                     int64_t *n_distinct, void *workspace, size_t workspace_bytes,
                     rqhip_stream_t stream);
/* p_unique_ids of RqVae.forward (modules/rqvae.py:159-167) as the reference returns it: *p_unique = float(n_distinct) / float(B), a device
 * fp32 scalar (torch: int64 tensor / int -> both to fp32, IEEE division); B >= 1; one fill + one kernel; workspace as rqhip_dedup_rank. */
int rqhip_unique_fraction(const int64_t *ids, int64_t B, int L, float *p_unique, void *workspace, size_t workspace_bytes,
                          rqhip_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Decoder-side consumers of semantic ids: exact integer tuple matching.
 *
 * Prefix index -- replaces EncoderDecoderRetrievalModel._check_valid_prefix (modules/model.py:169-182), the
 *   [N, P, h] equality tensor rebuilt at every beam step of generate() (model.py:349,364).
 *   corpus [N, H] int64, row stride ld elements (>= H): the model's `codebooks` buffer (model.py:75), i.e.
 *   tokenizer.cached_ids[:, :n_layers] (train_decoder.py:131).  Any int64 values are accepted.
 *   rqhip_prefix_index_build: fills `index` (rqhip_prefix_index_bytes(N, H) bytes, caller-owned, opaque) with
 *     the set of all prefixes of length 1..H of all corpus rows.  Build once per corpus.
 *   rqhip_prefix_lookup: valid[q] = 1 iff prefix[q, 0:h] equals corpus[n, 0:h] for some n; prefix [P, h]
 *     int64 with row stride ldp (>= h), 0 <= h <= H (h = 0: valid iff N > 0, as all()/any() give), valid [P]
 *     bytes (a torch.bool buffer).  `corpus`, N, H, ld must be the ones the index was built from: the index
 *     stores row numbers and every hit is confirmed against the corpus row itself.
 * Top-k match -- the array step of TopKAccumulator.accumulate (evaluate/metrics.py:16-25):
 *   rank[b] = the first k with top_k[b, k, :] == actual[b, :], or -1; actual [B, D], top_k [B, K, D], dense.
 */
#define RQHIP_MAX_PREFIX_LEN 16
size_t rqhip_prefix_index_bytes(int64_t N, int H);
int rqhip_prefix_index_build(const int64_t *corpus, int64_t N, int H, int64_t ld, void *index,
                             size_t index_bytes, rqhip_stream_t stream);
int rqhip_prefix_lookup(const void *index, size_t index_bytes, const int64_t *corpus, int64_t N, int H,
                        int64_t ld, const int64_t *prefix, int64_t P, int h, int64_t ldp, uint8_t *valid,
                        rqhip_stream_t stream);
int rqhip_topk_first_match(const int64_t *actual, const int64_t *top_k, int64_t B, int K, int D,
                           int64_t *rank, rqhip_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Reconstruction loss (modules/loss.py:5-10 ReconstructionLoss, called at modules/rqvae.py:152), fused.
 *   forward : out[b] = sum_d (x_hat[b,d] - x[b,d])^2        x_hat, x: [B,N] with row strides ld_* (elements, >= N)
 *   backward: g_x_hat[b,d] = 2 (x_hat[b,d] - x[b,d]) g_out[b]; g_x = -g_x_hat; either output may be NULL;
 *             outputs are dense [B,N].
 */
int rqhip_recon_loss_forward(const float *x_hat, int64_t ld_hat, const float *x, int64_t ld_x, int64_t B,
                             int N, float *out, rqhip_stream_t stream);
int rqhip_recon_loss_backward(const float *x_hat, int64_t ld_hat, const float *x, int64_t ld_x,
                              const float *g_out, int64_t B, int N, float *g_x_hat, float *g_x,
                              rqhip_stream_t stream);

/* Speculative form of the pair above for the training step: the forward also writes the gradient it expects to be
 * asked for, g_spec[b,:] = (2 (x_hat - x)) * row_scale -- `(reconstruction + quantize_loss).mean().backward()`
 * (modules/rqvae.py:152-154) sends row_scale = 1/B to every row -- and the backward only re-does rows whose upstream
 * gradient g_out[b] is NOT bit-identical to row_scale (exactly what rqhip_recon_loss_backward would write).  Same
 * results as the plain pair in every case; one HBM pass instead of two when the expectation holds.
 * Needs N and the strides multiples of 4 and 16-byte aligned pointers (else RQHIP_EUNSUPPORTED: use the plain pair). */
int rqhip_recon_loss_forward_spec(const float *x_hat, int64_t ld_hat, const float *x, int64_t ld_x, int64_t B, int N,
                                  float row_scale, float *out, float *g_spec, rqhip_stream_t stream);
int rqhip_recon_loss_backward_spec(const float *x_hat, int64_t ld_hat, const float *x, int64_t ld_x,
                                   const float *g_out, int64_t B, int N, float row_scale, float *g_spec,
                                   rqhip_stream_t stream);

/* The three batch means of RqVae.forward (modules/rqvae.py:154,171-172) in one launch:
 *   out3[0] = mean(recon + quant), out3[1] = mean(recon), out3[2] = mean(quant);  recon, quant [B] fp32, B >= 1. */
int rqhip_loss_means(const float *recon, const float *quant, int64_t B, float *out3, rqhip_stream_t stream);
/* the same means by many workgroups (one launch; block partials met by the last block to arrive, in block order: deterministic).
 * workspace: rqhip_loss_means_workspace_bytes() bytes, 16-byte aligned, zeroed ONCE by the caller and then reusable launch after launch
 * on one stream (the kernel re-arms its counter). */
size_t rqhip_loss_means_workspace_bytes(void);
int rqhip_loss_means_ws(const float *recon, const float *quant, int64_t B, float *out3, void *workspace, size_t workspace_bytes,
                        rqhip_stream_t stream);
/* Its backward (autograd of the three `.mean()`s): g_loss, g_recon_mean, g_quant_mean are device scalars (gradients wrt
 * out3[0..2]; each may be NULL = no gradient).  rows_recon[i] = (g_loss + g_recon_mean) * (1/B) and
 * rows_quant[i] = (g_loss + g_quant_mean) * (1/B) for every i < B, 1/B rounded to fp32 first as PyTorch's mean backward
 * does on the device (either output may be NULL). */
int rqhip_loss_means_backward(const float *g_loss, const float *g_recon_mean, const float *g_quant_mean, int64_t B,
                              float *rows_recon, float *rows_quant, rqhip_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Weight gradient of a bias-free Linear(+ReLU) layer with the ReLU backward fused in (SURVEY.md section 8 row f2;
 * reference modules/encoder.py:25-38, autograd of `relu(x @ W.T)`).
 *   g [M,N] gradient wrt the layer output, y [M,N] the layer's (ReLU) output or NULL (no ReLU), x [M,K] its input;
 *   all dense row-major fp32, 16-byte aligned.
 *   dW [N,K] = g_pre^T x  with  g_pre = g where y > 0 else 0  (g itself when y is NULL)      -- overwritten
 *   g_masked [M,N] or NULL: receives g_pre for the data-gradient GEMM that follows; may alias g.
 * The batch rows are split into ranges reduced in a fixed order (workspace holds the partial blocks): the result is
 * bit-reproducible.  Layer shapes: rqhip_linear_wgrad_supported(N, K) (N, K multiples of the 32..256 tile shapes
 * listed in csrc/wgrad.hip; every layer of the shipped 768-512-256-128-32 MLPs); others return RQHIP_EUNSUPPORTED
 * and the caller keeps the library GEMM.
 */
int rqhip_linear_wgrad_supported(int N, int K);
/* returns the tile configuration (>= 0) or -1, and the number of row ranges the kernel will reduce over (tests restate
 * the summation order with it) */
int rqhip_linear_wgrad_plan(int64_t M, int N, int K, int *msplit);
size_t rqhip_linear_wgrad_workspace_bytes(int64_t M, int N, int K);
int rqhip_linear_wgrad(const float *g, const float *y, const float *x, int64_t M, int N, int K,
                       float *g_masked, float *dW, void *workspace, size_t workspace_bytes,
                       rqhip_stream_t stream);
/* The same call with a kernel-selection flag (rqhip_linear_wgrad passes 0).  Layers whose dimensions are multiples of
 * 128 x 256 / 256 x 128 run by default on the bf16 matrix cores with every fp32 operand split into three bf16 pieces and
 * the six piece products that matter (dropped terms <= 2^-23 of a product: below fp32's own rounding of it; fp32
 * accumulation; csrc/wgrad_split.hip): same fixed row ranges and reduction tree, bit-reproducible run to run, but the
 * order inside the matrix instruction is not one the oracle can restate, so that result is held to "no less exact than the
 * library's fp32 GEMM against fp64", not bit-exactness.  RQHIP_WGRAD_FP32 selects the fp32-MFMA kernel whose summation
 * order oracle/rq_oracle.c:rqo_linear_wgrad restates bit for bit (and which small layers always use). */
#define RQHIP_WGRAD_FP32 0x1u
int rqhip_linear_wgrad_ex(const float *g, const float *y, const float *x, int64_t M, int N, int K,
                          float *g_masked, float *dW, void *workspace, size_t workspace_bytes,
                          unsigned flags, rqhip_stream_t stream);
/* The weight gradient in RQHIP_SPLIT_F16X2 arithmetic (the product path for the layers csrc/wgrad_split.hip tiles; see the
 * GEMM section below): the reduction runs over the batch rows, so the exact power-of-two scales are per COLUMN of g and of
 * x -- g_col_max [N], x_col_max [K]: bit patterns of the columns' largest |value| (rqhip_maxima, or the c_col_max of the
 * epilogue that wrote the matrix); two fp16 pieces per scaled value, products hh + hm + mh, the result multiplied back by
 * 2^(e_n + e_k).  With y given the ReLU mask is applied inside as above and g_col_max may hold the maxima of the unmasked
 * g (an upper bound costs low-order bits of the small elements only).  Same row ranges, reduction tree and workspace as
 * rqhip_linear_wgrad; layers the split kernel does not tile run the fp32 kernel and ignore the maxima. */
int rqhip_linear_wgrad_f16(const float *g, const float *y, const float *x, int64_t M, int N, int K,
                           const unsigned *g_col_max, const unsigned *x_col_max, float *g_masked, float *dW,
                           void *workspace, size_t workspace_bytes, rqhip_stream_t stream);
/* Several layers' weight gradients in ONE launch (ABI 462): dW_j [N_j, K_j] = g_j^T x_j for 2..4 layers over the same M rows, EITHER every dW
 * tiled 256 x 256 (N_j, K_j multiples of 256) OR every dW tiled 128 x 256 / 256 x 128 (one dimension a multiple of 256, the other of 128 but
 * not 256; the two kinds mix freely), g_j ALREADY masked by its layer's ReLU, f16x2 arithmetic under the column maxima (as
 * rqhip_linear_wgrad_f16).  The jobs share one number of row ranges (rqhip_linear_wgrad_f16_batch_plan: CUs / tiles of all jobs; 0 = not
 * batchable), so the launch writes and reduces one workgroup's worth of partial blocks per CU for ALL its layers instead of per layer --
 * the weight gradients of reference modules/encoder.py:25-38's Linear layers, which autograd forms one by one.  Results are those of
 * rqhip_linear_wgrad_f16 run with that number of ranges (bit-reproducible; another balanced tree than the per-layer call's). */
typedef struct {
    const float *g;              /* [M, N] */
    const float *x;              /* [M, K] */
    int N, K;
    const unsigned *g_col_max;   /* [N] */
    const unsigned *x_col_max;   /* [K] */
    float *dW;                   /* [N, K] */
} rqhip_wgrad_job;
int rqhip_linear_wgrad_f16_batch_plan(int64_t M, const int *N, const int *K, int n);
size_t rqhip_linear_wgrad_f16_batch_workspace_bytes(int64_t M, const int *N, const int *K, int n);
int rqhip_linear_wgrad_f16_batch(const rqhip_wgrad_job *jobs, int n, int64_t M, void *workspace, size_t workspace_bytes, rqhip_stream_t stream);

/* The weight gradients of SEVERAL layers in one launch, for the batch sizes the reference's gin files train with (64-640 rows:
 * configs/rqvae_ml32m.gin, rqvae_amazon.gin), where the kernels above are latency and launches (csrc/wgrad_jobs.hip).  Job i:
 * dW[i] [N[i], K[i]] = g[i]^T x[i] with g[i] [M, N[i]] ALREADY masked by the layer's ReLU and x[i] [M, K[i]], all row-major
 * fp32; every job has the same M.  g, x, dW, N, K are HOST arrays of n_jobs <= 8 entries (read during the call).  Every (job,
 * 64 x 64 block of dW) is one workgroup that reduces over all M rows -- no row ranges, no workspace, no reduction launch, a fixed
 * summation order, a job's bits independent of the other jobs -- in three-piece bf16 arithmetic: v = h + m + l exactly, the six
 * piece products that matter (dropped terms <= 2^-23 of a product), fp32 accumulation; an entry's error is within
 * (sqrt(M) + 8) 2^-24 of the sum of its terms' magnitudes (tests/test_gpu_wgrad.py).  Shapes: N and K multiples of 32
 * (rqhip_linear_wgrad_jobs_supported); anything else is RQHIP_EARG.  Correct for any M; meant for M up to a few thousand. */
int rqhip_linear_wgrad_jobs_supported(int N, int K);
int rqhip_linear_wgrad_jobs(const float *const *g, const float *const *x, float *const *dW, const int *N, const int *K,
                            int n_jobs, int64_t M, rqhip_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * The activation GEMMs of the MLPs (reference modules/encoder.py:25-38: `relu(x W^T)` forward; autograd's `g W` data
 * gradient and ReLU backward; modules/rqvae.py:146,152 + modules/loss.py:5-10: the last decoder layer with the
 * reconstruction loss) on the 16-bit matrix cores, carrying fp32's accuracy (csrc/gemm_split.hip).  Two arithmetics:
 *   RQHIP_SPLIT_F16X2  (the product path) every row of A and every weight row is scaled by an exact power of two (the
 *       exponent of its largest |value|), every scaled value is split into two fp16 pieces (11 + 11 significant bits and a
 *       sign), the product is hh + hm + mh (each piece product exact in fp32, fp32 accumulation), and the epilogue multiplies
 *       back by 2^(e_row + e_column).  Needs the row maxima of A: `a_row_max`, written by the kernel that produced A (every
 *       epilogue below can emit the row / column maxima of what it stores) or by rqhip_maxima.
 *   RQHIP_SPLIT_BF16X3 (round 3, kept for A/B) three exact bf16 pieces per operand, six products, no scaling.
 * Both are held to: max error against fp64 <= the library fp32 GEMM's on the same inputs (tests/test_gpu_gemm_split.py:
 * unit-norm rows, post-ReLU activations, 1e-5-scale masked gradients, twelve decades of row scales, five decades inside a
 * row, the worst-case mantissas of the 11-bit split, cancellation-heavy rows), results bit-reproducible run to run.
 *
 *   rqhip_weight_images : split weight matrices w [rows, cols] into the kernel's images, ALL jobs in one launch (an MLP's
 *       layers, both directions).  transpose = 0: the image of w itself (Nc = rows output columns, reduction R = cols: the
 *       forward, C = A w^T); transpose = 1: of w^T (Nc = cols, R = rows: the data gradient, C = A w).  `image`:
 *       rqhip_weight_image_bytes(Nc, R, arith) bytes, caller-owned, 16-byte aligned.  `jobs` is a HOST array.
 *   rqhip_maxima        : row_max [M] and / or col_max [R] (bit patterns of the largest |value|; col_max is maxed into
 *       atomically: zero it first) of A [M, R] in one pass; with Y given, of A masked by Y > 0 (the ReLU backward), which is
 *       also written to masked_out when that is not NULL.  R % 4 == 0, R <= 16384 (columns are taken in chunks of 1024).
 *   rqhip_gemm_split_ex : C [M, Nc] = epilogue(A [M, R] . image^T).  Needs Nc % 128 == 0 (256-column tiles when
 *       Nc % 256 == 0, else 128), R % 16 == 0 (rqhip_gemm_split_supported), 16-byte aligned pointers; one launch at a time
 *       per image (it holds the kernel's tile dispenser).
 */
#define RQHIP_SPLIT_F16X2 0
#define RQHIP_SPLIT_BF16X3 1
#define RQHIP_EPI_STORE 0
#define RQHIP_EPI_RELU 1  /* C = relu(A.B^T) */
#define RQHIP_EPI_RECON 2 /* the last decoder layer fused with ReconstructionLoss: x_hat = A.B^T is never stored; loss_rows[m] =
                             sum_n (x_hat - aux)^2 and C = (2 (x_hat - aux)) * row_scale, the gradient `(reconstruction +
                             quantize_loss).mean().backward()` sends back when row_scale = loss scale / B.  Nc % 256 == 0;
                             workspace: rqhip_gemm_split_recon_workspace_bytes(M, Nc) bytes */
#define RQHIP_EPI_MASK 3  /* C = A.B^T where aux > 0, else 0: a data gradient fused with the ReLU backward of the layer below
                             (aux = that layer's output); RQHIP_SPLIT_F16X2 only */
typedef struct {
    const float *w;      /* [rows, cols] */
    int rows, cols, transpose, arith;
    void *image;
    size_t image_bytes;
} rqhip_image_job;
typedef struct {
    const float *A;      /* [M, R] */
    int64_t M;
    int R;
    const void *image;   /* of B [Nc, R] (rqhip_weight_images) */
    int Nc;
    int arith;           /* RQHIP_SPLIT_* the image was built with */
    int epilogue;        /* RQHIP_EPI_* */
    int tile_rows;       /* 0 (tools only: staged-B kernels 256 / 128 force big tiles, 64 / 32 small ones; -9: 32-row leftover tiles allowed (A/B); -12: the atomic tile dispenser of rounds 4-5 instead of the static schedule (A/B); -8: gemm_f16_kernel with one
                            tile dispenser per XCD instead of one for the chip -- measured slower, profiles/r05_gemm_xcd_dispenser_ab.txt) */
    float *C;            /* [M, Nc] */
    const float *aux;    /* RQHIP_EPI_RECON: X [M, Nc]; RQHIP_EPI_MASK: Y [M, Nc]; else NULL */
    float row_scale;     /* RQHIP_EPI_RECON */
    float *loss_rows;    /* RQHIP_EPI_RECON: [M] */
    void *workspace;     /* RQHIP_EPI_RECON */
    size_t workspace_bytes;
    const unsigned *a_row_max; /* RQHIP_SPLIT_F16X2: [a_row_parts][M]; a row's largest |value| is the maximum over the parts */
    int a_row_parts;
    unsigned *c_row_max; /* optional output [column tiles][M] (Nc / 256 tiles when Nc % 256 == 0, else Nc / 128): largest |value| of
                            each row of C per column tile (plain stores) */
    unsigned *c_col_max; /* optional output [Nc]: largest |value| of each column of C, maxed into atomically (zero it first) */
} rqhip_gemm_args;
int rqhip_gemm_split_supported(int Nc, int R);
size_t rqhip_weight_image_bytes(int Nc, int R, int arith);
int rqhip_weight_images(const rqhip_image_job *jobs, int n_jobs, rqhip_stream_t stream);
int rqhip_maxima(const float *A, const float *Y, float *masked_out, int64_t M, int R, unsigned *row_max, unsigned *col_max,
                 rqhip_stream_t stream);
size_t rqhip_gemm_split_recon_workspace_bytes(int64_t M, int Nc);
int rqhip_gemm_split_ex(const rqhip_gemm_args *args, rqhip_stream_t stream);
/* the round-3 entry points: RQHIP_SPLIT_BF16X3 with one image per call */
size_t rqhip_weight_planes_bytes(int Nc, int R);
int rqhip_weight_planes(const float *w, int rows, int cols, int transpose, void *planes, size_t planes_bytes,
                        rqhip_stream_t stream);
int rqhip_gemm_split(const float *A, int64_t M, int R, const void *planes, int Nc, int relu, float *C,
                     rqhip_stream_t stream);
int rqhip_gemm_split_recon(const float *A, int64_t M, int R, const void *planes, int Nc, const float *X, float row_scale,
                           float *G, float *loss_rows, void *workspace, size_t workspace_bytes, rqhip_stream_t stream);
/* backward fix-up of RQHIP_EPI_RECON: rows of g_out that are not row_scale (bit compare) get G * (g / row_scale) */
int rqhip_recon_rescale_rows(const float *g_out, int64_t B, int N, float row_scale, float *g_spec, rqhip_stream_t stream);
/* ... which also brings the maxima the epilogue emitted for G up to date for the rows it changes: row_max [row_parts][B]
 * (the row's new maximum in part 0, the other parts cleared) and col_max [N] (maxed into); either may be NULL */
int rqhip_recon_rescale_rows_ex(const float *g_out, int64_t B, int N, float row_scale, float *g_spec, unsigned *row_max,
                                int row_parts, unsigned *col_max, rqhip_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * The RQ <-> MLP seam (first clause): the encoder's last Linear (128 -> D), every quantisation level and the
 * decoder's first Linear (D -> 128) + ReLU as ONE row-local launch -- reference modules/rqvae.py:118-139 (`res = self.encode(x)` ends
 * in that Linear, modules/encoder.py:25-38; the level loop) and :146 (`self.decode(embs.sum(axis=-1))` starts with the other).  D = 32
 * and a hidden width of 128 (the reference's configs/rqvae_amazon.gin); all levels' codebooks must fit the LDS next to the two weights
 * (rqhip_rq_seam_supported).  Either GEMM and the quantisation can be switched off, which makes the same kernel the stand-alone
 * 128 -> D / D -> 128 layer: the data gradients of the backward, with the ReLU backward applied on load (h_mask) or in the epilogue
 * (RQHIP_EPI_MASK).  Arithmetic: every GEMM output is ONE fp32 FMA chain over the input features in ascending order (the fp32 matrix
 * instruction, as the distance scan of rqhip_rq_forward): independent of batch size and launch form, restated by
 * oracle/rq_oracle.c:rqo_linear_chain; the quantisation is rqhip_rq_forward's, bit for bit, on the res0 the input GEMM produced.
 */
typedef struct {
    int64_t B;
    int D;                  /* 32 */
    int H;                  /* width of h / out: 128 */
    /* input side: h given -> res0 = h' . Win^T, h' = h where h_mask > 0 else 0 (h_mask NULL: h' = h); h NULL -> rows come from res0 */
    const float *h;         /* [B, H] or NULL */
    const float *h_mask;    /* [B, H] or NULL */
    const float *w_in;      /* [D, H] row-major (nn.Linear.weight of the H -> D layer); w_in_transposed: [H, D], used as its transpose */
    int w_in_transposed;
    const float *res0;      /* [B, D]: the rows when h == NULL */
    float *res0_out;        /* [B, D] or NULL: the input GEMM's result */
    /* quantisation: L == 0 skips it (the rows go straight to the output GEMM); else as rqhip_rq_forward (filtered scan, all levels resident) */
    const float *codebooks; /* [L, K, D] */
    int L, K, mode;         /* RQHIP_MODE_EVAL / STE / ROTATION */
    float beta;
    int64_t *ids;           /* [L, B] */
    float *emb_sum;         /* [B, D] or NULL */
    float *loss;            /* [B] or NULL */
    float *embs_norm;       /* [B, L] or NULL */
    /* output side: w_out given -> out = epilogue(s . Wout^T), s = the sum of the levels' outputs (L == 0: the rows) */
    const float *w_out;     /* [H, D] row-major (nn.Linear.weight of the D -> H layer); w_out_transposed: [D, H], used as its transpose; or NULL */
    int w_out_transposed;
    int out_epilogue;       /* RQHIP_EPI_STORE / RQHIP_EPI_RELU / RQHIP_EPI_MASK (out where out_mask > 0 else 0) */
    const float *out_mask;  /* [B, H]: RQHIP_EPI_MASK */
    float *out;             /* [B, H] */
    unsigned *out_row_max;  /* optional [H / 32][B]: bit patterns of the largest |value| of each row of `out` per 32-column block (the
                               a_row_max / a_row_parts = H / 32 of the rqhip_gemm_split_ex that reads `out` next) */
    unsigned *out_col_max;  /* optional [H]: column maxima of `out`, maxed into atomically (zero it first) */
} rqhip_seam_args;
int rqhip_rq_seam_supported(int D, int H, int L, int K);   /* 1 when rqhip_rq_seam takes these shapes on the current device */
int rqhip_rq_seam(const rqhip_seam_args *args, rqhip_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * The encoder / decoder Linear layers at the batch sizes the reference ships (M < 4096 rows: configs/rqvae_amazon.gin:7 batch 640,
 * rqvae_ml32m.gin:7 batch 64) -- reference modules/encoder.py:25-38 (`relu(x W^T)`) and its autograd data gradient `g W` with the ReLU
 * backward of the layer below.  out [M, N] = epilogue(a [M, Kr] . B), B = w^T for w [N, Kr] (w_kn = 0: the forward, nn.Linear.weight
 * as stored) or B = w for w [Kr, N] (w_kn = 1: the data gradient, the same nn.Linear.weight).  N, Kr multiples of 32; a, w, out, aux
 * 16-byte aligned, contiguous.  epilogue: RQHIP_EPI_STORE / RQHIP_EPI_RELU / RQHIP_EPI_MASK (out where aux [M, N] > 0, else 0).
 * Arithmetic: exact fp32 on the fp32 matrix instruction.  The reduction runs as `waves` contiguous ranges of 32-term groups (wave v:
 * groups [v ng / waves, (v + 1) ng / waves), ng = Kr / 32), each ONE fp32 FMA chain from +0 taking a group's terms in the order
 * 0 8 16 24 1 9 17 25 ... 7 15 23 31; out = ((p_0 + p_1) + ...) + p_{waves-1}, then the epilogue.  (col_blocks, waves) = (0, 0): chosen from the shape
 * (rqhip_linear_small_plan reports the choice: the same bits on every box, eager or replayed); restated by
 * oracle/rq_oracle.c:rqo_linear_small.  Valid overrides: col_blocks 1 | 2 (N % 64 == 0), waves 4 | 8 | 16 (16: col_blocks 1). */
int rqhip_linear_small_supported(int64_t M, int N, int Kr);
int rqhip_linear_small_plan(int64_t M, int N, int Kr, int *col_blocks, int *waves);
int rqhip_linear_small(const float *a, const float *w, int w_kn, float *out, int64_t M, int N, int Kr, int epilogue, const float *aux,
                       int col_blocks, int waves, rqhip_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * The AdamW update of all parameters in one launch (reference train_rqvae.py:136-138: AdamW with decoupled weight decay on every
 * parameter, codebooks included).  Arithmetic of torch's `_fused_adamw_` in fp32 (no amsgrad, no maximize); tensors p / g / m / v of
 * numel[i] contiguous fp32 elements, 16-byte aligned, caller-owned; `step`: device float scalar = steps taken so far, incremented by
 * the call (on the device, so a captured hipGraph advances it on replay); `scratch`: 8 device bytes (8-byte aligned) the call may overwrite. */
int rqhip_adamw_step(float *const *p, const float *const *g, float *const *m, float *const *v, const int64_t *numel, int n,
                     float *step, unsigned *scratch, float lr, float beta1, float beta2, float eps, float weight_decay,
                     rqhip_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Kernel timing for bench.py's roofline objects (no reference counterpart).  While enabled, the calls below bracket
 * their MAIN kernel(s) with a hipEvent pair recorded on the call's stream and note what the launch was: a tag, its
 * algorithmic FLOPs and its algorithmic bytes.  rqhip_profile_read_tagged synchronises the recorded events and returns
 * the per-launch records (at most `cap`), then clears the log; rqhip_profile_read returns the durations of the
 * RQHIP_PROF_RQ_FORWARD records only (round 1's interface) and clears the log too.
 */
#define RQHIP_PROF_RQ_FORWARD 1  /* rqhip_rq_forward(_ex): the scan kernel, not the codebook-norm prologue */
#define RQHIP_PROF_RQ_BACKWARD 2 /* rqhip_rq_backward: all of its kernels (flat kernel + table reduce) */
#define RQHIP_PROF_GEMM_SPLIT 3  /* rqhip_gemm_split_ex */
#define RQHIP_PROF_WGRAD 4       /* rqhip_linear_wgrad*: the kernel and its partial-sum reduction */
#define RQHIP_PROF_MAXIMA 5      /* rqhip_maxima */
#define RQHIP_PROF_IMAGES 6      /* rqhip_weight_images */
#define RQHIP_PROF_SEAM 7        /* rqhip_rq_seam: flops = the GEMMs' 2 B D H each + the levels' L (2 D K + 5 D) per row */
#define RQHIP_PROF_LINEAR_SMALL 8 /* rqhip_linear_small: flops = 2 M N Kr */
typedef struct {
    int tag;
    float ms;
    double flops, bytes;
} rqhip_profile_record;
int rqhip_profile_enable(int max_records); /* 0 disables and frees the events */
int rqhip_profile_select(unsigned tag_mask); /* bit t set: record launches tagged t (default: all) */
int rqhip_profile_read(float *ms_out, int cap, int *n_out);
int rqhip_profile_read_tagged(rqhip_profile_record *out, int cap, int *n_out);

#ifdef __cplusplus
}
#endif
#endif /* RQHIP_H */
