// gumbel_mfma.hip -- Gumbel-softmax quantisation level on the matrix instructions (gfx950), the fast path of
// gumbel.hip for the shipped latent width: D == 32, K in {32, 64, 128, 256}, 16-byte aligned rows.
//
// Same mathematics as gumbel.hip (reference modules/quantize.py:112-117,128,131-136,157 and
// distributions/gumbel.py:8-20; closed-form backward in oracle/rq_oracle.c:rqo_gumbel_backward), different shape:
// a wave owns 32 rows, not one, and every [rows x codes] x [codes x features] product is a chain of
// v_mfma_f32_32x32x2_f32:
//
//   dist  = |x|^2 + |c|^2 - 2 x.c     A = codes (LDS image [d-quad][parity][code][4], as rq_forward), B = rows in pair
//                                     layout: k runs over d = 0,1,2,... so the distances -- and therefore the
//                                     noise-free argmin ids -- are bit-identical to the oracle's FMA chain
//   y, w  = softmax((-dist + gumbel(U)) / T)   in the accumulator layout: lane (item il, h) holds, for every code
//                                     tile t, the 16 codes 32 t + 8 (j >> 2) + 4 h + (j & 3); K/32 x 16 registers
//   emb   = w @ C                     A = C row-major in LDS ([code][33]), B = w straight from those registers (the
//                                     accumulator layout of one product IS the B-operand layout of the next: no
//                                     transposition), output lane (item il, h) holds features 8 a + 4 h + b
//                                     ("quad layout", r = 4 a + b): rows are loaded / stored in that layout as float4
//
// Transcendentals and the summation orders of softmax / emb differ from the oracle's (which is a sequential loop):
// results agree to ~1e-6 relative like gumbel.hip's, ids are exact.
#include "gumbel_mfma.h"

#include <stdlib.h>
#include "rq_rowmath.h"
#include "rqhip_common.h"

namespace rqhip {




typedef float gm_f32x16 __attribute__((ext_vector_type(16)));
typedef float gm_f32x4 __attribute__((ext_vector_type(4)));

constexpr int kGmThreads = 256;
constexpr int kGmD = 32;

// LDS (floats): imgA [4][2][K][4] | csq [K] | crow [K][33]
__host__ __device__ inline size_t gm_lds_floats(int K) { return (size_t)K * 32 + K + (size_t)K * 33; }

__device__ __forceinline__ void gm_stage(float *sm, const float *__restrict__ cb, int K) {
    float *imgA = sm, *csq = sm + (size_t)K * 32, *crow = csq + K;
    // code c, float4 group d4 (features 4 d4 .. 4 d4 + 3): pair-layout image and the row-major copy
    for (int e = threadIdx.x; e < K * 8; e += kGmThreads) {
        const int c = e >> 3, d4 = e & 7;
        const gm_f32x4 v = *reinterpret_cast<const gm_f32x4 *>(cb + (size_t)c * kGmD + 4 * d4);
        const int q = d4 >> 1, j = (d4 & 1) * 2;  // kk = 2 d4, 2 d4 + 1 -> quad q = kk >> 2, slot kk & 3
        float *even = imgA + ((size_t)(q * 2 + 0) * K + c) * 4 + j;
        float *odd = imgA + ((size_t)(q * 2 + 1) * K + c) * 4 + j;
        even[0] = v.x; even[1] = v.z;
        odd[0] = v.y; odd[1] = v.w;
        float *r = crow + (size_t)c * 33 + 4 * d4;
        r[0] = v.x; r[1] = v.y; r[2] = v.z; r[3] = v.w;
    }
    __syncthreads();
    for (int c = threadIdx.x; c < K; c += kGmThreads) {  // sumsq2 of code c: parity accumulators (oracle's order)
        float a0 = 0.0f, a1 = 0.0f;
        for (int d = 0; d < kGmD; d += 2) {
            const float u = crow[(size_t)c * 33 + d], v = crow[(size_t)c * 33 + d + 1];
            a0 = a0 + u * u;
            a1 = a1 + v * v;
        }
        csq[c] = a0 + a1;
    }
    __syncthreads();
}

// rows in quad layout: lane (il, h) register r = 4 a + b holds feature 8 a + 4 h + b
__device__ __forceinline__ void gm_load_quad(const float *__restrict__ row_base, int h, float (&v)[16]) {
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const gm_f32x4 q = *reinterpret_cast<const gm_f32x4 *>(row_base + 8 * a + 4 * h);
        v[4 * a + 0] = q.x; v[4 * a + 1] = q.y; v[4 * a + 2] = q.z; v[4 * a + 3] = q.w;
    }
}
__device__ __forceinline__ void gm_store_quad(float *__restrict__ row_base, int h, const float (&v)[16]) {
#pragma unroll
    for (int a = 0; a < 4; ++a)
        *reinterpret_cast<gm_f32x4 *>(row_base + 8 * a + 4 * h) =
            gm_f32x4{v[4 * a + 0], v[4 * a + 1], v[4 * a + 2], v[4 * a + 3]};
}

// NT = K / 32 code tiles
template <int NT>
__global__ __launch_bounds__(kGmThreads) void gumbel_mfma_forward_kernel(const GumbelMfmaParams p) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    constexpr int K = NT * 32;
    const float *imgA = sm, *csq = sm + (size_t)K * 32, *crow = csq + K;
    gm_stage(sm, p.cb, K);
    const float inv_t = 1.0f / p.temperature;

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int il = lane & 31, h = lane >> 5;
    constexpr int kWaves = kGmThreads / 64;
    const long long nw = (long long)gridDim.x * kWaves;
    const gm_f32x4 *img4 = reinterpret_cast<const gm_f32x4 *>(imgA);

    for (long long tile = (long long)wave * gridDim.x + blockIdx.x; tile < p.n_tiles; tile += nw) {
        const long long row = tile * 32 + il;
        const bool ok = row < p.B;
        const long long rc = ok ? row : p.B - 1;
        float xp[16], xq[16];
        load_pair_row_vec<16>(p.x + (size_t)rc * kGmD, h, xp);
        gm_load_quad(p.x + (size_t)rc * kGmD, h, xq);
        const float xsq = pair_sumsq<16>(xp);

        float y[NT][16];
        float lbest = __builtin_inff();
        int lidx = 0x7fffffff, nanidx = 0x7fffffff;
        float mx = -__builtin_inff();
        const float *urow = p.U + (size_t)rc * K + 4 * h;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            gm_f32x4 u4[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) u4[g] = *reinterpret_cast<const gm_f32x4 *>(urow + 32 * t + 8 * g);
            gm_f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const gm_f32x4 a = img4[(size_t)(q * 2 + h) * K + t * 32 + il];
#pragma unroll
                for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], xp[4 * q + i], acc, 0, 0, 0);
            }
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int code = 32 * t + 8 * (j >> 2) + 4 * h + (j & 3);
                const float tt = xsq + csq[code];
                const float dv = tt - 2.0f * acc[j];  // quantize.py:113-117 (2*acc is exact)
                if (dv != dv) nanidx = min(nanidx, code);
                else if (dv < lbest) { lbest = dv; lidx = code; }  // codes ascend with (t, j): first minimum kept
                const float u = u4[j >> 2][j & 3];
                const float gn = gm_gumbel(u);                       // gumbel.py:10-11
                const float yy = ((-dv) + gn) * inv_t;               // gumbel.py:18
                y[t][j] = yy;
                mx = fmaxf(mx, yy);
            }
        }
        {  // the other half of the row's codes lives in lane ^ 32
            const int on = shfl_xor32(nanidx);
            const float ob = shfl_xor32(lbest);
            const int oi = shfl_xor32(lidx);
            nanidx = min(nanidx, on);
            if (ob < lbest || (ob == lbest && oi < lidx)) { lbest = ob; lidx = oi; }
            mx = fmaxf(mx, shfl_xor32(mx));
        }
        const int id = nanidx != 0x7fffffff ? nanidx : (lidx != 0x7fffffff ? lidx : 0);
        float zs = 0.0f;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const float ev = gm_exp(y[t][j] - mx);
                y[t][j] = ev;
                zs = zs + ev;
            }
        const float Z = zs + shfl_xor32(zs);
        const float rz = 1.0f / Z;

        // emb = w @ C: reduction over the codes, two per instruction (k = 0: this lane half's code, k = 1: the
        // other half's), 16 x NT instructions; A = C[code][d = lane & 31]
        gm_f32x16 eacc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int code = 32 * t + 8 * (j >> 2) + 4 * h + (j & 3);
                const float wv = y[t][j] * rz;  // softmax weight (gumbel.py:19)
                y[t][j] = wv;
                eacc = __builtin_amdgcn_mfma_f32_32x32x2f32(crow[(size_t)code * 33 + il], wv, eacc, 0, 0, 0);
            }
        // eacc[r]: feature 8 (r >> 2) + 4 h + (r & 3) of item il  == quad layout
        float e[16];
        float s = 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            e[r] = eacc[r];
            const float df = xq[r] - e[r];
            s = s + df * df;
        }
        const float ssum = s + shfl_xor32(s);
        if (ok) {
            gm_store_quad(p.emb + (size_t)row * kGmD, h, e);
            if (h == 0) {
                p.ids[row] = id;
                p.loss[row] = ssum + p.beta * ssum;  // loss.py:38-41
            }
        }
    }
}

// ---- backward ------------------------------------------------------------------------------------------------
// Per 32-row tile (after replaying the forward up to w and emb):
//   ge    = g_emb + 2 (emb - x) g_loss                                  quad layout
//   dw    = ge . C^T          A = C row-major (codes), B = ge           16 MFMAs per code tile, accumulator layout
//   sw    = sum_k w_k dw_k    needs every tile, dd below needs sw: the dw chain is run twice instead of keeping
//                             K/32 x 16 more registers alive
//   dd    = -(w (dw - sw)) / T
//   g_x   = 2 x sum_k dd_k - 2 dd @ C + 2 beta (x - emb) g_loss         A = C row-major (features), B = dd
//   gC   += w^T ge + dd^T (-2 x)   [codes x features], reduction over the tile's 32 ROWS: both operands have to
//                             be indexed by row along k, so w, dd, ge, -2x go through a wave-private LDS transpose
//                             ([row][33]); accumulated across all tiles of the wave in K/32 x 16 registers
//   S_k  += sum_rows dd_k     (the A operands of the second product, summed) for the  + 2 C_k S_k  term at the end
// Scratch per wave: 4 x 32 x 33 floats.  Workgroup tables meet in LDS at the end as in gumbel.hip.
constexpr int kGmScratch = 4 * 32 * 33;

__host__ __device__ inline size_t gm_lds_floats_bwd(int K) { return gm_lds_floats(K) + (size_t)(kGmThreads / 64) * kGmScratch; }

template <int NT>
__global__ __launch_bounds__(kGmThreads) void gumbel_mfma_backward_kernel(const GumbelMfmaParams p) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    constexpr int K = NT * 32;
    float *imgA = sm;
    const float *csq = sm + (size_t)K * 32, *crow = csq + K;
    gm_stage(sm, p.cb, K);
    const float inv_t = 1.0f / p.temperature;

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int il = lane & 31, h = lane >> 5;
    constexpr int kWaves = kGmThreads / 64;
    const long long nw = (long long)gridDim.x * kWaves;
    const gm_f32x4 *img4 = reinterpret_cast<const gm_f32x4 *>(imgA);
    float *scr = sm + gm_lds_floats(K) + (size_t)wave * kGmScratch;
    float *wT = scr, *dT = scr + 32 * 33, *geS = scr + 2 * 32 * 33, *xS = scr + 3 * 32 * 33;

    gm_f32x16 gc[NT];   // codebook-gradient accumulators: [code 32 t + 8 (j>>2) + 4 h + (j&3)][feature il]
    float scol[NT];     // sum over rows of dd for code 32 t + il, rows of parity h
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        gc[t] = gm_f32x16{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        scol[t] = 0.0f;
    }

    for (long long tile = (long long)wave * gridDim.x + blockIdx.x; tile < p.n_tiles; tile += nw) {
        const long long row = tile * 32 + il;
        const bool ok = row < p.B;
        const long long rc = ok ? row : p.B - 1;
        float xp[16], xq[16];
        load_pair_row_vec<16>(p.x + (size_t)rc * kGmD, h, xp);
        gm_load_quad(p.x + (size_t)rc * kGmD, h, xq);
        const float xsq = pair_sumsq<16>(xp);

        // ---- forward replay: weights w[t][j] and emb ------------------------------------------------------
        float w[NT][16];
        float mx = -__builtin_inff();
        const float *urow = p.U + (size_t)rc * K + 4 * h;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            gm_f32x4 u4[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) u4[g] = *reinterpret_cast<const gm_f32x4 *>(urow + 32 * t + 8 * g);
            gm_f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const gm_f32x4 a = img4[(size_t)(q * 2 + h) * K + t * 32 + il];
#pragma unroll
                for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], xp[4 * q + i], acc, 0, 0, 0);
            }
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int code = 32 * t + 8 * (j >> 2) + 4 * h + (j & 3);
                const float dv = (xsq + csq[code]) - 2.0f * acc[j];
                const float u = u4[j >> 2][j & 3];
                const float gn = gm_gumbel(u);
                const float yy = ((-dv) + gn) * inv_t;
                w[t][j] = yy;
                mx = fmaxf(mx, yy);
            }
        }
        mx = fmaxf(mx, shfl_xor32(mx));
        float zs = 0.0f;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const float ev = gm_exp(w[t][j] - mx);
                w[t][j] = ev;
                zs = zs + ev;
            }
        const float Z = zs + shfl_xor32(zs);
        const float rz = 1.0f / Z;
        gm_f32x16 eacc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int code = 32 * t + 8 * (j >> 2) + 4 * h + (j & 3);
                const float wv = w[t][j] * rz;
                w[t][j] = wv;
                eacc = __builtin_amdgcn_mfma_f32_32x32x2f32(crow[(size_t)code * 33 + il], wv, eacc, 0, 0, 0);
            }

        // ---- ge (quad layout); rows past the end contribute nothing ------------------------------------------
        const float gl = (ok && p.g_loss) ? p.g_loss[rc] : 0.0f;
        float ge[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) ge[r] = 0.0f;
        if (p.g_emb) gm_load_quad(p.g_emb + (size_t)rc * kGmD, h, ge);
#pragma unroll
        for (int r = 0; r < 16; ++r) ge[r] = (ok ? ge[r] : 0.0f) + (2.0f * (eacc[r] - xq[r])) * gl;

        // row-indexed copies for the codebook-gradient product: [row il][feature], features of this lane's quads
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int d = 8 * (r >> 2) + 4 * h + (r & 3);
            geS[il * 33 + d] = ge[r];
            xS[il * 33 + d] = -2.0f * xq[r];
        }

        // ---- pass 1: sw = sum_k w_k dw_k -------------------------------------------------------------------------
        float swl = 0.0f;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            gm_f32x16 dacc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int r = 0; r < 16; ++r)  // k = feature 8 (r>>2) + 4 kk + (r&3), kk = lane half
                dacc = __builtin_amdgcn_mfma_f32_32x32x2f32(crow[(size_t)(32 * t + il) * 33 + 8 * (r >> 2) + 4 * h + (r & 3)],
                                                           ge[r], dacc, 0, 0, 0);
#pragma unroll
            for (int j = 0; j < 16; ++j) swl = __builtin_fmaf(w[t][j], dacc[j], swl);
        }
        const float sw = swl + shfl_xor32(swl);

        // B operands of the codebook-gradient products, shared by all code tiles: rows 2 s + kk, feature il
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        float bge[16], bx[16];
#pragma unroll
        for (int s2 = 0; s2 < 16; ++s2) {
            bge[s2] = geS[(2 * s2 + h) * 33 + il];
            bx[s2] = xS[(2 * s2 + h) * 33 + il];
        }

        // ---- pass 2: dd, g_x product, codebook-gradient products ----------------------------------------------
        gm_f32x16 gacc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        float sddl = 0.0f;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            gm_f32x16 dacc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int r = 0; r < 16; ++r)
                dacc = __builtin_amdgcn_mfma_f32_32x32x2f32(crow[(size_t)(32 * t + il) * 33 + 8 * (r >> 2) + 4 * h + (r & 3)],
                                                           ge[r], dacc, 0, 0, 0);
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");  // previous tile's reads of wT / dT are done
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int cit = 8 * (j >> 2) + 4 * h + (j & 3);
                const float dy = (w[t][j] * (dacc[j] - sw)) * inv_t;
                const float ddk = -dy;
                sddl = sddl + ddk;
                gacc = __builtin_amdgcn_mfma_f32_32x32x2f32(crow[(size_t)(32 * t + cit) * 33 + il], ddk, gacc, 0, 0, 0);
                wT[il * 33 + cit] = w[t][j];
                dT[il * 33 + cit] = ddk;
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            float sc = scol[t];
#pragma unroll
            for (int s2 = 0; s2 < 16; ++s2) {  // k = rows 2 s + kk; A: [code il][row], B: [row][feature il]
                const float aw = wT[(2 * s2 + h) * 33 + il];
                const float ad = dT[(2 * s2 + h) * 33 + il];
                gc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(aw, bge[s2], gc[t], 0, 0, 0);
                gc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(ad, bx[s2], gc[t], 0, 0, 0);
                sc = sc + ad;
            }
            scol[t] = sc;
        }
        const float sdd = sddl + shfl_xor32(sddl);
        if (ok && p.g_x) {
            float gx[16];
#pragma unroll
            for (int r = 0; r < 16; ++r)
                gx[r] = __builtin_fmaf(-2.0f, gacc[r], (2.0f * xq[r]) * sdd) + ((2.0f * p.beta) * (xq[r] - eacc[r])) * gl;
            gm_store_quad(p.g_x + (size_t)row * kGmD, h, gx);
        }
    }

    // ---- the wave's table: add 2 C_k S_k, then meet the other waves in LDS (imgA is free now) -------------------
    __syncthreads();
    float *tab = imgA;  // [K][32]
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const float stot = scol[t] + shfl_xor32(scol[t]);  // S for code 32 t + il
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        if (h == 0) scr[il] = stot;
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int cit = 8 * (j >> 2) + 4 * h + (j & 3);
            gc[t][j] = __builtin_fmaf(2.0f * scr[cit], crow[(size_t)(32 * t + cit) * 33 + il], gc[t][j]);
        }
    }
    for (int wv = 0; wv < kWaves; ++wv) {
        if (wave == wv) {
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    float *cell = tab + (size_t)(32 * t + 8 * (j >> 2) + 4 * h + (j & 3)) * 32 + il;
                    *cell = (wv == 0) ? gc[t][j] : *cell + gc[t][j];
                }
        }
        __syncthreads();
    }
    float *out = p.partial + (size_t)blockIdx.x * K * kGmD;
    for (int e = threadIdx.x; e < K * kGmD; e += kGmThreads) out[e] = tab[e];
}

static int gm_grid(long long n_tiles, int wg_per_cu) {
    long long want = (n_tiles + 3) / 4;
    const long long cap = (long long)cu_count() * wg_per_cu;
    if (want < 1) want = 1;
    return (int)(want < cap ? want : cap);
}

static long long g_gumbel_mfma_min_rows = 4096;
long long gumbel_mfma_min_rows() { return g_gumbel_mfma_min_rows; }
void gumbel_mfma_set_min_rows(long long n) { g_gumbel_mfma_min_rows = n < 1 ? 1 : n; }

bool gumbel_mfma_supported(int D, int K, const void *x, const void *U, const void *a, const void *b) {
    auto al16 = [](const void *q) { return (reinterpret_cast<uintptr_t>(q) & 15u) == 0; };
    return D == kGmD && (K == 32 || K == 64 || K == 128 || K == 256) && al16(x) && al16(U) && al16(a) && al16(b);
}

int gumbel_mfma_backward_grid(long long B) { return gm_grid((B + 31) / 32, 1); }

int gumbel_mfma_backward(const GumbelMfmaParams &p0, hipStream_t s) {
    GumbelMfmaParams p = p0;
    p.n_tiles = (p.B + 31) / 32;
    const size_t lds = gm_lds_floats_bwd(p.K) * sizeof(float);
    const int grid = gumbel_mfma_backward_grid(p.B);
    auto go = [&](auto kern) -> int {
        static LdsGrant attr;
        RQ_RETURN_IF_HIP(attr.ensure(reinterpret_cast<const void *>(kern), 160 * 1024));
        hipLaunchKernelGGL(kern, dim3(grid), dim3(kGmThreads), lds, s, p);
        RQ_CHECK_LAUNCH("gumbel_mfma_backward_kernel");
        return 0;
    };
    switch (p.K) {
        case 32: return go(gumbel_mfma_backward_kernel<1>);
        case 64: return go(gumbel_mfma_backward_kernel<2>);
        case 128: return go(gumbel_mfma_backward_kernel<4>);
        default: return go(gumbel_mfma_backward_kernel<8>);
    }
}

int gumbel_mfma_forward(const GumbelMfmaParams &p0, hipStream_t s) {
    GumbelMfmaParams p = p0;
    p.n_tiles = (p.B + 31) / 32;
    const size_t lds = gm_lds_floats(p.K) * sizeof(float);
    const int grid = gm_grid(p.n_tiles, 2);
    auto go = [&](auto kern) -> int {
        static LdsGrant attr;
        RQ_RETURN_IF_HIP(attr.ensure(reinterpret_cast<const void *>(kern), 160 * 1024));
        hipLaunchKernelGGL(kern, dim3(grid), dim3(kGmThreads), lds, s, p);
        RQ_CHECK_LAUNCH("gumbel_mfma_forward_kernel");
        return 0;
    };
    switch (p.K) {
        case 32: return go(gumbel_mfma_forward_kernel<1>);
        case 64: return go(gumbel_mfma_forward_kernel<2>);
        case 128: return go(gumbel_mfma_forward_kernel<4>);
        default: return go(gumbel_mfma_forward_kernel<8>);
    }
}

}  // namespace rqhip
