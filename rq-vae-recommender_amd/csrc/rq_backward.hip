// rq_backward.hip -- backward of the fused residual-quantisation stack (gfx950).
//
// Closed form of what torch.autograd computes through the reference's level loop (modules/rqvae.py:
// 125-132), the STE / rotation-trick / eval branches of Quantize.forward (modules/quantize.py:137-161),
// the embedding lookup (:101-102) and QuantizeLoss (modules/loss.py:38-41); the recursion is written out
// in oracle/rq_oracle.c:rqo_rq_backward.  HBM-bound: per row it reads res0, the L ids and the upstream
// gradients once and writes g_res0 once; the codeword rows come from L2.
//
// One wave owns 32 rows in the pair layout of rq_rowmath.h.  Pass 1 replays the residual chain
// (bit-identical to the forward) and parks res_l, l >= 1, in the caller's workspace; pass 2 walks the
// levels backwards carrying G = dL/d res_l in registers and scatters each row's codeword gradient with
// global_atomic_add_f32 (order of accumulation across rows is therefore not fixed; g_res0 is exact).
#include "rq_rowmath.h"

namespace rqhip {

struct RqBwdParams {
    const float *res0, *cb;
    const int64_t *ids;
    const float *g_embs, *g_embsum, *g_resid, *g_loss;
    float *g_res0, *g_cb;
    float *ws;  // [(L-1), B, D] replayed residuals
    long long B, n_tiles;
    int D, L, K;
    float beta;
};

template <int KSTEPS, int MODE>
__global__ __launch_bounds__(256) void rq_backward_kernel(const RqBwdParams p) {
    const int lane = threadIdx.x & 63;
    const int il = lane & 31, h = lane >> 5;
    const int D = p.D, L = p.L, K = p.K;
    const long long waves = (long long)gridDim.x * (blockDim.x >> 6);
    const long long gw = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);

    for (long long tile = gw; tile < p.n_tiles; tile += waves) {
        const long long row = tile * 32 + il;
        const bool ok = row < p.B;
        const long long rc = ok ? row : p.B - 1;

        float r[KSTEPS], e[KSTEPS], o[KSTEPS];
        load_pair_row<KSTEPS>(p.res0 + (size_t)rc * D, D, h, r);
        // pass 1: replay res_1 .. res_{L-1}
        for (int l = 0; l + 1 < L; ++l) {
            const long long id = p.ids[(size_t)l * p.B + rc];
            load_pair_row<KSTEPS>(p.cb + ((size_t)l * K + id) * D, D, h, e);
            const float xsq = (MODE == RQHIP_MODE_ROTATION) ? pair_sumsq<KSTEPS>(r) : 0.0f;
            level_output<KSTEPS, MODE>(r, e, xsq, o);
            float *dst = p.ws + ((size_t)l * p.B + rc) * D;
#pragma unroll
            for (int kk = 0; kk < KSTEPS; ++kk) {
                r[kk] = r[kk] - o[kk];
                const int d = 2 * kk + h;
                if (ok && d < D) dst[d] = r[kk];
            }
        }
        // r now holds res_{L-1}
        const float gl = p.g_loss ? p.g_loss[rc] : 0.0f;
        float G[KSTEPS];
#pragma unroll
        for (int kk = 0; kk < KSTEPS; ++kk) G[kk] = 0.0f;

        for (int l = L - 1; l >= 0; --l) {
            if (l != L - 1) {
                const float *src = (l == 0) ? p.res0 + (size_t)rc * D : p.ws + ((size_t)(l - 1) * p.B + rc) * D;
                load_pair_row<KSTEPS>(src, D, h, r);
            }
            const long long id = p.ids[(size_t)l * p.B + rc];
            load_pair_row<KSTEPS>(p.cb + ((size_t)l * K + id) * D, D, h, e);
            const size_t lrow = ((size_t)l * p.B + rc) * D;
            float A[KSTEPS], gr[KSTEPS];
#pragma unroll
            for (int kk = 0; kk < KSTEPS; ++kk) {
                const int d = 2 * kk + h;
                float a = 0.0f;
                if (p.g_embs && d < D) a = p.g_embs[lrow + d];
                if (p.g_embsum) a = a + ((d < D) ? p.g_embsum[(size_t)rc * D + d] : 0.0f);
                A[kk] = a - G[kk];
                gr[kk] = (p.g_resid && d < D) ? p.g_resid[lrow + d] : 0.0f;
            }
            float *dE = p.g_cb ? p.g_cb + ((size_t)l * K + id) * D : nullptr;
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
                    const int d = 2 * kk + h;
                    if (dE && ok && d < D) atomicAdd(dE + d, embg);
                }
            } else {
#pragma unroll
                for (int kk = 0; kk < KSTEPS; ++kk) {
                    const float commit = (2.0f * p.beta) * (r[kk] - e[kk]) * gl;
                    const float embg = (2.0f * (e[kk] - r[kk])) * gl;
                    const int d = 2 * kk + h;
                    if (MODE == RQHIP_MODE_EVAL) {
                        const float contrib = A[kk] + embg;
                        G[kk] = (gr[kk] + G[kk]) + commit;
                        if (dE && ok && d < D) atomicAdd(dE + d, contrib);
                    } else {
                        G[kk] = ((gr[kk] + G[kk]) + A[kk]) + commit;
                        if (dE && ok && d < D) atomicAdd(dE + d, embg);
                    }
                }
            }
        }
        if (ok && p.g_res0) {
#pragma unroll
            for (int kk = 0; kk < KSTEPS; ++kk) {
                const int d = 2 * kk + h;
                if (d < D) p.g_res0[(size_t)row * D + d] = G[kk];
            }
        }
    }
}

template <int KSTEPS>
static int launch_bwd(const RqBwdParams &p, int mode, int grid, hipStream_t s) {
    switch (mode) {
        case RQHIP_MODE_EVAL:
            hipLaunchKernelGGL((rq_backward_kernel<KSTEPS, RQHIP_MODE_EVAL>), dim3(grid), dim3(256), 0, s, p);
            break;
        case RQHIP_MODE_STE:
            hipLaunchKernelGGL((rq_backward_kernel<KSTEPS, RQHIP_MODE_STE>), dim3(grid), dim3(256), 0, s, p);
            break;
        case RQHIP_MODE_ROTATION:
            hipLaunchKernelGGL((rq_backward_kernel<KSTEPS, RQHIP_MODE_ROTATION>), dim3(grid), dim3(256), 0, s, p);
            break;
        default:
            set_error("rq_backward: unsupported mode %d", mode);
            return RQHIP_EARG;
    }
    RQ_CHECK_LAUNCH("rq_backward_kernel");
    return 0;
}

}  // namespace rqhip

using namespace rqhip;

extern "C" size_t rqhip_rq_backward_workspace_bytes(int64_t B, int D, int L) {
    if (B <= 0 || D <= 0 || L <= 1) return 16;
    return (size_t)(L - 1) * (size_t)B * (size_t)D * sizeof(float);
}

extern "C" int rqhip_rq_backward(const float *res0, int64_t B, int D, const float *codebooks, int L, int K,
                                 int mode, float beta, const int64_t *ids, const float *g_embs,
                                 const float *g_embsum, const float *g_resid, const float *g_loss,
                                 float *g_res0, float *g_codebooks, void *workspace, size_t workspace_bytes,
                                 rqhip_stream_t stream) {
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
    if (L > 1 && B > 0 && (!workspace || workspace_bytes < rqhip_rq_backward_workspace_bytes(B, D, L))) {
        set_error("rq_backward: workspace too small");
        return RQHIP_EWORKSPACE;
    }
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (g_codebooks) RQ_RETURN_IF_HIP(hipMemsetAsync(g_codebooks, 0, sizeof(float) * (size_t)L * K * D, s));
    if (B == 0) return RQHIP_OK;
    RqBwdParams p;
    p.res0 = res0; p.cb = codebooks; p.ids = ids; p.g_embs = g_embs; p.g_embsum = g_embsum;
    p.g_resid = g_resid; p.g_loss = g_loss; p.g_res0 = g_res0; p.g_cb = g_codebooks;
    p.ws = reinterpret_cast<float *>(workspace);
    p.B = B; p.n_tiles = (B + 31) / 32; p.D = D; p.L = L; p.K = K; p.beta = beta;
    long long want = (p.n_tiles + 3) / 4;
    long long cap = (long long)cu_count() * 8;
    const int grid = (int)(want < cap ? want : cap);
    switch (ksteps_for(D)) {
        case 4: return launch_bwd<4>(p, mode, grid, s);
        case 8: return launch_bwd<8>(p, mode, grid, s);
        case 16: return launch_bwd<16>(p, mode, grid, s);
        case 32: return launch_bwd<32>(p, mode, grid, s);
        default: return launch_bwd<64>(p, mode, grid, s);
    }
}
