// recon_loss.hip -- fused reconstruction loss of the RQ-VAE (reference modules/loss.py:5-10, used at
// modules/rqvae.py:152): out[b] = sum_d (x_hat[b,d] - x[b,d])^2 and its backward, one pass each.
//
// HBM-bound: forward reads 8N bytes per row and writes 4; backward reads 8N + 4 and writes 4N.  PyTorch runs
// this as sub / pow / sum (and three more elementwise kernels backwards), each a full round trip of the
// [B, 768] tensor; at 100 000 x 768 that was ~0.7 ms of a 6.8 ms training step.
// One wave per row, 16-byte loads; the row sum has a fixed order (== oracle/rq_oracle.c:rqo_recon_loss):
// lane l adds its elements in ascending address order, then a 6-round xor butterfly.
#include "rqhip_common.h"

namespace rqhip {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float butterfly_sum(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = v + __shfl_xor(v, m, 64);
    return v;
}

__global__ __launch_bounds__(256) void recon_fwd_kernel(const float *__restrict__ xh, long long ldh,
                                                        const float *__restrict__ x, long long ldx, long long B, int N,
                                                        float *__restrict__ out) {
    const int lane = threadIdx.x & 63;
    const long long waves = (long long)gridDim.x * 4, gw = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const bool vec = (N & 3) == 0 && (ldh & 3) == 0 && (ldx & 3) == 0 &&
                     ((reinterpret_cast<uintptr_t>(xh) | reinterpret_cast<uintptr_t>(x)) & 15) == 0;
    for (long long row = gw; row < B; row += waves) {
        const float *a = xh + row * ldh, *b = x + row * ldx;
        float s = 0.0f;
        if (vec) {
            for (int i = lane; i < N / 4; i += 64) {
                const f32x4 u = *reinterpret_cast<const f32x4 *>(a + 4 * i);
                const f32x4 v = *reinterpret_cast<const f32x4 *>(b + 4 * i);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float d = u[j] - v[j];
                    s = s + d * d;
                }
            }
        } else {
            for (int i = lane; i < N; i += 64) {
                const float d = a[i] - b[i];
                s = s + d * d;
            }
        }
        s = butterfly_sum(s);
        if (lane == 0) out[row] = s;
    }
}

__global__ __launch_bounds__(256) void recon_bwd_kernel(const float *__restrict__ xh, long long ldh,
                                                        const float *__restrict__ x, long long ldx,
                                                        const float *__restrict__ g, long long B, int N,
                                                        float *__restrict__ gh, float *__restrict__ gx) {
    const int lane = threadIdx.x & 63;
    const long long waves = (long long)gridDim.x * 4, gw = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const bool vec = (N & 3) == 0 && (ldh & 3) == 0 && (ldx & 3) == 0 &&
                     ((reinterpret_cast<uintptr_t>(xh) | reinterpret_cast<uintptr_t>(x) |
                       reinterpret_cast<uintptr_t>(gh) | reinterpret_cast<uintptr_t>(gx)) & 15) == 0;
    for (long long row = gw; row < B; row += waves) {
        const float *a = xh + row * ldh, *b = x + row * ldx;
        const float gr = g[row];
        if (vec) {
            for (int i = lane; i < N / 4; i += 64) {
                const f32x4 u = *reinterpret_cast<const f32x4 *>(a + 4 * i);
                const f32x4 v = *reinterpret_cast<const f32x4 *>(b + 4 * i);
                f32x4 o;
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = (2.0f * (u[j] - v[j])) * gr;
                if (gh) *reinterpret_cast<f32x4 *>(gh + row * (long long)N + 4 * i) = o;
                if (gx) *reinterpret_cast<f32x4 *>(gx + row * (long long)N + 4 * i) = -o;
            }
        } else {
            for (int i = lane; i < N; i += 64) {
                const float o = (2.0f * (a[i] - b[i])) * gr;
                if (gh) gh[row * (long long)N + i] = o;
                if (gx) gx[row * (long long)N + i] = -o;
            }
        }
    }
}

// Forward that also writes the gradient it EXPECTS to be asked for: g_spec[b,:] = (2 (x_hat - x)) * row_scale, the
// value recon_bwd_kernel produces when the upstream gradient of row b equals row_scale -- which is what
// `(reconstruction + quantize_loss).mean().backward()` (rqvae.py:152-154) sends: 1/B for every row.  One pass
// (read x_hat, x; write g_spec) replaces the forward pass plus most of the backward pass.
__global__ __launch_bounds__(256) void recon_fwd_spec_kernel(const float *__restrict__ xh, long long ldh,
                                                             const float *__restrict__ x, long long ldx, long long B,
                                                             int N, float row_scale, float *__restrict__ out,
                                                             float *__restrict__ gs) {
    const int lane = threadIdx.x & 63;
    const long long waves = (long long)gridDim.x * 4, gw = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    for (long long row = gw; row < B; row += waves) {   // (caller guarantees the float4 layout)
        const float *a = xh + row * ldh, *b = x + row * ldx;
        float s = 0.0f;
        for (int i = lane; i < N / 4; i += 64) {
            const f32x4 u = *reinterpret_cast<const f32x4 *>(a + 4 * i);
            const f32x4 v = *reinterpret_cast<const f32x4 *>(b + 4 * i);
            f32x4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float d = u[j] - v[j];
                s = s + d * d;
                o[j] = (2.0f * d) * row_scale;
            }
            *reinterpret_cast<f32x4 *>(gs + row * (long long)N + 4 * i) = o;
        }
        s = butterfly_sum(s);
        if (lane == 0) out[row] = s;
    }
}

// Backward of the above: rows whose upstream gradient IS row_scale (bit for bit) are already right and are not
// touched; any other row is recomputed exactly as recon_bwd_kernel would.  In a training step this reads B floats.
__global__ __launch_bounds__(256) void recon_bwd_spec_kernel(const float *__restrict__ xh, long long ldh,
                                                             const float *__restrict__ x, long long ldx,
                                                             const float *__restrict__ g, long long B, int N,
                                                             float row_scale, float *__restrict__ gs) {
    const int lane = threadIdx.x & 63;
    const long long waves = (long long)gridDim.x * 4, gw = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    for (long long row = gw; row < B; row += waves) {
        const float gr = g[row];
        if (__float_as_uint(gr) == __float_as_uint(row_scale)) continue;
        const float *a = xh + row * ldh, *b = x + row * ldx;
        for (int i = lane; i < N / 4; i += 64) {
            const f32x4 u = *reinterpret_cast<const f32x4 *>(a + 4 * i);
            const f32x4 v = *reinterpret_cast<const f32x4 *>(b + 4 * i);
            f32x4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = (2.0f * (u[j] - v[j])) * gr;
            *reinterpret_cast<f32x4 *>(gs + row * (long long)N + 4 * i) = o;
        }
    }
}

// Backward of the reconstruction loss FUSED into the last decoder GEMM (csrc/gemm_split.hip, EPI 2): x_hat was never
// stored, so a row whose upstream gradient is not row_scale is rescaled, g_spec * (g / row_scale), instead of recomputed
// (one more rounding than (2 d) g; rows that match -- every row of a training step -- are not touched).
// row_max [parts][B] / col_max [N] (optional): the maxima the fused epilogue emitted for g_spec (the scales of the fp16 split
// kernels that read it next) are brought up to date for the rows that change: the row's new maximum goes to part 0 (the other
// parts are cleared), the columns are maxed into atomically (a stale, larger column maximum -- a row that was scaled down -- is a valid
// scale that costs the fp16 low piece one bit per binade of overestimate; the result then differs in the last bits from a fresh maxima pass).
__global__ __launch_bounds__(256) void recon_rescale_rows_kernel(const float *__restrict__ g, long long B, int N,
                                                                 float row_scale, float *__restrict__ gs,
                                                                 unsigned *__restrict__ row_max, int parts,
                                                                 unsigned *__restrict__ col_max) {
    const int lane = threadIdx.x & 63;
    const long long waves = (long long)gridDim.x * 4, gw = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    for (long long row = gw; row < B; row += waves) {
        const float gr = g[row];
        if (__float_as_uint(gr) == __float_as_uint(row_scale)) continue;
        const float f = gr / row_scale;
        unsigned rm = 0u;
        for (int i = lane; i < N / 4; i += 64) {
            f32x4 o = *reinterpret_cast<const f32x4 *>(gs + row * (long long)N + 4 * i);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                o[j] = o[j] * f;
                const unsigned b = __float_as_uint(o[j]) & 0x7fffffffu;
                rm = rm > b ? rm : b;
                if (col_max && b) atomicMax(col_max + 4 * i + j, b);
            }
            *reinterpret_cast<f32x4 *>(gs + row * (long long)N + 4 * i) = o;
        }
        if (row_max) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const unsigned other = (unsigned)__shfl_xor((int)rm, o, 64);
                rm = rm > other ? rm : other;
            }
            if (lane < parts) row_max[(size_t)lane * B + row] = lane == 0 ? rm : 0u;
        }
    }
}

// The three batch means RqVae.forward returns (modules/rqvae.py:154,171-172): mean(recon + quant), mean(recon),
// mean(quant) -- one launch instead of an elementwise add and three two-stage reductions (3 x 16.6 us at 100 000 rows).
// One workgroup; thread t adds the float4 groups t, t + 1024, ... in order, then a fixed LDS tree: deterministic.
__global__ __launch_bounds__(1024) void loss_means_kernel(const float *__restrict__ recon, const float *__restrict__ quant,
                                                          long long B, float *__restrict__ out) {
    __shared__ float red[3][1024];
    const int t = threadIdx.x;
    float s_sum = 0.0f, s_r = 0.0f, s_q = 0.0f;
    const bool vec = ((reinterpret_cast<uintptr_t>(recon) | reinterpret_cast<uintptr_t>(quant)) & 15) == 0;
    const long long n4 = vec ? B / 4 : 0;
    // (eight groups fetched per trip, added in the same order as with one per trip.  It did not change the 16 us this kernel takes
    // at 100 000 rows: ONE workgroup pulls 0.8 MB through one CU's L1 port; a two-stage form would take ~6 us -- 0.4 % of the step)
    for (long long i0 = t; i0 < n4; i0 += 8 * 1024) {
        f32x4 a[8], b[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const long long i = i0 + (long long)u * 1024;
            a[u] = i < n4 ? reinterpret_cast<const f32x4 *>(recon)[i] : f32x4{0.f, 0.f, 0.f, 0.f};
            b[u] = i < n4 ? reinterpret_cast<const f32x4 *>(quant)[i] : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (i0 + (long long)u * 1024 < n4) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    s_sum = s_sum + (a[u][j] + b[u][j]);
                    s_r = s_r + a[u][j];
                    s_q = s_q + b[u][j];
                }
            }
        }
    }
    for (long long i = 4 * n4 + t; i < B; i += 1024) {
        const float a = recon[i], b = quant[i];
        s_sum = s_sum + (a + b);
        s_r = s_r + a;
        s_q = s_q + b;
    }
    red[0][t] = s_sum; red[1][t] = s_r; red[2][t] = s_q;
    __syncthreads();
    for (int s = 512; s > 0; s >>= 1) {
        if (t < s) {
#pragma unroll
            for (int c = 0; c < 3; ++c) red[c][t] = red[c][t] + red[c][t + s];
        }
        __syncthreads();
    }
    if (t < 3) out[t] = red[t][0] / (float)B;
}

// The same three means by many workgroups (round 6): block b sums rows [b chunk, (b + 1) chunk) -- thread t its float4 groups t, t + 256, ...
// in order, then a fixed LDS tree -- into partial[b][0..2]; the LAST block to arrive (a counter behind the partials, re-armed for the
// next launch) adds the partials in block order and divides.  The grid depends on B only: deterministic; another summation order than
// the one-workgroup kernel's (both are within fp32 rounding of the exact means, tests/test_gpu_parity.py).
constexpr int kLmChunkRows = 4096;      // rows per block
constexpr int kLmMaxBlocks = 1024;
__global__ __launch_bounds__(256) void loss_means_blocks_kernel(const float *__restrict__ recon, const float *__restrict__ quant, long long B,
                                                               float *__restrict__ partial, unsigned *__restrict__ counter,
                                                               float *__restrict__ out) {
    __shared__ float red[3][256];
    __shared__ bool s_last;
    const int t = threadIdx.x, nb = gridDim.x;
    const long long per = ((B + nb - 1) / nb + 3) / 4 * 4;            // rows per block, a multiple of 4
    const long long lo = (long long)blockIdx.x * per, hi = lo + per < B ? lo + per : B;
    float s_sum = 0.0f, s_r = 0.0f, s_q = 0.0f;
    const bool vec = ((reinterpret_cast<uintptr_t>(recon) | reinterpret_cast<uintptr_t>(quant)) & 15) == 0;
    const long long n4 = (vec && hi > lo) ? (hi - lo) / 4 : 0;
    for (long long i = t; i < n4; i += 256) {
        const f32x4 a = reinterpret_cast<const f32x4 *>(recon + lo)[i], b = reinterpret_cast<const f32x4 *>(quant + lo)[i];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            s_sum = s_sum + (a[j] + b[j]);
            s_r = s_r + a[j];
            s_q = s_q + b[j];
        }
    }
    for (long long i = lo + 4 * n4 + t; i < hi; i += 256) {
        const float a = recon[i], b = quant[i];
        s_sum = s_sum + (a + b);
        s_r = s_r + a;
        s_q = s_q + b;
    }
    red[0][t] = s_sum; red[1][t] = s_r; red[2][t] = s_q;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (t < s) {
#pragma unroll
            for (int c = 0; c < 3; ++c) red[c][t] = red[c][t] + red[c][t + s];
        }
        __syncthreads();
    }
    // one device-scope release per WORKGROUP (thread 0: its three stores, the fence, the ticket), one acquire in the last block: a
    // __threadfence() executed by every wave of every block is a cache write-back + invalidate per wave
    if (t == 0) {
#pragma unroll
        for (int c = 0; c < 3; ++c) partial[(size_t)blockIdx.x * 3 + c] = red[c][0];
        __threadfence();
        s_last = atomicAdd(counter, 1u) == (unsigned)nb - 1;
        if (s_last) __threadfence();
    }
    __syncthreads();
    if (!s_last) return;
    if (t < 3) {
        float acc = 0.0f;
        for (int b = 0; b < nb; ++b) acc = acc + __builtin_nontemporal_load(partial + (size_t)b * 3 + t);
        out[t] = acc / (float)B;
    }
    if (t == 0) *counter = 0u;        // re-armed: the workspace stays usable launch after launch without a fill
}

// Backward of the three means: every row of `recon` receives (g_loss + g_recon_mean) * (1/B), every row of `quant`
// (g_loss + g_quant_mean) * (1/B) -- the arithmetic of PyTorch's own mean backward on the device, which multiplies by
// the fp32 reciprocal of a scalar divisor -- written as the two dense [B] vectors the next kernels read, in one launch
// instead of add / scale / expand-copy per vector.
__global__ __launch_bounds__(256) void loss_means_bwd_kernel(const float *__restrict__ g_loss, const float *__restrict__ g_recon,
                                                             const float *__restrict__ g_quant, long long B,
                                                             float *__restrict__ rows_recon, float *__restrict__ rows_quant) {
    const float inv = 1.0f / (float)B;
    const float gl = g_loss ? *g_loss : 0.0f;
    // one term: that term; two terms: their fp32 sum (g_loss first, as the autograd engine accumulates them)
    const float sr = (g_loss && g_recon) ? gl + *g_recon : (g_recon ? *g_recon : gl);
    const float sq = (g_loss && g_quant) ? gl + *g_quant : (g_quant ? *g_quant : gl);
    const float vr = sr * inv, vq = sq * inv;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < B; i += (long long)gridDim.x * blockDim.x) {
        if (rows_recon) rows_recon[i] = vr;
        if (rows_quant) rows_quant[i] = vq;
    }
}

static int row_grid(long long B) {
    long long want = (B + 3) / 4, cap = (long long)cu_count() * 8;
    if (want < 1) want = 1;
    return (int)(want < cap ? want : cap);
}

}  // namespace rqhip

using namespace rqhip;

extern "C" int rqhip_recon_loss_forward(const float *x_hat, int64_t ld_hat, const float *x, int64_t ld_x, int64_t B,
                                        int N, float *out, rqhip_stream_t stream) {
    if (B < 0 || N < 1 || ld_hat < N || ld_x < N || (B > 0 && (!x_hat || !x || !out))) {
        set_error("recon_loss_forward: bad arguments (B=%lld N=%d ld=%lld,%lld)", (long long)B, N, (long long)ld_hat,
                  (long long)ld_x);
        return RQHIP_EARG;
    }
    if (B == 0) return RQHIP_OK;
    hipLaunchKernelGGL(recon_fwd_kernel, dim3(row_grid(B)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x_hat,
                       (long long)ld_hat, x, (long long)ld_x, (long long)B, N, out);
    RQ_CHECK_LAUNCH("recon_fwd_kernel");
    return RQHIP_OK;
}

extern "C" int rqhip_recon_loss_backward(const float *x_hat, int64_t ld_hat, const float *x, int64_t ld_x,
                                         const float *g_out, int64_t B, int N, float *g_x_hat, float *g_x,
                                         rqhip_stream_t stream) {
    if (B < 0 || N < 1 || ld_hat < N || ld_x < N || (B > 0 && (!x_hat || !x || !g_out || (!g_x_hat && !g_x)))) {
        set_error("recon_loss_backward: bad arguments");
        return RQHIP_EARG;
    }
    if (B == 0) return RQHIP_OK;
    hipLaunchKernelGGL(recon_bwd_kernel, dim3(row_grid(B)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x_hat,
                       (long long)ld_hat, x, (long long)ld_x, g_out, (long long)B, N, g_x_hat, g_x);
    RQ_CHECK_LAUNCH("recon_bwd_kernel");
    return RQHIP_OK;
}

static bool recon_vec_ok(const void *a, const void *b, const void *c, int64_t ld_hat, int64_t ld_x, int N) {
    return (N & 3) == 0 && (ld_hat & 3) == 0 && (ld_x & 3) == 0 &&
           ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(c)) & 15) == 0;
}

extern "C" int rqhip_recon_loss_forward_spec(const float *x_hat, int64_t ld_hat, const float *x, int64_t ld_x,
                                             int64_t B, int N, float row_scale, float *out, float *g_spec,
                                             rqhip_stream_t stream) {
    if (B < 0 || N < 1 || ld_hat < N || ld_x < N || (B > 0 && (!x_hat || !x || !out || !g_spec))) {
        set_error("recon_loss_forward_spec: bad arguments");
        return RQHIP_EARG;
    }
    if (!recon_vec_ok(x_hat, x, g_spec, ld_hat, ld_x, N)) {
        set_error("recon_loss_forward_spec: needs N, strides multiples of 4 and 16-byte aligned pointers");
        return RQHIP_EUNSUPPORTED;
    }
    if (B == 0) return RQHIP_OK;
    hipLaunchKernelGGL(recon_fwd_spec_kernel, dim3(row_grid(B)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       x_hat, (long long)ld_hat, x, (long long)ld_x, (long long)B, N, row_scale, out, g_spec);
    RQ_CHECK_LAUNCH("recon_fwd_spec_kernel");
    return RQHIP_OK;
}

extern "C" int rqhip_recon_loss_backward_spec(const float *x_hat, int64_t ld_hat, const float *x, int64_t ld_x,
                                              const float *g_out, int64_t B, int N, float row_scale, float *g_spec,
                                              rqhip_stream_t stream) {
    if (B < 0 || N < 1 || ld_hat < N || ld_x < N || (B > 0 && (!x_hat || !x || !g_out || !g_spec))) {
        set_error("recon_loss_backward_spec: bad arguments");
        return RQHIP_EARG;
    }
    if (!recon_vec_ok(x_hat, x, g_spec, ld_hat, ld_x, N)) {
        set_error("recon_loss_backward_spec: needs N, strides multiples of 4 and 16-byte aligned pointers");
        return RQHIP_EUNSUPPORTED;
    }
    if (B == 0) return RQHIP_OK;
    hipLaunchKernelGGL(recon_bwd_spec_kernel, dim3(row_grid(B)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       x_hat, (long long)ld_hat, x, (long long)ld_x, g_out, (long long)B, N, row_scale, g_spec);
    RQ_CHECK_LAUNCH("recon_bwd_spec_kernel");
    return RQHIP_OK;
}

extern "C" int rqhip_recon_rescale_rows_ex(const float *g_out, int64_t B, int N, float row_scale, float *g_spec,
                                           unsigned *row_max, int row_parts, unsigned *col_max, rqhip_stream_t stream) {
    if (B < 0 || N < 4 || (N % 4) != 0 || (B > 0 && (!g_out || !g_spec)) || (reinterpret_cast<uintptr_t>(g_spec) & 15u) != 0 ||
        !(row_scale != 0.0f) || (row_max && (row_parts < 1 || row_parts > 64))) {
        set_error("recon_rescale_rows: bad arguments (N a multiple of 4, 16-byte aligned g_spec, non-zero row_scale, 1 .. 64 row-maxima parts)");
        return RQHIP_EARG;
    }
    if (B == 0) return RQHIP_OK;
    hipLaunchKernelGGL(recon_rescale_rows_kernel, dim3(row_grid(B)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       g_out, (long long)B, N, row_scale, g_spec, row_max, row_parts, col_max);
    RQ_CHECK_LAUNCH("recon_rescale_rows_kernel");
    return RQHIP_OK;
}

extern "C" int rqhip_recon_rescale_rows(const float *g_out, int64_t B, int N, float row_scale, float *g_spec,
                                        rqhip_stream_t stream) {
    return rqhip_recon_rescale_rows_ex(g_out, B, N, row_scale, g_spec, nullptr, 0, nullptr, stream);
}

extern "C" int rqhip_loss_means_backward(const float *g_loss, const float *g_recon_mean, const float *g_quant_mean,
                                         int64_t B, float *rows_recon, float *rows_quant, rqhip_stream_t stream) {
    if (B <= 0 || (!rows_recon && !rows_quant)) {
        set_error("loss_means_backward: bad arguments");
        return RQHIP_EARG;
    }
    long long g = (B + 255) / 256;
    if (g > 1024) g = 1024;
    hipLaunchKernelGGL(loss_means_bwd_kernel, dim3((int)g), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), g_loss,
                       g_recon_mean, g_quant_mean, (long long)B, rows_recon, rows_quant);
    RQ_CHECK_LAUNCH("loss_means_bwd_kernel");
    return RQHIP_OK;
}

extern "C" size_t rqhip_loss_means_workspace_bytes(void) { return (size_t)kLmMaxBlocks * 3 * sizeof(float) + 16; }

extern "C" int rqhip_loss_means_ws(const float *recon, const float *quant, int64_t B, float *out3, void *workspace, size_t workspace_bytes,
                                   rqhip_stream_t stream) {
    if (B <= 0 || !recon || !quant || !out3) {
        set_error("loss_means: bad arguments");
        return RQHIP_EARG;
    }
    if (!workspace || workspace_bytes < rqhip_loss_means_workspace_bytes() || (reinterpret_cast<uintptr_t>(workspace) & 15u)) {
        set_error("loss_means_ws: workspace of rqhip_loss_means_workspace_bytes() bytes, 16-byte aligned, ZEROED once by the caller");
        return RQHIP_EWORKSPACE;
    }
    long long nb = (B + kLmChunkRows - 1) / kLmChunkRows;
    if (nb > kLmMaxBlocks) nb = kLmMaxBlocks;
    float *partial = reinterpret_cast<float *>(workspace);
    unsigned *counter = reinterpret_cast<unsigned *>(partial + (size_t)kLmMaxBlocks * 3);
    hipLaunchKernelGGL(loss_means_blocks_kernel, dim3((unsigned)nb), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), recon, quant,
                       (long long)B, partial, counter, out3);
    RQ_CHECK_LAUNCH("loss_means_blocks_kernel");
    return RQHIP_OK;
}

extern "C" int rqhip_loss_means(const float *recon, const float *quant, int64_t B, float *out3, rqhip_stream_t stream) {
    if (B <= 0 || !recon || !quant || !out3) {
        set_error("loss_means: bad arguments");
        return RQHIP_EARG;
    }
    hipLaunchKernelGGL(loss_means_kernel, dim3(1), dim3(1024), 0, reinterpret_cast<hipStream_t>(stream), recon, quant,
                       (long long)B, out3);
    RQ_CHECK_LAUNCH("loss_means_kernel");
    return RQHIP_OK;
}
