// rq_rowmath.h -- per-row arithmetic shared by the forward and backward kernels.
//
// Lane layout ("pair layout"): a wave owns 32 rows; lane (il = lane&31, h = lane>>5) holds the features of
// row il with parity h: v[kk] = row[2*kk + h], kk < KSTEPS (zero beyond D).  A reduction over d is then
// one sequential chain per lane (== one of the oracle's two parity accumulators) plus ONE exchange with
// lane^32; a0 + a1 is commutative so both lanes obtain the oracle's value.
#pragma once
#include "rqhip_common.h"

namespace rqhip {

__device__ __forceinline__ float pair_sum(float a) { return a + shfl_xor32(a); }

// sumsq2: separately rounded square and add (reference: (v**2).sum())
template <int KSTEPS>
__device__ __forceinline__ float pair_sumsq(const float (&v)[KSTEPS]) {
    float a = 0.0f;
#pragma unroll
    for (int kk = 0; kk < KSTEPS; ++kk) a = a + v[kk] * v[kk];
    return pair_sum(a);
}

// dotp2: FMA chain per parity
template <int KSTEPS>
__device__ __forceinline__ float pair_dot(const float (&x)[KSTEPS], const float (&y)[KSTEPS]) {
    float a = 0.0f;
#pragma unroll
    for (int kk = 0; kk < KSTEPS; ++kk) a = __builtin_fmaf(x[kk], y[kk], a);
    return pair_sum(a);
}

// Rotation trick (reference modules/quantize.py:34-50,140-153) for one row in pair layout.
//   r = level input (carries the gradient), e = selected codeword, xsq = sumsq2(r)
//   o = emb_out; w, u (= r/(|r|+1e-8)), q (= e/(|e|+1e-8)) and scale are what backward needs.
template <int KSTEPS>
__device__ __forceinline__ void rotation_lane(const float (&r)[KSTEPS], const float (&e)[KSTEPS], float xsq,
                                              float (&o)[KSTEPS], float (&w)[KSTEPS], float (&u)[KSTEPS],
                                              float (&q)[KSTEPS], float &scale) {
    const float nx = __builtin_sqrtf(xsq);
    const float ne = __builtin_sqrtf(pair_sumsq<KSTEPS>(e));
    const float du = nx + 1e-8f, dq = ne + 1e-8f;
#pragma unroll
    for (int kk = 0; kk < KSTEPS; ++kk) {
        u[kk] = r[kk] / du;
        q[kk] = e[kk] / dq;
        w[kk] = u[kk] + q[kk];
    }
    const float nw = __builtin_sqrtf(pair_sumsq<KSTEPS>(w));
    const float den = nw > 1e-6f ? nw : 1e-6f;  // F.normalize(eps=1e-6)
#pragma unroll
    for (int kk = 0; kk < KSTEPS; ++kk) w[kk] = w[kk] / den;
    const float ew = pair_dot<KSTEPS>(r, w);
    const float eu = pair_dot<KSTEPS>(r, u);
    scale = ne / (nx + 1e-6f);
#pragma unroll
    for (int kk = 0; kk < KSTEPS; ++kk) {
        const float t1 = ew * w[kk];
        const float t2 = eu * q[kk];
        o[kk] = ((r[kk] - 2.0f * t1) + 2.0f * t2) * scale;
    }
}

// emb_out of one level (quantize.py:139 / :142-153 / :160)
template <int KSTEPS, int MODE>
__device__ __forceinline__ void level_output(const float (&r)[KSTEPS], const float (&e)[KSTEPS], float xsq,
                                             float (&o)[KSTEPS]) {
    if (MODE == RQHIP_MODE_EVAL) {
#pragma unroll
        for (int kk = 0; kk < KSTEPS; ++kk) o[kk] = e[kk];
    } else if (MODE == RQHIP_MODE_STE) {
#pragma unroll
        for (int kk = 0; kk < KSTEPS; ++kk) o[kk] = r[kk] + (e[kk] - r[kk]);
    } else {
        float w[KSTEPS], u[KSTEPS], q[KSTEPS], scale;
        rotation_lane<KSTEPS>(r, e, xsq, o, w, u, q, scale);
    }
}

// load this lane's half of a [*, D] row (zero padded)
template <int KSTEPS>
__device__ __forceinline__ void load_pair_row(const float *__restrict__ base, int D, int h, float (&v)[KSTEPS]) {
#pragma unroll
    for (int kk = 0; kk < KSTEPS; ++kk) {
        const int d = 2 * kk + h;
        v[kk] = (d < D) ? base[d] : 0.0f;
    }
}

typedef float rq_f32x4 __attribute__((ext_vector_type(4)));

// ---- row <-> "pair layout" conversion for the full-width kernels (D == 2*KSTEPS) ------------------------------
// The MFMA B operand wants lane (il, h) to hold the features d = 2 kk + h of row il.  Loading them directly makes every
// load/store instruction touch 32 cache lines 4 bytes at a time (16 instructions per 128-byte row).  Instead each
// lane half moves one contiguous half of the row as float4s -- lane (il,0) bytes [0, 2D), lane (il,1) bytes [2D, 4D)
// -- and the two halves trade the components of the wrong parity with v_permlane32_swap (gfx950): a quarter of the
// memory instructions, 16 bytes per lane each.
//   raw[4j+c]  (before) : lane (il,h) holds feature h*KSTEPS + 4j + c
//   swap(raw[4j+0], raw[4j+1]) -> r[2j],   r[KSTEPS/2 + 2j]      swap(raw[4j+2], raw[4j+3]) -> r[2j+1], r[KSTEPS/2 + 2j+1]
// The swap is its own inverse, so the same two instructions turn pair-layout registers back into row chunks.
__device__ __forceinline__ void rq_swap32(float a, float b, float &a_out, float &b_out) {
    // a_out = {lanes 0-31: a, lanes 32-63: b of lane-32};  b_out = {lanes 0-31: a of lane+32, lanes 32-63: b}
    // (inline asm: with __builtin_amdgcn_permlane32_swap this compiler drops the second result when two swaps
    // share a source register; the s_nop covers the VALU-write -> permlane-swap hazard the assembler cannot see)
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    a_out = a;
    b_out = b;
}
template <int KSTEPS>
__device__ __forceinline__ void rows_to_pairs(const float (&raw)[KSTEPS], float (&r)[KSTEPS]) {
#pragma unroll
    for (int j = 0; j < KSTEPS / 4; ++j) {
        rq_swap32(raw[4 * j + 0], raw[4 * j + 1], r[2 * j], r[KSTEPS / 2 + 2 * j]);
        rq_swap32(raw[4 * j + 2], raw[4 * j + 3], r[2 * j + 1], r[KSTEPS / 2 + 2 * j + 1]);
    }
}
template <int KSTEPS>
__device__ __forceinline__ void pairs_to_rows(const float (&r)[KSTEPS], float (&raw)[KSTEPS]) {
#pragma unroll
    for (int j = 0; j < KSTEPS / 4; ++j) {
        rq_swap32(r[2 * j], r[KSTEPS / 2 + 2 * j], raw[4 * j + 0], raw[4 * j + 1]);
        rq_swap32(r[2 * j + 1], r[KSTEPS / 2 + 2 * j + 1], raw[4 * j + 2], raw[4 * j + 3]);
    }
}
// store KSTEPS pair-layout registers of this lane as its half of row `dst_row` (dst_row = base + row*D, 16-byte aligned)
template <int KSTEPS>
__device__ __forceinline__ void store_pair_row(float *dst_row, int h, const float (&r)[KSTEPS]) {
    float raw[KSTEPS];
    pairs_to_rows<KSTEPS>(r, raw);
    rq_f32x4 *dst = reinterpret_cast<rq_f32x4 *>(dst_row + h * KSTEPS);
#pragma unroll
    for (int j = 0; j < KSTEPS / 4; ++j) dst[j] = rq_f32x4{raw[4 * j], raw[4 * j + 1], raw[4 * j + 2], raw[4 * j + 3]};
}

// this lane's half of a full-width row (D == 2*KSTEPS, 16-byte aligned) straight into pair layout
template <int KSTEPS>
__device__ __forceinline__ void load_pair_row_vec(const float *__restrict__ row_base, int h, float (&v)[KSTEPS]) {
    float raw[KSTEPS];
    const rq_f32x4 *src = reinterpret_cast<const rq_f32x4 *>(row_base + h * KSTEPS);
#pragma unroll
    for (int j = 0; j < KSTEPS / 4; ++j) {
        const rq_f32x4 q = src[j];
        raw[4 * j + 0] = q.x; raw[4 * j + 1] = q.y; raw[4 * j + 2] = q.z; raw[4 * j + 3] = q.w;
    }
    rows_to_pairs<KSTEPS>(raw, v);
}

inline int ksteps_for(int D) { return D <= 8 ? 4 : D <= 16 ? 8 : D <= 32 ? 16 : D <= 64 ? 32 : 64; }

}  // namespace rqhip
