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

inline int ksteps_for(int D) { return D <= 8 ? 4 : D <= 16 ? 8 : D <= 32 ? 16 : D <= 64 ? 32 : 64; }

}  // namespace rqhip
