// rqhip_common.h -- shared device/host helpers for librqhip (gfx950 only).
//
// Numerics contract (see include/rqhip.h and oracle/rq_oracle.c): the library is compiled with
// -ffp-contract=off, so the ONLY fused multiply-adds are the explicit __builtin_fmaf calls and the MFMA
// instructions (which are fp32 FMA chains in k order).  Reductions over the feature dimension use two
// accumulators split by the parity of d -- the split the 32x32x2 MFMA operand layout induces (lanes 0-31
// hold even d, lanes 32-63 odd d) -- combined as a0 + a1.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "rqhip.h"

#define RQ_WAVE 64

namespace rqhip {

void set_error(const char *fmt, ...);

inline int check_hip(hipError_t e, const char *what) {
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return (int)e;
    }
    return 0;
}

#define RQ_RETURN_IF_HIP(expr)                                                      \
    do {                                                                            \
        int _rc = ::rqhip::check_hip((expr), #expr);                                \
        if (_rc) return _rc;                                                        \
    } while (0)

#define RQ_CHECK_LAUNCH(name)                                                       \
    do {                                                                            \
        int _rc = ::rqhip::check_hip(hipGetLastError(), name);                      \
        if (_rc) return _rc;                                                        \
    } while (0)

// words behind a weight image (csrc/gemm_split.hip:weight_images_kernel) that hold the GEMM kernels' tile dispensers: zero between
// launches; the exponents of the rows of B follow them (csrc/gemm_split.hip)
constexpr int kWeightImageTailWords = 32;

int cu_count();          // compute units of the CURRENT device (cached per device)
constexpr int kMaxDevices = 16;
int current_device();    // hipGetDevice, 0 on error

// Raise a kernel's dynamic-LDS limit once per (kernel instantiation, DEVICE): the attribute belongs to each device's
// copy of the function, so a process that drives several GPUs must set it on every one of them.  Usage:
//   static LdsGrant g;  RQ_RETURN_IF_HIP(g.ensure(reinterpret_cast<const void *>(kernel), bytes));
struct LdsGrant {
    bool done[kMaxDevices] = {};
    hipError_t ensure(const void *kernel, int bytes) {
        const int dev = current_device();
        if (dev < kMaxDevices && done[dev]) return hipSuccess;
        const hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        if (e == hipSuccess && dev < kMaxDevices) done[dev] = true;
        return e;
    }
};

// bench-only kernel timing (capi.hip); no-ops unless rqhip_profile_enable(n > 0) was called
// (tag: RQHIP_PROF_*; flops / bytes: the ALGORITHMIC work of what is bracketed -- bench.py prices them against the rooflines)
void profile_begin(hipStream_t s, int tag, double flops, double bytes);
void profile_end(hipStream_t s);

// Fill `bytes` (a multiple of 4) at `dst` (4-byte aligned) with the 32-bit pattern `word`, as a KERNEL on stream s.
// Used instead of hipMemsetAsync everywhere in the library: memset nodes captured into a hipGraph broke replay
// (hang / memory access fault after a few hundred replays interleaved with eager launches, ROCm 7.0 runtime of
// torch 2.10; tools/graph_piece_probe.py), kernel nodes do not.
int fill_words(void *dst, uint32_t word, size_t bytes, hipStream_t s);

__device__ __forceinline__ float shfl_xor32(float v) { return __shfl_xor(v, 32, 64); }
__device__ __forceinline__ int shfl_xor32(int v) { return __shfl_xor(v, 32, 64); }

}  // namespace rqhip
