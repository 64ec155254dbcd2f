// Shared declarations for the gfx950 BTS hot-path kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
#include "../../include/bts_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// v_mfma_f32_32x32x2_f32: D[32x32] += A[32x2] * B[2x32], exact f32 (fmaf chain).
// lane l supplies A[i = l&31][k = l>>5] and B[k = l>>5][j = l&31];
// D register r of lane l is D[row = (r&3) + 8*(r>>2) + 4*(l>>5)][col = l&31].
__device__ __forceinline__ f32x16 mfma32x2(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// nn.ELU(alpha=1) and nn.Sigmoid on the hardware exponential (v_exp_f32 behind one multiply) and reciprocal: ~3 and ~5 VALU
// instructions where expm1f / expf + an IEEE division cost ~35 / ~30.  The decoder applies ELU to every output of every
// convolution (860 M elements per B=16 step) and 32 times per pixel group inside the reduction chains, so the accurate
// forms were a measurable share of those kernels' issue slots.  exp(x) - 1 is ATen's own ELU formula; its absolute error
// (<= 1e-7, relative to an O(1) activation scale) is what matters downstream -- the relative error near x = 0, where
// expm1f differs, does not survive the next layer's sum.  Parity: the oracle / golden tolerances are unchanged (tests).
__device__ __forceinline__ float elu1(float x) { return x > 0.f ? x : __expf(x) - 1.f; }
__device__ __forceinline__ float sigmoid1(float x) { return __frcp_rn(1.f + __expf(-x)); }

// Raise a kernel's dynamic-LDS limit above the 64 KB default.  hipFuncAttributeMaxDynamicSharedMemorySize is a property
// of (function, DEVICE): a process that drives several GPUs (nn.DataParallel: one Python thread per device,
// bts_test.py:91) must set it on each of them.  `done` is one bit per device ordinal, owned by the kernel
// instantiation's launcher; the only library state that is ever written after load, idempotent (setting the attribute
// twice is harmless) and lock-free, so concurrent callers on any mix of devices and streams are safe.
inline hipError_t bts_ensure_dynamic_lds(const void* kern, size_t bytes, std::atomic<unsigned long long>& done) {
    if (bytes < 64 * 1024) return hipSuccess;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const unsigned long long bit = (dev >= 0 && dev < 64) ? 1ull << dev : 0ull;      // ordinals >= 64: set every time
    if (bit && (done.load(std::memory_order_acquire) & bit)) return hipSuccess;
    e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) return e;
    if (bit) done.fetch_or(bit, std::memory_order_release);
    return hipSuccess;
}
