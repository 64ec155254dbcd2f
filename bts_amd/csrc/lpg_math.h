// LPG arithmetic shared by the stand-alone fused kernel (lpg.hip) and the reduction->LPG kernel (reduc.hip), so both
// produce the same bits: den = n1*u + n2*v + n3 with explicit FMAs, the +-1e-3 clamp of pytorch/bts.py:168-171, and
// depth/max_depth = n4 * rcp(den * max_depth) (one hardware reciprocal, <= ~1.5 ulp; the bit-exact module-level op is
// lpg_fwd_kernel in lpg.hip).  All expressions are explicit (fmaf / separate multiplies): -ffp-contract cannot change them.
#pragma once
#include <hip/hip_runtime.h>

__device__ __forceinline__ float lpg_clamp(float d) {
    const float eps = 1e-3f;
    if (d > 0.f && d < eps) d = eps;          // bts.py:170
    if (d < 0.f && d > -eps) d = -eps;        // bts.py:171
    return d;
}

// NaN-propagating |den| minimum (bts.py:167, torch.abs(divided).min()): int32 keys -- a non-negative float's bits keep
// the float order as a signed int, "saw a NaN" is the negative key 0xffc00000 (reads back as NaN from the float scalar).
constexpr int LPG_NAN_KEY = (int)0xffc00000u;
__device__ __forceinline__ int lpg_min_key(float amin, bool saw_nan) { return saw_nan ? LPG_NAN_KEY : __float_as_int(amin); }

// NOUT consecutive output columns starting at column c0 (c0 % K = first column inside the cell) of output row phase
// `rk` (0..K-1) of the cell with plane (n1,n2,n3,n4): res[i] = depth/max_depth; amin / saw_nan accumulate |den|.
template <int K, int NOUT>
__device__ __forceinline__ void lpg_cell_outputs(float n1, float n2, float n3, float n4, int rk, int ck0, float max_depth,
                                                 float (&res)[NOUT], float& amin, bool& saw_nan) {
    constexpr float invK = 1.0f / (float)K;
    const float v = ((float)rk - (float)(K - 1) * 0.5f) * invK;
    const float base = fmaf(n2, v, n3);                               // n2*v + n3
    const float u0 = ((float)ck0 - (float)(K - 1) * 0.5f) * invK;
#pragma unroll
    for (int i = 0; i < NOUT; ++i) {
        float d = fmaf(n1, u0 + (float)i * invK, base);               // n1*u + n2*v + n3   (bts.py:166)
        amin = fminf(amin, fabsf(d));                                 // bts.py:167
        saw_nan |= d != d;
        d = lpg_clamp(d);                                             // bts.py:168-171
        res[i] = n4 * __builtin_amdgcn_rcpf(d * max_depth);           // bts.py:173, 255
    }
}
