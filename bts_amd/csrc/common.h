// Shared declarations for the gfx950 BTS hot-path kernels.
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/bts_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// v_mfma_f32_32x32x2_f32: D[32x32] += A[32x2] * B[2x32], exact f32 (fmaf chain).
// lane l supplies A[i = l&31][k = l>>5] and B[k = l>>5][j = l&31];
// D register r of lane l is D[row = (r&3) + 8*(r>>2) + 4*(l>>5)][col = l&31].
__device__ __forceinline__ f32x16 mfma32x2(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ float elu1(float x) { return x > 0.f ? x : expm1f(x); }      // nn.ELU(alpha=1)
__device__ __forceinline__ float sigmoid1(float x) { return 1.f / (1.f + expf(-x)); }   // nn.Sigmoid
