// Layout movers between the NCHW module boundary (reference pytorch/bts.py:347-349 hands
// NCHW tensors in and expects NCHW out) and the NHWC interior of the hot path, plus the ABI's
// version/error helpers.  32x32 tiles through LDS: both the global read and the global write
// are 128-byte coalesced.  HBM-bound (8 B moved per element).
#include <hip/hip_runtime.h>
#include "common.h"

namespace {

// src [B][C][HW] -> dst [B*HW][stride] (+ optional ReLU)
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* __restrict__ src, int C, long HW,
                                                           float* __restrict__ dst, long stride, int relu) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z;
    const long p0 = (long)blockIdx.x * 32;
    const int c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = c0 + ty + i * 8;
        const long p = p0 + tx;
        float v = 0.f;
        if (c < C && p < HW) v = src[((long)b * C + c) * HW + p];
        if (relu) v = fmaxf(v, 0.f);
        tile[ty + i * 8][tx] = v;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const long p = p0 + ty + i * 8;
        const int c = c0 + tx;
        if (c < C && p < HW) dst[((long)b * HW + p) * stride + c] = tile[tx][ty + i * 8];
    }
}

// src [B][C<=4][HW] -> dst [B*HW][stride]: one thread per pixel, planar coalesced reads, one 16-byte store
// (the 3-channel image: the 32x32 tile kernel would leave 29 of 32 lanes idle on the write side)
__global__ __launch_bounds__(256) void nchw_to_nhwc_c4_kernel(const float* __restrict__ src, int C, long HW, long total,
                                                              float* __restrict__ dst, long stride, int relu) {
    for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < total; p += (long)gridDim.x * blockDim.x) {
        const long b = p / HW, q = p - b * HW;
        const float* s = src + b * C * HW + q;
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        for (int c = 0; c < C; ++c) v[c] = relu ? fmaxf(s[c * HW], 0.f) : s[c * HW];
        float* d = dst + p * stride;
        if (C == 4) *reinterpret_cast<float4*>(d) = make_float4(v[0], v[1], v[2], v[3]);
        else for (int c = 0; c < C; ++c) d[c] = v[c];
    }
}

// src [B*HW][stride] -> dst [B][C][HW]
__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const float* __restrict__ src, long stride, int C,
                                                           long HW, float* __restrict__ dst) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z;
    const long p0 = (long)blockIdx.x * 32;
    const int c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const long p = p0 + ty + i * 8;
        const int c = c0 + tx;
        float v = 0.f;
        if (c < C && p < HW) v = src[((long)b * HW + p) * stride + c];
        tile[ty + i * 8][tx] = v;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = c0 + ty + i * 8;
        const long p = p0 + tx;
        if (c < C && p < HW) dst[((long)b * C + c) * HW + p] = tile[tx][ty + i * 8];
    }
}

}  // namespace

extern "C" int bts_nchw_to_nhwc_f32(const float* src, int B, int C, long HW, float* dst, long dst_pix_stride,
                                    int relu, bts_stream_t stream) {
    if (!src || !dst || B <= 0 || C <= 0 || HW <= 0 || dst_pix_stride < C) return BTS_ERR_INVALID;
    if (C <= 4) {
        const long total = (long)B * HW;
        long blocks = (total + 255) / 256;
        if (blocks > 256L * 16) blocks = 256L * 16;
        hipLaunchKernelGGL(nchw_to_nhwc_c4_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, src, C, HW,
                           total, dst, dst_pix_stride, relu);
        return (int)hipGetLastError();
    }
    if (B > 65535 || (C + 31) / 32 > 65535) return BTS_ERR_UNSUPPORTED;
    dim3 grid((unsigned)((HW + 31) / 32), (unsigned)((C + 31) / 32), (unsigned)B);
    hipLaunchKernelGGL(nchw_to_nhwc_kernel, grid, dim3(256), 0, (hipStream_t)stream, src, C, HW, dst,
                       dst_pix_stride, relu);
    return (int)hipGetLastError();
}

extern "C" int bts_nhwc_to_nchw_f32(const float* src, long src_pix_stride, int B, int C, long HW, float* dst,
                                    bts_stream_t stream) {
    if (!src || !dst || B <= 0 || C <= 0 || HW <= 0 || src_pix_stride < C) return BTS_ERR_INVALID;
    if (B > 65535 || (C + 31) / 32 > 65535) return BTS_ERR_UNSUPPORTED;
    dim3 grid((unsigned)((HW + 31) / 32), (unsigned)((C + 31) / 32), (unsigned)B);
    hipLaunchKernelGGL(nhwc_to_nchw_kernel, grid, dim3(256), 0, (hipStream_t)stream, src, src_pix_stride, C, HW,
                       dst);
    return (int)hipGetLastError();
}

extern "C" int bts_hip_abi_version(void) { return BTS_HIP_ABI_VERSION; }

extern "C" const char* bts_hip_error_string(int code) {
    if (code == 0) return "ok";
    if (code == BTS_ERR_INVALID) return "bts_hip: invalid argument";
    if (code == BTS_ERR_UNSUPPORTED) return "bts_hip: configuration not built";
    if (code > 0) return hipGetErrorString((hipError_t)code);
    return "bts_hip: unknown error";
}
