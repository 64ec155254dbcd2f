// NHWC pooling kernels for the DenseNet encoder taps (caller side of the hot path, reference
// pytorch/bts.py:295-338 walks torchvision's DenseNet: pool0 = MaxPool2d(3,2,1), transitionN.pool =
// AvgPool2d(2,2)).  Both are HBM-bound streaming kernels: one thread = one output pixel x 4 channels
// (16-byte loads/stores, channels contiguous).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void maxpool3x3s2_kernel(const float* __restrict__ src, long ss, int B, int h, int w,
                                                           int C4, float* __restrict__ d1, long s1,
                                                           float* __restrict__ d2, long s2) {
    const int ho = (h + 1) / 2, wo = (w + 1) / 2;          // floor((h + 2 - 3)/2) + 1
    const long total = (long)B * ho * wo * C4;
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
        const int c4 = (int)(t % C4);
        long p = t / C4;
        const int x = (int)(p % wo); p /= wo;
        const int y = (int)(p % ho);
        const int b = (int)(p / ho);
        float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            const int yy = 2 * y - 1 + dy;
            if (yy < 0 || yy >= h) continue;
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const int xx = 2 * x - 1 + dx;
                if (xx < 0 || xx >= w) continue;
                const float4 v = *reinterpret_cast<const float4*>(src + (((long)b * h + yy) * w + xx) * ss + c4 * 4);
                m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
            }
        }
        const long po = ((long)b * ho + y) * wo + x;
        *reinterpret_cast<float4*>(d1 + po * s1 + c4 * 4) = m;
        if (d2) *reinterpret_cast<float4*>(d2 + po * s2 + c4 * 4) = m;
    }
}

// out = mean over the 2x2 window of relu(x*scale + shift)   (transition: norm -> relu -> [conv] -> pool;
// the 1x1 conv commutes with the average, so pooling first quarters the conv's work)
__global__ __launch_bounds__(256) void bn_relu_avgpool2_kernel(const float* __restrict__ src, long ss, int B, int h, int w,
                                                               int C4, const float* __restrict__ scale,
                                                               const float* __restrict__ shift,
                                                               float* __restrict__ dst, long sd) {
    const int ho = h / 2, wo = w / 2;
    const long total = (long)B * ho * wo * C4;
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
        const int c4 = (int)(t % C4);
        long p = t / C4;
        const int x = (int)(p % wo); p /= wo;
        const int y = (int)(p % ho);
        const int b = (int)(p / ho);
        const float4 sc = *reinterpret_cast<const float4*>(scale + c4 * 4);
        const float4 sh = *reinterpret_cast<const float4*>(shift + c4 * 4);
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                const float4 v = *reinterpret_cast<const float4*>(src + (((long)b * h + 2 * y + dy) * w + 2 * x + dx) * ss + c4 * 4);
                acc.x += fmaxf(v.x * sc.x + sh.x, 0.f); acc.y += fmaxf(v.y * sc.y + sh.y, 0.f);
                acc.z += fmaxf(v.z * sc.z + sh.z, 0.f); acc.w += fmaxf(v.w * sc.w + sh.w, 0.f);
            }
        *reinterpret_cast<float4*>(dst + (((long)b * ho + y) * wo + x) * sd + c4 * 4) =
            make_float4(acc.x * 0.25f, acc.y * 0.25f, acc.z * 0.25f, acc.w * 0.25f);
    }
}

inline unsigned grid_for(long total) {
    long blocks = (total + 255) / 256;
    if (blocks > 256L * 16) blocks = 256L * 16;
    return (unsigned)blocks;
}

}  // namespace

extern "C" int bts_maxpool3x3s2_nhwc_f32(const float* src, long src_pix_stride, int B, int h, int w, int C, float* dst,
                                         long dst_pix_stride, float* dst2, long dst2_pix_stride, bts_stream_t stream) {
    if (!src || !dst || B <= 0 || h <= 0 || w <= 0 || C <= 0 || (C & 3)) return BTS_ERR_INVALID;
    if ((src_pix_stride & 3) || (dst_pix_stride & 3) || (dst2 && (dst2_pix_stride & 3))) return BTS_ERR_INVALID;
    if (((uintptr_t)src & 15) || ((uintptr_t)dst & 15) || ((uintptr_t)dst2 & 15)) return BTS_ERR_INVALID;
    const long total = (long)B * ((h + 1) / 2) * ((w + 1) / 2) * (C / 4);
    hipLaunchKernelGGL(maxpool3x3s2_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, src, src_pix_stride,
                       B, h, w, C / 4, dst, dst_pix_stride, dst2, dst2_pix_stride);
    return (int)hipGetLastError();
}

extern "C" int bts_bn_relu_avgpool2_nhwc_f32(const float* src, long src_pix_stride, int B, int h, int w, int C,
                                             const float* scale, const float* shift, float* dst, long dst_pix_stride,
                                             bts_stream_t stream) {
    if (!src || !dst || !scale || !shift || B <= 0 || h <= 0 || w <= 0 || C <= 0 || (C & 3) || (h & 1) || (w & 1))
        return BTS_ERR_INVALID;
    if ((src_pix_stride & 3) || (dst_pix_stride & 3)) return BTS_ERR_INVALID;
    if (((uintptr_t)src & 15) || ((uintptr_t)dst & 15) || ((uintptr_t)scale & 15) || ((uintptr_t)shift & 15))
        return BTS_ERR_INVALID;
    const long total = (long)B * (h / 2) * (w / 2) * (C / 4);
    hipLaunchKernelGGL(bn_relu_avgpool2_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, src,
                       src_pix_stride, B, h, w, C / 4, scale, shift, dst, dst_pix_stride);
    return (int)hipGetLastError();
}
