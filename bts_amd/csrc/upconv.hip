// Tap-sum stage of the "tap GEMM" upconv (include/bts_hip.h: bts_upconv_combine_f32).
//
// upconv = nearest-2x upsample + 3x3 convolution (pytorch/bts.py:90-93).  On a SMALL, WIDE map (upconv5: 11x38 pixels,
// 2208 -> 512 channels) the cheapest exact formulation is neither the 3x3 on the upsampled map (9 taps x 4 output pixels
// per source pixel) nor the four 2x2 parity convolutions (16 tap-products per source pixel): it is ONE 1x1 convolution
// of the source map with all nine kernel taps side by side, P[s][t*C + n] = sum_c w[n][c][t] x[s][c] -- 9 tap-products per
// source pixel, a plain GEMM -- followed by this kernel, which gives every output pixel the nine P values its taps see:
//   y[2Y+py][2X+px][n] = sum_{ky,kx} P[Y + dy(py,ky)][X + dx(px,kx)][(3*ky+kx)*C + n],  d(0,.) = (-1,0,0), d(1,.) = (0,0,+1)
// (zero outside the source map = the zero padding of the upsampled map).  The sum runs in tap order 0..8 for every
// pixel: deterministic, and a frame's bits do not depend on its batch.  HBM/L2-bound: 9 reads of 16 B per 16 B written,
// every P line re-used by the four parity classes.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int ACT>
__global__ __launch_bounds__(256) void upconv_combine_kernel(const float* __restrict__ taps, long tps, int B, int h, int w, int c,
                                                             const float* __restrict__ e2s, const float* __restrict__ e2b,
                                                             float* __restrict__ y, long ys) {
    const int c4n = c >> 2;
    const long total = (long)B * (2 * h) * (2 * w) * c4n;
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
        const long op = t / c4n;                            // output pixel (b, oy, ox)
        const int n = (int)(t - op * c4n) * 4;
        const int ox = (int)(op % (2 * w));
        const long r = op / (2 * w);
        const int oy = (int)(r % (2 * h));
        const long b = r / (2 * h);
        const int Y = oy >> 1, py = oy & 1, X = ox >> 1, px = ox & 1;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int sy = Y + (py == 0 ? (ky == 0 ? -1 : 0) : (ky == 2 ? 1 : 0));
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int sx = X + (px == 0 ? (kx == 0 ? -1 : 0) : (kx == 2 ? 1 : 0));
                if (sy >= 0 && sy < h && sx >= 0 && sx < w)
                    v += *reinterpret_cast<const f32x4*>(taps + ((b * h + sy) * w + sx) * tps + (3 * ky + kx) * c + n);
            }
        }
        if (ACT == 1) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        if (ACT == 2) { v.x = elu1(v.x); v.y = elu1(v.y); v.z = elu1(v.z); v.w = elu1(v.w); }
        if (e2s) v = v * *reinterpret_cast<const f32x4*>(e2s + n) + *reinterpret_cast<const f32x4*>(e2b + n);
        *reinterpret_cast<f32x4*>(y + op * ys + n) = v;
    }
}

}  // namespace

extern "C" int bts_upconv_combine_f32(const float* taps, long taps_pix_stride, int B, int h, int w, int c,
                                      const float* e2_scale, const float* e2_shift, int act, float* y, long y_pix_stride,
                                      bts_stream_t stream) {
    if (!taps || !y || B < 0 || h <= 0 || w <= 0 || c <= 0 || (c & 3) || taps_pix_stride < 9L * c || (taps_pix_stride & 3) ||
        y_pix_stride < c || (y_pix_stride & 3) || act < 0 || act > 2 || ((e2_scale == nullptr) != (e2_shift == nullptr)))
        return BTS_ERR_INVALID;
    if (((uintptr_t)taps | (uintptr_t)y | (uintptr_t)e2_scale | (uintptr_t)e2_shift) & 15) return BTS_ERR_INVALID;
    const long total = (long)B * (2 * h) * (2 * w) * (c >> 2);
    if (total == 0) return 0;
    if ((long)B * h * w * taps_pix_stride > 0x7fffffffffffL) return BTS_ERR_INVALID;
    long blocks = (total + 255) / 256;
    if (blocks > 256L * 32) blocks = 256L * 32;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (act == 0) hipLaunchKernelGGL(upconv_combine_kernel<0>, dim3((unsigned)blocks), dim3(256), 0, s, taps, taps_pix_stride, B, h, w, c, e2_scale, e2_shift, y, y_pix_stride);
    else if (act == 1) hipLaunchKernelGGL(upconv_combine_kernel<1>, dim3((unsigned)blocks), dim3(256), 0, s, taps, taps_pix_stride, B, h, w, c, e2_scale, e2_shift, y, y_pix_stride);
    else hipLaunchKernelGGL(upconv_combine_kernel<2>, dim3((unsigned)blocks), dim3(256), 0, s, taps, taps_pix_stride, B, h, w, c, e2_scale, e2_shift, y, y_pix_stride);
    return (int)hipGetLastError();
}
