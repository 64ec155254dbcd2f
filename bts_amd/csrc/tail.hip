// The full-resolution tail of the decoder (reference pytorch/bts.py:286-291): concat1 assembly
// and get_depth (conv3x3 C->1 + sigmoid + max_depth/focal scaling).  Both are HBM-bound: a 1-output
// convolution has no GEMM shape worth an MFMA tile, so get_depth is a VALU stencil that reads the
// NCHW iconv1 planes (coalesced along x) exactly once from HBM.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "common.h"

namespace {

template <int N>
__global__ __launch_bounds__(256) void pack_planes_kernel(const float* __restrict__ p0, const float* __restrict__ p1,
                                                          const float* __restrict__ p2, const float* __restrict__ p3,
                                                          long npix, float* __restrict__ dst, long stride) {
    for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < npix; p += (long)gridDim.x * blockDim.x) {
        float* d = dst + p * stride;
        if (N == 4) {
            *reinterpret_cast<float4*>(d) = make_float4(p0[p], p1[p], p2[p], p3[p]);
        } else {
            d[0] = p0[p];
            if (N > 1) d[1] = p1[p];
            if (N > 2) d[2] = p2[p];
        }
    }
}

// Each thread: a 2-row x 8-column patch of outputs.  Per channel it reads 4 input rows x (two aligned float4 +
// two edge scalars) = 16 loads for 16 outputs x 9 taps; the earlier 1x4 patch issued 9 loads per 4 outputs and was
// bound by load issue, not HBM.  A block is a 2-D tile of 16 x 16 threads = 32 rows x 128 columns of one frame: the
// halo rows a block re-reads are 2 of 34 (1.06x), where the flat 1-D thread order of round 1 made every block a
// 3.4-row strip and fetched 1.64x the tensor (profiles/r01_pmc_traffic.json).
constexpr int GD_TX = 16, GD_TY = 16;                      // threads per block in x / y
template <int C>
__global__ __launch_bounds__(GD_TX * GD_TY) void get_depth_kernel(const float* __restrict__ in, const float* __restrict__ w,
                                                                  int B, int H, int W, float max_depth,
                                                                  const float* __restrict__ focal, float* __restrict__ out) {
    __shared__ float ws[C * 9];
    for (int i = threadIdx.x; i < C * 9; i += blockDim.x) ws[i] = w[i];
    __syncthreads();
    const long HW = (long)H * W;
    const int b = blockIdx.z;
    const int x0 = (blockIdx.x * GD_TX + (threadIdx.x % GD_TX)) * 8;
    const int y0 = (blockIdx.y * GD_TY + (threadIdx.x / GD_TX)) * 2;
    if (x0 >= W || y0 >= H) return;
    float acc[2][8];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[r][i] = 0.f;
    const float* base = in + (long)b * C * HW;
#pragma unroll 2
    for (int c = 0; c < C; ++c) {
        const float* pc = base + (long)c * HW;
        float row[4][10];                     // input rows y0-1 .. y0+2, columns x0-1 .. x0+8
#pragma unroll
        for (int ry = 0; ry < 4; ++ry) {
            const int yy = y0 - 1 + ry;
            if (yy >= 0 && yy < H) {
                const float* p = pc + (long)yy * W + x0;
                const float4 m0 = *reinterpret_cast<const float4*>(p);
                const float4 m1 = *reinterpret_cast<const float4*>(p + 4);
                row[ry][0] = x0 > 0 ? p[-1] : 0.f;
                row[ry][1] = m0.x; row[ry][2] = m0.y; row[ry][3] = m0.z; row[ry][4] = m0.w;
                row[ry][5] = m1.x; row[ry][6] = m1.y; row[ry][7] = m1.z; row[ry][8] = m1.w;
                row[ry][9] = x0 + 8 < W ? p[8] : 0.f;
            } else {
#pragma unroll
                for (int i = 0; i < 10; ++i) row[ry][i] = 0.f;
            }
        }
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            const float k0 = ws[c * 9 + dy * 3 + 0], k1 = ws[c * 9 + dy * 3 + 1], k2 = ws[c * 9 + dy * 3 + 2];
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    acc[r][i] += row[r + dy][i] * k0 + row[r + dy][i + 1] * k1 + row[r + dy][i + 2] * k2;
        }
    }
    const float fs = focal != nullptr ? focal[b] : 0.f;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        if (y0 + r >= H) continue;
        float o[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float v = max_depth * sigmoid1(acc[r][i]);                    // bts.py:289
            if (focal != nullptr) v = v * fs / 715.0873f;                 // bts.py:291
            o[i] = v;
        }
        float* q = out + ((long)b * H + y0 + r) * W + x0;
        *reinterpret_cast<float4*>(q) = make_float4(o[0], o[1], o[2], o[3]);
        *reinterpret_cast<float4*>(q + 4) = make_float4(o[4], o[5], o[6], o[7]);
    }
}

}  // namespace

extern "C" int bts_pack_planes_f32(const float* p0, const float* p1, const float* p2, const float* p3, int n_planes,
                                   long npix, float* dst, long dst_pix_stride, bts_stream_t stream) {
    if (!p0 || !dst || npix <= 0 || n_planes < 1 || n_planes > 4 || dst_pix_stride < n_planes) return BTS_ERR_INVALID;
    if ((n_planes > 1 && !p1) || (n_planes > 2 && !p2) || (n_planes > 3 && !p3)) return BTS_ERR_INVALID;
    if (n_planes == 4 && (((uintptr_t)dst & 15) || (dst_pix_stride & 3))) return BTS_ERR_INVALID;
    long blocks = (npix + 255) / 256;
    if (blocks > 256L * 16) blocks = 256L * 16;
    hipStream_t s = (hipStream_t)stream;
    dim3 g((unsigned)blocks), b(256);
    switch (n_planes) {
        case 1: hipLaunchKernelGGL(pack_planes_kernel<1>, g, b, 0, s, p0, p1, p2, p3, npix, dst, dst_pix_stride); break;
        case 2: hipLaunchKernelGGL(pack_planes_kernel<2>, g, b, 0, s, p0, p1, p2, p3, npix, dst, dst_pix_stride); break;
        case 3: hipLaunchKernelGGL(pack_planes_kernel<3>, g, b, 0, s, p0, p1, p2, p3, npix, dst, dst_pix_stride); break;
        default: hipLaunchKernelGGL(pack_planes_kernel<4>, g, b, 0, s, p0, p1, p2, p3, npix, dst, dst_pix_stride); break;
    }
    return (int)hipGetLastError();
}

extern "C" int bts_get_depth_f32(const float* iconv1, const float* w, int B, int C, int H, int W, float max_depth,
                                 const float* focal, float* final_depth, bts_stream_t stream) {
    if (!iconv1 || !w || !final_depth || B <= 0 || H <= 0 || W <= 0) return BTS_ERR_INVALID;
    if ((W & 7) || ((uintptr_t)iconv1 & 15) || ((uintptr_t)final_depth & 15)) return BTS_ERR_UNSUPPORTED;
    if (B > 65535) return BTS_ERR_UNSUPPORTED;
    const dim3 grid((unsigned)((W / 8 + GD_TX - 1) / GD_TX), (unsigned)(((H + 1) / 2 + GD_TY - 1) / GD_TY), (unsigned)B);
    const dim3 block(GD_TX * GD_TY);
    hipStream_t s = (hipStream_t)stream;
    if (C == 32)
        hipLaunchKernelGGL(get_depth_kernel<32>, grid, block, 0, s, iconv1, w, B, H, W, max_depth, focal, final_depth);
    else if (C == 16)
        hipLaunchKernelGGL(get_depth_kernel<16>, grid, block, 0, s, iconv1, w, B, H, W, max_depth, focal, final_depth);
    else
        return BTS_ERR_UNSUPPORTED;
    return (int)hipGetLastError();
}
