// Batch-statistic BatchNorm for the BTS training step (reference: nn.BatchNorm2d modules in train() mode,
// pytorch/bts.py:69-76, 182-202 and torchvision's DenseNet norm layers; driven by bts_main.py:476-500) on gfx950.
//
// NHWC rows [npix][C] (row stride >= C, so channel slices of concat buffers work in place).  All four kernels are
// HBM-bound streaming passes: a block covers up to 128 channels x 8 pixel rows per step with 16-byte lanes
// (fewer lanes, more rows for narrow tensors), several independent loads in flight per thread.  The per-channel reductions go through a two-level tree (thread ->
// block -> a finalize kernel that walks the block partials in a fixed order): deterministic, no atomics.
//   forward : bn_stats (Chan-merged (n, mean, M2) partials, shifted sums)  ->  bn_stats_finalize (mean, invstd,
//             fused scale/shift, running-stat update)  ->  bn_apply (y = [relu](x*scale + shift))
//   backward: bn_bwd_reduce (sum dy', sum dy'*xhat; dy' = dy masked by the fused ReLU)  ->  finalize
//             ->  bn_bwd_apply (dx = gamma*invstd*(dy' - mean(dy') - xhat*mean(dy'*xhat)))
#include "common.h"
#include <stdint.h>

namespace {


__device__ __forceinline__ void chan_merge(float& n, float& mean, float& m2, float nb, float meanb, float m2b) {
    if (nb == 0.f) return;
    const float nt = n + nb;
    const float d = meanb - mean;
    mean += d * (nb / nt);
    m2 += m2b + d * d * (n * nb / nt);
    n = nt;
}

// partial (n, mean, M2) of one pixel chunk for LX*4 channels -> ws[(chunk*3 + {0,1,2})*C + c].
// LX = float4 lanes per pixel row (32 / 16 / 8 for wide / 64.. / narrow tensors), 256/LX rows per block step.
template <int LX>
__global__ __launch_bounds__(256) void bn_stats_kernel(const float* __restrict__ x, long xs, long npix, int C,
                                                       long rows_per_chunk, float* __restrict__ ws) {
    constexpr int RYv = 256 / LX, CGv = LX * 4;
    const int cx = threadIdx.x % LX, ry = threadIdx.x / LX;
    const int c4 = blockIdx.x * CGv + cx * 4;
    const bool active = c4 < C;
    const long p0 = (long)blockIdx.y * rows_per_chunk;
    const long p1 = min(npix, p0 + rows_per_chunk);
    f32x4 s = (f32x4)(0.f), q = (f32x4)(0.f), x0 = (f32x4)(0.f);
    float n = 0.f;
    if (active) {
        x0 = *reinterpret_cast<const f32x4*>(x + p0 * xs + c4);      // shift: kills the sum-of-squares cancellation
        long p = p0 + ry;
        for (; p + 3 * RYv < p1; p += 4 * RYv) {                      // four independent 16-byte loads in flight
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(x + p * xs + c4) - x0;
            const f32x4 v1 = *reinterpret_cast<const f32x4*>(x + (p + RYv) * xs + c4) - x0;
            const f32x4 v2 = *reinterpret_cast<const f32x4*>(x + (p + 2 * RYv) * xs + c4) - x0;
            const f32x4 v3 = *reinterpret_cast<const f32x4*>(x + (p + 3 * RYv) * xs + c4) - x0;
            s += (v0 + v1) + (v2 + v3);
            q += (v0 * v0 + v1 * v1) + (v2 * v2 + v3 * v3);
            n += 4.f;
        }
        for (; p < p1; p += RYv) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(x + p * xs + c4) - x0;
            s += v;
            q += v * v;
            n += 1.f;
        }
    }
    __shared__ float red[RYv][CGv][3];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float mean = n > 0.f ? s[j] / n : 0.f;
        red[ry][cx * 4 + j][0] = n;
        red[ry][cx * 4 + j][1] = x0[j] + mean;
        red[ry][cx * 4 + j][2] = n > 0.f ? q[j] - s[j] * mean : 0.f;
    }
    __syncthreads();
    if (threadIdx.x < CGv) {
        const int c = blockIdx.x * CGv + threadIdx.x;
        if (c < C) {
            float nn = 0.f, mean = 0.f, m2 = 0.f;
#pragma unroll 8
            for (int r = 0; r < RYv; ++r) chan_merge(nn, mean, m2, red[r][threadIdx.x][0], red[r][threadIdx.x][1], red[r][threadIdx.x][2]);
            float* o = ws + (size_t)blockIdx.y * 3 * C;
            o[c] = nn;
            o[C + c] = mean;
            o[2 * C + c] = m2;
        }
    }
}

// Finalize kernels: a block owns 32 channels; 8 lanes per channel walk the chunk partials (strided, loads unrolled
// so they are independent of the merge chain), then one lane merges the 8 results in a fixed order.
__global__ __launch_bounds__(256) void bn_stats_finalize_kernel(const float* __restrict__ ws, int nchunks, int C,
                                                                const float* __restrict__ gamma,
                                                                const float* __restrict__ beta, float eps, float momentum,
                                                                float* __restrict__ running_mean,
                                                                float* __restrict__ running_var, float* __restrict__ mean_o,
                                                                float* __restrict__ invstd_o, float* __restrict__ scale_o,
                                                                float* __restrict__ shift_o) {
    const int cl = threadIdx.x & 31, lane = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cl;
    const int cc = c < C ? c : C - 1;
    float n = 0.f, mean = 0.f, m2 = 0.f;
    int k = lane;
    for (; k + 24 < nchunks; k += 32) {
        float pn[4], pm[4], pq[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float* p = ws + (size_t)(k + 8 * u) * 3 * C;
            pn[u] = p[cc]; pm[u] = p[C + cc]; pq[u] = p[2 * C + cc];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) chan_merge(n, mean, m2, pn[u], pm[u], pq[u]);
    }
    for (; k < nchunks; k += 8) {
        const float* p = ws + (size_t)k * 3 * C;
        chan_merge(n, mean, m2, p[cc], p[C + cc], p[2 * C + cc]);
    }
    __shared__ float red[8][32][3];
    red[lane][cl][0] = n; red[lane][cl][1] = mean; red[lane][cl][2] = m2;
    __syncthreads();
    if (lane != 0 || c >= C) return;
    n = 0.f; mean = 0.f; m2 = 0.f;
#pragma unroll
    for (int r = 0; r < 8; ++r) chan_merge(n, mean, m2, red[r][cl][0], red[r][cl][1], red[r][cl][2]);
    const float var = m2 / n;                                   // biased: what normalisation uses
    const float invstd = 1.f / sqrtf(var + eps);
    const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
    mean_o[c] = mean;
    invstd_o[c] = invstd;
    scale_o[c] = g * invstd;
    shift_o[c] = b - mean * g * invstd;
    if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
    if (running_var) running_var[c] = (1.f - momentum) * running_var[c] + momentum * (n > 1.f ? m2 / (n - 1.f) : var);
}

__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ x, long xs, long npix, int C,
                                                       const float* __restrict__ scale, const float* __restrict__ shift,
                                                       int relu, float* __restrict__ y, long ys) {
    const int c4n = C >> 2;
    const long total = npix * c4n;
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
        const long p = t / c4n;
        const int c = (int)(t - p * c4n) * 4;
        f32x4 v = *reinterpret_cast<const f32x4*>(x + p * xs + c);
        v = v * *reinterpret_cast<const f32x4*>(scale + c) + *reinterpret_cast<const f32x4*>(shift + c);
        if (relu) v = __builtin_elementwise_max(v, (f32x4)(0.f));
        *reinterpret_cast<f32x4*>(y + p * ys + c) = v;
    }
}

// partial (sum dy', sum dy'*xhat) -> ws[(chunk*2 + {0,1})*C + c]
template <int LX>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const float* __restrict__ x, long xs,
                                                            const float* __restrict__ dy, long dys, long npix, int C,
                                                            const float* __restrict__ mean, const float* __restrict__ invstd,
                                                            const float* __restrict__ scale, const float* __restrict__ shift,
                                                            int relu, long rows_per_chunk, float* __restrict__ ws) {
    constexpr int RYv = 256 / LX, CGv = LX * 4;
    const int cx = threadIdx.x % LX, ry = threadIdx.x / LX;
    const int c4 = blockIdx.x * CGv + cx * 4;
    const bool active = c4 < C;
    const long p0 = (long)blockIdx.y * rows_per_chunk;
    const long p1 = min(npix, p0 + rows_per_chunk);
    f32x4 s1 = (f32x4)(0.f), s2 = (f32x4)(0.f);
    if (active) {
        const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + c4), is = *reinterpret_cast<const f32x4*>(invstd + c4);
        const f32x4 sc = *reinterpret_cast<const f32x4*>(scale + c4), sh = *reinterpret_cast<const f32x4*>(shift + c4);
        auto one = [&](const f32x4 xv, f32x4 g) __attribute__((always_inline)) {
            if (relu) {
                const f32x4 z = xv * sc + sh;
#pragma unroll
                for (int j = 0; j < 4; ++j) g[j] = z[j] > 0.f ? g[j] : 0.f;
            }
            s1 += g;
            s2 += g * ((xv - mu) * is);
        };
        long p = p0 + ry;
        for (; p + RYv < p1; p += 2 * RYv) {                          // 2 x (x, dy): four independent loads in flight
            const f32x4 xa = *reinterpret_cast<const f32x4*>(x + p * xs + c4);
            const f32x4 ga = *reinterpret_cast<const f32x4*>(dy + p * dys + c4);
            const f32x4 xb = *reinterpret_cast<const f32x4*>(x + (p + RYv) * xs + c4);
            const f32x4 gb = *reinterpret_cast<const f32x4*>(dy + (p + RYv) * dys + c4);
            one(xa, ga);
            one(xb, gb);
        }
        for (; p < p1; p += RYv)
            one(*reinterpret_cast<const f32x4*>(x + p * xs + c4), *reinterpret_cast<const f32x4*>(dy + p * dys + c4));
    }
    __shared__ float red[RYv][CGv][2];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        red[ry][cx * 4 + j][0] = s1[j];
        red[ry][cx * 4 + j][1] = s2[j];
    }
    __syncthreads();
    if (threadIdx.x < CGv) {
        const int c = blockIdx.x * CGv + threadIdx.x;
        if (c < C) {
            float a = 0.f, b = 0.f;
#pragma unroll 8
            for (int r = 0; r < RYv; ++r) { a += red[r][threadIdx.x][0]; b += red[r][threadIdx.x][1]; }
            float* o = ws + (size_t)blockIdx.y * 2 * C;
            o[c] = a;
            o[C + c] = b;
        }
    }
}

__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(const float* __restrict__ ws, int nchunks, int C,
                                                              float* __restrict__ sum_dy, float* __restrict__ sum_dy_xhat) {
    const int cl = threadIdx.x & 31, lane = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cl;
    const int cc = c < C ? c : C - 1;
    float a = 0.f, b = 0.f;
#pragma unroll 4
    for (int k = lane; k < nchunks; k += 8) {
        a += ws[(size_t)k * 2 * C + cc];
        b += ws[(size_t)k * 2 * C + C + cc];
    }
    __shared__ float red[8][32][2];
    red[lane][cl][0] = a; red[lane][cl][1] = b;
    __syncthreads();
    if (lane != 0 || c >= C) return;
    a = 0.f; b = 0.f;
#pragma unroll
    for (int r = 0; r < 8; ++r) { a += red[r][cl][0]; b += red[r][cl][1]; }
    sum_dy[c] = a;            // = d loss / d beta
    sum_dy_xhat[c] = b;       // = d loss / d gamma
}

__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ x, long xs,
                                                           const float* __restrict__ dy, long dys, long npix, int C,
                                                           const float* __restrict__ mean, const float* __restrict__ invstd,
                                                           const float* __restrict__ scale, const float* __restrict__ shift,
                                                           int relu, const float* __restrict__ sum_dy,
                                                           const float* __restrict__ sum_dy_xhat, float* __restrict__ dx,
                                                           long dxs) {
    const int c4n = C >> 2;
    const long total = npix * c4n;
    const float inv_n = 1.f / (float)npix;
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
        const long p = t / c4n;
        const int c = (int)(t - p * c4n) * 4;
        const f32x4 xv = *reinterpret_cast<const f32x4*>(x + p * xs + c);
        f32x4 g = *reinterpret_cast<const f32x4*>(dy + p * dys + c);
        const f32x4 sc = *reinterpret_cast<const f32x4*>(scale + c);
        if (relu) {
            const f32x4 z = xv * sc + *reinterpret_cast<const f32x4*>(shift + c);
#pragma unroll
            for (int j = 0; j < 4; ++j) g[j] = z[j] > 0.f ? g[j] : 0.f;
        }
        const f32x4 xhat = (xv - *reinterpret_cast<const f32x4*>(mean + c)) * *reinterpret_cast<const f32x4*>(invstd + c);
        const f32x4 r = sc * (g - *reinterpret_cast<const f32x4*>(sum_dy + c) * inv_n -
                              xhat * (*reinterpret_cast<const f32x4*>(sum_dy_xhat + c) * inv_n));
        *reinterpret_cast<f32x4*>(dx + p * dxs + c) = r;
    }
}

// float4 lanes per pixel row for a channel count: the widest of 32/16/8 that the tensor fills
inline int lanes_for(int C) { return C >= 112 ? 32 : (C >= 48 ? 16 : 8); }

// chunking shared by both reductions: ~1024 blocks in flight, at least 64 rows per chunk, at most 1024 chunks
inline void plan_chunks(long npix, int C, long* rows_per_chunk, int* nchunks) {
    const int lx = lanes_for(C), ryv = 256 / lx;
    const int cgroups = (C + lx * 4 - 1) / (lx * 4);
    long want = (1024 + cgroups - 1) / cgroups;
    const long max_chunks = (npix + 63) / 64;
    if (want > max_chunks) want = max_chunks;
    if (want > 1024) want = 1024;
    if (want < 1) want = 1;
    long rows = ((npix + want - 1) / want + ryv - 1) / ryv * ryv;
    *rows_per_chunk = rows;
    *nchunks = (int)((npix + rows - 1) / rows);
}

inline bool bad_rows(const float* p, long stride, int C) {
    return !p || (stride & 3) || stride < C || ((uintptr_t)p & 15);
}

inline unsigned elem_blocks(long total) {
    long b = (total + 255) / 256;
    return (unsigned)(b > 8192 ? 8192 : (b < 1 ? 1 : b));
}

}  // namespace

extern "C" long bts_bn_train_ws_floats(long npix, int C) {
    long rows; int nchunks;
    if (npix <= 0 || C <= 0) return 0;
    plan_chunks(npix, C, &rows, &nchunks);
    return (long)nchunks * 3 * C;
}

extern "C" int bts_bn_train_stats_f32(const float* x, long x_pix_stride, long npix, int C, const float* gamma,
                                      const float* beta, float eps, float momentum, float* running_mean,
                                      float* running_var, float* ws, long ws_floats, float* mean, float* invstd,
                                      float* scale, float* shift, bts_stream_t stream) {
    if (npix <= 0 || C <= 0 || (C & 3) || bad_rows(x, x_pix_stride, C)) return BTS_ERR_INVALID;
    if (!ws || !mean || !invstd || !scale || !shift) return BTS_ERR_INVALID;
    if (((uintptr_t)scale & 15) || ((uintptr_t)shift & 15) || ((uintptr_t)mean & 15) || ((uintptr_t)invstd & 15))
        return BTS_ERR_INVALID;
    long rows; int nchunks;
    plan_chunks(npix, C, &rows, &nchunks);
    if (ws_floats < (long)nchunks * 3 * C) return BTS_ERR_INVALID;
    hipStream_t s = (hipStream_t)stream;
    const int lx = lanes_for(C);
    const dim3 grid((C + lx * 4 - 1) / (lx * 4), nchunks);
    if (lx == 32) hipLaunchKernelGGL(bn_stats_kernel<32>, grid, dim3(256), 0, s, x, x_pix_stride, npix, C, rows, ws);
    else if (lx == 16) hipLaunchKernelGGL(bn_stats_kernel<16>, grid, dim3(256), 0, s, x, x_pix_stride, npix, C, rows, ws);
    else hipLaunchKernelGGL(bn_stats_kernel<8>, grid, dim3(256), 0, s, x, x_pix_stride, npix, C, rows, ws);
    hipLaunchKernelGGL(bn_stats_finalize_kernel, dim3((C + 31) / 32), dim3(256), 0, s, ws, nchunks, C, gamma, beta, eps,
                       momentum, running_mean, running_var, mean, invstd, scale, shift);
    return (int)hipGetLastError();
}

extern "C" int bts_bn_apply_nhwc_f32(const float* x, long x_pix_stride, long npix, int C, const float* scale,
                                     const float* shift, int relu, float* y, long y_pix_stride, bts_stream_t stream) {
    if (npix <= 0 || C <= 0 || (C & 3) || bad_rows(x, x_pix_stride, C) || bad_rows(y, y_pix_stride, C)) return BTS_ERR_INVALID;
    if (!scale || !shift || ((uintptr_t)scale & 15) || ((uintptr_t)shift & 15)) return BTS_ERR_INVALID;
    hipLaunchKernelGGL(bn_apply_kernel, dim3(elem_blocks(npix * (C >> 2))), dim3(256), 0, (hipStream_t)stream, x,
                       x_pix_stride, npix, C, scale, shift, relu, y, y_pix_stride);
    return (int)hipGetLastError();
}

extern "C" int bts_bn_train_bwd_f32(const float* x, long x_pix_stride, const float* dy, long dy_pix_stride, long npix,
                                    int C, const float* mean, const float* invstd, const float* scale,
                                    const float* shift, int relu, float* ws, long ws_floats, float* dgamma,
                                    float* dbeta, float* dx, long dx_pix_stride, bts_stream_t stream) {
    if (npix <= 0 || C <= 0 || (C & 3) || bad_rows(x, x_pix_stride, C) || bad_rows(dy, dy_pix_stride, C)) return BTS_ERR_INVALID;
    if (!mean || !invstd || !scale || !shift || !ws || !dgamma || !dbeta) return BTS_ERR_INVALID;
    if (((uintptr_t)mean & 15) || ((uintptr_t)invstd & 15) || ((uintptr_t)scale & 15) || ((uintptr_t)shift & 15) ||
        ((uintptr_t)dgamma & 15) || ((uintptr_t)dbeta & 15))
        return BTS_ERR_INVALID;
    if (dx && bad_rows(dx, dx_pix_stride, C)) return BTS_ERR_INVALID;
    long rows; int nchunks;
    plan_chunks(npix, C, &rows, &nchunks);
    if (ws_floats < (long)nchunks * 2 * C) return BTS_ERR_INVALID;
    hipStream_t s = (hipStream_t)stream;
    const int lx = lanes_for(C);
    const dim3 grid((C + lx * 4 - 1) / (lx * 4), nchunks);
    if (lx == 32) hipLaunchKernelGGL(bn_bwd_reduce_kernel<32>, grid, dim3(256), 0, s, x, x_pix_stride, dy, dy_pix_stride,
                                     npix, C, mean, invstd, scale, shift, relu, rows, ws);
    else if (lx == 16) hipLaunchKernelGGL(bn_bwd_reduce_kernel<16>, grid, dim3(256), 0, s, x, x_pix_stride, dy, dy_pix_stride,
                                          npix, C, mean, invstd, scale, shift, relu, rows, ws);
    else hipLaunchKernelGGL(bn_bwd_reduce_kernel<8>, grid, dim3(256), 0, s, x, x_pix_stride, dy, dy_pix_stride,
                            npix, C, mean, invstd, scale, shift, relu, rows, ws);
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((C + 31) / 32), dim3(256), 0, s, ws, nchunks, C, dbeta, dgamma);
    if (dx)
        hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(elem_blocks(npix * (C >> 2))), dim3(256), 0, s, x, x_pix_stride, dy,
                           dy_pix_stride, npix, C, mean, invstd, scale, shift, relu, dbeta, dgamma, dx, dx_pix_stride);
    return (int)hipGetLastError();
}
