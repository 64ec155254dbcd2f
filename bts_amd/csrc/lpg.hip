// Local planar guidance (LPG) forward for gfx950.
//
// Restates local_planar_guidance.forward (reference pytorch/bts.py:149-173) as ONE pass:
// the reference materialises two repeat_interleave copies, CPU-built u/v grids (copied
// H2D every call), ~20 elementwise passes and a min-reduction; here each thread produces
// four adjacent output columns (one 16-byte store) straight from the 1/k-resolution plane
// coefficients.  The op is HBM-write bound (4 B written per ~0.06-1 B read).
//
// Bit-exactness: den = (n1*u + n2*v) + n3 is evaluated with separately rounded mul/add (FMA
// contraction disabled for this file), the clamp follows bts.py:168-171 and the divisions are
// IEEE (hipcc's default correctly-rounded fp32 divide), so on identical inputs the result
// equals the torch CPU reference bit for bit.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "common.h"
#include "lpg_math.h"

// hipcc defaults to -ffp-contract=fast, which would fuse n1*u + ... into FMAs and break bit-parity
// with the reference's separately rounded mul/add kernels.  (HIP's __fmul_rn/__fadd_rn do NOT help:
// they are header inlines compiled with the contract flag.)  This file is also built with
// -ffp-contract=off (Makefile).
#pragma clang fp contract(off)

namespace {

__device__ __forceinline__ float lpg_den(float n1, float n2, float n3, float u, float v) {
    return (n1 * u + n2 * v) + n3;   // contraction is off for this file
}

// block-min of |den| then ONE atomic per block: same-address atomics serialise in L2 (~12 ns each on
// MI355X), so blocks are fat (16 waves) and the grid is capped at 512.
// NaN: the reference's `torch.abs(divided).min()` (bts.py:167) returns NaN as soon as one denominator is NaN, and
// bts_main.py:484-486 logs abs_min precisely to hunt NaNs -- so a NaN must win the reduction.  fminf drops NaNs;
// the reduction therefore runs on int32 KEYS: a non-negative float's bit pattern as a signed int keeps the float
// order, and "saw a NaN" is the negative key 0xffc00000 (a quiet NaN with the sign bit set), which is below every
// real key and still reads back as NaN from the caller's float scalar.
constexpr int LPG_ROWS = 16;                       // block = 64 x 16 threads
__device__ __forceinline__ void publish_abs_min(float m, bool saw_nan, int* abs_min_key) {
    __shared__ int wave_min[LPG_ROWS];
    int k = lpg_min_key(m, saw_nan);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) k = min(k, __shfl_xor(k, off, 64));
    const int tid = threadIdx.y * blockDim.x + threadIdx.x;
    if ((tid & 63) == 0) wave_min[tid >> 6] = k;
    __syncthreads();
    if (tid < 64) {
        k = tid < LPG_ROWS ? wave_min[tid] : 0x7f800000;
#pragma unroll
        for (int off = 8; off > 0; off >>= 1) k = min(k, __shfl_xor(k, off, 64));
        if (tid == 0) atomicMin(abs_min_key, k);
    }
}

// K = upratio (1,2,4,8).  PLANAR: input [B,4,h,w]; else cell-interleaved [B*h*w,4].
// FUSED: optional normalize, divide by max_depth, strided nearest-downsample side output.
template <int K, int V, bool PLANAR, bool FUSED>
__global__ __launch_bounds__(64 * LPG_ROWS) void lpg_fwd_kernel(const float* __restrict__ plane, int B, int h, int w,
                                                      int normalize, float max_depth,
                                                      float* __restrict__ out, float* __restrict__ ds_out,
                                                      int ds_factor, long ds_pix_stride,
                                                      int* __restrict__ abs_min_bits) {
    const int W = w * K, H = h * K;
    const int W4 = W / V;                        // V = 4 when W % 4 == 0 (16-byte stores), else 1
    const int nrows = B * H;
    float amin = __uint_as_float(0x7f800000u);
    bool saw_nan = false;
    // block = 64 x 16: a wave walks one output row in 64-wide strides (coalesced 1 KiB stores), so the
    // (b, r) decode costs one division per row instead of two per element.
    [[maybe_unused]] const float inv_scale_den = max_depth;         // FUSED: depth/max_depth == n4 / (den * max_depth)
    for (int row = blockIdx.x * LPG_ROWS + threadIdx.y; row < nrows; row += gridDim.x * LPG_ROWS)
    for (int c4 = threadIdx.x; c4 < W4; c4 += 64) {
        const int r = row % H;
        const int b = row / H;
        const int cr = r / K;
        const float v = ((float)(r % K) - (float)(K - 1) * 0.5f) / (float)K;   // bts.py:160-161
        float res[V];
        constexpr int CELLS = (K >= V) ? 1 : (V / K);    // cells covered by V columns
        constexpr int PER = V / CELLS;
#pragma unroll
        for (int ci = 0; ci < CELLS; ++ci) {
            const int c0 = c4 * V + ci * PER;
            const int cc = c0 / K;
            float n1, n2, n3, n4;
            if (PLANAR) {
                const long hw = (long)h * w;
                const float* p = plane + ((long)b * 4) * hw + (long)cr * w + cc;
                n1 = p[0]; n2 = p[hw]; n3 = p[2 * hw]; n4 = p[3 * hw];
            } else {
                const float4 q = *reinterpret_cast<const float4*>(plane + (((long)b * h + cr) * w + cc) * 4);
                n1 = q.x; n2 = q.y; n3 = q.z; n4 = q.w;
            }
            if (FUSED && normalize) {                       // F.normalize(p=2, dim=1, eps=1e-12), bts.py:251
                const float nn = sqrtf((n1 * n1 + n2 * n2) + n3 * n3);
                const float dn = fmaxf(nn, 1e-12f);
                n1 = n1 / dn; n2 = n2 / dn; n3 = n3 / dn;
            }
#pragma unroll
            for (int i = 0; i < PER; ++i) {
                const int c = c0 + i;
                const float u = ((float)(c % K) - (float)(K - 1) * 0.5f) / (float)K;  // bts.py:157-158
                float d = lpg_den(n1, n2, n3, u, v);                                   // bts.py:166
                amin = fminf(amin, fabsf(d));                                          // bts.py:167
                saw_nan |= d != d;
                d = lpg_clamp(d);
                float y = n4 / d;                                                     // bts.py:173
                if (FUSED) y = y / max_depth;                                         // bts.py:255
                res[ci * PER + i] = y;
            }
        }
        if constexpr (V == 4)
            *reinterpret_cast<float4*>(out + ((long)b * H + r) * W + (long)c4 * 4) =
                make_float4(res[0], res[1], res[2], res[3]);
        else
            out[((long)b * H + r) * W + c4] = res[0];
        if (FUSED && ds_out != nullptr && (r % ds_factor) == 0) {   // nearest, scale 1/ds_factor == [::f, ::f]
            const int Hd = H / ds_factor, Wd = W / ds_factor;
            const long rowbase = ((long)b * Hd + r / ds_factor) * Wd;
#pragma unroll
            for (int i = 0; i < V; ++i) {
                const int c = c4 * V + i;
                if ((c % ds_factor) == 0) ds_out[(rowbase + c / ds_factor) * ds_pix_stride] = res[i];
            }
        }
    }
    if (abs_min_bits != nullptr) publish_abs_min(amin, saw_nan, abs_min_bits);
}

// Lean kernel for the decoder pipeline's case (planes already normalised by the reduction epilogue, W % 4 == 0,
// K in {2,4,8}): ~45 VALU per 16-byte store instead of ~200 in the general kernel (whose runtime `normalize`
// branch carries IEEE divisions and whose indices are 64-bit) -- the general kernel was VALU-bound at 2.7 TB/s
// where a plain fill reaches 6.9 TB/s on this device.  Not the bit-exact path: den uses FMAs and one hardware
// reciprocal replaces the two IEEE divisions (<= ~1.5 ulp; the module-level op stays exact).
template <int K>
__global__ __launch_bounds__(64 * LPG_ROWS) void lpg_fused_lean_kernel(const float4* __restrict__ plane4, int nrows, int H,
                                                                       int h, int w, float max_depth,
                                                                       float* __restrict__ out, float* __restrict__ ds_out,
                                                                       int ds_factor, int ds_pix_stride,
                                                                       int* __restrict__ abs_min_bits) {
    const int W = w * K, W4 = W >> 2;
    constexpr int CELLS = K >= 4 ? 1 : 2;
    float amin = __uint_as_float(0x7f800000u);
    bool saw_nan = false;
    for (int row = blockIdx.x * LPG_ROWS + threadIdx.y; row < nrows; row += gridDim.x * LPG_ROWS) {
        const int b = row / H, r = row - b * H;
        const int cr = r / K;
        const float4* prow = plane4 + (size_t)(b * h + cr) * w;
        float* orow = out + (size_t)row * W;
        const bool ds_row = ds_out != nullptr && (r % ds_factor) == 0;
        float* dsrow = ds_row ? ds_out + (size_t)((b * (H / ds_factor) + r / ds_factor) * (W / ds_factor)) * ds_pix_stride : nullptr;
        for (int c4 = threadIdx.x; c4 < W4; c4 += 64) {
            const int c0 = c4 * 4;
            float res[4];
#pragma unroll
            for (int ci = 0; ci < CELLS; ++ci) {
                const int cb = c0 + ci * (4 / CELLS);
                const float4 q = prow[cb / K];
                float part[4 / CELLS];
                lpg_cell_outputs<K, 4 / CELLS>(q.x, q.y, q.z, q.w, r - cr * K, cb % K, max_depth, part, amin, saw_nan);
#pragma unroll
                for (int i = 0; i < 4 / CELLS; ++i) res[ci * (4 / CELLS) + i] = part[i];
            }
            *reinterpret_cast<float4*>(orow + c0) = make_float4(res[0], res[1], res[2], res[3]);
            if (ds_row) {                                                   // nearest [::f, ::f]   (bts.py:256,270)
                if (ds_factor == 4) dsrow[(size_t)c4 * ds_pix_stride] = res[0];
                else if (ds_factor == 2) { dsrow[(size_t)(2 * c4) * ds_pix_stride] = res[0]; dsrow[(size_t)(2 * c4 + 1) * ds_pix_stride] = res[2]; }
                else {
#pragma unroll
                    for (int i = 0; i < 4; ++i) dsrow[(size_t)(c0 + i) * ds_pix_stride] = res[i];
                }
            }
        }
    }
    if (abs_min_bits != nullptr) publish_abs_min(amin, saw_nan, abs_min_bits);
}

template <bool PLANAR, bool FUSED>
int launch_lpg(const float* plane, int B, int h, int w, int k, int normalize, float max_depth, float* out,
               float* ds_out, int ds_factor, long ds_pix_stride, float* abs_min, hipStream_t s) {
    if (!plane || !out || B <= 0 || h <= 0 || w <= 0) return BTS_ERR_INVALID;
    if (k != 1 && k != 2 && k != 4 && k != 8) return BTS_ERR_UNSUPPORTED;
    if (ds_out && (ds_factor <= 0 || (h * k) % ds_factor || (w * k) % ds_factor)) return BTS_ERR_INVALID;
    if (FUSED && !(max_depth > 0.f)) return BTS_ERR_INVALID;
    int* bits = reinterpret_cast<int*>(abs_min);
    if (bits) {
        hipError_t e = hipMemsetD32Async((hipDeviceptr_t)bits, 0x7f800000, 1, s);
        if (e != hipSuccess) return (int)e;
    }
    const bool vec4 = (((long)w * k) % 4) == 0;
    if ((long)B * h * k > 0x7fffffffL) return BTS_ERR_UNSUPPORTED;
    long blocks = ((long)B * h * k + LPG_ROWS - 1) / LPG_ROWS;
    if (blocks > 512) blocks = 512;                  // <= 512 same-address atomics (see publish_abs_min); 2 blocks/CU
    dim3 grid((unsigned)blocks), block(64, LPG_ROWS);
    if (FUSED && !PLANAR && !normalize && vec4 && k >= 2 && (ds_out == nullptr || ds_factor == 1 || ds_factor == 2 || ds_factor == 4) &&
        ds_pix_stride <= 0x7fffffffL && (long)B * h * k * w * k <= 0x7fffffffL) {
        const float4* p4 = reinterpret_cast<const float4*>(plane);
        const int nrows = B * h * k, H = h * k, dss = (int)ds_pix_stride;
        switch (k) {
            case 2: hipLaunchKernelGGL(lpg_fused_lean_kernel<2>, grid, block, 0, s, p4, nrows, H, h, w, max_depth, out, ds_out, ds_factor, dss, bits); break;
            case 4: hipLaunchKernelGGL(lpg_fused_lean_kernel<4>, grid, block, 0, s, p4, nrows, H, h, w, max_depth, out, ds_out, ds_factor, dss, bits); break;
            default: hipLaunchKernelGGL(lpg_fused_lean_kernel<8>, grid, block, 0, s, p4, nrows, H, h, w, max_depth, out, ds_out, ds_factor, dss, bits); break;
        }
        return (int)hipGetLastError();
    }
#define LPG_LAUNCH(KK)                                                                                          \
    if (vec4)                                                                                                   \
        hipLaunchKernelGGL((lpg_fwd_kernel<KK, 4, PLANAR, FUSED>), grid, block, 0, s, plane, B, h, w, normalize, \
                           max_depth, out, ds_out, ds_factor, ds_pix_stride, bits);                             \
    else                                                                                                        \
        hipLaunchKernelGGL((lpg_fwd_kernel<KK, 1, PLANAR, FUSED>), grid, block, 0, s, plane, B, h, w, normalize, \
                           max_depth, out, ds_out, ds_factor, ds_pix_stride, bits)
    switch (k) {
        case 1: LPG_LAUNCH(1); break;
        case 2: LPG_LAUNCH(2); break;
        case 4: LPG_LAUNCH(4); break;
        default: LPG_LAUNCH(8); break;
    }
#undef LPG_LAUNCH
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------ backward
// One thread per low-resolution cell accumulates the k*k output gradients it produced (no atomics, the
// mapping of the reference's LocalPlanarGuidanceGradFunctor, local_planar_guidance.cu:95-150).  The derivative
// is the TRUE one of bts.py:149-173 (what autograd gives): d/dn4 = 1/den_c, d/dn{1,2,3} = -n4*{u,v,1}/den^2 on
// un-clamped pixels and 0 where the +-1e-3 clamp replaced den by a constant -- the TF op's gradient omits the
// n4 factor (cu:143-145) and has no clamp.
template <int K>
__global__ __launch_bounds__(256) void lpg_bwd_kernel(const float* __restrict__ plane, const float* __restrict__ gout,
                                                      int B, int h, int w, float* __restrict__ gplane) {
    const long ncell = (long)B * h * w;
    const int W = w * K, H = h * K;
    const long hw = (long)h * w;
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < ncell; t += (long)gridDim.x * blockDim.x) {
        const int cx = (int)(t % w);
        const long r = t / w;
        const int cy = (int)(r % h);
        const int b = (int)(r / h);
        const float* p = plane + ((long)b * 4) * hw + (long)cy * w + cx;
        const float n1 = p[0], n2 = p[hw], n3 = p[2 * hw], n4 = p[3 * hw];
        float g1 = 0.f, g2 = 0.f, g3 = 0.f, g4 = 0.f;
#pragma unroll
        for (int dy = 0; dy < K; ++dy) {
            const float v = ((float)dy - (float)(K - 1) * 0.5f) / (float)K;
            const float* grow = gout + ((long)b * H + cy * K + dy) * W + (long)cx * K;
#pragma unroll
            for (int dx = 0; dx < K; ++dx) {
                const float u = ((float)dx - (float)(K - 1) * 0.5f) / (float)K;
                const float g = grow[dx];
                const float den = lpg_den(n1, n2, n3, u, v);
                const float dc = lpg_clamp(den);
                const bool clamped = dc != den;
                g4 += g / dc;
                const float gd = clamped ? 0.f : -g * n4 / (den * den);     // d(n4/den)/d(den)
                g1 += gd * u; g2 += gd * v; g3 += gd;
            }
        }
        float* q = gplane + ((long)b * 4) * hw + (long)cy * w + cx;
        q[0] = g1; q[hw] = g2; q[2 * hw] = g3; q[3 * hw] = g4;
    }
}

}  // namespace

extern "C" int bts_lpg_bwd_f32(const float* plane_eq, const float* grad_depth, int B, int h, int w, int upratio,
                               float* grad_plane_eq, bts_stream_t stream) {
    if (!plane_eq || !grad_depth || !grad_plane_eq || B <= 0 || h <= 0 || w <= 0) return BTS_ERR_INVALID;
    const long ncell = (long)B * h * w;
    long blocks = (ncell + 255) / 256;
    if (blocks > 256L * 16) blocks = 256L * 16;
    hipStream_t s = (hipStream_t)stream;
    dim3 grid((unsigned)blocks), block(256);
    switch (upratio) {
        case 1: hipLaunchKernelGGL(lpg_bwd_kernel<1>, grid, block, 0, s, plane_eq, grad_depth, B, h, w, grad_plane_eq); break;
        case 2: hipLaunchKernelGGL(lpg_bwd_kernel<2>, grid, block, 0, s, plane_eq, grad_depth, B, h, w, grad_plane_eq); break;
        case 4: hipLaunchKernelGGL(lpg_bwd_kernel<4>, grid, block, 0, s, plane_eq, grad_depth, B, h, w, grad_plane_eq); break;
        case 8: hipLaunchKernelGGL(lpg_bwd_kernel<8>, grid, block, 0, s, plane_eq, grad_depth, B, h, w, grad_plane_eq); break;
        default: return BTS_ERR_UNSUPPORTED;
    }
    return (int)hipGetLastError();
}

namespace {
}  // namespace

extern "C" int bts_lpg_fwd_f32(const float* plane_eq, int B, int h, int w, int upratio, float* depth,
                               float* abs_min, bts_stream_t stream) {
    return launch_lpg<true, false>(plane_eq, B, h, w, upratio, 0, 1.f, depth, nullptr, 1, 0, abs_min,
                                   (hipStream_t)stream);
}

extern "C" int bts_lpg_fused_fwd_f32(const float* plane4, int B, int h, int w, int upratio, int normalize,
                                     float max_depth, float* depth_scaled, float* ds_out, int ds_factor,
                                     long ds_pix_stride, float* abs_min, bts_stream_t stream) {
    if (((uintptr_t)plane4 & 15) != 0) return BTS_ERR_INVALID;
    return launch_lpg<false, true>(plane4, B, h, w, upratio, normalize, max_depth, depth_scaled, ds_out,
                                   ds_factor, ds_pix_stride, abs_min, (hipStream_t)stream);
}
