// Convolution weight gradient for the BTS training step (reference: autograd of every nn.Conv2d on the path,
// pytorch/bts.py:70-77, 87-93, 108-119, 180-221; training caller bts_main.py:476-500) on gfx950.
//
//   dw[co][tap][ci] = sum over output pixels p of  dy[p][co] * x[q(p,tap)][ci]
//
// As a GEMM:  C[M = c_out][N = taps*c_in] = A^T[K = pixels][M] (dy, NHWC)  x  B[K = pixels][N] (gathered x, NHWC).
// Both operands are K-major with their M/N index contiguous in memory -- exactly how NHWC stores them -- so a
// tile row (one pixel) is loaded with coalesced 16-byte reads and lands in LDS as [k][m] / [k][n]; an MFMA operand
// is then one conflict-free ds_read_b32 (lane l reads row 2*kk + (l>>5), column base + (l&31)).
// K is the long axis (B*H*W up to 6.8 M pixels) and the output is small, so the launch always splits K over
// gridDim.y; partials go to a workspace and are summed in a fixed order (deterministic, no atomics).
#include "common.h"
#include <stdint.h>
#include <stdlib.h>

namespace {

constexpr int WBK = 32;   // pixels per K-step

struct WgradArgs {
    const float* __restrict__ x;
    const float* __restrict__ dy;
    float* __restrict__ out;     // dw (ksplit == 1) or the partial-sum workspace [ksplit][c_out][N]
    long x_pix_stride, dy_pix_stride;
    int c_in, c_out, N;          // N = taps * c_in
    int B, h_in, w_in, Hs, Ws, ups, H, W, ksize, dil, stride, pad;
    unsigned M;                  // B*H*W output pixels
    unsigned pix_per_split;      // multiple of WBK
    int n_ntiles;
};

// One workgroup = 4 waves arranged WM x WN over a BM x BN tile of dw; each wave owns (BM/WM) x (BN/WN).
template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(const WgradArgs a) {
    static_assert(WM * WN == 4, "4 waves");
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    static_assert(TM >= 1 && TN >= 1, "wave tile must be at least 32x32");
    constexpr int ACOLS = BM / 4, BCOLS = BN / 4;          // float4 columns per tile row
    constexpr int AROWS = 256 / ACOLS, BROWS = 256 / BCOLS;  // rows covered per pass
    constexpr int PA = WBK / AROWS, PB = WBK / BROWS;
    static_assert(PA >= 1 && PB >= 1, "tile too wide for 256 loader threads");

    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* As = lds;                       // [2][WBK][BM]
    float* Bs = lds + 2 * WBK * BM;        // [2][WBK][BN]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm0 = (wave / WN) * (BM / WM), wn0 = (wave % WN) * (BN / WN);
    const int l31 = lane & 31, khalf = lane >> 5;
    const int tile_n = blockIdx.x % a.n_ntiles, tile_m = blockIdx.x / a.n_ntiles;
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    // ---- loader roles (fixed for the whole K walk) ----
    const int a_col = (tid % ACOLS) * 4, a_row = tid / ACOLS;
    const int b_col = (tid % BCOLS) * 4, b_row = tid / BCOLS;
    const bool a_ok = m0 + a_col < a.c_out;                 // c_out % 4 == 0: a float4 is all real or all padding
    const int a_m = a_ok ? m0 + a_col : 0;
    const int n4 = n0 + b_col;
    const bool b_ok = n4 < a.N;
    const int nn = b_ok ? n4 : 0;
    const int tap = nn / a.c_in, ci = nn - tap * a.c_in;    // c_in % 4 == 0: a float4 never straddles a tap
    const int ky = tap / a.ksize, kx = tap - ky * a.ksize;
    const int off_y = ky * a.dil - a.pad, off_x = kx * a.dil - a.pad;

    const unsigned k_begin = blockIdx.y * a.pix_per_split;
    const unsigned k_end = min(a.M, k_begin + a.pix_per_split);
    const int n_it = k_begin < k_end ? (int)((k_end - k_begin + WBK - 1) / WBK) : 0;
    const unsigned HW = (unsigned)a.H * (unsigned)a.W;

    f32x4 ra[PA], rb[PB];
    auto issue = [&](int it) __attribute__((always_inline)) {
        const unsigned base = k_begin + (unsigned)it * WBK;
#pragma unroll
        for (int p = 0; p < PA; ++p) {
            const unsigned pix = base + a_row + p * AROWS;
            const bool ok = a_ok && pix < k_end;
            const float* src = a.dy + (size_t)(ok ? pix : 0u) * a.dy_pix_stride + a_m;
            f32x4 v = *reinterpret_cast<const f32x4*>(src);
            ra[p] = ok ? v : (f32x4)(0.f);
        }
#pragma unroll
        for (int p = 0; p < PB; ++p) {
            const unsigned pix = base + b_row + p * BROWS;
            const unsigned pp = pix < k_end ? pix : k_begin;
            const unsigned b = pp / HW, rem = pp - b * HW;
            const unsigned y = rem / (unsigned)a.W, xq = rem - y * (unsigned)a.W;
            const int iy = (int)y * a.stride + off_y, ix = (int)xq * a.stride + off_x;
            const bool ok = b_ok && pix < k_end && iy >= 0 && iy < a.Hs && ix >= 0 && ix < a.Ws;
            const int sy = ok ? (iy >> a.ups) : 0, sx = ok ? (ix >> a.ups) : 0;
            const float* src = a.x + ((size_t)(b * a.h_in + sy) * a.w_in + sx) * a.x_pix_stride + ci;
            f32x4 v = *reinterpret_cast<const f32x4*>(src);
            rb[p] = ok ? v : (f32x4)(0.f);
        }
    };
    auto stage = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int p = 0; p < PA; ++p)
            *reinterpret_cast<f32x4*>(&As[(buf * WBK + a_row + p * AROWS) * BM + a_col]) = ra[p];
#pragma unroll
        for (int p = 0; p < PB; ++p)
            *reinterpret_cast<f32x4*>(&Bs[(buf * WBK + b_row + p * BROWS) * BN + b_col]) = rb[p];
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (f32x16)(0.f);

    if (n_it > 0) {
        issue(0);
        stage(0);
    }
    __syncthreads();
    for (int it = 0; it < n_it; ++it) {
        const int cur = it & 1;
        if (it + 1 < n_it) issue(it + 1);
        const float* Ac = As + cur * WBK * BM + wm0 + l31;
        const float* Bc = Bs + cur * WBK * BN + wn0 + l31;
#pragma unroll
        for (int kk = 0; kk < WBK / 2; ++kk) {
            float av[TM], bv[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) av[i] = Ac[(2 * kk + khalf) * BM + 32 * i];
#pragma unroll
            for (int j = 0; j < TN; ++j) bv[j] = Bc[(2 * kk + khalf) * BN + 32 * j];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = mfma32x2(av[i], bv[j], acc[i][j]);
        }
        if (it + 1 < n_it) stage(cur ^ 1);
        __syncthreads();
    }

    // ---- store: D register r of lane l is row (r&3) + 8*(r>>2) + 4*(l>>5), column l&31 ----
    float* out = a.out + (size_t)blockIdx.y * a.c_out * a.N;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + wn0 + 32 * j + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm0 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * khalf;
                if (m < a.c_out && n < a.N) out[(size_t)m * a.N + n] = acc[i][j][r];
            }
        }
}

__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw,
                                                           long count4, int ksplit) {
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < count4; t += (long)gridDim.x * blockDim.x) {
        f32x4 v = reinterpret_cast<const f32x4*>(ws)[t];
        for (int s = 1; s < ksplit; ++s) v += reinterpret_cast<const f32x4*>(ws)[(size_t)s * count4 + t];
        reinterpret_cast<f32x4*>(dw)[t] = v;
    }
}

template <int BM, int BN, int WM, int WN>
int launch_wgrad(WgradArgs a, float* dw, float* ws, long ws_floats, hipStream_t s) {
    const int m_tiles = (a.c_out + BM - 1) / BM;
    a.n_ntiles = (a.N + BN - 1) / BN;
    const long tiles = (long)m_tiles * a.n_ntiles;
    const long per = (long)a.c_out * a.N;
    // Split the pixel axis until ~4 workgroups per CU are in flight (256 CUs), keeping >= 8 K-steps per split and
    // the partials within the caller's workspace.  A function of the geometry only, so results are reproducible.
    static const long target = getenv("BTS_WGRAD_TARGET") ? atol(getenv("BTS_WGRAD_TARGET")) : 1024;
    long split = (target + tiles - 1) / tiles;
    const long max_by_k = ((long)a.M + 8 * WBK - 1) / (8 * WBK);
    if (split > max_by_k) split = max_by_k;
    if (split > 1024) split = 1024;
    if (ws == nullptr || ws_floats < 2 * per) split = 1;
    else if (split * per > ws_floats) split = ws_floats / per;
    if (split < 1) split = 1;
    long pps = (((long)a.M + split - 1) / split + WBK - 1) / WBK * WBK;
    split = ((long)a.M + pps - 1) / pps;
    a.pix_per_split = (unsigned)pps;
    a.out = split > 1 ? ws : dw;
    const size_t lds_bytes = (size_t)2 * WBK * (BM + BN) * sizeof(float);
    auto kern = conv_wgrad_kernel<BM, BN, WM, WN>;
    static bool attr_done = false;     // per instantiation
    if (!attr_done) {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) != hipSuccess)
            return (int)hipGetLastError();
        attr_done = true;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)tiles, (unsigned)split), dim3(256), lds_bytes, s, a);
    if (split > 1) {
        const long count4 = per / 4;      // N % 4 == 0
        long blocks = (count4 + 255) / 256;
        if (blocks > 2048) blocks = 2048;
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, s, ws, dw, count4, (int)split);
    }
    return (int)hipGetLastError();
}

}  // namespace

extern "C" int bts_conv_wgrad_f32(const bts_conv_wgrad_desc* d, bts_stream_t stream) {
    if (!d || !d->x || !d->dy || !d->dw) return BTS_ERR_INVALID;
    if (d->B <= 0 || d->h_in <= 0 || d->w_in <= 0 || d->c_in <= 0 || d->c_out <= 0) return BTS_ERR_INVALID;
    if (d->up != 1 && d->up != 2) return BTS_ERR_UNSUPPORTED;
    if (d->ksize < 1 || d->ksize > 7 || !(d->ksize & 1)) return BTS_ERR_UNSUPPORTED;
    if (d->dil < 1 || d->stride < 1 || d->pad < 0) return BTS_ERR_INVALID;
    if (d->up == 2 && d->stride != 1) return BTS_ERR_UNSUPPORTED;
    if ((d->c_in & 3) || (d->c_out & 3)) return BTS_ERR_INVALID;
    if ((d->x_pix_stride & 3) || d->x_pix_stride < d->c_in) return BTS_ERR_INVALID;
    if ((d->dy_pix_stride & 3) || d->dy_pix_stride < d->c_out) return BTS_ERR_INVALID;
    if (((uintptr_t)d->x & 15) || ((uintptr_t)d->dy & 15) || ((uintptr_t)d->dw & 15)) return BTS_ERR_INVALID;
    if (d->ws && (((uintptr_t)d->ws & 15) || d->ws_floats < 0)) return BTS_ERR_INVALID;
    WgradArgs a;
    a.x = d->x; a.dy = d->dy; a.out = d->dw;
    a.x_pix_stride = d->x_pix_stride; a.dy_pix_stride = d->dy_pix_stride;
    a.c_in = d->c_in; a.c_out = d->c_out;
    a.B = d->B; a.h_in = d->h_in; a.w_in = d->w_in; a.ups = d->up == 2 ? 1 : 0;
    a.Hs = d->h_in * d->up; a.Ws = d->w_in * d->up;
    a.ksize = d->ksize; a.dil = d->dil; a.stride = d->stride; a.pad = d->pad;
    a.H = (a.Hs + 2 * d->pad - d->dil * (d->ksize - 1) - 1) / d->stride + 1;
    a.W = (a.Ws + 2 * d->pad - d->dil * (d->ksize - 1) - 1) / d->stride + 1;
    if (a.H <= 0 || a.W <= 0) return BTS_ERR_INVALID;
    const double mpix = (double)d->B * a.H * a.W;
    if (mpix >= 4294967296.0 - 65536.0) return BTS_ERR_UNSUPPORTED;
    const double nflat = (double)d->ksize * d->ksize * d->c_in;
    if (nflat >= 2147483648.0) return BTS_ERR_UNSUPPORTED;
    a.M = (unsigned)mpix;
    a.N = d->ksize * d->ksize * d->c_in;
    a.pix_per_split = 0; a.n_ntiles = 0;
    hipStream_t s = (hipStream_t)stream;
    if (d->c_out > 64) return launch_wgrad<128, 128, 2, 2>(a, d->dw, d->ws, d->ws_floats, s);
    if (d->c_out > 32) return launch_wgrad<64, 128, 1, 4>(a, d->dw, d->ws, d->ws_floats, s);
    return launch_wgrad<32, 128, 1, 4>(a, d->dw, d->ws, d->ws_floats, s);
}
