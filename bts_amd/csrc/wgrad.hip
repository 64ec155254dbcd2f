// Convolution weight gradient for the BTS training step (reference: autograd of every nn.Conv2d on the path,
// pytorch/bts.py:70-77, 87-93, 108-119, 180-221; training caller bts_main.py:476-500) on gfx950.
//
//   dw[co][tap][ci] = sum over output pixels p of  dy[p][co] * x[q(p,tap)][ci]
//
// As a GEMM:  C[M = c_out][N = taps*c_in] = A^T[K = pixels][M] (dy, NHWC)  x  B[K = pixels][N] (gathered x, NHWC).
// Both operands are K-major with their M/N index contiguous in memory -- exactly how NHWC stores them -- so a
// tile row (one pixel) is loaded with coalesced 16-byte reads and lands in LDS as [k][m] / [k][n]; an MFMA operand
// is then one conflict-free ds_read_b32 (lane l reads row 2*kk + (l>>5), column base + (l&31)).
// K is the long axis (B*H*W up to 6.8 M pixels) and the output is small, so the launch splits K over gridDim.y;
// partials go to a workspace and are summed in a fixed order (deterministic, no atomics); plan_wgrad picks the tile
// (128x128 / 64x128 / 32x128 / 64x64) and the split together.
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
    const float* pre_scale; const float* pre_shift; int pre_relu;   // optional per-input-channel affine (+ReLU) applied to x
                                 // in the gather: the forward convolution saw relu(x*scale + shift) (a folded norm layer)
    int n_bundles;               // grouped convolution as channel bundles (blockIdx.z): bundle j uses input channels
                                 // [j*c_in, ..), gradient channels [j*c_out, ..) and writes dense block j of [c_out][N]
};

// One workgroup = 4 waves arranged WM x WN over a BM x BN tile of dw; each wave owns (BM/WM) x (BN/WN).
template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(256, 2) void conv_wgrad_kernel(const WgradArgs a) {
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
    const float* __restrict__ xg = a.x + (size_t)blockIdx.z * a.c_in;      // this bundle's channel slice
    const float* __restrict__ dyg = a.dy + (size_t)blockIdx.z * a.c_out;

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

    // Two staging register sets: the loads of tile it+2 are issued at the top of step it and consumed (written to LDS)
    // at the bottom of step it+1, so every global load has two full MFMA steps to land.
    f32x4 ra0[PA], rb0[PB], ra1[PA], rb1[PB];
    f32x4 psc = {1.f, 1.f, 1.f, 1.f}, psh = {0.f, 0.f, 0.f, 0.f};          // this thread's four input channels never change
    const bool has_pre = a.pre_scale != nullptr;
    if (has_pre) {
        const int cpre = blockIdx.z * a.c_in + ci;
        psc = *reinterpret_cast<const f32x4*>(a.pre_scale + cpre);
        psh = *reinterpret_cast<const f32x4*>(a.pre_shift + cpre);
    }
    int pb[PB], py[PB], px[PB];             // (frame, row, column) of each gathered row's output pixel at the next step
#pragma unroll
    for (int p = 0; p < PB; ++p) {
        const unsigned pix = k_begin + b_row + p * BROWS;
        const unsigned b = pix / HW, rem = pix - b * HW;
        pb[p] = (int)b;
        py[p] = (int)(rem / (unsigned)a.W);
        px[p] = (int)(rem - (unsigned)py[p] * (unsigned)a.W);
    }
    auto issue = [&](int it, f32x4 (&ra)[PA], f32x4 (&rb)[PB]) __attribute__((always_inline)) {
        const unsigned base = k_begin + (unsigned)it * WBK;
#pragma unroll
        for (int p = 0; p < PA; ++p) {
            const unsigned pix = base + a_row + p * AROWS;
            const bool ok = a_ok && pix < k_end;
            const float* src = dyg + (size_t)(ok ? pix : 0u) * a.dy_pix_stride + a_m;
            f32x4 v = *reinterpret_cast<const f32x4*>(src);
            ra[p] = ok ? v : (f32x4)(0.f);
        }
#pragma unroll
        for (int p = 0; p < PB; ++p) {
            const unsigned pix = base + b_row + p * BROWS;
            const int iy = py[p] * a.stride + off_y, ix = px[p] * a.stride + off_x;
            const bool ok = b_ok && pix < k_end && iy >= 0 && iy < a.Hs && ix >= 0 && ix < a.Ws;
            const int sb = ok ? pb[p] : 0, sy = ok ? (iy >> a.ups) : 0, sx = ok ? (ix >> a.ups) : 0;
            const float* src = xg + ((size_t)(sb * a.h_in + sy) * a.w_in + sx) * a.x_pix_stride + ci;
            f32x4 v = *reinterpret_cast<const f32x4*>(src);
            if (has_pre) {
                v = v * psc + psh;
                if (a.pre_relu) v = __builtin_elementwise_max(v, (f32x4)(0.f));
            }
            rb[p] = ok ? v : (f32x4)(0.f);                               // zero padding AFTER the prologue
            // advance this row's output pixel by one K-step without dividing (issue() is called for it = 0, 1, 2, ...)
            px[p] += WBK;
            while (px[p] >= a.W) { px[p] -= a.W; ++py[p]; }
            while (py[p] >= a.H) { py[p] -= a.H; ++pb[p]; }
        }
    };
    auto stage = [&](int buf, const f32x4 (&ra)[PA], const f32x4 (&rb)[PB]) __attribute__((always_inline)) {
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

    auto mfma_step = [&](int cur) __attribute__((always_inline)) {
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
    };

    if (n_it > 0) {
        issue(0, ra0, rb0);
        if (n_it > 1) issue(1, ra1, rb1);
        stage(0, ra0, rb0);
    }
    __syncthreads();
    for (int it = 0; it < n_it; it += 2) {
        // even step: LDS[0] holds tile it, set 1 holds tile it+1, set 0 is free
        if (it + 2 < n_it) issue(it + 2, ra0, rb0);
        mfma_step(0);
        if (it + 1 < n_it) stage(1, ra1, rb1);
        __syncthreads();
        if (it + 1 >= n_it) break;
        // odd step: LDS[1] holds tile it+1, set 0 holds tile it+2, set 1 is free
        if (it + 3 < n_it) issue(it + 3, ra1, rb1);
        mfma_step(1);
        if (it + 2 < n_it) stage(0, ra0, rb0);
        __syncthreads();
    }

    // ---- store: D register r of lane l is row (r&3) + 8*(r>>2) + 4*(l>>5), column l&31 ----
    float* out = a.out + ((size_t)blockIdx.y * a.n_bundles + blockIdx.z) * a.c_out * a.N;
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

// Sum the ksplit partial tiles in a fixed order.  The output is small and the split count can be in the hundreds, so
// the walk over splits is itself spread over 16 lanes per element (a serial walk is pure load latency): a block owns
// 16 consecutive float4 elements, lane j adds splits j, j+16, ... (four independent loads in flight), then lane 0 adds
// the 16 lane sums in order.
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw,
                                                           long count4, int ksplit) {
    const int el = threadIdx.x & 15, lane = threadIdx.x >> 4;
    const long t = (long)blockIdx.x * 16 + el;
    const long tt = t < count4 ? t : count4 - 1;
    const f32x4* p = reinterpret_cast<const f32x4*>(ws) + tt;
    f32x4 v = (f32x4)(0.f);
    int s = lane;
    for (; s + 48 < ksplit; s += 64) {
        const f32x4 a0 = p[(size_t)s * count4], a1 = p[(size_t)(s + 16) * count4];
        const f32x4 a2 = p[(size_t)(s + 32) * count4], a3 = p[(size_t)(s + 48) * count4];
        v += (a0 + a1) + (a2 + a3);
    }
    for (; s < ksplit; s += 16) v += p[(size_t)s * count4];
    __shared__ f32x4 red[16][16];
    red[lane][el] = v;
    __syncthreads();
    if (lane == 0 && t < count4) {
        f32x4 r = red[0][el];
#pragma unroll
        for (int j = 1; j < 16; ++j) r += red[j][el];
        reinterpret_cast<f32x4*>(dw)[t] = r;
    }
}

template <int BM, int BN, int WM, int WN>
int launch_wgrad(WgradArgs a, long split, float* dw, float* ws, hipStream_t s) {
    const int m_tiles = (a.c_out + BM - 1) / BM;
    a.n_ntiles = (a.N + BN - 1) / BN;
    const long tiles = (long)m_tiles * a.n_ntiles;
    const long per = (long)a.c_out * a.N * a.n_bundles;
    long pps = (((long)a.M + split - 1) / split + WBK - 1) / WBK * WBK;
    split = ((long)a.M + pps - 1) / pps;                         // no empty splits
    a.pix_per_split = (unsigned)pps;
    a.out = split > 1 ? ws : dw;
    const size_t lds_bytes = (size_t)2 * WBK * (BM + BN) * sizeof(float);
    auto kern = conv_wgrad_kernel<BM, BN, WM, WN>;
    static std::atomic<unsigned long long> lds_set{0};     // per instantiation: one bit per device (common.h)
    if (hipError_t e = bts_ensure_dynamic_lds((const void*)kern, lds_bytes, lds_set); e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(kern, dim3((unsigned)tiles, (unsigned)split, (unsigned)a.n_bundles), dim3(256), lds_bytes, s, a);
    if (split > 1) {
        const long count4 = per / 4;      // N % 4 == 0
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((count4 + 15) / 16)), dim3(256), 0, s, ws, dw, count4,
                           (int)split);
    }
    return (int)hipGetLastError();
}

// Tile + split plan.  The output (c_out x taps*c_in) is small and the pixel axis long, so parallelism comes from
// splitting pixels; every split costs one partial tile written and re-read.  Candidates are scored by
//   useful fraction of the padded tile grid  x  per-tile efficiency  x  chip fill (workgroups / 512, capped at 1)
//   x  operand bytes / (operand bytes + partial-tile bytes)
// with the split chosen to reach ~768 workgroups within [>= 4 K-steps per split, <= 1024 splits, workspace size].
struct WgradPlan { int variant; long split; };

inline WgradPlan plan_wgrad(int c_out, int N, long M, long ws_floats, bool have_ws, int n_bundles) {
    static const int bm[4] = {128, 64, 32, 64}, bn[4] = {128, 128, 128, 64};
    static const double eff[4] = {1.0, 0.9, 0.75, 0.8};
    static const long target = getenv("BTS_WGRAD_TARGET") ? atol(getenv("BTS_WGRAD_TARGET")) : 768;   // immutable once read
    static const int force_variant = getenv("BTS_WGRAD_VARIANT") ? (atoi(getenv("BTS_WGRAD_VARIANT")) & 3) : -1;
    const long per = (long)c_out * N * n_bundles;
    long max_split = M / (4 * WBK);
    if (max_split > 1024) max_split = 1024;
    if (!have_ws || ws_floats < 2 * per) max_split = 1;
    else if (max_split * per > ws_floats) max_split = ws_floats / per;
    if (max_split < 1) max_split = 1;
    const double in_bytes = 4.0 * (double)M * (c_out + N);           // operands read once (taps re-read from cache)
    WgradPlan best = {0, 1};
    double best_score = -1.0;
    for (int v = 0; v < 4; ++v) {
        const long mt = (c_out + bm[v] - 1) / bm[v], nt = (N + bn[v] - 1) / bn[v];
        const long tiles = mt * nt * n_bundles;
        long split = (target + tiles - 1) / tiles;
        if (split > max_split) split = max_split;
        const double useful = (double)c_out * N / ((double)mt * bm[v] * nt * bn[v]);
        double fill = (double)(tiles * split) / 512.0;
        if (fill > 1.0) fill = 1.0;
        const double partial_bytes = split > 1 ? 8.0 * (double)split * (double)per : 0.0;   // written once, read once
        const double score = useful * eff[v] * fill * in_bytes / (in_bytes + partial_bytes);
        if (score > best_score * 1.0001) { best_score = score; best = {v, split}; }
    }
    if (force_variant >= 0) best.variant = force_variant;
    return best;
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
    a.pre_scale = d->pre_scale; a.pre_shift = d->pre_shift; a.pre_relu = d->pre_relu;
    if (d->pre_scale && (!d->pre_shift || ((uintptr_t)d->pre_scale & 15) || ((uintptr_t)d->pre_shift & 15))) return BTS_ERR_INVALID;
    a.n_bundles = d->n_bundles > 1 ? d->n_bundles : 1;
    if (d->n_bundles < 0 || a.n_bundles > 65535) return BTS_ERR_INVALID;
    if (d->x_pix_stride < (long)a.n_bundles * d->c_in || d->dy_pix_stride < (long)a.n_bundles * d->c_out) return BTS_ERR_INVALID;
    const WgradPlan p = plan_wgrad(d->c_out, a.N, (long)a.M, d->ws_floats, d->ws != nullptr, a.n_bundles);
    switch (p.variant) {
        case 0: return launch_wgrad<128, 128, 2, 2>(a, p.split, d->dw, d->ws, s);
        case 1: return launch_wgrad<64, 128, 1, 4>(a, p.split, d->dw, d->ws, s);
        case 2: return launch_wgrad<32, 128, 1, 4>(a, p.split, d->dw, d->ws, s);
        default: return launch_wgrad<64, 64, 2, 2>(a, p.split, d->dw, d->ws, s);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Batched weight packing for the training step: every step the optimiser rewrites the OIHW parameters, and both the
// forward kernel ([c_out_pad][k_pad], K = tap*c_in_ld + c) and the input-gradient pass (the same layout of the
// flipped, transposed kernel) need them re-laid.  One launch re-packs every registered weight from a device table.
namespace {

struct PackEntry {          // 12 x int64, filled by the host (bts_amd/train.py)
    const float* src;       // OIHW parameter
    float* dst;             // [rows_pad][k_pad]
    long rows, inner;       // packed rows (c_out, or c_in for the transposed layout) and real inner channels
    long k, c_in_ld, rows_pad, k_pad;
    long s_row, s_c;        // element strides in src for the packed row / inner channel
    long flip;              // 0: taps as stored, 1: spatially flipped (input gradient)
    long first_block;       // prefix sum of blocks over the table
    long cg, gmode;         // grouped weight [C][cg][k][k] packed block-diagonally into a bundle (src points at the
                            // bundle's first output channel): cg = channels per group; gmode 1 = forward (row = output
                            // channel, inner = bundle-local input channel), 2 = input gradient (row = bundle-local input
                            // channel, inner = output channel); entries outside a row's own group are zero.  0 = dense
};

constexpr int PACK_PER_BLOCK = 256 * 4 * 4;

__global__ __launch_bounds__(256) void pack_weights_kernel(const PackEntry* __restrict__ table, int n) {
    int lo = 0, hi = n - 1;                       // the entry whose block range holds blockIdx.x
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (table[mid].first_block <= (long)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const PackEntry e = table[lo];
    const long total = e.rows_pad * e.k_pad;
    const long base = ((long)blockIdx.x - e.first_block) * PACK_PER_BLOCK;
    const long kk2 = e.k * e.k;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const long i0 = base + ((long)u * 256 + threadIdx.x) * 4;
        if (i0 >= total) continue;
        const long row = i0 / e.k_pad, col0 = i0 - row * e.k_pad;
        f32x4 v = (f32x4)(0.f);
        if (row < e.rows) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const long col = col0 + j;
                const long tap = col / e.c_in_ld, c = col - tap * e.c_in_ld;
                if (tap < kk2 && c < e.inner) {
                    const long t = e.flip ? kk2 - 1 - tap : tap;
                    if (e.gmode == 0) {
                        v[j] = e.src[row * e.s_row + c * e.s_c + t];
                    } else if (row / e.cg == c / e.cg) {                 // same group: the only non-zero block
                        const long g0 = (row / e.cg) * e.cg;
                        const long r = e.gmode == 2 ? row - g0 : row, cc = e.gmode == 1 ? c - g0 : c;
                        v[j] = e.src[r * e.s_row + cc * e.s_c + t];
                    }
                }
            }
        }
        *reinterpret_cast<f32x4*>(e.dst + i0) = v;
    }
}

// Winograd F(2x2,3x3) form of a packed 3x3 weight, in conv_wino_kernel's B-fragment order (include/bts_hip.h,
// bts_pack_wino_f32).  One thread = one output f32x4 (four consecutive k of one (xi, n)): U = G g G^T in fp64, rounded once.
__global__ __launch_bounds__(256) void pack_wino_kernel(const float* __restrict__ w, long k_pad, int c_in_ld, int c_main, int n_ct,
                                                        int mf16, float* __restrict__ out, long total4) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total4) return;
    const int lane = (int)(i & 63);
    long r = i >> 6;
    const int gpc = mf16 ? 2 : 4;
    const int g = (int)(r % gpc); r /= gpc;
    const int ct = (int)(r % n_ct); r /= n_ct;
    const int nchunks = c_main / 32;
    const int chunk = (int)(r % nchunks);
    const int xi = (int)(r / nchunks);
    const int n = mf16 ? 16 * ct + (lane & 15) : 32 * ct + (lane & 31);
    const int k0 = mf16 ? 32 * chunk + 16 * g + 4 * (lane >> 4) : 32 * chunk + 8 * g + 4 * (lane >> 5);
    const int wi = xi >> 2, wj = xi & 3;
    // rows of G: {1,0,0}, {.5,.5,.5}, {.5,-.5,.5}, {0,0,1}
    const double Gi[3] = {wi == 0 ? 1.0 : (wi == 3 ? 0.0 : 0.5), wi == 1 ? 0.5 : (wi == 2 ? -0.5 : 0.0), wi == 3 ? 1.0 : (wi == 0 ? 0.0 : 0.5)};
    const double Gj[3] = {wj == 0 ? 1.0 : (wj == 3 ? 0.0 : 0.5), wj == 1 ? 0.5 : (wj == 2 ? -0.5 : 0.0), wj == 3 ? 1.0 : (wj == 0 ? 0.0 : 0.5)};
    f32x4 o;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float* src = w + (long)n * k_pad + (k0 + q);
        double acc = 0.0;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            double row = 0.0;
#pragma unroll
            for (int b = 0; b < 3; ++b) row += (double)src[(long)(3 * a + b) * c_in_ld] * Gj[b];
            acc += Gi[a] * row;
        }
        o[q] = (float)acc;
    }
    *reinterpret_cast<f32x4*>(out + 4 * i) = o;
}

}  // namespace

extern "C" long bts_pack_wino_floats(int c_out_pad, int c_in_ld, int n_tail, int c_out16) {
    const int c_main = c_in_ld - (n_tail > 0 ? 4 : 0);
    if (c_out_pad <= 0 || (c_out_pad & 31) || c_main <= 0 || (c_main & 31) || c_out16 < 0 || (c_out16 & 15) || c_out16 > c_out_pad) return -1;
    return 16L * c_main * (c_out16 > 0 ? c_out16 : c_out_pad);
}

extern "C" int bts_pack_wino_f32(const float* w_packed, int c_out_pad, long k_pad, int c_in_ld, int n_tail, int c_out16,
                                 float* out, bts_stream_t stream) {
    const long total = bts_pack_wino_floats(c_out_pad, c_in_ld, n_tail, c_out16);
    if (!w_packed || !out || total <= 0 || k_pad < 9L * c_in_ld || ((uintptr_t)out & 15)) return BTS_ERR_INVALID;
    const int c_main = c_in_ld - (n_tail > 0 ? 4 : 0);
    const int mf16 = c_out16 > 0 ? 1 : 0;
    const int n_ct = mf16 ? c_out16 / 16 : c_out_pad / 32;
    const long total4 = total / 4, blocks = (total4 + 255) / 256;
    if (blocks > 2147483647L) return BTS_ERR_INVALID;
    hipLaunchKernelGGL(pack_wino_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, w_packed, k_pad, c_in_ld, c_main,
                       n_ct, mf16, out, total4);
    return (int)hipGetLastError();
}

extern "C" long bts_pack_weights_blocks(long rows_pad, long k_pad) {
    return (rows_pad * k_pad + PACK_PER_BLOCK - 1) / PACK_PER_BLOCK;
}

extern "C" int bts_pack_weights_f32(const void* table, int n_entries, long total_blocks, bts_stream_t stream) {
    if (!table || n_entries <= 0 || total_blocks <= 0 || total_blocks > 2147483647L) return BTS_ERR_INVALID;
    hipLaunchKernelGGL(pack_weights_kernel, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream,
                       (const PackEntry*)table, n_entries);
    return (int)hipGetLastError();
}
