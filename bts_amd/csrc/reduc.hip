// reduction_1x1 forward for gfx950: the whole 1x1-conv(+ELU) chain of reference
// pytorch/bts.py:97-136 in one kernel, activations never leaving registers.
//
// Mapping.  One wave owns a tile of 32 pixels.  Every layer is computed TRANSPOSED,
// Y^T[Cout x 32px] = W[Cout x Cin] * X^T[Cin x 32px], on v_mfma_f32_32x32x2_f32 with the
// weights as the A operand and the activations as the B operand.  The D tile then has the
// pixel on the lane and the output channel in the register -- which is exactly the B-operand
// shape of the next layer (register s of a D tile feeds MFMA k-step s), so a layer's output
// (after ELU) is consumed by the next layer with no LDS round trip and no lane movement.
// The k order this induces (lane half h, step s=4g+q  <->  channel 8g+4h+q) is baked into the
// weight fragments, which the host packs once (bts_amd/ops.py:pack_reduc_weights) so that one
// conflict-free lane-linear ds_read_b128 yields the A operands of four consecutive k-steps.
// Weights live in LDS (<=114 KB for the 8x8 chain), copied once per workgroup.
//
// Roofline: 2x2 / 1x1 chains are HBM-read bound (AI 20 / 9.8 FLOP/B), 8x8 / 4x4 are (small)
// MFMA-bound; see DESIGN.md.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "common.h"
#include "lpg_math.h"

namespace {

// ---- reduction -> LPG hand-off on chip (reference bts.py:249-256, 263-270, 277-283) --------------------
// The chain's epilogue leaves the plane (n1,n2,n3,n4) of cell p0+i on lane i (i < 32) of the wave.  Instead of sending
// it through HBM to a second launch, the same wave writes the cell's K x K block of depth/max_depth: per output row the
// 32 cells x K columns are spread over the lanes in memory order (one shuffle set per kernel, hoisted out of the row
// loop), so a wave store is 1 KB (K=8), 2 x 512 B (K=4: two rows per instruction) or 2 x 256 B (K=2) of contiguous
// floats per row segment.  The nearest-downsampled side output ([::K/2, ::K/2] of the map, bts.py:256, 270) is a dense
// plane; |den| minima go through one int-key atomic per block (NaN-propagating, lpg_math.h).
struct LpgTail {
    float* depth;        // [B, h*K, w*K] depth / max_depth
    float* ds;           // optional [B, 2h, 2w] nearest-downsampled copy (K = 8: factor 4, K = 4: factor 2); NULL for K = 2
    int* abs_min_key;    // optional
    int h, w;            // low-resolution map size (cells)
};

template <int K>
__device__ __forceinline__ void lpg_emit(const LpgTail& t, long p0, long npix, int lane, float n1, float n2, float n3,
                                         float n4, float max_depth, float& amin, bool& saw_nan) {
    // which cell this lane writes for, and which part of the cell's row
    constexpr int LPC = K == 8 ? 2 : 1;                     // lanes per cell per output row
    const int src = K == 8 ? (lane >> 1) : (lane & 31);
    const float c1 = __shfl(n1, src, 64), c2 = __shfl(n2, src, 64), c3 = __shfl(n3, src, 64), c4 = __shfl(n4, src, 64);
    const long p = p0 + src;
    if (p >= npix) return;
    const int cx = (int)(p % t.w);
    const long rowi = p / t.w;
    const int cy = (int)(rowi % t.h);
    const int b = (int)(rowi / t.h);
    const int W = t.w * K, H = t.h * K;
    if constexpr (K == 8) {
        const int ck0 = (lane & (LPC - 1)) * 4;             // columns ck0..ck0+3 of the cell
        float* o = t.depth + ((long)b * H + (long)cy * K) * W + (long)cx * K + ck0;
#pragma unroll
        for (int r = 0; r < K; ++r) {
            float res[4];
            lpg_cell_outputs<K, 4>(c1, c2, c3, c4, r, ck0, max_depth, res, amin, saw_nan);
            *reinterpret_cast<float4*>(o + (long)r * W) = make_float4(res[0], res[1], res[2], res[3]);
            if (t.ds != nullptr && (r & 3) == 0)            // out rows 0 and 4, columns 0 and 4 -> ds (2cy + r/4, 2cx + lane&1)
                t.ds[((long)b * (2 * t.h) + 2 * cy + (r >> 2)) * (2 * t.w) + 2 * cx + (lane & 1)] = res[0];
        }
    } else if constexpr (K == 4) {
        const int rh = lane >> 5;                           // lanes 0-31: rows 0 / 2, lanes 32-63: rows 1 / 3
        float* o = t.depth + ((long)b * H + (long)cy * K) * W + (long)cx * K;
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int r = 2 * it + rh;
            float res[4];
            lpg_cell_outputs<K, 4>(c1, c2, c3, c4, r, 0, max_depth, res, amin, saw_nan);
            *reinterpret_cast<float4*>(o + (long)r * W) = make_float4(res[0], res[1], res[2], res[3]);
            if (t.ds != nullptr && rh == 0)                 // out rows 0 and 2, columns 0 and 2
                *reinterpret_cast<float2*>(t.ds + ((long)b * (2 * t.h) + 2 * cy + it) * (2 * t.w) + 2 * cx) = make_float2(res[0], res[2]);
        }
    } else {
        static_assert(K == 2, "upratio 8, 4 or 2");
        const int r = lane >> 5;
        float res[2];
        lpg_cell_outputs<K, 2>(c1, c2, c3, c4, r, 0, max_depth, res, amin, saw_nan);
        *reinterpret_cast<float2*>(t.depth + ((long)b * H + (long)cy * K + r) * W + (long)cx * K) = make_float2(res[0], res[1]);
    }
}

// block minimum of the waves' int keys, one atomic per block
__device__ __forceinline__ void lpg_publish_key(int key, int* abs_min_key) {
    __shared__ int wave_key[16];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) key = min(key, __shfl_xor(key, off, 64));
    const int wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
    if ((threadIdx.x & 63) == 0) wave_key[wv] = key;
    __syncthreads();
    if (threadIdx.x == 0) {
        int k = wave_key[0];
        for (int i = 1; i < nw; ++i) k = min(k, wave_key[i]);
        atomicMin(abs_min_key, k);
    }
}

// One dense layer: acc[mt] (32 out-rows each) = W * x, K real input channels (multiple of 8).
// wf points at this layer's fragments: float4 index ((mt*(K/8) + g)*64 + lane).
template <int K, int MT, int NX>
__device__ __forceinline__ void dense_layer(const float4* __restrict__ wf, int lane, const float (&x)[NX],
                                            f32x16 (&acc)[MT]) {
    static_assert(NX >= K / 2, "activation registers");
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mt][r] = 0.f;
    // software-prefetch one g-step of weight fragments; the sched_barrier keeps hipcc from
    // hoisting every ds_read of the layer to its top (which spills: 64 x b128 for 128->128)
    float4 wn[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) wn[mt] = wf[(mt * (K / 8)) * 64 + lane];
#pragma unroll
    for (int g = 0; g < K / 8; ++g) {
        float4 w[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) w[mt] = wn[mt];
        if (g + 1 < K / 8) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) wn[mt] = wf[(mt * (K / 8) + g + 1) * 64 + lane];
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            acc[mt] = mfma32x2(w[mt].x, x[4 * g + 0], acc[mt]);
            acc[mt] = mfma32x2(w[mt].y, x[4 * g + 1], acc[mt]);
            acc[mt] = mfma32x2(w[mt].z, x[4 * g + 2], acc[mt]);
            acc[mt] = mfma32x2(w[mt].w, x[4 * g + 3], acc[mt]);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// Mirrors the while-loop of reduction_1x1.__init__ (bts.py:105-122): K = num_in_filters,
// M = num_out_filters.  Leaves the last layer's first rows in o[0..2] (valid on lanes < 32).
template <int K, int M, int NX>
__device__ __forceinline__ void chain(const float4* __restrict__ wf, int lane, const float (&x)[NX], float (&o)[3]) {
    if constexpr (M < 8) {                        // plane_params (3 outs) or final (1 out)
        f32x16 acc[1];
        dense_layer<K, 1, NX>(wf, lane, x, acc);
        o[0] = acc[0][0]; o[1] = acc[0][1]; o[2] = acc[0][2];
    } else {
        constexpr int MT = (M + 31) / 32;
        f32x16 acc[MT];
        dense_layer<K, MT, NX>(wf, lane, x, acc);
        constexpr int NY = (M / 2 < 4) ? 4 : M / 2;
        float y[NY];
#pragma unroll
        for (int i = 0; i < NY; ++i) y[i] = elu1(acc[i / 16][i % 16]);      // conv + ELU, bts.py:116-119
        chain<M, M / 2, NY>(wf + MT * (K / 8) * 64, lane, y, o);
    }
}

template <int C0, int M0>
constexpr long chain_frag_float4s() {
    long n = 0;
    int k = C0, m = M0;
    while (m >= 8) { n += (long)((m + 31) / 32) * (k / 8) * 64; k = m; m = m / 2; }
    n += (long)(k / 8) * 64;
    return n;
}

template <int C0, int M0, bool FINAL, int LPGK = 0>
__global__ __launch_bounds__(512, 2) void reduc_fwd_kernel(const float* __restrict__ x, long x_pix_stride,
                                                           long npix, const float4* __restrict__ w_frag,
                                                           float max_depth, int normalize,
                                                           float* __restrict__ out, const LpgTail lt) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float4* wl = reinterpret_cast<float4*>(smem_raw);
    constexpr long NW = chain_frag_float4s<C0, M0>();
    for (long i = threadIdx.x; i < NW; i += blockDim.x) wl[i] = w_frag[i];
    __syncthreads();

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int waves_per_block = blockDim.x >> 6;
    const int j = lane & 31, h = lane >> 5;
    const long ntiles = (npix + 31) / 32;
    [[maybe_unused]] float amin = __uint_as_float(0x7f800000u);
    [[maybe_unused]] bool saw_nan = false;
    for (long tile = (long)blockIdx.x * waves_per_block + wave; tile < ntiles;
         tile += (long)gridDim.x * waves_per_block) {
        const long p = tile * 32 + j;
        const bool live = p < npix;
        float xr[C0 / 2];
        const float* xp = x + (live ? p : 0) * x_pix_stride + 4 * h;
#pragma unroll
        for (int t = 0; t < C0 / 8; ++t) {         // lane (j,h): channels 4*(2t+h) .. +3
            float4 v = live ? *reinterpret_cast<const float4*>(xp + 8 * t) : make_float4(0.f, 0.f, 0.f, 0.f);
            xr[4 * t + 0] = v.x; xr[4 * t + 1] = v.y; xr[4 * t + 2] = v.z; xr[4 * t + 3] = v.w;
        }
        float o[3];
        chain<C0, M0, C0 / 2>(wl, lane, xr, o);
        if constexpr (FINAL) {
            if (live && h == 0) out[p] = sigmoid1(o[0]);                        // bts.py:108-110
        } else {
            const float PI = 3.14159265358979323846f;
            const float theta = sigmoid1(o[0]) * PI / 3.f;                      // bts.py:127
            const float phi = sigmoid1(o[1]) * PI * 2.f;                        // bts.py:128
            const float dist = sigmoid1(o[2]) * max_depth;                      // bts.py:129
            float n1 = sinf(theta) * cosf(phi);                                 // bts.py:130
            float n2 = sinf(theta) * sinf(phi);                                 // bts.py:131
            float n3 = cosf(theta);                                             // bts.py:132
            if (normalize) {                                                    // bts.py:251
                const float nn = fmaxf(sqrtf(n1 * n1 + n2 * n2 + n3 * n3), 1e-12f);
                n1 /= nn; n2 /= nn; n3 /= nn;
            }
            if (live && h == 0 && out != nullptr) *reinterpret_cast<float4*>(out + p * 4) = make_float4(n1, n2, n3, dist);
            if constexpr (LPGK > 0) lpg_emit<LPGK>(lt, tile * 32, npix, lane, n1, n2, n3, dist, max_depth, amin, saw_nan);
        }
    }
    if constexpr (LPGK > 0) {
        if (lt.abs_min_key != nullptr) lpg_publish_key(lpg_min_key(amin, saw_nan), lt.abs_min_key);
    }
}

template <int C0, int M0, bool FINAL, int LPGK = 0>
int launch_reduc(const float* x, long stride, long npix, const float* w_frag, long w_frag_floats, float max_depth,
                 int normalize, float* out, hipStream_t s, const LpgTail lt = LpgTail{nullptr, nullptr, nullptr, 0, 0}) {
    constexpr long NW = chain_frag_float4s<C0, M0>();
    if (w_frag_floats != NW * 4) return BTS_ERR_INVALID;
    const size_t lds = (size_t)NW * 16;
    auto kern = reduc_fwd_kernel<C0, M0, FINAL, LPGK>;
    static std::atomic<unsigned long long> lds_set{0};     // per instantiation: one bit per device (common.h)
    if (hipError_t e = bts_ensure_dynamic_lds((const void*)kern, lds, lds_set); e != hipSuccess) return (int)e;
    const long ntiles = (npix + 31) / 32;
    const int per_cu = lds > 80 * 1024 ? 1 : 2;
    long blocks = (ntiles + 7) / 8;
    if (blocks > 256L * per_cu) blocks = 256L * per_cu;
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(512), lds, s, x, stride, npix,
                       reinterpret_cast<const float4*>(w_frag), max_depth, normalize, out, lt);
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------------------------
// 16x16x4-MFMA variant for the narrow chains (2x2: 64->32->16->8->3, 1x1: 32->16->8->1).  Same chaining trick:
// D of v_mfma_f32_16x16x4_f32 has the pixel on lane&15 and out-channel 4*(lane>>4)+r in register r, which is
// exactly the B operand of the next layer's k-step r (channel 16*mt + 4*kq + r).  A wave owns 64 pixels as four
// 16-pixel tiles that share every weight fragment; layers of <= 16 outputs cost one 16-row tile instead of a
// padded 32-row one (1x1 chain: 32 instead of 56 MFMA-cycles per pixel), a lane loads 64-byte pixel segments
// (32-byte ones in the 32x32 kernel), and the sigmoid/sin/cos epilogue is re-spread over all 64 lanes.
typedef float f32x4v __attribute__((ext_vector_type(4)));
#ifndef RT2
#define RT2 2     // 16-pixel tiles per wave for the 2x2 chain
#endif
#ifndef RT1
#define RT1 2     // ... for the 1x1 chain
#endif

template <int NT, int K, int MT, int NX>
__device__ __forceinline__ void dense_layer16(const float4* __restrict__ wf, int lane, const float (&x)[NT][NX],
                                              f32x4v (&acc)[MT][NT]) {
    constexpr int G = (K + 15) / 16;
    static_assert(NX >= 4 * G, "activation registers");
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[mt][t] = f32x4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int g = 0; g < G; ++g) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const float4 w = wf[(mt * G + g) * 64 + lane];
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                acc[mt][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(w.x, x[t][4 * g + 0], acc[mt][t], 0, 0, 0);
                acc[mt][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(w.y, x[t][4 * g + 1], acc[mt][t], 0, 0, 0);
                acc[mt][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(w.z, x[t][4 * g + 2], acc[mt][t], 0, 0, 0);
                acc[mt][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(w.w, x[t][4 * g + 3], acc[mt][t], 0, 0, 0);
            }
        }
    }
}

// (K, M) walk as in reduction_1x1.__init__ (bts.py:105-122); o[t][c] = channel c of the last layer for pixel tile t
// (valid on lanes < 16: out-channel 4*(lane>>4)+r lives in register r).
template <int NT, int K, int M, int NX>
__device__ __forceinline__ void chain16(const float4* __restrict__ wf, int lane, const float (&x)[NT][NX], float (&o)[NT][3]) {
    constexpr int G = (K + 15) / 16;
    if constexpr (M < 8) {
        f32x4v acc[1][NT];
        dense_layer16<NT, K, 1, NX>(wf, lane, x, acc);
#pragma unroll
        for (int t = 0; t < NT; ++t) { o[t][0] = acc[0][t][0]; o[t][1] = acc[0][t][1]; o[t][2] = acc[0][t][2]; }
    } else {
        constexpr int MT = (M + 15) / 16;
        f32x4v acc[MT][NT];
        dense_layer16<NT, K, MT, NX>(wf, lane, x, acc);
        float y[NT][4 * MT];
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) y[t][4 * mt + r] = elu1(acc[mt][t][r]);     // rows >= M are exact zeros
        chain16<NT, M, M / 2, 4 * MT>(wf + MT * G * 64, lane, y, o);
    }
}

template <int C0, int M0>
constexpr long chain16_frag_float4s() {
    long n = 0;
    int k = C0, m = M0;
    while (m >= 8) { n += (long)((m + 15) / 16) * ((k + 15) / 16) * 64; k = m; m = m / 2; }
    n += (long)((k + 15) / 16) * 64;
    return n;
}

// NT = 16-pixel tiles per wave (4: 64 pixels, all lanes busy in the epilogue; 2: half the registers, more waves)
// The 2x2 chain (C0 = 64) is compiled for FOUR waves per SIMD (128 VGPRs; the LPG variant spills two dwords): its MFMA,
// VALU (29 ELU + sigmoid / sin / cos evaluations per 32 pixels) and HBM times are all ~60-80 us at B = 16 and only
// overlap across waves -- 3 -> 4 waves per SIMD took the fused launch from 204 to 175 us (gpurun_out/r3l).  A register
// double-buffer that prefetches the next pixel group instead cost occupancy and measured -3 % fused / +13 % stand-alone:
// dropped.
template <int C0, int M0, bool FINAL, int NT, int LPGK = 0>
__global__ __launch_bounds__(512, (C0 == 64 ? 4 : 1)) void reduc16_fwd_kernel(const float* __restrict__ x, long x_pix_stride, long npix,
                                                          const float4* __restrict__ w_frag, float max_depth,
                                                          int normalize, float* __restrict__ out, const LpgTail lt) {
    static_assert(LPGK == 0 || NT == 2, "the LPG tail expects 32 cells per wave");
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float4* wl = reinterpret_cast<float4*>(smem_raw);
    constexpr long NW = chain16_frag_float4s<C0, M0>();
    for (long i = threadIdx.x; i < NW; i += blockDim.x) wl[i] = w_frag[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwav = blockDim.x >> 6;
    const int j = lane & 15, q = lane >> 4;
    constexpr int PXW = 16 * NT;                          // pixels per wave-group
    const long ngroups = (npix + PXW - 1) / PXW;
    [[maybe_unused]] float amin = __uint_as_float(0x7f800000u);
    [[maybe_unused]] bool saw_nan = false;
    for (long grp = (long)blockIdx.x * nwav + wave; grp < ngroups; grp += (long)gridDim.x * nwav) {
        const long p0 = grp * PXW;
        float xr[NT][C0 / 4];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const long p = p0 + 16 * t + j;
            const bool live = p < npix;
            const float* xp = x + (live ? p : 0) * x_pix_stride + 4 * q;
#pragma unroll
            for (int g = 0; g < C0 / 16; ++g) {          // lane (j,q): channels 16g + 4q .. +3
                const float4 v = live ? *reinterpret_cast<const float4*>(xp + 16 * g) : make_float4(0.f, 0.f, 0.f, 0.f);
                xr[t][4 * g + 0] = v.x; xr[t][4 * g + 1] = v.y; xr[t][4 * g + 2] = v.z; xr[t][4 * g + 3] = v.w;
            }
        }
        float o[NT][3];
        chain16<NT, C0, M0, C0 / 4>(wl, lane, xr, o);
        // pixel p0 + lane's parameters sit on lane (lane & 15) of tile (lane >> 4): one shuffle per value
        float c0 = 0.f, c1 = 0.f, c2 = 0.f;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const float a0 = __shfl(o[t][0], j, 64), a1 = __shfl(o[t][1], j, 64), a2 = __shfl(o[t][2], j, 64);
            if (q == t) { c0 = a0; c1 = a1; c2 = a2; }
        }
        const long p = p0 + lane;
        const bool mine = lane < PXW && p < npix;
        if constexpr (FINAL) {
            if (mine) out[p] = sigmoid1(c0);                                    // bts.py:108-110
        } else {
            const float PI = 3.14159265358979323846f;
            const float theta = sigmoid1(c0) * PI / 3.f;                        // bts.py:127
            const float phi = sigmoid1(c1) * PI * 2.f;                          // bts.py:128
            const float dist = sigmoid1(c2) * max_depth;                        // bts.py:129
            float n1 = sinf(theta) * cosf(phi);                                 // bts.py:130
            float n2 = sinf(theta) * sinf(phi);                                 // bts.py:131
            float n3 = cosf(theta);                                             // bts.py:132
            if (normalize) {                                                    // bts.py:251
                const float nn = fmaxf(sqrtf(n1 * n1 + n2 * n2 + n3 * n3), 1e-12f);
                n1 /= nn; n2 /= nn; n3 /= nn;
            }
            if (mine && out != nullptr) *reinterpret_cast<float4*>(out + p * 4) = make_float4(n1, n2, n3, dist);
            if constexpr (LPGK > 0) lpg_emit<LPGK>(lt, p0, npix, lane, n1, n2, n3, dist, max_depth, amin, saw_nan);
        }
    }
    if constexpr (LPGK > 0) {
        if (lt.abs_min_key != nullptr) lpg_publish_key(lpg_min_key(amin, saw_nan), lt.abs_min_key);
    }
}

template <int C0, int M0, bool FINAL, int NT, int LPGK = 0>
int launch_reduc16(const float* x, long stride, long npix, const float* w_frag, long w_frag_floats, float max_depth,
                   int normalize, float* out, hipStream_t s, const LpgTail lt = LpgTail{nullptr, nullptr, nullptr, 0, 0}) {
    constexpr long NW = chain16_frag_float4s<C0, M0>();
    if (w_frag_floats != NW * 4) return BTS_ERR_INVALID;
    const size_t lds = (size_t)NW * 16;
    const long ngroups = (npix + 16 * NT - 1) / (16 * NT);
    long blocks = (ngroups + 7) / 8;
    if (blocks > 256L * 4) blocks = 256L * 4;
    hipLaunchKernelGGL((reduc16_fwd_kernel<C0, M0, FINAL, NT, LPGK>), dim3((unsigned)blocks), dim3(512), lds, s, x, stride, npix,
                       reinterpret_cast<const float4*>(w_frag), max_depth, normalize, out, lt);
    return (int)hipGetLastError();
}

}  // namespace

extern "C" int bts_reduc_fwd_f32(const float* x, long x_pix_stride, long npix, int c_in, int c_first_out,
                                 const float* w_frag, long w_frag_floats, float max_depth, int is_final,
                                 int normalize, float* out, bts_stream_t stream) {
    if (!x || !w_frag || !out || npix <= 0) return BTS_ERR_INVALID;
    if ((x_pix_stride & 3) || ((uintptr_t)x & 15) || ((uintptr_t)w_frag & 15) || ((uintptr_t)out & 15))
        return BTS_ERR_INVALID;
    if (x_pix_stride < c_in) return BTS_ERR_INVALID;
    hipStream_t s = (hipStream_t)stream;
    if (!is_final) {
        if (c_in == 128 && c_first_out == 128)
            return launch_reduc<128, 128, false>(x, x_pix_stride, npix, w_frag, w_frag_floats, max_depth, normalize, out, s);
        if (c_in == 128 && c_first_out == 64)
            return launch_reduc<128, 64, false>(x, x_pix_stride, npix, w_frag, w_frag_floats, max_depth, normalize, out, s);
        if (c_in == 64 && c_first_out == 32)       // narrow chains: 16x16x4-MFMA kernel (fragments from pack_reduc_weights)
            return launch_reduc16<64, 32, false, RT2>(x, x_pix_stride, npix, w_frag, w_frag_floats, max_depth, normalize, out, s);
        // bts_size 256 (bts.py:198-217 with num_features = 256): reduc8x8 64 -> 64.., reduc2x2 32 -> 16.. (reduc4x4 is the 64 -> 32 chain above)
        if (c_in == 64 && c_first_out == 64)
            return launch_reduc16<64, 64, false, RT2>(x, x_pix_stride, npix, w_frag, w_frag_floats, max_depth, normalize, out, s);
        if (c_in == 32 && c_first_out == 16)
            return launch_reduc16<32, 16, false, RT2>(x, x_pix_stride, npix, w_frag, w_frag_floats, max_depth, normalize, out, s);
    } else {
        if (c_in == 32 && c_first_out == 16)
            return launch_reduc16<32, 16, true, RT1>(x, x_pix_stride, npix, w_frag, w_frag_floats, max_depth, normalize, out, s);
        if (c_in == 16 && c_first_out == 8)        // bts_size 256: reduc1x1 16 -> 8 -> 1
            return launch_reduc16<16, 8, true, RT1>(x, x_pix_stride, npix, w_frag, w_frag_floats, max_depth, normalize, out, s);
    }
    return BTS_ERR_UNSUPPORTED;
}


// reduction_1x1 (non-final) -> F.normalize -> local_planar_guidance -> /max_depth -> nearest downsample, ONE launch
// (reference pytorch/bts.py:249-256 / 263-270 / 277-283 run these as ~45 ATen launches per scale).
extern "C" int bts_reduc_lpg_fwd_f32(const float* x, long x_pix_stride, int B, int h, int w, int c_in, int c_first_out,
                                     const float* w_frag, long w_frag_floats, float max_depth, int upratio,
                                     float* plane4, float* depth_scaled, float* ds_out, float* abs_min, bts_stream_t stream) {
    if (!x || !w_frag || !depth_scaled || B <= 0 || h <= 0 || w <= 0) return BTS_ERR_INVALID;
    if ((x_pix_stride & 3) || ((uintptr_t)x & 15) || ((uintptr_t)w_frag & 15) || ((uintptr_t)plane4 & 15) ||
        ((uintptr_t)depth_scaled & 15) || ((uintptr_t)ds_out & 7))
        return BTS_ERR_INVALID;
    if (x_pix_stride < c_in || !(max_depth > 0.f)) return BTS_ERR_INVALID;
    if (upratio == 2 && ds_out != nullptr) return BTS_ERR_INVALID;          // the 2x2 scale has no downsampled side output
    const long npix = (long)B * h * w;
    if ((double)npix * upratio * upratio >= 9.0e18) return BTS_ERR_UNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    int* key = reinterpret_cast<int*>(abs_min);
    if (key) {
        hipError_t e = hipMemsetD32Async((hipDeviceptr_t)key, 0x7f800000, 1, s);
        if (e != hipSuccess) return (int)e;
    }
    const LpgTail lt{depth_scaled, ds_out, key, h, w};
    if (c_in == 128 && c_first_out == 128 && upratio == 8)
        return launch_reduc<128, 128, false, 8>(x, x_pix_stride, npix, w_frag, w_frag_floats, max_depth, 1, plane4, s, lt);
    if (c_in == 128 && c_first_out == 64 && upratio == 4)
        return launch_reduc<128, 64, false, 4>(x, x_pix_stride, npix, w_frag, w_frag_floats, max_depth, 1, plane4, s, lt);
    if (c_in == 64 && c_first_out == 32 && upratio == 2)
        return launch_reduc16<64, 32, false, RT2, 2>(x, x_pix_stride, npix, w_frag, w_frag_floats, max_depth, 1, plane4, s, lt);
    // bts_size 256: the three scales on the narrow kernel
    if (c_in == 64 && c_first_out == 64 && upratio == 8)
        return launch_reduc16<64, 64, false, RT2, 8>(x, x_pix_stride, npix, w_frag, w_frag_floats, max_depth, 1, plane4, s, lt);
    if (c_in == 64 && c_first_out == 32 && upratio == 4)
        return launch_reduc16<64, 32, false, RT2, 4>(x, x_pix_stride, npix, w_frag, w_frag_floats, max_depth, 1, plane4, s, lt);
    if (c_in == 32 && c_first_out == 16 && upratio == 2)
        return launch_reduc16<32, 16, false, RT2, 2>(x, x_pix_stride, npix, w_frag, w_frag_floats, max_depth, 1, plane4, s, lt);
    return BTS_ERR_UNSUPPORTED;
}
