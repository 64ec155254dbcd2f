// NHWC implicit-GEMM convolution on fp32-input MFMA for gfx950 (v_mfma_f32_32x32x2_f32).
//
// One kernel family serves every convolution of the BTS decoder (reference pytorch/bts.py):
// atrous_conv's 1x1 and dilated 3x3 (bts.py:65-80), upconv's nearest-2x + 3x3 (bts.py:83-94),
// conv5..conv1 and daspp_conv (bts.py:183-221).  The reference runs these as separate ATen
// launches (BN, ReLU, conv, BN, ReLU, conv, ELU, cat ...); here the per-channel affine + ReLU
// of the INPUT is applied while the tile sits in registers on its way to LDS (prologue) and
// the BN/activation/BN of the OUTPUT is applied on the accumulators (epilogue); concat is
// free because outputs are written into channel slices of a preallocated NHWC buffer.
//
// GEMM view: M = B*H*W output pixels, N = c_out, K = taps * c_in (flattened, tap-major).  Workgroup = 256 threads
// (4 waves) computing BM x BN with BK = 32 per step.  Both operands are staged K-contiguous
// in LDS with a 36-float row stride (144 B: the 16 rows a ds_read_b128 lane group touches fall
// in 16 distinct 16-B bank slots).  A lane reads 4 consecutive k of its row with ONE
// ds_read_b128 and feeds them to 4 consecutive MFMA k-steps: step q of group g multiplies
// k = 8g+q (lanes 0-31) and k = 8g+4+q (lanes 32-63) -- the K order is immaterial as long as
// both operands agree.  Taps are gathered per lane with bounds masks (dilation 24 on a 44-row
// map leaves no contiguous halo to exploit); zero padding is applied AFTER the prologue, as the
// reference pads the post-BN-ReLU tensor.  Stride and a folded nearest-2x upsample live in the
// gather index, so upconv and the encoder stem need no extra pass.
//
// fp32-input MFMA is exact f32 (a k-ordered fmaf chain), so parity with the CPU reference is
// at rounding-order level (~1e-6), well inside the 1e-3 budget that bf16 operands break.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <type_traits>
#include "common.h"

namespace {

constexpr int BK = 32;

// Tuning / A-B knobs from the environment, read ONCE (thread-safe function-local static) into an immutable struct:
// nothing in this file mutates state after that, whichever thread, device or stream calls in.
struct ConvKnobs {
    int split_max; long split_below; long split_target;   // BTS_CONV_SPLITK / _BELOW / _TARGET
    long lds_bytes;                                        // BTS_CONV_LDS_KB: inflate LDS to cap workgroups per CU (0 = off)
    int no48, force_bm;                                    // BTS_CONV_NO48, BTS_CONV_BM
    int precision, emu_sb;                                 // BTS_CONV_PRECISION, BTS_CONV_EMU_SB (-1 = unset)
    int w8, w8s;                                           // BTS_CONV_W8, BTS_CONV_W8S
    int k1x1; long k1x1_min_tiles;                         // BTS_CONV_1X1 (0 off, 1 = wide-tile 1x1 kernel for c_out % 192 == 0, 2 = also 256/128 wide), BTS_CONV_1X1_MIN_TILES
    int k1x1_sb;                                           // BTS_CONV_1X1_SB: 1 (default) = the 64-row wide tile keeps one weight buffer (three workgroups per CU), 0 = two
    int k1x1_rows;                                         // BTS_CONV_1X1_ROWS: 0 = by K (default), 64 / 128 force the wide kernel's row tile
    int stem;                                              // BTS_CONV_STEM: 0 = the stem on the generic row-tiled kernel (A/B), default 1
    int tapskip;                                           // BTS_CONV_TAPSKIP: 0 = run every tap of every tile (A/B), default 1
    int halo_sb;                                           // BTS_CONV_HALO_SB: 1 (default) = the 128- and 32-wide halo tiles (no planar tail) keep one weight buffer: two / three workgroups per CU instead of one / two
    int halo48_w8; long halo48_w8_below;                   // BTS_CONV_HALO48_W8 (0 = never) / _BELOW: 8-wave 48-wide halo tile for declared launches below this many workgroups (default: all)
    long halo_fill;                                        // BTS_CONV_HALO_FILL: declared-launch workgroups from which the halo kernel replaces split-K (200)
    int fill_frames;                                       // BTS_CONV_FILL_FRAMES: default of bts_conv_desc.fill_frames (8)
    int wino;                                              // BTS_CONV_WINO: 1 (default) = eligible stride-1 3x3 convolutions whose caller supplies Winograd-form weights take the fused F(2x2,3x3) kernel (conv_wino.inc); 0 = direct kernels (A/B)
    int halo_emu;                                          // BTS_CONV_HALO_EMU: 1 (default) = precision-1 launches with pre-split weights take the bf16x3 halo-tile kernel where eligible, 0 = row-tiled emulation (A/B)
    int halo_dil;                                          // BTS_CONV_HALO_DIL: 1 (default) = the dilation-3 3x3 convolution (ASPP daspp_3) on the dilated halo tile, 2 = also dilation 6 / 12, 0 = none (row-tiled kernel with tap skipping)
    int stagger;                                           // BTS_CONV_STAGGER: 1 (default) = the eight-wave 48-wide halo tile staggers the staging block of waves 4..7 against their SIMD partners 0..3 (A/B)
    int halo;                                              // BTS_CONV_HALO: 0 off, 1 = halo-tile kernel where eligible unless split-K applies, 2 = also instead of split-K.  Default 1: on the deep 22x76 maps a frame has only 15 spatial tiles, so at batch 1 split-K fills the chip 7x better (53 vs 16 us per layer), and the choice may not depend on the batch (a frame's bits must not)
};
inline long env_long(const char* name, long dflt) { const char* v = getenv(name); return v ? atol(v) : dflt; }
const ConvKnobs& knobs() {
    static const ConvKnobs k = {(int)env_long("BTS_CONV_SPLITK", 8), env_long("BTS_CONV_SPLITK_BELOW", 700),
                                env_long("BTS_CONV_SPLITK_TARGET", 1024), env_long("BTS_CONV_LDS_KB", 0) * 1024,
                                (int)env_long("BTS_CONV_NO48", 0), (int)env_long("BTS_CONV_BM", 0),
                                (int)env_long("BTS_CONV_PRECISION", -1), (int)env_long("BTS_CONV_EMU_SB", -1),
                                (int)env_long("BTS_CONV_W8", 1), (int)env_long("BTS_CONV_W8S", 1),
                                (int)env_long("BTS_CONV_1X1", 1), env_long("BTS_CONV_1X1_MIN_TILES", 150),
                                (int)env_long("BTS_CONV_1X1_SB", 1), (int)env_long("BTS_CONV_1X1_ROWS", 0), (int)env_long("BTS_CONV_STEM", 1), (int)env_long("BTS_CONV_TAPSKIP", 1), (int)env_long("BTS_CONV_HALO_SB", 1), (int)env_long("BTS_CONV_HALO48_W8", 1), env_long("BTS_CONV_HALO48_W8_BELOW", 1L << 40), env_long("BTS_CONV_HALO_FILL", 200),
                                (int)(env_long("BTS_CONV_FILL_FRAMES", 8) > 0 ? env_long("BTS_CONV_FILL_FRAMES", 8) : 8),
                                (int)env_long("BTS_CONV_WINO", 1), (int)env_long("BTS_CONV_HALO_EMU", 1), (int)env_long("BTS_CONV_HALO_DIL", 1), (int)env_long("BTS_CONV_STAGGER", 1),
                                (int)env_long("BTS_CONV_HALO", 1)};
    return k;
}
// floats per LDS row (32 + pad), chosen per MFMA shape so that the 16 rows a ds_read_b128 lane group touches
// land in 16 distinct 16-B bank slots: 36 (slot = 9*row) for the 32x32 lane map, 40 (slot = 10*row + k-quarter)
// for the 16x16 one -- 36 is 2-way conflicted there (SQ_LDS_BANK_CONFLICT, profiles/r01_pmc_mfma.json)
template <int MF> struct LdsLd { static constexpr int value = MF == 32 ? 36 : 40; };

struct ConvArgs {
    const float* x; long x_pix_stride; int c_in_ld; int k_pad;   // k_pad: padded flattened K (taps*c_in_ld rounded to 32)
    int B, h_in, w_in, ups;          // ups = log2(up)
    int ksize, dil, stride, pad;
    int subpix;                      // 1: sub-pixel upconv (4 parity classes of 2x2 kernels, output scattered x2)
    int k_flat;                      // ksize*ksize*c_in_ld
    unsigned magic_c, magic_ks;      // floor(2^32/d)+1 for d = c_in_ld, ksize: exact n/d for n,d < 2^16
    const float* w; int c_out, c_out_pad;
    const float* pre_scale; const float* pre_shift; int pre_relu;
    const float* e1_scale; const float* e1_shift; int act;
    const float* e2_scale; const float* e2_shift;
    float* y; long y_pix_stride;
    float* y2; long y2_pix_stride;   // optional second NHWC destination
    int H, W;                        // output spatial size
    int Hs, Ws;                      // (upsampled) source extent the taps index: h_in*up, w_in*up
    long M;                          // B*H*W
    int n_ntiles;
    int tiles_per_class;
    // split-K (under-filled launches): split s of `ksplit` covers K-steps [s*its_per_split, ...) and stores raw
    // fp32 partial sums to ws[s][m][n] (row length ws_ld); splitk_reduce_kernel sums them and applies the epilogue
    int ksplit, its_per_split, ws_ld;
    float* ws;
    // grouped convolution as `n_classes` independent channel bundles (bts_conv_desc.n_bundles): bundle j reads input
    // channels [j*c_in_ld, ..), writes output channels [j*c_out, ..), uses weight block j and the j-th c_out_pad / c_in_ld
    // slice of the epilogue / prologue vectors.  bundled == 0: plain (n_classes 1) or sub-pixel (n_classes 4).
    int n_classes, bundled;
    const float* res; long res_pix_stride;   // optional residual added after e1, before the activation (NHWC)
    const float* tail[4];                    // planar tail operand (conv_halo.inc); tail[j] = plane 0 for unused slots
    int n_tail;
    int fill_frames;                         // frames assumed to share a launch (bts_conv_desc.fill_frames, resolved)
    int halo_single_a;                       // halo-tile kernel: one channel chunk, one A buffer (set by launch_halo)
    int tapskip;                             // 1: a tile skips the taps that fall outside the map for ALL of its pixels (tile_tapmask)
    const float* w_wino;                     // Winograd-form weights (bts_conv_desc.w_wino) or null
    const void* w_split;                     // precision 1: weights pre-split into bf16 planes [classes][3][c_out_pad][k_pad] (bts_conv_desc.w_split) or null
    int stagger;                             // halo-tile kernel, eight-wave 48-wide tile: waves 4..7 stage half a step after their SIMD partners 0..3 (set by launch_halo)
};

// Taps of a ksize x ksize convolution that can touch the map for at least one pixel of the row tile [m0, m0 + bm) -- a
// conservative superset computed from the tile's first and last pixel only (wave-uniform, SALU).  A dilated ASPP branch
// (bts.py:65-80: dilation 3..24 on a 44x152 map) reads zero padding for a third of its taps over most of the map:
// with dilation 24 the three taps of kernel row 0 lie above the map for every pixel of rows 0..23.  Such a K-step
// multiplies an all-zero A tile; skipping it leaves every accumulator bit unchanged (fma(0, w, acc) == acc).
// Stride 1 only.  Shared by the kernel and by the host-side FLOP accounting (bts_conv_plan_ksteps_f32).
__host__ __device__ inline unsigned long long tile_tapmask(int H, int W, int Hs, int Ws, int ksize, int dil, int pad_y, int pad_x,
                                                           long M, long m0, int bm) {
    const unsigned HW = (unsigned)(H * W);
    const long mend = m0 + bm < M ? m0 + bm : M;
    const unsigned mf = (unsigned)m0, ml = (unsigned)(mend - 1);
    const unsigned b0 = mf / HW, b1 = ml / HW;
    const unsigned yx0 = mf - b0 * HW, yx1 = ml - b1 * HW;
    const int y0 = (int)(yx0 / (unsigned)W), y1 = (int)(yx1 / (unsigned)W);
    const int x0 = (int)yx0 - y0 * W, x1 = (int)yx1 - y1 * W;
    const bool one_frame = b0 == b1, one_row = one_frame && y0 == y1;
    const int ylo = one_frame ? y0 : 0, yhi = one_frame ? y1 : H - 1;
    const int xlo = one_row ? x0 : 0, xhi = one_row ? x1 : W - 1;
    unsigned long long mask = 0;
    for (int t = 0; t < ksize * ksize; ++t) {
        const int ky = t / ksize, kx = t - ky * ksize;
        const int dy = ky * dil - pad_y, dx = kx * dil - pad_x;
        // some y in [ylo, yhi] with 0 <= y + dy < Hs, some x in [xlo, xhi] with 0 <= x + dx < Ws
        const int ya = ylo > -dy ? ylo : -dy, yb = yhi < Hs - 1 - dy ? yhi : Hs - 1 - dy;
        const int xa = xlo > -dx ? xlo : -dx, xb = xhi < Ws - 1 - dx ? xhi : Ws - 1 - dx;
        if (ya <= yb && xa <= xb) mask |= 1ull << t;
    }
    return mask ? mask : 1ull;                 // never empty: a tile always runs at least one (all-zero) tap
}

__device__ __forceinline__ float apply_act(float v, int act) {
    if (act == 1) return fmaxf(v, 0.f);
    if (act == 2) return elu1(v);
    if (act == 3) return sigmoid1(v);
    return v;
}


// ---- compact epilogues ------------------------------------------------------------------------------------------------
// The straightforward epilogue -- per accumulator element: optional affine, `apply_act` on a runtime code, optional
// affine, 64-bit `m * stride + n`, bounds tests, up to two stores -- unrolls to ~135 instructions per element: 6 500 lines
// (40 KB) of straight-line code behind a 128x192 tile, executed once per workgroup.  Ablation on the DenseNet block-1
// 1x1 layers: 163 of 340 us per layer were epilogue, of which the stores themselves 26.  The fast paths below are
// selected ONCE per workgroup on the uniform facts (activation, second destination) and keep ~6 instructions per
// element: the activation is a template parameter, the affines are unconditional FMAs with identity defaults (x*1+0 is
// exact), a row's address is a wave-uniform 64-bit base (SALU) plus a per-lane 32-bit offset computed once per column
// tile.  Residual adds and sigmoid stay on the generic code.
template <int ACT> __device__ __forceinline__ float act_fn(float v) {
    if constexpr (ACT == 1) return fmaxf(v, 0.f);
    else if constexpr (ACT == 2) return elu1(v);
    else if constexpr (ACT == 3) return sigmoid1(v);
    else return v;
}

// One wave's TM x TN accumulator tiles to NHWC.  Row slot q (0..MF-1) of row tile i is output pixel
// pix0[i] + q * pixstep (pix0 wave-uniform), valid iff q < nvalid[i]; column tile j covers channels ncol0 + j*MF + li.
// ev[j] = {e1 scale, e1 shift, e2 scale, e2 shift} of this lane's channel (identity when absent).
template <int ACT, bool Y2, int TM, int TN, int MF, int NACC, typename AccT>
__device__ __forceinline__ void store_tiles_nhwc(const ConvArgs& a, const AccT (&acc)[TM][TN], const long (&pix0)[TM],
                                                 const int (&nvalid)[TM], int pixstep, int ncol0, int li, int lh,
                                                 const float (&ev)[TN][4]) {
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = ncol0 + j * MF + li;
        const bool nok = n < a.c_out;
        const float s1 = ev[j][0], b1 = ev[j][1], s2 = ev[j][2], b2 = ev[j][3];
        // per-lane part of the address: the lane's k-half rows (4*lh) and its channel
        const unsigned lane_y = (unsigned)(4 * lh * pixstep) * (unsigned)a.y_pix_stride + (unsigned)n;
        [[maybe_unused]] const unsigned lane_y2 = (unsigned)(4 * lh * pixstep) * (unsigned)a.y2_pix_stride + (unsigned)n;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            float* const yb = a.y + pix0[i] * a.y_pix_stride;                           // wave-uniform
            [[maybe_unused]] float* const y2b = Y2 ? a.y2 + pix0[i] * a.y2_pix_stride : nullptr;
            const int nv = nvalid[i] - 4 * lh;                                          // slots this lane's half may write
#pragma unroll
            for (int r = 0; r < NACC; ++r) {
                const int qu = MF == 32 ? (r & 3) + 8 * (r >> 2) : r;                   // uniform part of the row slot
                float v = fmaf(acc[i][j][r], s1, b1);
                v = act_fn<ACT>(v);
                v = fmaf(v, s2, b2);
                if (nok && qu < nv) {
                    (yb + (long)(qu * pixstep) * a.y_pix_stride)[lane_y] = v;
                    if constexpr (Y2) (y2b + (long)(qu * pixstep) * a.y2_pix_stride)[lane_y2] = v;
                }
            }
        }
    }
}

// The same for NCHW output (rows = channels in the registers, lanes = pixels): element (channel n, pixel) lives at
// ((b*c_out + n) * HW + yx); no per-channel affines on this path (callers with e1/e2 take the generic code).
template <int ACT, int TM, int TN, int MF, int NACC, typename AccT>
__device__ __forceinline__ void store_tiles_nchw(const ConvArgs& a, const AccT (&acc)[TM][TN], const long (&chan_base)[TM],
                                                 const unsigned (&lane_pix)[TM], const bool (&pix_ok)[TM], long HW, int ncol0, int lh) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int nb = ncol0 + j * MF + 4 * lh;                                     // this lane's first channel of the tile
            float* const yb = a.y + (chan_base[i] + ncol0 + j * MF) * HW;               // wave-uniform
            const unsigned lane_off = (unsigned)(4 * lh) * (unsigned)HW + lane_pix[i];
#pragma unroll
            for (int r = 0; r < NACC; ++r) {
                const int qu = MF == 32 ? (r & 3) + 8 * (r >> 2) : r;
                const float v = act_fn<ACT>(acc[i][j][r]);
                if (pix_ok[i] && nb + qu < a.c_out) (yb + (long)qu * HW)[lane_off] = v;
            }
        }
}

// Uniform dispatch onto the fast paths: returns false when the generic epilogue has to run instead.
template <int TM, int TN, int MF, int NACC, typename AccT>
__device__ __forceinline__ bool fast_epilogue_nhwc(const ConvArgs& a, const AccT (&acc)[TM][TN], const long (&pix0)[TM],
                                                   const int (&nvalid)[TM], int pixstep, int ncol0, int li, int lh) {
    if (a.res != nullptr || a.act == 3) return false;
    // offsets inside one row tile must fit 32 bits
    if ((double)(MF * pixstep) * (double)(a.y_pix_stride > a.y2_pix_stride ? a.y_pix_stride : a.y2_pix_stride) >= 2147483648.0) return false;
    float ev[TN][4];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = ncol0 + j * MF + li;
        const bool in = n < a.c_out_pad;
        ev[j][0] = (a.e1_scale != nullptr && in) ? a.e1_scale[n] : 1.f;
        ev[j][1] = (a.e1_scale != nullptr && in) ? a.e1_shift[n] : 0.f;
        ev[j][2] = (a.e2_scale != nullptr && in) ? a.e2_scale[n] : 1.f;
        ev[j][3] = (a.e2_scale != nullptr && in) ? a.e2_shift[n] : 0.f;
    }
#define BTS_EPI(ACTV)                                                                                               \
    if (a.y2 != nullptr) store_tiles_nhwc<ACTV, true, TM, TN, MF, NACC>(a, acc, pix0, nvalid, pixstep, ncol0, li, lh, ev); \
    else                 store_tiles_nhwc<ACTV, false, TM, TN, MF, NACC>(a, acc, pix0, nvalid, pixstep, ncol0, li, lh, ev)
    if (a.act == 1) { BTS_EPI(1); }
    else if (a.act == 2) { BTS_EPI(2); }
    else { BTS_EPI(0); }
#undef BTS_EPI
    return true;
}

// Issue one K-step's global loads into staging registers.  Everything here is UNCONDITIONAL and
// branch-free: coordinates are clamped into the image so the address is always valid, and the
// "was this row real" bit goes to `okmask`; nothing consumes the data until stage_to_lds(), which
// runs after the MFMA block, so the loads' latency hides under the previous step's MFMAs.
template <int PA, int PB>
__device__ __forceinline__ void issue_loads(const ConvArgs& a, const float* __restrict__ wbase, int pad_y, int pad_x,
                                            int it, int lk, const int (&ay)[PA],
                                            const int (&ax)[PA], const unsigned (&abase)[PA], unsigned avalid,
                                            const unsigned (&wrow)[PB], f32x4 (&ra)[PA], f32x4 (&rb)[PB],
                                            f32x4& ps, f32x4& pb, unsigned& okmask) {
    // K is flattened over (tap, channel): k = tap*c_in_ld + c.  A lane's 4 consecutive k never straddle
    // a tap (c_in_ld % 4 == 0), so each lane decodes its own tap; channel counts that are not multiples
    // of 32 (36, 164, 228, DenseNet's 48i) then cost no per-tap padding.
    const unsigned k4 = (unsigned)(it * BK + lk);
    const bool kok = k4 < (unsigned)a.k_flat;
    const unsigned kk = kok ? k4 : 0u;
    const unsigned tap = __umulhi(kk, a.magic_c);
    const int c = (int)(kk - tap * (unsigned)a.c_in_ld);
    const int ky = (int)__umulhi(tap, a.magic_ks);
    const int kx = (int)tap - ky * a.ksize;
    const int dy = ky * a.dil - pad_y, dx = kx * a.dil - pad_x;
    if (a.pre_scale != nullptr) {
        ps = *reinterpret_cast<const f32x4*>(a.pre_scale + c);
        pb = *reinterpret_cast<const f32x4*>(a.pre_shift + c);
    }
    unsigned m = 0;
#pragma unroll
    for (int p = 0; p < PA; ++p) {
        const int yy = ay[p] + dy, xx = ax[p] + dx;        // ay/ax already carry the stride
        const bool ok = kok && ((avalid >> p) & 1u) && (unsigned)yy < (unsigned)a.Hs && (unsigned)xx < (unsigned)a.Ws;
        const int yc = min(max(yy, 0), a.Hs - 1) >> a.ups, xc = min(max(xx, 0), a.Ws - 1) >> a.ups;
        const unsigned off = (abase[p] + (unsigned)(yc * a.w_in + xc)) * (unsigned)a.x_pix_stride + (unsigned)c;
        ra[p] = *reinterpret_cast<const f32x4*>(a.x + off);
        m |= (ok ? 1u : 0u) << p;
    }
    okmask = m;
    const unsigned wofs = (unsigned)(it * BK);
#pragma unroll
    for (int p = 0; p < PB; ++p) rb[p] = *reinterpret_cast<const f32x4*>(wbase + (wofs + wrow[p]));
}

// Lean loader for the common case c_in_ld % 32 == 0 and no folded upsample: the whole K-step then sits
// in ONE tap, so tap decode and the tap's spatial offset are wave-uniform (SALU), the bounds checks collapse
// to one precomputed per-row bitmask over taps, and each row's address is base[p] + (ok ? delta : c).
// Per-wave VALU between MFMAs is the main tax on MFMA issue (scripts/ubench/mfma_ablate.cpp), so this
// matters more than it looks: ~5 VALU per staged row instead of ~20.
template <int PA, int PB>
__device__ __forceinline__ void issue_loads_fast(const ConvArgs& a, const float* __restrict__ wbase, int pad_y, int pad_x,
                                                 int it, int tap, int kc, const unsigned (&abase)[PA],
                                                 const unsigned long long (&vmask)[PA], const unsigned (&wrow)[PB],
                                                 f32x4 (&ra)[PA], f32x4 (&rb)[PB], f32x4& ps, f32x4& pb,
                                                 unsigned& okmask, int lk) {
    const int ky = tap / a.ksize, kx = tap - ky * a.ksize;                 // scalar
    const int dy = ky * a.dil - pad_y, dx = kx * a.dil - pad_x;
    const int c0 = kc * BK;
    const unsigned d_c = (unsigned)c0;
    const unsigned d_full = (unsigned)((dy * a.w_in + dx) * (int)a.x_pix_stride + c0);
    if (a.pre_scale != nullptr) {
        ps = *reinterpret_cast<const f32x4*>(a.pre_scale + c0 + lk);
        pb = *reinterpret_cast<const f32x4*>(a.pre_shift + c0 + lk);
    }
    unsigned m = 0;
#pragma unroll
    for (int p = 0; p < PA; ++p) {
        const unsigned ok = (unsigned)(vmask[p] >> tap) & 1u;
        const unsigned off = abase[p] + (ok ? d_full : d_c);
        ra[p] = *reinterpret_cast<const f32x4*>(a.x + off);
        m |= ok << p;
    }
    okmask = m;
    const unsigned wofs = (unsigned)(it * BK);
#pragma unroll
    for (int p = 0; p < PB; ++p) rb[p] = *reinterpret_cast<const f32x4*>(wbase + (wofs + wrow[p]));
}

// Prologue (BN affine + ReLU, reference bts.py:70,72) and zero padding, applied on the way to LDS.
template <int BM, int BN, int RPP, int PA, int PB, int LDS_LD>
__device__ __forceinline__ void stage_to_lds(const ConvArgs& a, float* __restrict__ As, int lrow, int lk,
                                             const f32x4 (&ra)[PA], const f32x4 (&rb)[PB], const f32x4& ps,
                                             const f32x4& pb, unsigned okmask) {
    float* Bs = As + BM * LDS_LD;
    const bool has_pre = a.pre_scale != nullptr;
#pragma unroll
    for (int p = 0; p < PA; ++p) {
        f32x4 v = ra[p];
        if (has_pre) v = v * ps + pb;
        if (a.pre_relu) {
            v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
        }
        const bool ok = (okmask >> p) & 1u;                  // zero padding AFTER the prologue
        v.x = ok ? v.x : 0.f; v.y = ok ? v.y : 0.f; v.z = ok ? v.z : 0.f; v.w = ok ? v.w : 0.f;
        *reinterpret_cast<f32x4*>(As + (p * RPP + lrow) * LDS_LD + lk) = v;
    }
#pragma unroll
    for (int p = 0; p < PB; ++p)
        if ((p + 1) * RPP <= BN || p * RPP + lrow < BN)       // BN = 48: the last pass covers half its rows
            *reinterpret_cast<f32x4*>(Bs + (p * RPP + lrow) * LDS_LD + lk) = rb[p];
}

// ---- "fp32 on the bf16 matrix cores" (bts_conv_desc.precision = 1) ---------------------------------------------
// Each fp32 operand x is split on the way to LDS into three bf16 pieces h + m + l (truncation split: h = top 16 bits
// of x, m = top 16 bits of x - h, l = top 16 bits of x - h - m; the subtractions are exact), stored as three bf16
// planes; a 32x32 block of the product then takes six v_mfma_f32_32x32x16_bf16 per 16 k (hh, hm, mh, hl, lh, mm --
// the three dropped cross terms are below 2^-24 of the product), accumulated in fp32.  Products of bf16 pairs are
// exact in fp32, so the only roundings are the fp32 accumulation (as in the fp32-MFMA path, but 8x fewer partial
// sums) and the 2^-24-relative tail of the split: measured error 1.2e-6 of max|result| on K = 2304 dot products vs
// 1.1e-6 for the v_mfma_f32_32x32x2_f32 chain (scripts/bf16x3_numerics.py).  The bf16 pipe runs 16x the fp32-MFMA rate, six products cost 6/16.
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int EMU_ROW_BYTES = 64;             // 32 bf16 of one K-step, unpadded; the four 16-B chunks of a row are
// XOR-swizzled with bits 2-3 of the row index: ds_read_b128 serves the non-contiguous lane groups {0-3,12-15,20-27},
// {4-11,16-19,28-31} (MI355X_MICROARCH.md, LDS), whose 16 rows then land on 16 distinct 16-B slots of the 256-B bank
// row, and a ds_write_b64 group (16 contiguous lanes = two consecutive rows) covers the two 64-B halves of the 128-B
// write bank row.  (A padded 80-B row was conflict-free for the reads only: SQ_LDS_BANK_CONFLICT = 1/3 of LDS cycles.)
__device__ __forceinline__ int emu_off(int row, int k) {      // byte offset of element (row, k) inside one plane
    return row * EMU_ROW_BYTES + ((((k >> 3) ^ (row >> 2)) & 3) << 4) + ((k & 7) << 1);
}

__device__ __forceinline__ void split_store(char* plane0, int plane_bytes, int byte_off, const f32x4 v) {
    unsigned h[4], m[4], l[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const unsigned xb = __float_as_uint(v[i]);
        h[i] = xb & 0xffff0000u;
        const float r = v[i] - __uint_as_float(h[i]);
        m[i] = __float_as_uint(r) & 0xffff0000u;
        l[i] = __float_as_uint(r - __uint_as_float(m[i])) & 0xffff0000u;
    }
    // bf16 pairs: low half = element 2i, high half = element 2i+1
    u32x2 ph = {__builtin_amdgcn_perm(h[1], h[0], 0x07060302u), __builtin_amdgcn_perm(h[3], h[2], 0x07060302u)};
    u32x2 pm = {__builtin_amdgcn_perm(m[1], m[0], 0x07060302u), __builtin_amdgcn_perm(m[3], m[2], 0x07060302u)};
    u32x2 pl = {__builtin_amdgcn_perm(l[1], l[0], 0x07060302u), __builtin_amdgcn_perm(l[3], l[2], 0x07060302u)};
    *reinterpret_cast<u32x2*>(plane0 + byte_off) = ph;
    *reinterpret_cast<u32x2*>(plane0 + plane_bytes + byte_off) = pm;
    *reinterpret_cast<u32x2*>(plane0 + 2 * plane_bytes + byte_off) = pl;
}

// same role as stage_to_lds: prologue + zero padding, then the split into the three bf16 planes of one LDS buffer
// (layout [plane][A rows BM | B rows BN][EMU_ROW_BYTES])
template <int BM, int BN, int RPP, int PA, int PB>
__device__ __forceinline__ void stage_to_lds_emu(const ConvArgs& a, char* __restrict__ buf, int lrow, int lk,
                                                 const f32x4 (&ra)[PA], const f32x4 (&rb)[PB], const f32x4& ps,
                                                 const f32x4& pb, unsigned okmask) {
    constexpr int PLANE = (BM + BN) * EMU_ROW_BYTES;
    const bool has_pre = a.pre_scale != nullptr;
#pragma unroll
    for (int p = 0; p < PA; ++p) {
        f32x4 v = ra[p];
        if (has_pre) v = v * ps + pb;
        if (a.pre_relu) {
            v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
        }
        const bool ok = (okmask >> p) & 1u;
        v.x = ok ? v.x : 0.f; v.y = ok ? v.y : 0.f; v.z = ok ? v.z : 0.f; v.w = ok ? v.w : 0.f;
        split_store(buf, PLANE, emu_off(p * RPP + lrow, lk), v);
    }
#pragma unroll
    for (int p = 0; p < PB; ++p)
        if ((p + 1) * RPP <= BN || p * RPP + lrow < BN)
            split_store(buf, PLANE, emu_off(BM + p * RPP + lrow, lk), rb[p]);
}

// MF = 32: v_mfma_f32_32x32x2_f32 tiles (default).  MF = 16: v_mfma_f32_16x16x4_f32 tiles, same FLOP rate
// but 16-column granularity -- used for c_out = 48 (DenseNet growth) where a 64-wide tile wastes 25 %.
template <int BM, int BN, int WM, int WN, int MF, bool NCHW_OUT, int PREC = 0>
__global__ __launch_bounds__(WM * WN * 64) void conv_fwd_kernel(const ConvArgs a0) {
    static_assert(PREC == 0 || MF == 32, "the bf16x3 emulation uses the 32x32x16 bf16 MFMA");
    static_assert(PREC >= 0 && PREC <= 2, "PREC: 0 fp32 MFMA, 1 bf16x3, 2 bf16x3 with a single LDS buffer");
    constexpr int NT = WM * WN * 64;          // threads per workgroup (4 or 8 waves)
    constexpr int RPP = NT / 8;               // tile rows staged per pass (8 lanes x 16 B per row)
    constexpr int TM = BM / WM / MF, TN = BN / WN / MF;
    constexpr int PA = BM / RPP, PB = (BN + RPP - 1) / RPP;
    constexpr int LDS_LD = LdsLd<MF>::value;
    constexpr int NACC = MF == 32 ? 16 : 4;   // accumulator registers per tile
    typedef float acc_t __attribute__((ext_vector_type(NACC)));
    static_assert(BM % RPP == 0 && BM % (WM * MF) == 0 && BN % (WN * MF) == 0, "tile/wave layout");
    static_assert(TM >= 1 && TN >= 1 && PA >= 1 && PB >= 1, "tile/wave layout");
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* smem = reinterpret_cast<float*>(smem_raw);
    // layout: [buf][A rows BM | B rows BN][LDS_LD]
    constexpr int BUF_FLOATS = PREC == 0 ? (BM + BN) * LDS_LD : 3 * (BM + BN) * EMU_ROW_BYTES / 4;
    constexpr bool EMU_SB = PREC == 2;        // bf16x3 with ONE LDS buffer (half the footprint -> two workgroups per CU)

    // XCD-aware block remap (bijective): blocks sharing an XCD (bid % 8) get a contiguous range of
    // tiles, so neighbouring pixel tiles (shared halo rows) and the N tiles of one M tile share an L2.
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int tiles_all0 = a0.tiles_per_class * a0.n_classes;
    ConvArgs a = a0;
    if (a0.bundled) {                         // wave-uniform: re-base every per-channel pointer onto this block's bundle
        const int xcd0 = bid & 7, qq0 = nwg >> 3, rr0 = nwg & 7;
        const int swz0 = (xcd0 < rr0 ? xcd0 * (qq0 + 1) : rr0 * (qq0 + 1) + (xcd0 - rr0) * qq0) + (bid >> 3);
        const int j = (swz0 % tiles_all0) / a0.tiles_per_class;
        a.x += (size_t)j * a0.c_in_ld;
        a.y += (size_t)j * a0.c_out;
        if (a0.y2) a.y2 += (size_t)j * a0.c_out;
        if (a0.res) a.res += (size_t)j * a0.c_out;
        if (a0.pre_scale) { a.pre_scale += (size_t)j * a0.c_in_ld; a.pre_shift += (size_t)j * a0.c_in_ld; }
        if (a0.e1_scale) { a.e1_scale += (size_t)j * a0.c_out_pad; a.e1_shift += (size_t)j * a0.c_out_pad; }
        if (a0.e2_scale) { a.e2_scale += (size_t)j * a0.c_out_pad; a.e2_shift += (size_t)j * a0.c_out_pad; }
    }
    const int xcd = bid & 7, qq = nwg >> 3, rr = nwg & 7;
    const int swz = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (bid >> 3);
    // sub-pixel upconv: the grid covers 4 parity classes (py,px) of output pixels, each its own GEMM over the
    // SOURCE pixels with a 2x2 kernel (weights pre-summed per class) -- see bts_conv_desc.subpixel
    const int tiles_all = a.tiles_per_class * a.n_classes;
    const int split = swz / tiles_all;
    const int srem = swz - split * tiles_all;
    // sub-pixel: class-minor order -- the four parity classes of one source tile run back to back, so the source pixels
    // they all gather are fetched from HBM once and re-read from L2 (class-major order re-fetched them 4x, PMC);
    // bundles (disjoint channels) stay bundle-major
    const int cls = a.subpix ? (srem & 3) : srem / a.tiles_per_class;
    const int tcl = a.subpix ? (srem >> 2) : srem - cls * a.tiles_per_class;
    const int mt_idx = tcl / a.n_ntiles, nt_idx = tcl % a.n_ntiles;
    const long m0 = (long)mt_idx * BM;
    const int n0 = nt_idx * BN;
    const int spy = cls >> 1, spx = cls & 1;
    const int pad_y = a.subpix ? 1 - spy : a.pad, pad_x = a.subpix ? 1 - spx : a.pad;
    const float* __restrict__ wbase = a.w + (size_t)cls * a.c_out_pad * a.k_pad;

    const int tid = threadIdx.x;
    const int lrow = tid >> 3, lk = (tid & 7) * 4;
    const int HW = a.H * a.W;

    // per-thread A rows: decode the output pixel once (host guarantees every element offset < 2^32)
    int ay[PA], ax[PA];
    unsigned abase[PA], avalid = 0;
#pragma unroll
    for (int p = 0; p < PA; ++p) {
        const long m = m0 + p * RPP + lrow;
        const bool v = m < a.M;
        avalid |= (v ? 1u : 0u) << p;
        const long mm = v ? m : 0;
        // 32-bit unsigned divisions (the host guarantees M < 2^31): a 64-bit divide costs ~10x as many instructions, and
        // short-K layers (K = 96..324) run only 3-11 K-steps per tile, so this prologue is not negligible there
        const unsigned mu = (unsigned)mm;
        const unsigned bu = mu / (unsigned)HW;
        const unsigned yxu = mu - bu * (unsigned)HW;
        const unsigned yu = yxu / (unsigned)a.W;
        const int b = (int)bu;
        ay[p] = (int)yu * a.stride;
        ax[p] = (int)(yxu - yu * (unsigned)a.W) * a.stride;
        abase[p] = (unsigned)b * (unsigned)(a.h_in * a.w_in);
    }
    // lean-loader state: centre-pixel element offset (+ lane channel) and a tap-validity bitmask per row
    const bool fastk = (a.c_in_ld % BK) == 0 && a.ups == 0;
    unsigned fbase[PA];
    unsigned long long vmask[PA];
    const int kchunks = a.c_in_ld / BK;          // K-steps per tap (lean path)
#pragma unroll
    for (int p = 0; p < PA; ++p) {
        fbase[p] = (abase[p] + (unsigned)(ay[p] * a.w_in + ax[p])) * (unsigned)a.x_pix_stride + (unsigned)lk;
        unsigned long long vm = 0;
        if (fastk && ((avalid >> p) & 1u)) {
            for (int t = 0; t < a.ksize * a.ksize; ++t) {
                const int ky = t / a.ksize, kx = t - ky * a.ksize;
                const int yy = ay[p] + ky * a.dil - pad_y, xx = ax[p] + kx * a.dil - pad_x;
                if ((unsigned)yy < (unsigned)a.Hs && (unsigned)xx < (unsigned)a.Ws) vm |= 1ull << t;
            }
        }
        vmask[p] = vm;
    }
    unsigned wrow[PB];
#pragma unroll
    for (int p = 0; p < PB; ++p) {
        int n = n0 + p * RPP + lrow;
        n = n < a.c_out_pad ? n : a.c_out_pad - 1;     // rows past c_out_pad feed outputs that are never stored
        wrow[p] = (unsigned)n * (unsigned)a.k_pad + (unsigned)lk;
    }

    const int it0 = split * a.its_per_split;                       // this workgroup's K-step range
    // taps this tile runs (lean path, no split-K): see tile_tapmask.  Otherwise every tap.
    const bool skipping = fastk && a.tapskip != 0;
    const unsigned long long tapmask = skipping ? tile_tapmask(a.H, a.W, a.Hs, a.Ws, a.ksize, a.dil, pad_y, pad_x, a.M, m0, BM) : ~0ull;
    const int nit = skipping ? __builtin_popcountll(tapmask) * kchunks : min(a.k_pad / BK - it0, a.its_per_split);

    f32x4 ra[PA], rb[PB];
    f32x4 ps = {1.f, 1.f, 1.f, 1.f}, pb = {0.f, 0.f, 0.f, 0.f};
    unsigned okmask = 0;

    const int wv = tid >> 6, lane = tid & 63;
    const int wm = wv / WN, wn = wv % WN;
    // fragment coordinates: lane = (li, lh); MF 32: li = row in tile (0..31), lh = k half (0..1);
    //                                        MF 16: li = row in tile (0..15), lh = k quarter (0..3)
    const int li = lane & (MF - 1), lh = lane / MF;

    acc_t acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < NACC; ++r) acc[i][j][r] = 0.f;

    // Prefetch distance 2 with ONE staging register set: during step `it` the registers hold tile it+1 (its loads
    // were issued a full step ago, so they have landed even on an HBM miss): write it to the other LDS buffer,
    // then reuse the registers for the loads of tile it+2, which get a whole step of MFMAs to arrive.
    int tap = fastk ? (skipping ? __builtin_ctzll(tapmask) : it0 / kchunks) : 0;   // lean path: (tap, channel chunk) of the K-step being loaded
    int kc = fastk && !skipping ? it0 - tap * kchunks : 0;
    auto issue = [&](int t) {                     // t = local step index; global K-step = tap * kchunks + kc (lean path)
        if (fastk) {
            if (t > 0 && ++kc == kchunks) {       // next tap of this tile (tapmask is all ones unless skipping)
                kc = 0;
                const unsigned long long rest = tap < 63 ? tapmask >> (tap + 1) : 0ull;
                tap += rest ? 1 + __builtin_ctzll(rest) : 1;
            }
            issue_loads_fast<PA, PB>(a, wbase, pad_y, pad_x, tap * kchunks + kc, tap, kc, fbase, vmask, wrow, ra, rb, ps, pb, okmask, lk);
        } else {
            issue_loads<PA, PB>(a, wbase, pad_y, pad_x, it0 + t, lk, ay, ax, abase, avalid, wrow, ra, rb, ps, pb, okmask);
        }
    };
    auto stage = [&](float* dst) __attribute__((always_inline)) {
        if constexpr (PREC == 0) stage_to_lds<BM, BN, RPP, PA, PB, LDS_LD>(a, dst, lrow, lk, ra, rb, ps, pb, okmask);
        else stage_to_lds_emu<BM, BN, RPP, PA, PB>(a, reinterpret_cast<char*>(dst), lrow, lk, ra, rb, ps, pb, okmask);
    };
    issue(0);
    stage(smem);
    if (nit > 1) issue(1);
    __syncthreads();

    if constexpr (PREC == 0) {
        constexpr int KG = 256 / MF;              // k covered by one ds_read_b128 per lane set: 8 (MF 32) / 16 (MF 16)
        constexpr int NG = BK / KG;               // fragment groups per K-step: 4 / 2
        const int a_off = (wm * TM * MF + li) * LDS_LD + 4 * lh;
        const int b_off = BM * LDS_LD + (wn * TN * MF + li) * LDS_LD + 4 * lh;
        // Fragments are double-buffered in registers: group g+1 is read from LDS while group g's MFMAs issue, and
        // the first group of the NEXT K-step is read right after the barrier.
        f32x4 fa[2][TM], fb[2][TN];
        auto read_frags = [&](const float* base, int g, int slot) {
    #pragma unroll
            for (int i = 0; i < TM; ++i) fa[slot][i] = *reinterpret_cast<const f32x4*>(base + a_off + i * MF * LDS_LD + KG * g);
    #pragma unroll
            for (int j = 0; j < TN; ++j) fb[slot][j] = *reinterpret_cast<const f32x4*>(base + b_off + j * MF * LDS_LD + KG * g);
        };
        read_frags(smem, 0, 0);

        for (int it = 0; it < nit; ++it) {
            const int buf = it & 1;
            const float* cur = smem + buf * BUF_FLOATS;
    #pragma unroll
            for (int g = 0; g < NG; ++g) {
                if (g + 1 < NG) read_frags(cur, g + 1, (g + 1) & 1);
                // q outermost: consecutive MFMAs go to different accumulators (no back-to-back dependent issue)
    #pragma unroll
                for (int q = 0; q < 4; ++q)
    #pragma unroll
                    for (int i = 0; i < TM; ++i)
    #pragma unroll
                        for (int j = 0; j < TN; ++j) {
                            // NCHW_OUT: rows = channels, lanes = pixels; else rows = pixels, lanes = channels
                            const float ra_ = NCHW_OUT ? fb[g & 1][j][q] : fa[g & 1][i][q];
                            const float rb_ = NCHW_OUT ? fa[g & 1][i][q] : fb[g & 1][j][q];
                            if constexpr (MF == 32) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(ra_, rb_, acc[i][j], 0, 0, 0);
                            else                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(ra_, rb_, acc[i][j], 0, 0, 0);
                        }
                if (g == 0) {     // after the first MFMA group: the staging VALU / ds_writes / address math issue in the
                                  // shadow of the remaining groups; the new loads still get ~a full step to land
                    if (it + 1 < nit) stage(smem + (buf ^ 1) * BUF_FLOATS);
                    if (it + 2 < nit) issue(it + 2);
                }
            }
            __syncthreads();
            if (it + 1 < nit) read_frags(smem + (buf ^ 1) * BUF_FLOATS, 0, 0);
        }
    } else {
        // bf16x3: per k16 step every 32x32 block takes six bf16 MFMAs on the (h, m, l) planes
        constexpr int PLANE = (BM + BN) * EMU_ROW_BYTES;
        // fragment of k16 step ks: 8 bf16 at k = 16*ks + 8*lh -> chunk (2*ks + lh), swizzled with the row's bits 2-3
        const int a_row = (wm * TM * 32 + li) * EMU_ROW_BYTES, b_row = (BM + wn * TN * 32 + li) * EMU_ROW_BYTES;
        const int sw = (li >> 2) & 3;
        for (int it = 0; it < nit; ++it) {
            const int buf = EMU_SB ? 0 : (it & 1);
            const char* cur = reinterpret_cast<const char*>(smem + buf * BUF_FLOATS);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                u32x4 fa[3][TM], fb[3][TN];
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
                    for (int i = 0; i < TM; ++i)
                        fa[pl][i] = *reinterpret_cast<const u32x4*>(cur + pl * PLANE + a_row + i * 32 * EMU_ROW_BYTES + (((2 * ks + lh) ^ sw) << 4));
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        fb[pl][j] = *reinterpret_cast<const u32x4*>(cur + pl * PLANE + b_row + j * 32 * EMU_ROW_BYTES + (((2 * ks + lh) ^ sw) << 4));
                }
                // product order hh, hm, mh, hl, lh, mm; the pair index outermost so consecutive MFMAs hit different
                // accumulators
                constexpr int PA_[6] = {0, 0, 1, 0, 2, 1}, PB_[6] = {0, 1, 0, 2, 0, 1};
#pragma unroll
                for (int t = 0; t < 6; ++t)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j) {
                            const u32x4 xa = NCHW_OUT ? fb[PB_[t]][j] : fa[PA_[t]][i];
                            const u32x4 xb = NCHW_OUT ? fa[PA_[t]][i] : fb[PB_[t]][j];
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, xa),
                                                                                __builtin_bit_cast(bf16x8, xb), acc[i][j], 0, 0, 0);
                        }
                if (!EMU_SB && ks == 0) {
                    if (it + 1 < nit) stage(smem + (buf ^ 1) * BUF_FLOATS);
                    if (it + 2 < nit) issue(it + 2);
                }
            }
            __syncthreads();
            if (EMU_SB) {                      // everyone has read the tile: overwrite it with the next one
                if (it + 1 < nit) stage(smem);
                if (it + 2 < nit) issue(it + 2);
                __syncthreads();
            }
        }
    }

    // ---------------------------------------------------------------- epilogue
    // D register r of lane (li, lh) is D[row][col = li] with row = (r&3) + 8*(r>>2) + 4*lh (MF 32) or 4*lh + r (MF 16)
    if (a.ksplit > 1) {                                  // partial sums only: ws[split][m][n]
        float* __restrict__ wsp = a.ws + (size_t)split * a.M * a.ws_ld;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < NACC; ++r) {
                    const int drow = MF == 32 ? (r & 3) + 8 * (r >> 2) + 4 * lh : 4 * lh + r;
                    const long m = m0 + (wm * TM + i) * MF + (NCHW_OUT ? li : drow);
                    const int n = n0 + (wn * TN + j) * MF + (NCHW_OUT ? drow : li);
                    if (m < a.M && n < a.c_out) wsp[m * a.ws_ld + n] = acc[i][j][r];
                }
        return;
    }
    {   // compact epilogues (flat pixel tiling: a row tile is MF consecutive output pixels unless sub-pixel scatters them)
        const int wms = __builtin_amdgcn_readfirstlane(wm), wns = __builtin_amdgcn_readfirstlane(wn);
        if constexpr (!NCHW_OUT) {
            if (!a.subpix) {
                long pix0[TM];
                int nvalid[TM];
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    pix0[i] = m0 + (wms * TM + i) * MF;
                    const long left = a.M - pix0[i];
                    nvalid[i] = left >= MF ? MF : (left > 0 ? (int)left : 0);
                }
                if (fast_epilogue_nhwc<TM, TN, MF, NACC>(a, acc, pix0, nvalid, 1, n0 + wns * TN * MF, li, lh)) return;
            }
        }
    }
    const bool has_e1 = a.e1_scale != nullptr, has_e2 = a.e2_scale != nullptr;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            if (!NCHW_OUT) {
                const int n = n0 + (wn * TN + j) * MF + li;
                float s1 = 1.f, b1 = 0.f, s2 = 1.f, b2 = 0.f;
                const bool nok = n < a.c_out;
                if (has_e1 && n < a.c_out_pad) { s1 = a.e1_scale[n]; b1 = a.e1_shift[n]; }
                if (has_e2 && n < a.c_out_pad) { s2 = a.e2_scale[n]; b2 = a.e2_shift[n]; }
#pragma unroll
                for (int r = 0; r < NACC; ++r) {
                    const int drow = MF == 32 ? (r & 3) + 8 * (r >> 2) + 4 * lh : 4 * lh + r;
                    const long m = m0 + (wm * TM + i) * MF + drow;
                    float v = acc[i][j][r];
                    if (has_e1) v = v * s1 + b1;
                    if (a.res != nullptr && nok && m < a.M) v += a.res[m * a.res_pix_stride + n];
                    v = apply_act(v, a.act);
                    if (has_e2) v = v * s2 + b2;
                    if (nok && m < a.M) {
                        long op = m;
                        if (a.subpix) {            // source pixel (b,Y,X) -> output pixel (b, 2Y+py, 2X+px)
                            const unsigned mu = (unsigned)m;
                            const int b = (int)(mu / (unsigned)HW), yx = (int)(mu - (unsigned)b * (unsigned)HW);
                            const int Y = (int)((unsigned)yx / (unsigned)a.W), X = yx - Y * a.W;
                            op = ((long)b * (2 * a.H) + 2 * Y + spy) * (2 * a.W) + 2 * X + spx;
                        }
                        a.y[op * a.y_pix_stride + n] = v;
                        if (a.y2) a.y2[op * a.y2_pix_stride + n] = v;
                    }
                }
            } else {
                const long m = m0 + (wm * TM + i) * MF + li;
                const bool mok = m < a.M;
                const long mm = mok ? m : 0;
                const long b = (long)((unsigned)mm / (unsigned)HW), yx = (long)((unsigned)mm - (unsigned)b * (unsigned)HW);
#pragma unroll
                for (int r = 0; r < NACC; ++r) {
                    const int drow = MF == 32 ? (r & 3) + 8 * (r >> 2) + 4 * lh : 4 * lh + r;
                    const int n = n0 + (wn * TN + j) * MF + drow;
                    float v = acc[i][j][r];
                    if (n < a.c_out_pad) {
                        if (has_e1) v = v * a.e1_scale[n] + a.e1_shift[n];
                        v = apply_act(v, a.act);
                        if (has_e2) v = v * a.e2_scale[n] + a.e2_shift[n];
                    }
                    if (mok && n < a.c_out) a.y[(b * a.c_out + n) * HW + yx] = v;
                }
            }
        }
}

// bts_conv_plan_f32 runs the real dispatch with this set: launch_* then report their choice instead of launching
struct ConvChoice { int kind, bm, bn, ksplit; long ksteps_issued = 0, ksteps_dense = 0; };   // ksteps: tap-steps over all row tiles (tap skipping)
thread_local ConvChoice* g_dry = nullptr;

#include "conv_halo.inc"
#include "conv_halo_emu.inc"
#include "conv_wino.inc"
#include "conv_1x1.inc"
#include "conv_stem.inc"

// Second pass of a split-K convolution: out = E(sum_s ws[s][m][n]) in a FIXED order (deterministic), then the
// same epilogue / destinations as the fused path.
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const ConvArgs a, int nchw) {
    const long total = a.M * (long)a.c_out;
    const long HW = (long)a.H * a.W;
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
        const long m = t / a.c_out;
        const int n = (int)(t - m * a.c_out);
        float v = 0.f;
        for (int sidx = 0; sidx < a.ksplit; ++sidx) v += a.ws[((size_t)sidx * a.M + m) * a.ws_ld + n];
        if (a.e1_scale) v = v * a.e1_scale[n] + a.e1_shift[n];
        if (a.res) v += a.res[m * a.res_pix_stride + n];
        v = apply_act(v, a.act);
        if (a.e2_scale) v = v * a.e2_scale[n] + a.e2_shift[n];
        if (nchw) {
            const long b = m / HW, yx = m % HW;
            a.y[(b * a.c_out + n) * HW + yx] = v;
        } else {
            a.y[m * a.y_pix_stride + n] = v;
            if (a.y2) a.y2[m * a.y2_pix_stride + n] = v;
        }
    }
}

// Split K over several workgroups when the grid would leave most of the 256 CUs idle (M-starved deep encoder layers)
// and the caller lent a workspace.  The decision is a function of the PER-FRAME geometry only (H*W, c_out) -- sized
// for the nominal 8-frame sub-batch -- never of the batch size: an output element's summation order, hence its bits,
// must not depend on how many frames share the launch (frames are independent, bts.py:223-293).
inline bool wants_split(const ConvArgs& a) {
    if (a.n_classes != 1 || a.ws == nullptr || knobs().split_max <= 1) return false;
    const long tiles64 = (((long)a.fill_frames * a.H * a.W + 63) / 64) * a.n_ntiles;
    return tiles64 < knobs().split_below;
}

// The split factor launch_conv will use (1 = no split): the one place that decides it, so the dispatch can ask "will this
// layer really split?" before it picks a kernel family.  a.n_ntiles must be set; ws_floats = the workspace the caller lent.
inline int split_factor(const ConvArgs& a, long ws_floats) {
    if (!wants_split(a)) return 1;
    const int nit_all = a.k_pad / BK;
    const long tiles64 = (((long)a.fill_frames * a.H * a.W + 63) / 64) * a.n_ntiles;     // 64-row tiles of a nominal launch
    long sp = knobs().split_target / tiles64;
    if (sp > knobs().split_max) sp = knobs().split_max;
    if (sp > nit_all / 4) sp = nit_all / 4;
    const long ws_ld = (a.c_out + 3) & ~3;
    if (sp <= 1 || sp * a.M * ws_ld > ws_floats) return 1;
    return (int)sp;
}

template <int BM, int BN, int WM, int WN, int MF = 32, int PREC = 0>
int launch_conv(const ConvArgs& a0, bool nchw, hipStream_t s, long ws_floats) {
    ConvArgs a = a0;
    const long n_mtiles = (a.M + BM - 1) / BM;
    a.n_ntiles = (a.c_out + BN - 1) / BN;            // tiles over REAL channels; wrow clamps into c_out_pad
    a.tiles_per_class = (int)(n_mtiles * a.n_ntiles);
    const long tiles = n_mtiles * a.n_ntiles * a.n_classes;
    // split-K when the grid would leave most of the 256 CUs idle (M-starved deep encoder layers) and the caller
    // lent a workspace.  The split factor is a function of the PER-FRAME geometry only (H*W, c_out, K) -- sized for
    // the nominal 8-frame sub-batch -- never of the batch size: an output element's summation order, hence its
    // bits, must not depend on how many frames share the launch (frames are independent, bts.py:223-293).
    const int nit_all = a.k_pad / BK;
    a.ksplit = 1; a.its_per_split = nit_all; a.ws_ld = (a.c_out + 3) & ~3;
    if (const int sp = split_factor(a, ws_floats); sp > 1) {
        a.its_per_split = (nit_all + sp - 1) / sp;
        a.ksplit = (nit_all + a.its_per_split - 1) / a.its_per_split;     // no empty splits
    }
    const long nwg = tiles * a.ksplit;
    if (nwg > 0x7fffffffL) return BTS_ERR_INVALID;
    const bool lean = (a.c_in_ld % BK) == 0 && a.ups == 0;
    a.tapskip = (knobs().tapskip && lean && a.ksplit == 1 && a.stride == 1 && a.ksize > 1 && a.ksize * a.ksize <= 49) ? 1 : 0;
    if (g_dry) {
        *g_dry = ConvChoice{0, BM, BN, a.ksplit};
        const long taps = (long)a.ksize * a.ksize;
        g_dry->ksteps_dense = n_mtiles * a.n_classes * taps;
        g_dry->ksteps_issued = g_dry->ksteps_dense;
        if (a.tapskip) {                           // the kernel's own tile rule, summed over the row tiles (x classes)
            long issued = 0;
            for (int cls = 0; cls < a.n_classes; ++cls) {
                const int pad_y = a.subpix ? 1 - (cls >> 1) : a.pad, pad_x = a.subpix ? 1 - (cls & 1) : a.pad;
                for (long mt = 0; mt < n_mtiles; ++mt)
                    issued += __builtin_popcountll(tile_tapmask(a.H, a.W, a.Hs, a.Ws, a.ksize, a.dil, pad_y, pad_x, a.M, mt * BM, BM));
            }
            g_dry->ksteps_issued = issued;
        }
        return 0;
    }
    size_t lds = PREC == 0 ? (size_t)2 * (BM + BN) * LdsLd<MF>::value * sizeof(float)
                           : (size_t)(PREC == 2 ? 1 : 2) * 3 * (BM + BN) * EMU_ROW_BYTES;
    if ((size_t)knobs().lds_bytes > lds && knobs().lds_bytes <= 160 * 1024) lds = (size_t)knobs().lds_bytes;
    hipError_t e;
    if (nchw) {
        auto k = conv_fwd_kernel<BM, BN, WM, WN, MF, true, PREC>;
        static std::atomic<unsigned long long> lds_set{0};    // per instantiation: one bit per device (bts_ensure_dynamic_lds)
        if ((e = bts_ensure_dynamic_lds((const void*)k, lds, lds_set)) != hipSuccess) return (int)e;
        hipLaunchKernelGGL(k, dim3((unsigned)nwg), dim3(WM * WN * 64), lds, s, a);
    } else {
        auto k = conv_fwd_kernel<BM, BN, WM, WN, MF, false, PREC>;
        static std::atomic<unsigned long long> lds_set{0};
        if ((e = bts_ensure_dynamic_lds((const void*)k, lds, lds_set)) != hipSuccess) return (int)e;
        hipLaunchKernelGGL(k, dim3((unsigned)nwg), dim3(WM * WN * 64), lds, s, a);
    }
    if (a.ksplit > 1) {
        const long total = a.M * (long)a.c_out;
        long blocks = (total + 255) / 256;
        if (blocks > 256L * 8) blocks = 256L * 8;
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, s, a, nchw ? 1 : 0);
    }
    return (int)hipGetLastError();
}

// Tile choice.  BN follows c_out; BM trades per-tile efficiency against wave quantisation: the chip
// finishes ceil(n_wg / 256 CUs) "tile rounds", so 836 tiles of 128 rows cost 4 rounds (82 % busy)
// while 1672 tiles of 64 rows cost 7 half-rounds (93 %).  The 64-row tile pays ~5 % more staging.
void choose_tile(long M, int c_out, int* bm, int* bn) {
    // BN: least padded width, weighted by how well each tile shape runs.  48 = three 16x16x4 MFMA tiles.
    const int cand[4] = {128, 64, 32, 48};
    const double eff[4] = {1.0, 1.05, 1.2, 1.12};
    double best = 1e30;
    for (int i = 0; i < 4; ++i) {
        if (cand[i] == 48 && c_out % 48 != 0) continue;
        const double cost = (double)((c_out + cand[i] - 1) / cand[i] * cand[i]) * eff[i];
        if (cost < best) { best = cost; *bn = cand[i]; }
    }
    if (knobs().no48 && *bn == 48) *bn = 64;
    *bm = 128;
    if (*bn != 32) {
        const long nt = (c_out + *bn - 1) / *bn;
        const long wg128 = ((M + 127) / 128) * nt, wg64 = ((M + 63) / 64) * nt;
        const double t128 = (double)((wg128 + 255) / 256) * 128.0;
        const double t64 = (double)((wg64 + 255) / 256) * 64.0 * 1.05;
        if (t64 < t128) *bm = 64;
    }
    {                                                  // tuning aid: force the row tile (64 / 128)
        const int v = knobs().force_bm;
        if ((v == 64 && *bn != 32) || v == 128) *bm = v;
    }
}

}  // namespace

namespace {
int conv_dispatch(const bts_conv_desc* d, bts_stream_t stream);
}

extern "C" int bts_conv_fwd_f32(const bts_conv_desc* d, bts_stream_t stream) { return conv_dispatch(d, stream); }

namespace {
int conv_dispatch(const bts_conv_desc* d, bts_stream_t stream) {
    if (!d || !d->x || !d->w || !d->y) return BTS_ERR_INVALID;
    if (d->B <= 0 || d->h_in <= 0 || d->w_in <= 0 || d->c_out <= 0) return BTS_ERR_INVALID;
    if (d->up != 1 && d->up != 2) return BTS_ERR_UNSUPPORTED;
    if (d->subpixel) {   // 3x3 conv on a nearest-2x upsampled input, as four 2x2 convs on the source
        if (d->ksize != 2 || d->up != 1 || d->stride != 1 || d->dil != 1 || d->y_nchw) return BTS_ERR_INVALID;
    } else if (d->ksize < 1 || d->ksize > 7 || !(d->ksize & 1)) return BTS_ERR_UNSUPPORTED;
    if (d->dil < 1 || d->stride < 1 || d->pad < 0) return BTS_ERR_INVALID;
    if (d->up == 2 && d->stride != 1) return BTS_ERR_UNSUPPORTED;
    if (d->c_in_ld <= 0 || (d->c_in_ld & 3) || (d->k_pad % BK) || d->k_pad < d->ksize * d->ksize * d->c_in_ld)
        return BTS_ERR_INVALID;
    if (d->k_pad >= 65536) return BTS_ERR_UNSUPPORTED;
    if ((d->x_pix_stride & 3) || d->x_pix_stride < d->c_in_ld - (d->n_tail > 0 ? 4 : 0)) return BTS_ERR_INVALID;
    if ((d->c_out_pad & 31) || d->c_out_pad < d->c_out) return BTS_ERR_INVALID;
    if (((uintptr_t)d->x & 15) || ((uintptr_t)d->w & 15)) return BTS_ERR_INVALID;
    if ((d->pre_scale && (((uintptr_t)d->pre_scale & 15) || !d->pre_shift || ((uintptr_t)d->pre_shift & 15))))
        return BTS_ERR_INVALID;
    if ((d->e1_scale && !d->e1_shift) || (d->e2_scale && !d->e2_shift)) return BTS_ERR_INVALID;
    if (!d->y_nchw && d->y_pix_stride < d->c_out) return BTS_ERR_INVALID;
    if (d->act < 0 || d->act > 3) return BTS_ERR_INVALID;
    if (d->w_split && ((uintptr_t)d->w_split & 15)) return BTS_ERR_INVALID;
    if (d->w_wino && ((uintptr_t)d->w_wino & 15)) return BTS_ERR_INVALID;
    // the kernel addresses both operands with 32-bit element offsets
    if ((double)d->B * d->h_in * d->w_in * (double)d->x_pix_stride >= 4294967296.0) return BTS_ERR_UNSUPPORTED;
    if ((double)d->c_out_pad * (double)d->k_pad * (d->subpixel ? 4.0 : 1.0) >= 4294967296.0) return BTS_ERR_UNSUPPORTED;

    ConvArgs a;
    a.x = d->x; a.x_pix_stride = d->x_pix_stride; a.c_in_ld = d->c_in_ld; a.k_pad = d->k_pad;
    a.B = d->B; a.h_in = d->h_in; a.w_in = d->w_in; a.ups = d->up == 2 ? 1 : 0;
    a.ksize = d->ksize; a.dil = d->dil; a.stride = d->stride; a.pad = d->pad; a.subpix = d->subpixel ? 1 : 0;
    a.k_flat = d->ksize * d->ksize * d->c_in_ld;
    a.magic_c = (unsigned)(4294967296ULL / (unsigned)d->c_in_ld) + 1u;
    a.magic_ks = (unsigned)(4294967296ULL / (unsigned)d->ksize) + 1u;
    a.w = d->w; a.c_out = d->c_out; a.c_out_pad = d->c_out_pad;
    a.pre_scale = d->pre_scale; a.pre_shift = d->pre_shift; a.pre_relu = d->pre_relu;
    a.e1_scale = d->e1_scale; a.e1_shift = d->e1_shift; a.act = d->act;
    a.e2_scale = d->e2_scale; a.e2_shift = d->e2_shift;
    a.y = d->y; a.y_pix_stride = d->y_pix_stride;
    a.y2 = d->y2; a.y2_pix_stride = d->y2_pix_stride;
    if (d->y2 && (d->y_nchw || d->y2_pix_stride < d->c_out)) return BTS_ERR_INVALID;
    a.Hs = d->h_in * d->up; a.Ws = d->w_in * d->up;
    a.H = (a.Hs + 2 * d->pad - d->dil * (d->ksize - 1) - 1) / d->stride + 1;
    a.W = (a.Ws + 2 * d->pad - d->dil * (d->ksize - 1) - 1) / d->stride + 1;
    if (a.subpix) { a.H = d->h_in; a.W = d->w_in; }      // the GEMM's M walks SOURCE pixels, per parity class
    if (a.H <= 0 || a.W <= 0) return BTS_ERR_INVALID;
    a.M = (long)d->B * a.H * a.W;
    if (a.M >= 2147483648L) return BTS_ERR_UNSUPPORTED;          // the kernel decodes pixel indices with 32-bit arithmetic
    a.n_ntiles = 0; a.tiles_per_class = 0;
    a.n_classes = a.subpix ? 4 : 1; a.bundled = 0;
    a.res = d->res; a.res_pix_stride = d->res_pix_stride;
    a.n_tail = d->n_tail;
    if (d->fill_frames < 0 || d->fill_frames > 4096) return BTS_ERR_INVALID;
    a.fill_frames = d->fill_frames > 0 ? d->fill_frames : knobs().fill_frames;
    for (int j = 0; j < 4; ++j) a.tail[j] = d->n_tail > 0 ? d->tail_planes[j < d->n_tail ? j : 0] : nullptr;
    if (d->n_tail < 0 || d->n_tail > 4) return BTS_ERR_INVALID;
    if (d->n_tail > 0) {
        // the planes supply channels [c_in_ld-4, c_in_ld-4+n_tail) of the packed K axis; only the halo-tile kernel reads them
        if (d->ksize != 3 || d->pad != 1 || d->stride != 1 || d->dil != 1 || d->up != 1 || d->subpixel || d->n_bundles > 1)
            return BTS_ERR_UNSUPPORTED;
        if (d->c_in_ld < 8 || d->x_pix_stride < d->c_in_ld - 4) return BTS_ERR_INVALID;
        for (int j = 0; j < d->n_tail; ++j) if (!d->tail_planes[j] || ((uintptr_t)d->tail_planes[j] & 3)) return BTS_ERR_INVALID;
    }
    if (d->res && (d->y_nchw || d->res_pix_stride < d->c_out)) return BTS_ERR_INVALID;
    if (d->n_bundles > 1) {
        if (d->subpixel || d->y_nchw || d->up != 1) return BTS_ERR_INVALID;
        if (d->c_out_pad != d->c_out) return BTS_ERR_INVALID;                 // bundle outputs tile the channel axis exactly
        if (d->x_pix_stride < (long)d->n_bundles * d->c_in_ld || d->y_pix_stride < (long)d->n_bundles * d->c_out)
            return BTS_ERR_INVALID;
        if (d->y2 && d->y2_pix_stride < (long)d->n_bundles * d->c_out) return BTS_ERR_INVALID;
        if (d->res && d->res_pix_stride < (long)d->n_bundles * d->c_out) return BTS_ERR_INVALID;
        if ((double)d->n_bundles * d->c_out_pad * (double)d->k_pad >= 4294967296.0) return BTS_ERR_UNSUPPORTED;
        a.n_classes = d->n_bundles; a.bundled = 1;
    } else if (d->n_bundles < 0) return BTS_ERR_INVALID;
    a.ksplit = 1; a.its_per_split = 0; a.ws_ld = 0; a.tapskip = 0; a.halo_single_a = 0; a.stagger = 0; a.w_split = d->w_split; a.w_wino = d->w_wino;
    a.ws = d->splitk_ws;
    const long wsf = d->splitk_ws ? d->splitk_ws_floats : 0;
    if (d->splitk_ws && (((uintptr_t)d->splitk_ws & 15) || d->splitk_ws_floats < 0)) return BTS_ERR_INVALID;
    hipStream_t s = (hipStream_t)stream;
    const bool nchw = d->y_nchw != 0;
    int bm, bn;
    choose_tile(a.M * a.n_classes, d->c_out, &bm, &bn);
    // precision: 0 = v_mfma_f32_32x32x2_f32 (exact fp32 products), 1 = fp32 emulated on the bf16 matrix cores
    // (three-way split, six products; see split_store).  BTS_CONV_PRECISION overrides the descriptor (A/B runs).
    const int prec_env = knobs().precision;
    const int prec = prec_env >= 0 ? prec_env : d->precision;
    if (prec != 0 && prec != 1) return BTS_ERR_INVALID;
    // fused Winograd F(2x2,3x3) (conv_wino.inc), planar-tail layers included (conv3 / conv2: NHWC output)
    if (a.n_tail > 0 && prec == 0 && knobs().wino && (bn == 128 || bn == 64) && a.c_out_pad % (bn == 128 ? 128 : 64) == 0 && wino_eligible(a, nchw))
        return bn == 128 ? launch_wino<128, true>(a, s) : launch_wino<64, true>(a, s);
    if (a.n_tail > 0) {   // planar tail operand: always the halo-tile kernel (fp32-input MFMA whatever `precision` says)
        if (bn == 128) return launch_halo<128, 4, 2, 32, 3, true>(a, nchw, s);
        if (bn == 32) return launch_halo<32, 4, 1, 32, 3, true>(a, nchw, s);
        return launch_halo<64, 4, 2, 32, 3, true>(a, nchw, s);
    }
    if (prec == 1) {
        if (bn == 48) bn = 64;                       // the 16x16x4 48-wide tile has no bf16x3 twin: pad to 64
        // stride-1 3x3 / sub-pixel 2x2 convolutions on maps that tile well, whole 32-channel chunks, pre-split weights:
        // the bf16x3 halo-tile kernel (conv_halo_emu.inc); same geometry-only gating as the fp32 halo kernel below
        if (knobs().halo && knobs().halo_emu && (bn == 128 || bn == 64) && halo_emu_eligible(a, nchw)) {
            ConvArgs probe = a;
            probe.n_ntiles = (a.c_out + bn - 1) / bn;
            const long halo_wgs = (long)a.fill_frames * ((a.H + 3) / 4) * ((a.W + 31) / 32) * probe.n_ntiles * a.n_classes;
            if (knobs().halo >= 2 || split_factor(probe, wsf) <= 1 || halo_wgs >= knobs().halo_fill) {
                if (a.subpix) return bn == 128 ? launch_halo_emu<128, 2>(a, s) : launch_halo_emu<64, 2>(a, s);
                return bn == 128 ? launch_halo_emu<128, 3>(a, s) : launch_halo_emu<64, 3>(a, s);
            }
        }
        // LDS buffering: the planes take 6 B per element, and with ONE buffer per workgroup (two barriers per K-step,
        // the other resident workgroup fills the gaps) every tile runs faster than double-buffered with fewer
        // workgroups per CU (total over the decoder layers 134.7 vs 126.6 TFLOP/s-equivalent).  BTS_CONV_EMU_SB=0 = double.
        // (Forcing a third workgroup per CU with an 80-VGPR cap on the narrow tiles: same layer rates, whole model
        // 39.7 vs 38.3 ms -- dropped.)
        const int emu_sb_env = knobs().emu_sb;
        const int emu_sb = emu_sb_env >= 0 ? emu_sb_env : 1;
        if (emu_sb) {                                  // one LDS buffer: half the footprint, two workgroups per CU
            if (bn == 128) return bm == 128 ? launch_conv<128, 128, 2, 4, 32, 2>(a, nchw, s, wsf) : launch_conv<64, 128, 2, 4, 32, 2>(a, nchw, s, wsf);
            if (bn == 64) return bm == 128 ? launch_conv<128, 64, 4, 2, 32, 2>(a, nchw, s, wsf) : launch_conv<64, 64, 2, 2, 32, 2>(a, nchw, s, wsf);
            return launch_conv<128, 32, 4, 1, 32, 2>(a, nchw, s, wsf);
        }
        if (bn == 128) return bm == 128 ? launch_conv<128, 128, 2, 4, 32, 1>(a, nchw, s, wsf) : launch_conv<64, 128, 2, 4, 32, 1>(a, nchw, s, wsf);
        if (bn == 64) return bm == 128 ? launch_conv<128, 64, 4, 2, 32, 1>(a, nchw, s, wsf) : launch_conv<64, 64, 2, 2, 32, 1>(a, nchw, s, wsf);
        return launch_conv<128, 32, 4, 1, 32, 1>(a, nchw, s, wsf);
    }
    // the encoder stem (7x7 / stride 2 on the 3-channel image): its own kernel (conv_stem.inc)
    if (knobs().stem && stem_eligible(a, nchw, prec)) return a.c_out == 96 ? launch_stem<96>(a, s) : launch_stem<64>(a, s);
    // plain 1x1 convolutions with a wide output: one workgroup per 128 (or 64) pixels x 192 channels (conv_1x1.inc).
    // Layers that will really split K stay on the row-tiled kernel (split_factor, not the mere tile-count threshold:
    // DenseNet block 3's bottlenecks sit under the threshold but end up with a split factor of 1).
    if (knobs().k1x1 && a.ksize == 1) {
        const int wide = a.c_out % 192 == 0 ? 192 : (knobs().k1x1 == 2 ? (a.c_out % 256 == 0 ? 256 : (a.c_out % 128 == 0 ? 128 : 0)) : 0);
        ConvArgs probe = a;
        probe.n_ntiles = (a.c_out + bn - 1) / bn;
        if (wide && split_factor(probe, wsf) <= 1 && conv1x1_eligible(a, nchw, wide)) {
            if (wide == 192) {
                // row tile by K (per-layer geometry, never the batch): a short K loop cannot amortise a 128x192 tile's
                // prologue and 96 KB of output stores with nothing else resident on the CU -- the 64-row four-wave tile
                // (74 KB of LDS: two workgroups per CU) overlaps them; from ~24 K-steps on the fatter tile's lower
                // staging-per-MFMA wins (measured at B=16: block 2, K 192..720: 2.62 -> 2.33 ms with 64 rows;
                // block 3, K 384..2064: 4.86 ms with 128 rows, 5.06 with 64)
                const int rows = knobs().k1x1_rows ? knobs().k1x1_rows : (a.c_in_ld <= 768 ? 64 : 128);
                if (rows == 64) return knobs().k1x1_sb ? launch_conv1x1<192, 2, true>(a, s) : launch_conv1x1<192, 2>(a, s);
                return launch_conv1x1<192>(a, s);
            }
            if (wide == 256) return launch_conv1x1<256>(a, s);
            return launch_conv1x1<128>(a, s);
        }
    }
    // fused Winograd F(2x2,3x3) (conv_wino.inc): same geometry-only gating as the halo kernel (never instead of split-K
    // unless the declared launch fills the chip)
    if (knobs().wino && ((bn == 128 && a.c_out_pad % 128 == 0) || (bn == 64 && a.c_out_pad % 64 == 0) || (bn == 48 && a.c_out % 48 == 0)) &&
        wino_eligible(a, nchw)) {
        ConvArgs probe = a;
        probe.n_ntiles = (a.c_out + bn - 1) / bn;
        const long wgs = (long)a.fill_frames * ((a.H + 7) / 8) * ((a.W + 15) / 16) * ((a.c_out + 127) / 128);
        if (split_factor(probe, wsf) <= 1 || wgs >= knobs().halo_fill)
            return bn == 128 ? launch_wino<128>(a, s) : (bn == 64 ? launch_wino<64>(a, s) : launch_wino<48>(a, s));
    }
    // stride-1 3x3 (and sub-pixel 2x2) convolutions on maps that tile well: the halo-tile kernel (conv_halo.inc).  The
    // choice depends on per-frame geometry and the DECLARED frames per launch only (never on B), like the split-K
    // decision.  Where a layer would split K, the halo kernel still wins once the declared launch brings enough spatial
    // tiles (DenseNet block 3 at fill_frames 16: 240 workgroups, 83 -> 55 us per layer at B=16).  The threshold (200) sits
    // ABOVE the library's default fill_frames of 8 (120 workgroups) on purpose: a caller that never declares anything and
    // runs one frame per call (bts_test.py) would get 15 workgroups per launch from it -- split-K fills the chip 7x
    // better there (whole forward at batch 1: 8.9 ms with split-K, 12.0 ms with the halo / wide-tile choices).
    if (knobs().halo) {
        ConvArgs probe = a;
        probe.n_ntiles = (a.c_out + bn - 1) / bn;
        const int mf = bn == 48 ? 16 : 32;
        const long halo_wgs = (long)a.fill_frames * ((a.H + 128 / mf - 1) / (128 / mf)) * ((a.W + mf - 1) / mf) * probe.n_ntiles * a.n_classes;
        if ((knobs().halo >= 2 || split_factor(probe, wsf) <= 1 || halo_wgs >= knobs().halo_fill) && halo_eligible(a, true, mf, nullptr)) {
            if (a.subpix) {
                if (bn == 128) return knobs().halo_sb ? launch_halo<128, 4, 2, 32, 2, false, true>(a, nchw, s) : launch_halo<128, 4, 2, 32, 2>(a, nchw, s);
                if (bn == 64) return launch_halo<64, 4, 2, 32, 2>(a, nchw, s);
                if (bn == 32) return knobs().halo_sb ? launch_halo<32, 4, 1, 32, 2, false, true>(a, nchw, s) : launch_halo<32, 4, 1, 32, 2>(a, nchw, s);
            } else if (a.dil != 1) {
                // dilated halo tiles (ASPP branches, c_out 128): one eight-wave workgroup per CU, single weight buffer
                if (bn == 128 && !nchw) {
                    if (a.dil == 3) return launch_halo<128, 4, 2, 32, 3, false, true, 3>(a, nchw, s);
                    if (a.dil == 6) return launch_halo<128, 4, 2, 32, 3, false, true, 6>(a, nchw, s);
                    if (a.dil == 12) return launch_halo<128, 4, 2, 32, 3, false, true, 12>(a, nchw, s);
                }
            } else {
                if (bn == 128) return knobs().halo_sb ? launch_halo<128, 4, 2, 32, 3, false, true>(a, nchw, s) : launch_halo<128, 4, 2, 32, 3>(a, nchw, s);
                if (bn == 64) return launch_halo<64, 4, 2, 32, 3>(a, nchw, s);
                if (bn == 32) return knobs().halo_sb ? launch_halo<32, 4, 1, 32, 3, false, true>(a, nchw, s) : launch_halo<32, 4, 1, 32, 3>(a, nchw, s);
                if (bn == 48) {
                    // eight waves per workgroup (one 16-pixel row each) instead of four: 16 instead of 8 waves per CU at two
                    // workgroups per CU.  Worth -10 % where the launch is under-filled (DenseNet block 3: 240 workgroups at
                    // fill_frames 16) and -2 % on the chip-filling block 1 / 2 launches; BTS_CONV_HALO48_W8_BELOW restricts it
                    // to declared launches below that many workgroups (A/B)
                    if (knobs().halo48_w8 && halo_wgs < knobs().halo48_w8_below) return launch_halo<48, 8, 1, 16, 3>(a, nchw, s);
                    return launch_halo<48, 4, 1, 16, 3>(a, nchw, s);
                }
            }
        }
    }
    if (bn == 48) return bm == 128 ? launch_conv<128, 48, 4, 1, 16>(a, nchw, s, wsf) : launch_conv<64, 48, 4, 1, 16>(a, nchw, s, wsf);
    // 8-wave workgroups (two waves per SIMD from the same tile) for the 128-row tiles: +2 % end to end over the
    // 4-wave layout on MI355X (more waves to cover each other's staging); BTS_CONV_W8=0 selects the 4-wave kernels
    const int w8 = knobs().w8, w8s = knobs().w8s;
    if (bn == 128) {
        if (bm == 128) return w8 ? launch_conv<128, 128, 2, 4>(a, nchw, s, wsf) : launch_conv<128, 128, 2, 2>(a, nchw, s, wsf);
        return w8s ? launch_conv<64, 128, 2, 4>(a, nchw, s, wsf) : launch_conv<64, 128, 2, 2>(a, nchw, s, wsf);
    }
    if (bn == 64) {
        if (bm == 128) return w8 ? launch_conv<128, 64, 4, 2>(a, nchw, s, wsf) : launch_conv<128, 64, 2, 2>(a, nchw, s, wsf);
        return launch_conv<64, 64, 2, 2>(a, nchw, s, wsf);
    }
    return launch_conv<128, 32, 4, 1>(a, nchw, s, wsf);
}

}  // namespace

extern "C" int bts_conv_plan_f32(const bts_conv_desc* d, int* bm, int* bn, int* kind) {
    if (!d || !bm || !bn || !kind) return BTS_ERR_INVALID;
    ConvChoice c{0, 0, 0, 1};
    g_dry = &c;
    const int rc = conv_dispatch(d, nullptr);          // the real decision path; launch_* fill `c` instead of launching
    g_dry = nullptr;
    *bm = c.bm; *bn = c.bn; *kind = c.kind + (c.ksplit > 1 ? 16 : 0);
    return rc;
}

extern "C" int bts_conv_plan_ksteps_f32(const bts_conv_desc* d, long* issued, long* dense) {
    if (!d || !issued || !dense) return BTS_ERR_INVALID;
    ConvChoice c{0, 0, 0, 1};
    g_dry = &c;
    const int rc = conv_dispatch(d, nullptr);
    g_dry = nullptr;
    *issued = c.ksteps_issued; *dense = c.ksteps_dense;      // both 0 for the kernel families that never skip
    return rc;
}
