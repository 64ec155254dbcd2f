// Depth-evaluation metrics as GPU reductions.
//
// Restates the per-sample body of online_eval / compute_errors (reference pytorch/bts_main.py:87-108, 221-251;
// the same code sits in bts_eval.py:81-102, 237-307): un-crop the prediction into the ground-truth frame
// (do_kb_crop), clamp it to [min_depth_eval, max_depth_eval] with inf -> max and NaN -> min, build the valid mask
// (gt inside the range AND the Garg / Eigen crop rectangle), and reduce the nine error measures over the valid
// pixels.  The reference moves both maps to the host and runs ~25 NumPy passes per sample; here a sample is ONE
// pass over the two maps (HBM-bound: 8 B per ground-truth pixel) that leaves ten sums per frame, and a tiny second
// kernel turns them into the nine measures and adds them to the running accumulator that online_eval all-reduces
// (bts_main.py:253-260).  Everything is accumulated in fp64 and in a fixed order (no atomics): results are
// reproducible run to run and more exact than the reference's float32 NumPy reductions.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "common.h"

namespace {

constexpr int EV_NS = 10;          // n, d1, d2, d3, sum sq, sum err^2, sum |d|/gt, sum d^2/gt, sum err, sum |log10 ratio|
constexpr int EV_THREADS = 256;

struct EvalArgs {
    const float* pred; const float* gt;
    int B, Hp, Wp, Hg, Wg, top, left;
    float dmin, dmax;
    int y0, y1, x0, x1;            // crop rectangle in ground-truth coordinates, [y0,y1) x [x0,x1)
    int nblk;                      // blocks per frame
};

__global__ __launch_bounds__(EV_THREADS) void eval_partial_kernel(const EvalArgs a, double* __restrict__ ws) {
    const int b = blockIdx.y;
    const long npix = (long)a.Hg * a.Wg;
    const float* __restrict__ gt = a.gt + (long)b * npix;
    const float* __restrict__ pr = a.pred + (long)b * a.Hp * a.Wp;
    double s[EV_NS];
#pragma unroll
    for (int i = 0; i < EV_NS; ++i) s[i] = 0.0;
    for (long p = (long)blockIdx.x * EV_THREADS + threadIdx.x; p < npix; p += (long)a.nblk * EV_THREADS) {
        const int y = (int)(p / a.Wg), x = (int)(p - (long)y * a.Wg);
        const float g = gt[p];
        // valid_mask (bts_main.py:234) AND the crop rectangle (bts_main.py:236-249)
        if (!(g > a.dmin && g < a.dmax) || y < a.y0 || y >= a.y1 || x < a.x0 || x >= a.x1) continue;
        const int py = y - a.top, px = x - a.left;                       // kb-crop canvas (bts_main.py:221-227): 0 outside
        float v = (py >= 0 && py < a.Hp && px >= 0 && px < a.Wp) ? pr[(long)py * a.Wp + px] : 0.f;
        if (v < a.dmin) v = a.dmin;                                       // bts_main.py:229
        if (v > a.dmax) v = a.dmax;                                       // bts_main.py:230 (+inf lands here, :231)
        if (v != v) v = a.dmin;                                           // bts_main.py:232
        const double gd = (double)g, pd = (double)v;
        const double r1 = gd / pd, r2 = pd / gd;
        const double th = r1 > r2 ? r1 : r2;                              // bts_main.py:88
        const double d = gd - pd;
        const double err = log(pd) - log(gd);                             // bts_main.py:102
        s[0] += 1.0;
        s[1] += th < 1.25 ? 1.0 : 0.0;                                    // bts_main.py:89-91
        s[2] += th < 1.25 * 1.25 ? 1.0 : 0.0;
        s[3] += th < 1.25 * 1.25 * 1.25 ? 1.0 : 0.0;
        s[4] += d * d;                                                    // rms, bts_main.py:93-94
        s[5] += err * err;                                                // log_rms and silog, bts_main.py:96-97, 103
        s[6] += fabs(d) / gd;                                             // abs_rel, bts_main.py:99
        s[7] += d * d / gd;                                               // sq_rel, bts_main.py:100
        s[8] += err;
        s[9] += fabs(err) * 0.43429448190325182765;                       // |log10 pred - log10 gt|, bts_main.py:105-106
    }
    // fixed-order block reduction: lanes (shuffle tree), then waves (LDS)
    __shared__ double red[EV_THREADS / 64][EV_NS];
#pragma unroll
    for (int i = 0; i < EV_NS; ++i) {
        double v = s[i];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][i] = v;
    }
    __syncthreads();
    if (threadIdx.x < EV_NS) {
        double v = 0.0;
        for (int w = 0; w < EV_THREADS / 64; ++w) v += red[w][threadIdx.x];
        ws[((long)b * a.nblk + blockIdx.x) * EV_NS + threadIdx.x] = v;
    }
}

// one block per frame: sum the partials in block order, derive the nine measures (bts_main.py:87-108), and -- as
// online_eval does per sample (bts_main.py:253-254) -- add them to the running accumulator, frames in index order
__global__ __launch_bounds__(64) void eval_finalize_kernel(const double* __restrict__ ws, int B, int nblk,
                                                           double* __restrict__ per_frame, double* __restrict__ accum) {
    __shared__ double tot[EV_NS];
    for (int b = 0; b < B; ++b) {
        if (threadIdx.x < EV_NS) {
            double v = 0.0;
            for (int k = 0; k < nblk; ++k) v += ws[((long)b * nblk + k) * EV_NS + threadIdx.x];
            tot[threadIdx.x] = v;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            const double n = tot[0];
            double m[9];
            if (n > 0.0) {
                const double me = tot[8] / n;
                // silog: the variance of the log error.  In exact arithmetic E[e^2] - E[e]^2 >= 0; in floating point it can
                // come out a few ulp below zero when all log errors are (almost) equal -- one valid pixel, or a constant
                // scale error -- and sqrt would turn that into a NaN that then poisons the running accumulator: clamp
                m[0] = sqrt(fmax(tot[5] / n - me * me, 0.0)) * 100.0;
                m[1] = tot[6] / n;                            // abs_rel
                m[2] = tot[9] / n;                            // log10
                m[3] = sqrt(tot[4] / n);                      // rms
                m[4] = tot[7] / n;                            // sq_rel
                m[5] = sqrt(tot[5] / n);                      // log_rms
                m[6] = tot[1] / n; m[7] = tot[2] / n; m[8] = tot[3] / n;
            } else {
                for (int i = 0; i < 9; ++i) m[i] = 0.0;
            }
            for (int i = 0; i < 9; ++i) per_frame[(long)b * 10 + i] = m[i];
            per_frame[(long)b * 10 + 9] = n;
            // A frame whose mask leaves NO pixel (n == 0) is left out of the accumulator.  Deliberate deviation: the reference
            // skips a sample only on its has_valid_depth flag (bts_main.py:201-203); a sample that passes the flag but has
            // an empty mask would make its compute_errors (bts_main.py:87-108) average empty arrays -- NaN for all nine
            // measures, which the running sums never recover from.  per_frame still reports it, with n = 0.
            if (accum != nullptr && n > 0.0) {
                for (int i = 0; i < 9; ++i) accum[i] += m[i];
                accum[9] += 1.0;
            }
        }
        __syncthreads();
    }
}

}  // namespace

extern "C" long bts_eval_ws_doubles(int B, int Hg, int Wg) {
    if (B <= 0 || Hg <= 0 || Wg <= 0) return 0;
    long nblk = ((long)Hg * Wg + EV_THREADS * 8 - 1) / (EV_THREADS * 8);
    if (nblk > 512) nblk = 512;
    if (nblk < 1) nblk = 1;
    return (long)B * nblk * EV_NS;
}

extern "C" int bts_eval_depth_metrics_f32(const float* pred, int B, int Hp, int Wp, const float* gt, int Hg, int Wg,
                                          int top, int left, float min_depth_eval, float max_depth_eval,
                                          int y0, int y1, int x0, int x1, double* ws, long ws_doubles,
                                          double* per_frame, double* accum, bts_stream_t stream) {
    if (!pred || !gt || !ws || !per_frame || B <= 0 || Hp <= 0 || Wp <= 0 || Hg <= 0 || Wg <= 0) return BTS_ERR_INVALID;
    if (!(max_depth_eval > min_depth_eval) || y0 < 0 || x0 < 0 || y1 > Hg || x1 > Wg || y0 > y1 || x0 > x1)
        return BTS_ERR_INVALID;
    if (B > 65535) return BTS_ERR_UNSUPPORTED;
    const long need = bts_eval_ws_doubles(B, Hg, Wg);
    if (ws_doubles < need || ((uintptr_t)ws & 7) || ((uintptr_t)per_frame & 7) || ((uintptr_t)accum & 7)) return BTS_ERR_INVALID;
    EvalArgs a;
    a.pred = pred; a.gt = gt; a.B = B; a.Hp = Hp; a.Wp = Wp; a.Hg = Hg; a.Wg = Wg; a.top = top; a.left = left;
    a.dmin = min_depth_eval; a.dmax = max_depth_eval; a.y0 = y0; a.y1 = y1; a.x0 = x0; a.x1 = x1;
    a.nblk = (int)(need / ((long)B * EV_NS));
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(eval_partial_kernel, dim3((unsigned)a.nblk, (unsigned)B), dim3(EV_THREADS), 0, s, a, ws);
    hipLaunchKernelGGL(eval_finalize_kernel, dim3(1), dim3(64), 0, s, ws, B, a.nblk, per_frame, accum);
    return (int)hipGetLastError();
}
