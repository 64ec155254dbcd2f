// A C++ caller of the C ABI with no Python and no torch in the process: the drop-in boundary as a native host
// (a TensorFlow custom-op kernel, a C++ inference server) would use it.  include/bts_hip.h is the only interface.
//
//   1. bts_lpg_fwd_f32 on seeded plane equations, k = 8 / 4 / 2, against the oracle's C restatement of
//      pytorch/bts.py:149-173 (oracle/lpg_oracle.c) -- bit-exact, incl. abs_min;
//   2. bts_conv_fwd_f32: a 3x3 dilated convolution with BN+ReLU epilogue against plain host loops;
//   3. error behaviour: bad arguments come back as negative codes, never as a launch.
//
// Built by tests/cabi/Makefile (hipcc), run by tests/test_cabi_native.py on the GPU box.  Exit code 0 = all checks hold.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../../include/bts_hip.h"

extern "C" void lpg_oracle_fwd(const float* plane_eq, int B, int h, int w, int k, float* depth, float* abs_min);

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 2; } } while (0)

static unsigned lcg_state = 12345u;
static float frand() { lcg_state = lcg_state * 1664525u + 1013904223u; return (float)((lcg_state >> 8) & 0xFFFFFF) / 16777216.0f; }

static int test_lpg(int k) {
    const int B = 2, h = 5, w = 7, H = h * k, W = w * k;
    std::vector<float> pe((size_t)B * 4 * h * w), ref((size_t)B * H * W), got(ref.size());
    for (int b = 0; b < B; ++b)
        for (int i = 0; i < h * w; ++i) {
            float n1 = frand() - 0.5f, n2 = frand() - 0.5f, n3 = frand() * 0.9f + 0.1f;
            const float inv = 1.0f / std::sqrt(n1 * n1 + n2 * n2 + n3 * n3);
            if (i % 11 == 3) { n1 = 0.f; n2 = 0.f; n3 = 5e-4f / inv; }          // inside the +-1e-3 clamp band
            pe[((size_t)b * 4 + 0) * h * w + i] = n1 * inv;
            pe[((size_t)b * 4 + 1) * h * w + i] = n2 * inv;
            pe[((size_t)b * 4 + 2) * h * w + i] = n3 * inv;
            pe[((size_t)b * 4 + 3) * h * w + i] = frand() * 80.f;
        }
    float am_ref = 0.f;
    lpg_oracle_fwd(pe.data(), B, h, w, k, ref.data(), &am_ref);
    float *d_pe, *d_out, *d_am;
    CK(hipMalloc(&d_pe, pe.size() * 4)); CK(hipMalloc(&d_out, ref.size() * 4)); CK(hipMalloc(&d_am, 4));
    CK(hipMemcpy(d_pe, pe.data(), pe.size() * 4, hipMemcpyHostToDevice));
    const int rc = bts_lpg_fwd_f32(d_pe, B, h, w, k, d_out, d_am, nullptr);
    if (rc != 0) { std::printf("bts_lpg_fwd_f32 k=%d rc=%d (%s)\n", k, rc, bts_hip_error_string(rc)); return 1; }
    CK(hipDeviceSynchronize());
    float am_got = -1.f;
    CK(hipMemcpy(got.data(), d_out, got.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(&am_got, d_am, 4, hipMemcpyDeviceToHost));
    CK(hipFree(d_pe)); CK(hipFree(d_out)); CK(hipFree(d_am));
    if (std::memcmp(got.data(), ref.data(), got.size() * 4) != 0 || am_got != am_ref) {
        std::printf("LPG k=%d: not bit-exact against the C oracle (abs_min %g vs %g)\n", k, am_got, am_ref);
        return 1;
    }
    std::printf("LPG k=%d: %zu pixels bit-exact, abs_min %g\n", k, got.size(), am_got);
    return 0;
}

static int test_conv() {
    const int B = 2, h = 9, w = 11, cin = 32, cout = 32, ks = 3, dil = 2, pad = 2;
    const int kpad = ks * ks * cin;                          // 288: already a multiple of 32
    std::vector<float> x((size_t)B * h * w * cin), wt((size_t)cout * kpad), sc(cout), sh(cout), ref((size_t)B * h * w * cout), got(ref.size());
    for (auto& v : x) v = frand() - 0.5f;
    for (auto& v : wt) v = (frand() - 0.5f) * 0.2f;           // packed layout: [cout][tap*cin + c]
    for (int n = 0; n < cout; ++n) { sc[n] = 0.5f + frand(); sh[n] = frand() - 0.5f; }
    for (int b = 0; b < B; ++b)
        for (int y = 0; y < h; ++y)
            for (int xx = 0; xx < w; ++xx)
                for (int n = 0; n < cout; ++n) {
                    double s = 0.0;
                    for (int ky = 0; ky < ks; ++ky)
                        for (int kx = 0; kx < ks; ++kx) {
                            const int iy = y + ky * dil - pad, ix = xx + kx * dil - pad;
                            if (iy < 0 || iy >= h || ix < 0 || ix >= w) continue;
                            const float* px = &x[(((size_t)b * h + iy) * w + ix) * cin];
                            const float* pw = &wt[(size_t)n * kpad + (ky * ks + kx) * cin];
                            for (int c = 0; c < cin; ++c) s += (double)px[c] * pw[c];
                        }
                    const float v = (float)s * sc[n] + sh[n];
                    ref[(((size_t)b * h + y) * w + xx) * cout + n] = v > 0.f ? v : 0.f;
                }
    float *d_x, *d_w, *d_sc, *d_sh, *d_y;
    CK(hipMalloc(&d_x, x.size() * 4)); CK(hipMalloc(&d_w, wt.size() * 4)); CK(hipMalloc(&d_sc, cout * 4));
    CK(hipMalloc(&d_sh, cout * 4)); CK(hipMalloc(&d_y, ref.size() * 4));
    CK(hipMemcpy(d_x, x.data(), x.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_w, wt.data(), wt.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_sc, sc.data(), cout * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_sh, sh.data(), cout * 4, hipMemcpyHostToDevice));
    bts_conv_desc d;
    std::memset(&d, 0, sizeof d);
    d.x = d_x; d.x_pix_stride = cin; d.c_in_ld = cin; d.k_pad = kpad;
    d.B = B; d.h_in = h; d.w_in = w; d.up = 1; d.ksize = ks; d.dil = dil; d.stride = 1; d.pad = pad;
    d.w = d_w; d.c_out = cout; d.c_out_pad = cout;
    d.e1_scale = d_sc; d.e1_shift = d_sh; d.act = 1;
    d.y = d_y; d.y_pix_stride = cout;
    int rc = bts_conv_fwd_f32(&d, nullptr);
    if (rc != 0) { std::printf("bts_conv_fwd_f32 rc=%d (%s)\n", rc, bts_hip_error_string(rc)); return 1; }
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(got.data(), d_y, got.size() * 4, hipMemcpyDeviceToHost));
    double worst = 0.0, scale = 0.0;
    for (size_t i = 0; i < ref.size(); ++i) { worst = std::fmax(worst, std::fabs((double)got[i] - ref[i])); scale = std::fmax(scale, std::fabs((double)ref[i])); }
    std::printf("conv 3x3 dil 2 + BN + ReLU: max abs err %.3g (scale %.3g)\n", worst, scale);
    if (!(worst <= 2e-5 * scale)) return 1;
    // error behaviour: an unaligned channel count and a null output are refused on the host
    d.c_in_ld = 30;
    if (bts_conv_fwd_f32(&d, nullptr) >= 0) { std::printf("c_in_ld=30 was not refused\n"); return 1; }
    d.c_in_ld = cin; d.y = nullptr;
    if (bts_conv_fwd_f32(&d, nullptr) >= 0) { std::printf("null output was not refused\n"); return 1; }
    if (bts_lpg_fwd_f32(d_x, 1, 4, 4, 3, d_y, nullptr, nullptr) >= 0) { std::printf("upratio 3 was not refused\n"); return 1; }
    CK(hipFree(d_x)); CK(hipFree(d_w)); CK(hipFree(d_sc)); CK(hipFree(d_sh)); CK(hipFree(d_y));
    return 0;
}

int main() {
    if (bts_hip_abi_version() != BTS_HIP_ABI_VERSION) { std::printf("ABI version mismatch\n"); return 1; }
    int fails = 0;
    for (int k : {8, 4, 2}) fails += test_lpg(k);
    fails += test_conv();
    std::printf(fails ? "FAILED\n" : "all C-ABI host checks passed\n");
    return fails ? 1 : 0;
}
