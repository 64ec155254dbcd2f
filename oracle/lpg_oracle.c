/*
 * lpg_oracle.c -- CPU ORACLE (C restatement) of local planar guidance.  TEST INFRASTRUCTURE, NOT PRODUCT:
 * only tests/ may load oracle/_build/liblpg_oracle.so.
 *
 * Loop structure follows the reference's only native code, the TF custom op's CPU kernel
 * (tensorflow/custom_layer/local_planar_guidance.cc:74-115: one serial loop over output pixels,
 * u from the column, v from the row), with the two differences of the PyTorch module that is the
 * parity target (pytorch/bts.py:149-173): input is NCHW planar [B,4,h,w] (the TF op reads NHWC,
 * .cc:101) and the denominator is clamped to +-1e-3 (bts.py:168-171; the TF op does not clamp).
 * abs_min follows bts.py:167.  Pinned by tests/golden/lpg_tables.npz (generated from the reference).
 * Build: make -C oracle   (gcc -O2 -ffp-contract=off: no FMA, as the reference's separate torch ops)
 */
#include <math.h>

void lpg_oracle_fwd(const float* plane_eq, int B, int h, int w, int k, float* depth, float* abs_min) {
    const int H = h * k, W = w * k;
    float amin = INFINITY;
    for (long index = 0; index < (long)B * H * W; ++index) {
        long t = index;
        const int col = (int)(t % W); t /= W;
        const int row = (int)(t % H); t /= H;
        const int b = (int)t;
        const int ir = row / k, ic = col / k;
        const float v = ((float)(row % k) - (float)(k - 1) * 0.5f) / (float)k;   /* .cc:97, bts.py:160-161 */
        const float u = ((float)(col % k) - (float)(k - 1) * 0.5f) / (float)k;   /* .cc:98, bts.py:157-158 */
        const long hw = (long)h * w;
        const float* p = plane_eq + (long)b * 4 * hw + (long)ir * w + ic;
        const float n1 = p[0], n2 = p[hw], n3 = p[2 * hw], n4 = p[3 * hw];
        float den = n1 * u;
        const float t2 = n2 * v;
        den = den + t2;
        den = den + n3;                                                           /* bts.py:166 */
        const float a = fabsf(den);
        if (a < amin) amin = a;                                                   /* bts.py:167 */
        const float eps = 1e-3f;
        if (den > 0.f && den < eps) den = eps;                                    /* bts.py:170 */
        if (den < 0.f && den > -eps) den = -eps;                                  /* bts.py:171 */
        depth[index] = n4 / den;                                                  /* bts.py:173 */
    }
    if (abs_min) *abs_min = amin;
}
