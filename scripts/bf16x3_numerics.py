#!/usr/bin/env python3
"""CPU (NumPy) statement of the numerics behind bts_conv_desc.precision = 1: error of K = 2304 dot products against
fp64 for (a) fp32 accumulated two products at a time like the v_mfma_f32_32x32x2_f32 chain, (b) the three-way bf16
split with six products accumulated per 16 k in fp32, (c) the cheaper two-piece split with three products.
    python scripts/bf16x3_numerics.py"""
import numpy as np

rng = np.random.default_rng(0)


def bf16_trunc(x):
    return (x.astype(np.float32).view(np.uint32) & 0xffff0000).view(np.float32)


def split3(x):
    h = bf16_trunc(x)
    r = (x - h).astype(np.float32)
    m = bf16_trunc(r)
    low = bf16_trunc((r - m).astype(np.float32))
    return h, m, low


def main():
    K, M, N = 2304, 256, 64
    A = rng.standard_normal((M, K)).astype(np.float32)
    B = (rng.standard_normal((K, N)) * 0.05).astype(np.float32)
    ref = A.astype(np.float64) @ B.astype(np.float64)
    chain = np.zeros((M, N), np.float32)
    for k in range(0, K, 2):
        chain = (chain + (A[:, k:k + 2].astype(np.float64) @ B[k:k + 2, :].astype(np.float64)).astype(np.float32)).astype(np.float32)
    ah, am, al = split3(A)
    bh, bm, bl = split3(B)

    def emu(pairs):
        acc = np.zeros((M, N), np.float32)
        for k in range(0, K, 16):
            for x, y in pairs:      # bf16 x bf16 products are exact in fp32; each MFMA adds its 16-term sum to the accumulator
                acc = (acc + (x[:, k:k + 16].astype(np.float64) @ y[k:k + 16, :].astype(np.float64)).astype(np.float32)).astype(np.float32)
        return acc
    six = emu(((ah, bh), (ah, bm), (am, bh), (ah, bl), (al, bh), (am, bm)))
    three = emu(((ah, bh), (ah, bm), (am, bh)))
    sc = np.abs(ref).max()
    for name, v in (("fp32, two products per accumulate (fp32-MFMA chain)", chain), ("bf16 three-way split, six products", six),
                    ("bf16 two-way split, three products", three)):
        print("%-52s max err / max|ref| = %.2e" % (name, np.abs(v - ref).max() / sc))


if __name__ == "__main__":
    main()
