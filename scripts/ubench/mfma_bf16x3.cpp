// Micro-benchmark for "fp32 emulated on the bf16 matrix cores" (each fp32 operand split into
// three bf16 pieces, six of the nine cross products kept: error 1.2e-6 of max|result| on K=2304 dot products, the same as the
// fp32 MFMA chain's 1.1e-6 -- see DESIGN.md section 3b).  What fp32-EQUIVALENT rate does the device sustain
//   (a) from registers only (6 x v_mfma_f32_32x32x16_bf16 per 16 k of a 32x32 block),
//   (b) with the 12 ds_read_b128 fragment reads per k16 step of a 64x64 wave tile,
//   (c) with (b) plus splitting one operand from fp32 on the fly (truncation split + v_perm packing)?
// Build: hipcc --offload-arch=gfx950 -O3.  fp32-equivalent FLOP = 2*M*N*K of the emulated product.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ bf16x8 as_bf(u32x4 v) { return __builtin_bit_cast(bf16x8, v); }

// split 8 fp32 (two f32x4) into hi/mid/lo bf16x8 by truncation: x = h + m + l + O(2^-24 x)
__device__ __forceinline__ void split8(const f32x4 x0, const f32x4 x1, u32x4& h, u32x4& m, u32x4& l) {
    float xs[8] = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
    unsigned hh[8], mm[8], ll[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const unsigned xb = __float_as_uint(xs[i]);
        hh[i] = xb & 0xffff0000u;
        const float r = xs[i] - __uint_as_float(hh[i]);
        mm[i] = __float_as_uint(r) & 0xffff0000u;
        ll[i] = __float_as_uint(r - __uint_as_float(mm[i])) & 0xffff0000u;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {        // pack element pairs: low half = element 2i, high half = element 2i+1
        h[i] = __builtin_amdgcn_perm(hh[2 * i + 1], hh[2 * i], 0x07060302u);
        m[i] = __builtin_amdgcn_perm(mm[2 * i + 1], mm[2 * i], 0x07060302u);
        l[i] = __builtin_amdgcn_perm(ll[2 * i + 1], ll[2 * i], 0x07060302u);
    }
}

#define MFMA(a, b, c) c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf(a), as_bf(b), c, 0, 0, 0)
#define SIX(A, B, c) MFMA(A[0], B[0], c); MFMA(A[0], B[1], c); MFMA(A[1], B[0], c); MFMA(A[0], B[2], c); MFMA(A[2], B[0], c); MFMA(A[1], B[1], c)

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    // LDS: bf16 planes [3][256 rows][32 k] (64 B rows + 16 B pad) for A rows 0..127 / B rows 128..255, and fp32 A rows
    __shared__ __attribute__((aligned(16))) unsigned planes[3][256][20];
    __shared__ __attribute__((aligned(16))) float afp32[128][36];
    for (int i = threadIdx.x; i < 3 * 256 * 20; i += 256) (&planes[0][0][0])[i] = 0x3f803f80u + (i % 5);
    for (int i = threadIdx.x; i < 128 * 36; i += 256) (&afp32[0][0])[i] = 1.0f + (i % 7) * 0.125f;
    __syncthreads();
    f32x16 acc[4];
    for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, r31 = lane & 31, kh = lane >> 5;
    u32x4 A0[3], A1[3], B0[3], B1[3];
    for (int p = 0; p < 3; ++p) { A0[p] = (u32x4)(0x3f803f80u + lane + p); A1[p] = A0[p] + 1u; B0[p] = A0[p] + 2u; B1[p] = A0[p] + 3u; }
    const int ar0 = (w >> 1) * 64 + r31, br0 = 128 + (w & 1) * 64 + r31;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {                 // BK = 32 = two k16 steps
            if (MODE >= 1) {
#pragma unroll
                for (int p = 0; p < 3; ++p) {
                    B0[p] = *reinterpret_cast<const u32x4*>(&planes[p][br0][ks * 8 + kh * 4]);
                    B1[p] = *reinterpret_cast<const u32x4*>(&planes[p][br0 + 32][ks * 8 + kh * 4]);
                    if (MODE == 1) {
                        A0[p] = *reinterpret_cast<const u32x4*>(&planes[p][ar0][ks * 8 + kh * 4]);
                        A1[p] = *reinterpret_cast<const u32x4*>(&planes[p][ar0 + 32][ks * 8 + kh * 4]);
                    }
                }
            }
            if (MODE == 2) {                               // A arrives as fp32 and is split here
                const f32x4 x0 = *reinterpret_cast<const f32x4*>(&afp32[ar0][ks * 16 + kh * 8]);
                const f32x4 x1 = *reinterpret_cast<const f32x4*>(&afp32[ar0][ks * 16 + kh * 8 + 4]);
                const f32x4 y0 = *reinterpret_cast<const f32x4*>(&afp32[ar0 + 32][ks * 16 + kh * 8]);
                const f32x4 y1 = *reinterpret_cast<const f32x4*>(&afp32[ar0 + 32][ks * 16 + kh * 8 + 4]);
                split8(x0, x1, A0[0], A0[1], A0[2]);
                split8(y0, y1, A1[0], A1[1], A1[2]);
            }
            SIX(A0, B0, acc[0]); SIX(A0, B1, acc[1]); SIX(A1, B0, acc[2]); SIX(A1, B1, acc[3]);
        }
    }
    float s = 0.f;
    for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
void run(const char* name, int blocks_per_cu, int iters) {
    float* out;
    const int blocks = 256 * blocks_per_cu;
    hipMalloc(&out, sizeof(float) * blocks * 256);
    hipEvent_t s, e;
    hipEventCreate(&s); hipEventCreate(&e);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters);
    hipDeviceSynchronize();
    float best = 1e9f;
    for (int r = 0; r < 5; ++r) {
        hipEventRecord(s);
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters);
        hipEventRecord(e);
        hipEventSynchronize(e);
        float ms;
        hipEventElapsedTime(&ms, s, e);
        best = ms < best ? ms : best;
    }
    // per wave per iteration: 64x64 tile x 32 k of emulated fp32 product
    const double eq = 2.0 * 64 * 64 * 32 * 4.0 * blocks * (double)iters;
    std::printf("%-52s %d wg/CU: %8.3f ms  %7.1f fp32-equivalent TFLOP/s  (%7.1f bf16 TFLOP/s issued)\n", name, blocks_per_cu, best,
                eq / best / 1e9, 6.0 * eq / best / 1e9);
    hipFree(out);
}

int main() {
    for (int bpc : {1, 2}) {
        run<0>("bf16x3, 6 products, registers only", bpc, 2000);
        run<1>("  + ds_read_b128 of all 12 fragments per k16", bpc, 2000);
        run<2>("  + A split from fp32 on the fly (B pre-split)", bpc, 2000);
    }
    return 0;
}
