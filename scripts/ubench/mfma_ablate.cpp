// Ablation by construction: start from the 155 TF MFMA+LDS-read loop and add the conv kernel's other
// per-K-step ingredients one at a time: barrier, LDS writes (double buffer), global loads, address VALU.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// FLAGS bit0: __syncthreads per step; bit1: 8 ds_write_b128 per step; bit2: 8 global_load_dwordx4 per step;
//       bit3: ~200 dependent-free integer VALU ops per step (address arithmetic stand-in)
template <int FLAGS>
__global__ __launch_bounds__(256) void k(const float* __restrict__ src, long src_floats, float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) float lds[];   // 2 x 256 x 36 floats
    for (int i = threadIdx.x; i < 2 * 256 * 36; i += 256) lds[i] = (float)(i % 7) * 0.125f;
    __syncthreads();
    f32x16 acc[4];
    for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int wm = w >> 1, wn = w & 1;
    const int lrow = threadIdx.x >> 3, lk = (threadIdx.x & 7) * 4;
    f32x4 st[8];
    for (int p = 0; p < 8; ++p) st[p] = f32x4{1.f, 2.f, 3.f, 4.f};
    unsigned addr = (blockIdx.x * 977u + threadIdx.x * 16u) % (unsigned)(src_floats - 64 * 1024);
    int junk = threadIdx.x;
    for (int it = 0; it < iters; ++it) {
        const int buf = it & 1;
        if (FLAGS & 4) {
#pragma unroll
            for (int p = 0; p < 8; ++p) st[p] = *reinterpret_cast<const f32x4*>(src + ((addr + p * 4096u * 4u) & ~3u));
            addr = (addr + 32u) % (unsigned)(src_floats - 64 * 1024);
        }
        if (FLAGS & 8) {
#pragma unroll
            for (int v = 0; v < 200; ++v) junk = (junk * 3 + v) ^ (junk >> 3);
        }
        const float* base = lds + buf * 256 * 36;
        const float* pa = base + ((wm * 64 + (lane & 31)) * 36 + 4 * (lane >> 5));
        const float* pb = base + ((128 + wn * 64 + (lane & 31)) * 36 + 4 * (lane >> 5));
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f32x4 fa0 = *reinterpret_cast<const f32x4*>(pa + 8 * g);
            f32x4 fa1 = *reinterpret_cast<const f32x4*>(pa + 32 * 36 + 8 * g);
            f32x4 fb0 = *reinterpret_cast<const f32x4*>(pb + 8 * g);
            f32x4 fb1 = *reinterpret_cast<const f32x4*>(pb + 32 * 36 + 8 * g);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa0[q], fb0[q], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa0[q], fb1[q], acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa1[q], fb0[q], acc[2], 0, 0, 0);
                acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa1[q], fb1[q], acc[3], 0, 0, 0);
            }
        }
        if (FLAGS & 2) {
            float* nb = lds + (buf ^ 1) * 256 * 36;
#pragma unroll
            for (int p = 0; p < 8; ++p) *reinterpret_cast<f32x4*>(nb + (p * 32 + lrow) * 36 + lk) = st[p] * 0.999f;
        }
        if (FLAGS & 1) __syncthreads();
    }
    float s = (float)junk * 1e-30f;
    for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int F>
void run(const char* name, const float* src, long n, int iters) {
    float* out;
    const int blocks = 256 * 2;
    hipMalloc(&out, sizeof(float) * blocks * 256);
    hipEvent_t s, e;
    hipEventCreate(&s); hipEventCreate(&e);
    const size_t ldsz = 2 * 256 * 36 * 4;
    hipFuncSetAttribute((const void*)k<F>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsz);
    hipLaunchKernelGGL(k<F>, dim3(blocks), dim3(256), ldsz, 0, src, n, out, iters);
    hipDeviceSynchronize();
    float best = 1e9f;
    for (int r = 0; r < 5; ++r) {
        hipEventRecord(s);
        hipLaunchKernelGGL(k<F>, dim3(blocks), dim3(256), ldsz, 0, src, n, out, iters);
        hipEventRecord(e);
        hipEventSynchronize(e);
        float ms; hipEventElapsedTime(&ms, s, e);
        if (ms < best) best = ms;
    }
    const double flops = (double)blocks * 4 * iters * 64.0 * 4096.0;
    printf("%-52s %8.3f ms  %7.1f TFLOP/s\n", name, best, flops / best / 1e9);
    hipFree(out);
}

int main() {
    const long n = 64L * 1024 * 1024;     // 256 MB source
    float* src; hipMalloc(&src, n * 4); hipMemset(src, 0, n * 4);
    const int it = 1500;
    run<0>("mfma + lds reads", src, n, it);
    run<1>("+ barrier/step", src, n, it);
    run<3>("+ barrier + 8 ds_write_b128", src, n, it);
    run<4>("+ 8 global loads (no consumer)", src, n, it);
    run<7>("+ barrier + ds_write + global loads (staged)", src, n, it);
    run<8>("+ 200 VALU", src, n, it);
    run<15>("all", src, n, it);
    return 0;
}
