// Wave specialisation check: 8-wave workgroup, waves 0-3 = consumers (LDS fragment reads + MFMA only),
// waves 4-7 = producers (global loads + ~700 VALU + ds_write into the other LDS buffer); one barrier per step.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int VALU_ITERS>
__global__ __launch_bounds__(512) void k(const float* __restrict__ src, long src_floats, float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) float lds[];   // 2 x 256 x 36 floats
    for (int i = threadIdx.x; i < 2 * 256 * 36; i += 512) lds[i] = (float)(i % 7) * 0.125f;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float s = 0.f;
    if (w < 4) {
        f32x16 acc[4];
        for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
        const int wm = w >> 1, wn = w & 1;
        for (int it = 0; it < iters; ++it) {
            const float* base = lds + (it & 1) * 256 * 36;
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
            __syncthreads();
        }
        for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
    } else {
        const int ptid = threadIdx.x - 256;
        const int lrow = ptid >> 3, lk = (ptid & 7) * 4;
        unsigned addr = (blockIdx.x * 977u + ptid * 16u) % (unsigned)(src_floats - 64 * 1024);
        int junk = ptid;
        f32x4 st[8];
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int p = 0; p < 8; ++p) st[p] = *reinterpret_cast<const f32x4*>(src + ((addr + p * 4096u * 4u) & ~3u));
            addr = (addr + 32u) % (unsigned)(src_floats - 64 * 1024);
#pragma unroll
            for (int v = 0; v < VALU_ITERS; ++v) junk = (junk * 3 + v) ^ (junk >> 3);
            float* nb = lds + ((it & 1) ^ 1) * 256 * 36;
#pragma unroll
            for (int p = 0; p < 8; ++p) *reinterpret_cast<f32x4*>(nb + (p * 32 + lrow) * 36 + lk) = st[p] * 0.999f;
            __syncthreads();
        }
        s = (float)junk * 1e-30f;
    }
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int V>
void run(const char* name, const float* src, long n, int iters, int bpc) {
    float* out;
    const int blocks = 256 * bpc;
    hipMalloc(&out, sizeof(float) * blocks * 512);
    hipEvent_t s, e;
    hipEventCreate(&s); hipEventCreate(&e);
    const size_t ldsz = 2 * 256 * 36 * 4;
    hipFuncSetAttribute((const void*)k<V>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsz);
    hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(512), ldsz, 0, src, n, out, iters);
    hipDeviceSynchronize();
    float best = 1e9f;
    for (int r = 0; r < 5; ++r) {
        hipEventRecord(s);
        hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(512), ldsz, 0, src, n, out, iters);
        hipEventRecord(e);
        hipEventSynchronize(e);
        float ms; hipEventElapsedTime(&ms, s, e);
        if (ms < best) best = ms;
    }
    const double flops = (double)blocks * 4 * iters * 64.0 * 4096.0;
    printf("%-44s WG/CU %d: %8.3f ms  %7.1f TFLOP/s\n", name, bpc, best, flops / best / 1e9);
    hipFree(out);
}

int main() {
    const long n = 64L * 1024 * 1024;
    float* src; hipMalloc(&src, n * 4); hipMemset(src, 0, n * 4);
    const int it = 1500;
    for (int bpc = 1; bpc <= 2; ++bpc) {
        run<0>("specialised, producers: loads+ds_write", src, n, it, bpc);
        run<60>("specialised, + ~200 VALU in producers", src, n, it, bpc);
        run<200>("specialised, + ~700 VALU in producers", src, n, it, bpc);
    }
    return 0;
}
