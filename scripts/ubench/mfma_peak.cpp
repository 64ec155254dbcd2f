// Micro-benchmark: what fp32-input MFMA rate does THIS device sustain (a) from registers only,
// (b) with the conv kernel's LDS fragment reads in the loop?  Build: hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int WITH_LDS>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    __shared__ __attribute__((aligned(16))) float lds[256 * 36];
    for (int i = threadIdx.x; i < 256 * 36; i += 256) lds[i] = (float)(i % 7) * 0.125f;
    __syncthreads();
    f32x16 acc[4];
    for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    float a = 1.0f + lane * 1e-3f, b = 0.5f - lane * 1e-3f;
    const float* base = lds + ((w * 32 + (lane & 31)) * 36 + 4 * (lane >> 5));
    for (int it = 0; it < iters; ++it) {
        if (WITH_LDS) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 fa0 = *reinterpret_cast<const f32x4*>(base + 8 * g);
                f32x4 fa1 = *reinterpret_cast<const f32x4*>(base + 32 * 36 + 8 * g);
                f32x4 fb0 = *reinterpret_cast<const f32x4*>(base + 128 * 36 + 8 * g);
                f32x4 fb1 = *reinterpret_cast<const f32x4*>(base + 160 * 36 + 8 * g);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa0[q], fb0[q], acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa0[q], fb1[q], acc[1], 0, 0, 0);
                    acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa1[q], fb0[q], acc[2], 0, 0, 0);
                    acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa1[q], fb1[q], acc[3], 0, 0, 0);
                }
            }
        } else {
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, a, acc[2], 0, 0, 0);
                acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(b, b, acc[3], 0, 0, 0);
            }
        }
    }
    float s = 0.f;
    for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int L>
void run(const char* name, int blocks_per_cu, int iters) {
    float* out;
    const int blocks = 256 * blocks_per_cu;
    hipMalloc(&out, sizeof(float) * blocks * 256);
    hipEvent_t s, e;
    hipEventCreate(&s); hipEventCreate(&e);
    hipLaunchKernelGGL(k<L>, dim3(blocks), dim3(256), 0, 0, out, iters);
    hipDeviceSynchronize();
    float best = 1e9f;
    for (int r = 0; r < 5; ++r) {
        hipEventRecord(s);
        hipLaunchKernelGGL(k<L>, dim3(blocks), dim3(256), 0, 0, out, iters);
        hipEventRecord(e);
        hipEventSynchronize(e);
        float ms; hipEventElapsedTime(&ms, s, e);
        if (ms < best) best = ms;
    }
    const double flops = (double)blocks * 4 /*waves*/ * iters * 64.0 * 4096.0;
    printf("%-28s blocks/CU %d: %8.3f ms  %7.1f TFLOP/s\n", name, blocks_per_cu, best, flops / best / 1e9);
    hipFree(out);
}

int main() {
    for (int bpc = 1; bpc <= 2; ++bpc) {
        run<0>("registers only", bpc, 4000);
        run<1>("with LDS fragment reads", bpc, 4000);
    }
    return 0;
}
