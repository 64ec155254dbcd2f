// Construction benchmark, closer to conv_fwd_kernel's K-step: 64 MFMAs + 16 fragment ds_reads per wave,
// 8 global_load_dwordx4 per thread staged through registers into the other LDS buffer, one barrier.
// Variants: source footprint (L2-resident vs HBM), prefetch distance (consume same step / next step),
// address arithmetic per load (cheap / conv-like).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int DIST, int ADDR>
__global__ __launch_bounds__(256) void k(const float* __restrict__ src, unsigned mask, float* out, int iters, int w_in, int stride) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    for (int i = threadIdx.x; i < 2 * 256 * 36; i += 256) lds[i] = (float)(i % 7) * 0.125f;
    __syncthreads();
    f32x16 acc[4];
    for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int wm = w >> 1, wn = w & 1;
    const int lrow = threadIdx.x >> 3, lk = (threadIdx.x & 7) * 4;
    f32x4 st[8];
    unsigned base[8];
    int py[8], px[8];
    for (int p = 0; p < 8; ++p) {
        const unsigned row = blockIdx.x * 256u + p * 32u + lrow;
        base[p] = (row * 128u + lk);                 // 512-B rows
        py[p] = (row / 152u) % 44u; px[p] = row % 152u;
        st[p] = f32x4{1.f, 2.f, 3.f, 4.f};
    }
    auto issue = [&](int it) {
        const int tap = (it / 8) % 9, kc = it % 8;
        const int dy = (tap / 3 - 1) * 3, dx = (tap % 3 - 1) * 3;
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            unsigned off;
            if (ADDR == 0) {
                off = base[p] + (unsigned)(it * 32);
            } else {   // conv-like: bounds test, clamp, multiply-add
                const int yy = py[p] + dy, xx = px[p] + dx;
                const bool ok = (unsigned)yy < 44u && (unsigned)xx < 152u;
                const int yc = min(max(yy, 0), 43), xc = min(max(xx, 0), 151);
                off = (unsigned)(yc * w_in + xc) * (unsigned)stride + (unsigned)(kc * 32 + lk) + (ok ? 0u : 4u) + base[p];
            }
            st[p] = *reinterpret_cast<const f32x4*>(src + (off & mask));
        }
    };
    auto stage = [&](int buf) {
        float* nb = lds + buf * 256 * 36;
#pragma unroll
        for (int p = 0; p < 8; ++p) *reinterpret_cast<f32x4*>(nb + (p * 32 + lrow) * 36 + lk) = st[p];
    };
    if (DIST == 1) issue(0);
    for (int it = 0; it < iters; ++it) {
        const int buf = it & 1;
        if (DIST == 0) issue(it);
        const float* b0 = lds + buf * 256 * 36;
        const float* pa = b0 + ((wm * 64 + (lane & 31)) * 36 + 4 * (lane >> 5));
        const float* pb = b0 + ((128 + wn * 64 + (lane & 31)) * 36 + 4 * (lane >> 5));
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
            if (DIST == 1 && g == 0) { stage(buf ^ 1); issue(it + 1); }
        }
        if (DIST == 0) stage(buf ^ 1);
        __syncthreads();
    }
    float s = 0.f;
    for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int D, int A>
void run(const char* name, const float* src, unsigned mask, int iters) {
    float* out;
    const int blocks = 256 * 2;
    hipMalloc(&out, sizeof(float) * blocks * 256);
    hipEvent_t s, e;
    hipEventCreate(&s); hipEventCreate(&e);
    const size_t ldsz = 2 * 256 * 36 * 4;
    hipFuncSetAttribute((const void*)k<D, A>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsz);
    hipLaunchKernelGGL((k<D, A>), dim3(blocks), dim3(256), ldsz, 0, src, mask, out, iters, 152, 448);
    hipDeviceSynchronize();
    float best = 1e9f;
    for (int r = 0; r < 5; ++r) {
        hipEventRecord(s);
        hipLaunchKernelGGL((k<D, A>), dim3(blocks), dim3(256), ldsz, 0, src, mask, out, iters, 152, 448);
        hipEventRecord(e);
        hipEventSynchronize(e);
        float ms; hipEventElapsedTime(&ms, s, e);
        if (ms < best) best = ms;
    }
    const double flops = (double)blocks * 4 * iters * 64.0 * 4096.0;
    printf("%-64s %8.3f ms  %7.1f TFLOP/s\n", name, best, flops / best / 1e9);
    hipFree(out);
}

int main() {
    const long n = 128L * 1024 * 1024;   // 512 MB
    float* src; hipMalloc(&src, n * 4); hipMemset(src, 0, n * 4);
    const unsigned small = (4u << 20) / 4 - 4, big = (unsigned)n - 4;   // 4 MB (L2) vs 512 MB (HBM) footprints (masks need 2^k-4)
    const int it = 1500;
    run<0, 0>("consume same step, cheap addr, 4 MB footprint", src, (1u << 20) - 4, it);
    run<0, 0>("consume same step, cheap addr, 512 MB footprint", src, (1u << 27) - 4, it);
    run<1, 0>("prefetch 1 step, cheap addr, 4 MB footprint", src, (1u << 20) - 4, it);
    run<1, 0>("prefetch 1 step, cheap addr, 512 MB footprint", src, (1u << 27) - 4, it);
    run<1, 1>("prefetch 1 step, conv-like addr, 4 MB footprint", src, (1u << 20) - 4, it);
    run<1, 1>("prefetch 1 step, conv-like addr, 512 MB footprint", src, (1u << 27) - 4, it);
    return 0;
}
