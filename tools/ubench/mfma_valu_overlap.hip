// Micro-benchmark: do an MFMA-only wave and a VALU-only wave that share a SIMD overlap?  (DESIGN.md "additive model")
// 512-thread workgroups = two waves per SIMD; waves 0-3 run matrix work, waves 4-7 run vector work.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_valu_overlap mfma_valu_overlap.hip && ./mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

template <int KIND>   // 0: v_mfma_f32_16x16x4_f32   1: v_mfma_f32_32x32x2_f32   2: v_mfma_f32_16x16x16_bf16
__global__ void __launch_bounds__(512) k(float* out, int n_mfma, int n_valu, int mode)
{
    const int wave = threadIdx.x >> 6;
    const bool mat = (mode & 4) ? wave >= 4 : wave < 4;
    if ((mode & 8) && !mat) __builtin_amdgcn_s_setprio(3);
    if ((mode & 16) && mat) __builtin_amdgcn_s_setprio(3);
    if (mat && !(mode & 1)) return;
    if (!mat && !(mode & 2)) return;
    float r = 0.f;
    if (mat) {
        if constexpr (KIND == 1) {
            f32x16 a0 = {}, a1 = {}, a2 = {}, a3 = {};
            const float x = threadIdx.x * 1e-3f, y = 1.0f;
            for (int i = 0; i < n_mfma; ++i) {
                a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0); a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a1, 0, 0, 0);
                a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a2, 0, 0, 0); a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a3, 0, 0, 0);
            }
            r = a0[0] + a1[1] + a2[2] + a3[3];
        } else if constexpr (KIND == 0) {
            f32x4 a[8] = {};
            const float x = threadIdx.x * 1e-3f, y = 1.0f;
            for (int i = 0; i < n_mfma; ++i) {
#pragma unroll
                for (int j = 0; j < 8; ++j) a[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a[j], 0, 0, 0);
            }
            for (int j = 0; j < 8; ++j) r += a[j][j & 3];
        } else {
            f32x4 a[8] = {};
            const s16x4 x = {(short)threadIdx.x, 1, 2, 3}, y = {0x3f80, 0x3f80, 0x3f80, 0x3f80};
            for (int i = 0; i < n_mfma; ++i) {
#pragma unroll
                for (int j = 0; j < 8; ++j) a[j] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(x, y, a[j], 0, 0, 0);
            }
            for (int j = 0; j < 8; ++j) r += a[j][j & 3];
        }
    } else {
        float v[16];
        for (int j = 0; j < 16; ++j) v[j] = threadIdx.x * 1e-3f + j;
        const float c = 1.0001f, d = 1e-7f;
        for (int i = 0; i < n_valu; ++i) {
#pragma unroll
            for (int j = 0; j < 16; ++j) v[j] = __builtin_fmaf(v[j], c, d);
        }
        for (int j = 0; j < 16; ++j) r += v[j];
    }
    if (r == 12345.678f) out[threadIdx.x] = r;
}

template <int KIND>
static float run(float* out, int nm, int nv, int mode)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(512), 0, 0, out, nm, nv, mode);
    hipEventRecord(e0);
    for (int w = 0; w < 5; ++w) hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(512), 0, 0, out, nm, nv, mode);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms / 5 * 1e3f;
}
template <int KIND>
static void study(const char* name, float* out, int per_iter_mfma, int cycles_per_mfma, int flags = 0)
{
    const int nm = 4000;                                   // iterations of the matrix loop
    const float t_m = run<KIND>(out, nm, 0, 1 | flags);
    // choose the vector loop so that it takes about as long as the matrix loop when alone
    int nv = 2000; float t_v = run<KIND>(out, 0, nv, 2 | flags);
    nv = (int)(nv * t_m / t_v); t_v = run<KIND>(out, 0, nv, 2 | flags);
    const float t_b = run<KIND>(out, nm, nv, 3 | flags);
    printf("%-28s matrix alone %7.1f us (%.1f cyc/MFMA @2.4GHz, nominal %d) | vector alone %7.1f us | both %7.1f us  -> overlap %.0f %% (0 = additive, 100 = max)\n",
           name, t_m, t_m * 2400.f / (nm * (float)per_iter_mfma), cycles_per_mfma, t_v, t_b, 100.f * (t_m + t_v - t_b) / (t_m < t_v ? t_m : t_v));
}
int main()
{
    float* out; hipMalloc(&out, 4096);
    study<0>("v_mfma_f32_16x16x4_f32", out, 8, 32);
    study<1>("v_mfma_f32_32x32x2_f32", out, 4, 64);
    study<2>("v_mfma_f32_16x16x16_bf16", out, 8, 16);
    study<0>("16x16x4_f32, vector older", out, 8, 32, 4);
    study<0>("16x16x4_f32, vector prio 3", out, 8, 32, 8);
    study<0>("16x16x4_f32, v.older+prio", out, 8, 32, 12);
    study<0>("16x16x4_f32, matrix prio 3", out, 8, 32, 16);
    study<2>("16x16x16_bf16, vector older", out, 8, 16, 4);
    study<2>("16x16x16_bf16, v.older+prio", out, 8, 16, 12);
    return 0;
}
