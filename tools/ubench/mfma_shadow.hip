// Micro-benchmark 2: does another wave's non-FMA work (integer, transcendental, LDS, global loads) overlap a dense MFMA stream on the
// same SIMD?  (pair<> below.)  The same-wave part (shadow<>) is NOT conclusive: the compiler hoists the MFMAs into one run whatever
// the source order -- mfma_pacing.hip / mfma_bf16_shadow.hip repeat it with the order forced by inline asm.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_shadow mfma_shadow.hip && ./mfma_shadow
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

template <int BF, int KV>     // one wave per SIMD (256 threads): per MFMA, KV independent v_fma_f32
__global__ void __launch_bounds__(256) shadow(float* out, int n)
{
    f32x4 a[8] = {};
    float v[8]; for (int j = 0; j < 8; ++j) v[j] = threadIdx.x * 1e-3f + j;
    const float x = threadIdx.x * 1e-3f, y = 1.0f;
    const s16x4 xb = {(short)threadIdx.x, 1, 2, 3}, yb = {0x3f80, 0x3f80, 0x3f80, 0x3f80};
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if constexpr (BF) a[j] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(xb, yb, a[j], 0, 0, 0);
            else a[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a[j], 0, 0, 0);
#pragma unroll
            for (int q = 0; q < KV; ++q) v[(j + q) & 7] = __builtin_fmaf(v[(j + q) & 7], 1.0001f, 1e-7f);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float r = 0.f; for (int j = 0; j < 8; ++j) r += a[j][j & 3] + v[j];
    if (r == 12345.678f) out[threadIdx.x] = r;
}

template <int VK>             // two waves per SIMD: waves 0-3 dense fp32 MFMA, waves 4-7 vector work of kind VK (0 fma, 1 int add, 2 exp, 3 ds_read_b128, 4 global load)
__global__ void __launch_bounds__(512) pair(float* out, const float* in, int n_m, int n_v, int mode)
{
    __shared__ float lds[4096];
    const int wave = threadIdx.x >> 6; const bool mat = wave < 4;
    for (int i = threadIdx.x; i < 4096; i += 512) lds[i] = i;
    __syncthreads();
    if (mat && !(mode & 1)) return;
    if (!mat && !(mode & 2)) return;
    float r = 0.f;
    if (mat) {
        f32x4 a[8] = {}; const float x = threadIdx.x * 1e-3f, y = 1.0f;
        for (int i = 0; i < n_m; ++i) {
#pragma unroll
            for (int j = 0; j < 8; ++j) a[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a[j], 0, 0, 0);
        }
        for (int j = 0; j < 8; ++j) r += a[j][j & 3];
    } else {
        float v[16]; int u[16];
        for (int j = 0; j < 16; ++j) { v[j] = threadIdx.x * 1e-3f + j; u[j] = threadIdx.x + j; }
        for (int i = 0; i < n_v; ++i) {
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                if constexpr (VK == 0) v[j] = __builtin_fmaf(v[j], 1.0001f, 1e-7f);
                else if constexpr (VK == 1) u[j] = (u[j] + 77) ^ u[(j + 1) & 15];
                else if constexpr (VK == 2) v[j] = __builtin_amdgcn_exp2f(v[j]);
                else if constexpr (VK == 3) { const f32x4 t = *reinterpret_cast<const f32x4*>(&lds[((threadIdx.x & 63) * 4 + 256 * (j & 7) + (i & 1) * 2048) & 4095]); v[j] += t[0]; }
                else { v[j] += in[(size_t)(blockIdx.x * 64 + (threadIdx.x & 63)) + (size_t)((i * 16 + j) & 1023) * 16384]; }
            }
        }
        for (int j = 0; j < 16; ++j) r += v[j] + u[j];
    }
    if (r == 12345.678f) out[threadIdx.x] = r;
}

template <class F> static float tm(F f) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    f(); f(); (void)hipEventRecord(e0); for (int w = 0; w < 5; ++w) f(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); return ms / 5 * 1e3f;
}
template <int BF, int KV> static void sh(float* out) {
    const int n = 4000; const float t = tm([&] { hipLaunchKernelGGL((shadow<BF, KV>), dim3(256), dim3(256), 0, 0, out, n); });
    printf("  %s + %2d v_fma per MFMA (same wave): %7.1f us = %5.1f cycles per MFMA slot\n", BF ? "16x16x16_bf16" : "16x16x4_f32 ", KV, t, t * 2400.f / (n * 8.f));
}
template <int VK> static void pr(const char* name, float* out, const float* in) {
    const int nm = 4000; const float t_m = tm([&] { hipLaunchKernelGGL(pair<VK>, dim3(256), dim3(512), 0, 0, out, in, nm, 0, 1); });
    int nv = 500; float t_v = tm([&] { hipLaunchKernelGGL(pair<VK>, dim3(256), dim3(512), 0, 0, out, in, 0, nv, 2); });
    nv = (int)(nv * t_m / t_v); t_v = tm([&] { hipLaunchKernelGGL(pair<VK>, dim3(256), dim3(512), 0, 0, out, in, 0, nv, 2); });
    const float t_b = tm([&] { hipLaunchKernelGGL(pair<VK>, dim3(256), dim3(512), 0, 0, out, in, nm, nv, 3); });
    printf("  dense fp32 MFMA wave + %-14s wave on one SIMD: alone %6.1f / %6.1f us, together %6.1f us -> overlap %3.0f %%\n", name, t_m, t_v, t_b,
           100.f * (t_m + t_v - t_b) / (t_m < t_v ? t_m : t_v));
}
int main() {
    float *out, *in; (void)hipMalloc(&out, 4096); (void)hipMalloc(&in, (size_t)1024 * 16384 * 4 + 65536); (void)hipMemset(in, 0, (size_t)1024 * 16384 * 4 + 65536);
    sh<0, 0>(out); sh<0, 2>(out); sh<0, 4>(out); sh<0, 6>(out); sh<0, 7>(out); sh<0, 8>(out); sh<0, 12>(out);
    sh<1, 0>(out); sh<1, 2>(out); sh<1, 3>(out); sh<1, 4>(out); sh<1, 6>(out);
    pr<0>("v_fma_f32", out, in); pr<1>("integer", out, in); pr<2>("v_exp_f32", out, in); pr<3>("ds_read_b128", out, in); pr<4>("global_load", out, in);
    return 0;
}
