// Micro-benchmark 3: a dense MFMA stream blocks the co-resident wave of its SIMD (mfma_valu_overlap.hip).  Does PACING the stream
// -- idle wait states (s_nop) or own independent instructions after each MFMA, so that the next MFMA is presented only when the matrix
// pipe is about to be free -- let the partner wave issue?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int PACE>           // PACE: 0 dense; N > 0: N x "s_nop 3" (4 wait states each) after every MFMA
__global__ void __launch_bounds__(512) pair(float* out, int n_m, int n_v, int mode)
{
    const int wave = threadIdx.x >> 6; const bool mat = wave < 4;
    if (mat && !(mode & 1)) return;
    if (!mat && !(mode & 2)) return;
    float r = 0.f;
    if (mat) {
        f32x4 a[8] = {}; const float x = threadIdx.x * 1e-3f, y = 1.0f;
        for (int i = 0; i < n_m; ++i) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(a[j]) : "v"(x), "v"(y));
#pragma unroll
                for (int q = 0; q < PACE; ++q) asm volatile("s_nop 3");
            }
        }
        for (int j = 0; j < 8; ++j) r += a[j][j & 3];
    } else {
        float v[16];
        for (int j = 0; j < 16; ++j) v[j] = threadIdx.x * 1e-3f + j;
        for (int i = 0; i < n_v; ++i) {
#pragma unroll
            for (int j = 0; j < 16; ++j) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[j]) : "v"(1.0001f), "v"(1e-7f));
        }
        for (int j = 0; j < 16; ++j) r += v[j];
    }
    if (r == 12345.678f) out[threadIdx.x] = r;
}
template <int KV>             // one wave per SIMD: every MFMA followed by KV independent v_fma of the SAME wave (forced order)
__global__ void __launch_bounds__(256) shadow(float* out, int n)
{
    f32x4 a[8] = {}; float v[8]; for (int j = 0; j < 8; ++j) v[j] = threadIdx.x * 1e-3f + j;
    const float x = threadIdx.x * 1e-3f, y = 1.0f;
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(a[j]) : "v"(x), "v"(y));
#pragma unroll
            for (int q = 0; q < KV; ++q) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[(j + q) & 7]) : "v"(1.0001f), "v"(1e-7f));
        }
    }
    float r = 0.f; for (int j = 0; j < 8; ++j) r += a[j][j & 3] + v[j];
    if (r == 12345.678f) out[threadIdx.x] = r;
}
template <class F> static float tm(F f) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    f(); f(); (void)hipEventRecord(e0); for (int w = 0; w < 5; ++w) f(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); return ms / 5 * 1e3f;
}
template <int PACE> static void pr(float* out) {
    const int nm = 4000; const float t_m = tm([&] { hipLaunchKernelGGL(pair<PACE>, dim3(256), dim3(512), 0, 0, out, nm, 0, 1); });
    int nv = 500; float t_v = tm([&] { hipLaunchKernelGGL(pair<PACE>, dim3(256), dim3(512), 0, 0, out, 0, nv, 2); });
    nv = (int)(nv * t_m / t_v); t_v = tm([&] { hipLaunchKernelGGL(pair<PACE>, dim3(256), dim3(512), 0, 0, out, 0, nv, 2); });
    const float t_b = tm([&] { hipLaunchKernelGGL(pair<PACE>, dim3(256), dim3(512), 0, 0, out, nm, nv, 3); });
    printf("  MFMA + %d x s_nop 3 | v_fma wave: alone %6.1f (%.1f cyc/MFMA) / %6.1f us, together %6.1f us -> overlap %3.0f %%\n", PACE, t_m, t_m * 2400.f / (nm * 8.f), t_v, t_b,
           100.f * (t_m + t_v - t_b) / (t_m < t_v ? t_m : t_v));
}
template <int KV> static void sh(float* out) {
    const int n = 4000; const float t = tm([&] { hipLaunchKernelGGL(shadow<KV>, dim3(256), dim3(256), 0, 0, out, n); });
    printf("  16x16x4_f32 + %2d v_fma after each MFMA (same wave, forced order): %7.1f us = %5.1f cycles per MFMA slot\n", KV, t, t * 2400.f / (n * 8.f));
}
int main() {
    float* out; (void)hipMalloc(&out, 4096);
    sh<0>(out); sh<1>(out); sh<2>(out); sh<3>(out); sh<4>(out); sh<5>(out); sh<6>(out); sh<7>(out); sh<8>(out); sh<10>(out); sh<12>(out);
    pr<0>(out); pr<2>(out); pr<4>(out); pr<5>(out); pr<6>(out); pr<7>(out); pr<8>(out);
    return 0;
}
