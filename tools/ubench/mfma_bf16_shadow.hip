// Micro-benchmark 4: is the bf16 matrix pipe independent of the vector ALUs?  One wave per SIMD, every
// v_mfma_f32_32x32x16_bf16 (8 passes = 32 cycles) followed by KV independent VALU instructions of the same wave, order forced.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

template <int KIND, int KV, int VK>   // KIND 0: 32x32x16 bf16 (8 regs operands)  1: 16x16x32 bf16   2: 32x32x2 f32 ; VK 0 v_fma_f32, 1 v_cvt_pk_bf16_f32, 2 ds_read_b128
__global__ void __launch_bounds__(256) shadow(float* out, int n)
{
    __shared__ float lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = i;
    __syncthreads();
    f32x16 a[4] = {}; f32x4 c[4] = {};
    float v[8]; for (int j = 0; j < 8; ++j) v[j] = threadIdx.x * 1e-3f + j;
    unsigned pk[8] = {};
    f32x4 ld[4] = {};
    const s16x8 xb = {(short)threadIdx.x, 1, 2, 3, 4, 5, 6, 7}, yb = {0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80};
    const float x = threadIdx.x * 1e-3f, y = 1.0f;
    const unsigned la = (threadIdx.x & 63) * 16;
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if constexpr (KIND == 0) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(a[j]) : "v"(xb), "v"(yb));
            else if constexpr (KIND == 1) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c[j]) : "v"(xb), "v"(yb));
            else asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(a[j]) : "v"(x), "v"(y));
#pragma unroll
            for (int q = 0; q < KV; ++q) {
                if constexpr (VK == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[(j + q) & 7]) : "v"(1.0001f), "v"(1e-7f));
                else if constexpr (VK == 1) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(pk[(j + q) & 7]) : "v"(v[q & 7]), "v"(v[(q + 1) & 7]));
                else asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(ld[q & 3]) : "v"(la), "n"(1024 * (q & 7)));
            }
        }
        if constexpr (VK == 2) asm volatile("s_waitcnt lgkmcnt(0)");
    }
    float r = 0.f; for (int j = 0; j < 4; ++j) r += a[j][j] + c[j][j & 3] + ld[j][0]; for (int j = 0; j < 8; ++j) r += v[j] + pk[j];
    if (r == 12345.678f) out[threadIdx.x] = r;
}
template <class F> static float tm(F f) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    f(); f(); (void)hipEventRecord(e0); for (int w = 0; w < 5; ++w) f(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); return ms / 5 * 1e3f;
}
template <int KIND, int KV, int VK> static void sh(float* out) {
    const int n = 8000; const float t = tm([&] { hipLaunchKernelGGL((shadow<KIND, KV, VK>), dim3(256), dim3(256), 0, 0, out, n); });
    const char* kn[] = {"32x32x16_bf16", "16x16x32_bf16", "32x32x2_f32  "}; const char* vn[] = {"v_fma_f32", "v_cvt_pk_bf16_f32", "ds_read_b128"};
    printf("  %s + %2d %-18s per MFMA: %7.1f us = %5.1f cycles per MFMA slot\n", kn[KIND], KV, vn[VK], t, t * 2400.f / (n * 4.f));
}
int main() {
    float* out; (void)hipMalloc(&out, 4096);
    sh<0, 0, 0>(out); sh<0, 2, 0>(out); sh<0, 4, 0>(out); sh<0, 6, 0>(out); sh<0, 8, 0>(out); sh<0, 12, 0>(out); sh<0, 16, 0>(out);
    sh<0, 4, 1>(out); sh<0, 8, 1>(out); sh<0, 2, 2>(out); sh<0, 4, 2>(out);
    sh<1, 0, 0>(out); sh<1, 2, 0>(out); sh<1, 4, 0>(out); sh<1, 8, 0>(out);
    sh<2, 0, 0>(out); sh<2, 4, 0>(out); sh<2, 8, 0>(out); sh<2, 12, 0>(out); sh<2, 4, 2>(out);
    return 0;
}
