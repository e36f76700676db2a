// st_common.h -- shared device/host helpers for libsignaltrain_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/signaltrain_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define ST_WAVE 64

// error plumbing (st_api.hip)
int st_fail(int code, const char* fmt, ...);
int st_check_launch(const char* what);

static inline hipStream_t st_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

__host__ __device__ static inline int st_kp_of(int F) { return 2 * ((F + 15) / 16 * 16); }
__host__ __device__ static inline int st_round_up(int a, int b) { return (a + b - 1) / b * b; }

// ---------------------------------------------------------------- wave / block reductions
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// Sum over the block; result valid in thread 0.  `red` = LDS float[NWAVES].
template <int NWAVES>
__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) red[w] = v;
    __syncthreads();
    float t = 0.f;
    if (threadIdx.x == 0) {
#pragma unroll
        for (int i = 0; i < NWAVES; ++i) t += red[i];
    }
    __syncthreads();
    return t;
}

// sin / cos of a phase on the hardware path (v_sin_f32 / v_cos_f32 after a multiply by 1/2pi): absolute error ~1e-6 for
// the |phase| < ~100 rad the model produces, against a 1e-4 parity tolerance; the full-precision sincosf costs ~10x the
// instructions and sits on the critical path of the issue-bound autoencoder kernels.
__device__ __forceinline__ void st_sincos(float x, float& sn, float& cs) { sn = __sinf(x); cs = __cosf(x); }

// atan2 for the polar epilogue of the analysis GEMM (nn_proc.py:310): odd minimax polynomial of degree 17 for atan on [0, 1] (max error
// 1.1e-7 rad evaluated in fp32) on min/max of |x|, |y| through v_rcp_f32 (1 ulp), octant / quadrant fix-ups, the sign of y copied
// (so atan2(-0, x < 0) = -pi as IEEE has it).  Total error <= 3e-7 rad against the 1e-4 parity tolerance: ~25 instructions instead
// of the ~150 of the library atan2f -- the epilogue runs 24 of these per lane with nothing to overlap (fixed cost of the kernel).
__device__ __forceinline__ float st_atan2f(const float y, const float x)
{
    const float ax = __builtin_fabsf(x), ay = __builtin_fabsf(y);
    const float mx = __builtin_amdgcn_fmed3f(ax, ay, 3.0e38f), mn = __builtin_amdgcn_fmed3f(ax, ay, 0.f);     // max / min of two non-negative values
    const float t = mn * __builtin_amdgcn_rcpf(__builtin_amdgcn_fmed3f(mx, 2.0e-38f, 3.0e38f));               // x = y = 0 -> 0
    const float s = t * t;
    float p = 0.0028340641874819994f;
    p = __builtin_fmaf(p, s, -0.016005029901862144f);
    p = __builtin_fmaf(p, s, 0.042587608098983765f);
    p = __builtin_fmaf(p, s, -0.07495445758104324f);
    p = __builtin_fmaf(p, s, 0.10636754333972931f);
    p = __builtin_fmaf(p, s, -0.14202570915222168f);
    p = __builtin_fmaf(p, s, 0.19992484152317047f);
    p = __builtin_fmaf(p, s, -0.3333306610584259f);
    p = __builtin_fmaf(p, s, 1.0f);
    float r = t * p;
    r = ay > ax ? 1.57079632679489662f - r : r;
    r = x < 0.f ? 3.14159265358979324f - r : r;
    return __builtin_copysignf(r, y);
}

// ELU(a) = a > 0 ? a : exp(a) - 1 (nn_proc.py:31, alpha = 1).  With u = exp(a) - 1:  u >= a for every a, and u <= 0 exactly when
// a <= 0, so the value is the MEDIAN of {a, u, 0} -- one v_med3_f32 instead of a compare and a select (5 -> 4 instructions per
// activation; 3 with the packed multiply / add of elu4).  Differs from the select form only where the fp32 rounding of exp()
// puts u slightly below a small positive a (a < ~1e-3): by at most the rounding error of exp(a) - 1 itself, <= 1.2e-7 absolute.
__device__ __forceinline__ float elu_f(float a) { return __builtin_amdgcn_fmed3f(a, __expf(a) - 1.0f, 0.f); }
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x4 elu4(const f32x4 a)
{
    // explicit register PAIRS: v_pk_mul_f32 / v_pk_add_f32 (the 4-wide form is scalarised); __expf(a) is exactly v_exp_f32(a * log2(e))
    const f32x2 c = {1.44269504088896341f, 1.44269504088896341f}, one = {1.0f, 1.0f};
    const f32x2 t0 = (f32x2){a[0], a[1]} * c, t1 = (f32x2){a[2], a[3]} * c;
    const f32x2 e0 = {__builtin_amdgcn_exp2f(t0[0]), __builtin_amdgcn_exp2f(t0[1])}, e1 = {__builtin_amdgcn_exp2f(t1[0]), __builtin_amdgcn_exp2f(t1[1])};
    const f32x2 u0 = e0 - one, u1 = e1 - one;
    return (f32x4){__builtin_amdgcn_fmed3f(a[0], u0[0], 0.f), __builtin_amdgcn_fmed3f(a[1], u0[1], 0.f),
                   __builtin_amdgcn_fmed3f(a[2], u1[0], 0.f), __builtin_amdgcn_fmed3f(a[3], u1[1], 0.f)};
}
// ELU'(a) through h = ELU(a):  1 if h > 0 else h + 1 (= exp(a)).
__device__ __forceinline__ float elu_grad_from_out(float h) { return h > 0.f ? 1.0f : h + 1.0f; }

// ---------------------------------------------------------------- 16-bit GEMM operands (st_gemm16.h)
// Round an fp32 value to the 16-bit operand type of the call: ht = 1 bfloat16 (round to nearest even), 2 = IEEE float16 (round to
// nearest even, overflow -> inf: the gradient norm then turns non-finite and the optimizer kernel skips the step).  The SAME
// conversions gemm_half_kernel applied while staging fp32 operands (v_cvt_pk_bf16_f32 / v_cvt_pk_f16_f32) -- and the oracle's
// bf16_round / fp16_round -- so storing an operand pre-rounded changes no value.
__device__ __forceinline__ unsigned short st_to_h16(const float x, const int ht)
{
    if (ht == 2) { const _Float16 h = (_Float16)x; return __builtin_bit_cast(unsigned short, h); }
    const __bf16 b = (__bf16)x; return __builtin_bit_cast(unsigned short, b);
}
__device__ __forceinline__ uint2 st_to_h16x4(const float4 v, const int ht)
{
    return make_uint2((unsigned)st_to_h16(v.x, ht) | ((unsigned)st_to_h16(v.y, ht) << 16), (unsigned)st_to_h16(v.z, ht) | ((unsigned)st_to_h16(v.w, ht) << 16));
}
