// st_misc.h -- HBM-bound helper kernels of the SignalTrain step (gfx950).
#pragma once
#include "st_common.h"

namespace stm {

// ---------------------------------------------------------------- synthesis basis fold
// Sfold[KP,N]: rows [0,F) = Sr[k] (+ Sr[N-k], 1<=k<=F-2); rows [KP/2, KP/2+F) = Si[k] (- Si[N-k]); rest 0.
// Weight-side form of the Hermitian extension at cls_fe_dft.py:109-110.
__device__ __forceinline__ void fold_row(const float* __restrict__ Sr, const float* __restrict__ Si, float* __restrict__ Sfold,
                                         const int N, const int F, const int KP, const int row)
{
    const int half = KP / 2;
    const bool is_im = row >= half;
    const int k = is_im ? row - half : row;
    const float* S = is_im ? Si : Sr;
    const bool valid = k < F;
    const bool paired = valid && k >= 1 && k <= F - 2;
    const float sgn = is_im ? -1.f : 1.f;
    for (int n4 = threadIdx.x; n4 < N / 4; n4 += blockDim.x) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (valid) {
            v = reinterpret_cast<const float4*>(S + (size_t)k * N)[n4];
            if (paired) {
                const float4 u = reinterpret_cast<const float4*>(S + (size_t)(N - k) * N)[n4];
                v.x += sgn * u.x; v.y += sgn * u.y; v.z += sgn * u.z; v.w += sgn * u.w;
            }
        }
        reinterpret_cast<float4*>(Sfold + (size_t)row * N)[n4] = v;
    }
}
__global__ void __launch_bounds__(256)
fold_kernel(const float* __restrict__ Sr, const float* __restrict__ Si, float* __restrict__ Sfold, int N, int F, int KP)
{
    fold_row(Sr, Si, Sfold, N, F, KP, blockIdx.x);                      // 0..KP-1
}

// Frames that lie entirely in the Conv1d zero padding: re = im = mag = 0 and phs = atan2(0, 1e-7) = 0 exactly.
__device__ __forceinline__ void zero_dead_frame(float* __restrict__ a0, float* __restrict__ a1, float* __restrict__ a2, float* __restrict__ a3,
                                                const int T, const int F, const int t_lo, const int Tv, const int blk)
{
    const int nd = T - Tv;
    const int b = blk / nd, j = blk - b * nd;
    const int t = j < t_lo ? j : j + Tv;
    const size_t base = ((size_t)b * T + t) * F;
    for (int f = threadIdx.x; f < F; f += 256) {
        if (a0) a0[base + f] = 0.f;
        if (a1) a1[base + f] = 0.f;
        if (a2) a2[base + f] = 0.f;
        if (a3) a3[base + f] = 0.f;
    }
}
__global__ void __launch_bounds__(256)
zero_dead_frames_kernel(float* __restrict__ a0, float* __restrict__ a1, float* __restrict__ a2, float* __restrict__ a3,
                        int T, int F, int t_lo, int Tv)
{
    zero_dead_frame(a0, a1, a2, a3, T, F, t_lo, Tv, blockIdx.x);
}

// ---------------------------------------------------------------- overlap-add + residual + log-cosh
// y_hat[b,j] = 2 * sum_t frs[b,t, N + j - H t] + x[b, L-y+j]     (cls_fe_dft.py:112-113, nn_proc.py:332,340)
// loss partial = sum log cosh(y - y_hat) ; dsyn = 2 * (-tanh(y - y_hat)) * inv_count  (loss_functions.py:9-10)
__global__ void __launch_bounds__(256)
ola_loss_kernel(const float* __restrict__ frs, const float* __restrict__ x, const float* __restrict__ y_true,
                float* __restrict__ y_hat, float* __restrict__ dsyn, float* __restrict__ loss_partial,
                int L, int N, int H, int OT, int ysz, float inv_count, int nslab, size_t slab, int dsyn_pad,
                unsigned short* __restrict__ dsyn16 = nullptr, int ht = 0)      // 16-bit configurations: d syn goes out rounded to the GEMM operand type (same padded layout), INSTEAD of fp32
{
    __shared__ float red[4];
    const int b = blockIdx.y;
    const int j = blockIdx.x * 256 + threadIdx.x;
    float lc = 0.f;
    if (j < ysz) {
        int t0 = j / H + 1;                           // first frame with N + j - H t < N
        int t1 = (N + j) / H;                         // last frame with N + j - H t >= 0
        if (t1 > OT - 1) t1 = OT - 1;
        const float* fb = frs + (size_t)b * OT * N;
        float s = 0.f;
        for (int t = t0; t <= t1; ++t) {            // up to 6 split-K slabs of the synthesis GEMM, loads issued together, fixed order (all frames' loads at once measured 7 % slower: 24 loads for 9 live ones)
            const size_t o = (size_t)t * N + (N + j - H * t);
            float v[6];
#pragma unroll
            for (int z = 0; z < 6; ++z) v[z] = fb[(z < nslab ? z * slab : 0) + o];
#pragma unroll
            for (int z = 0; z < 6; ++z) s += z < nslab ? v[z] : 0.f;
        }
        const float out = x ? 2.0f * (s + 0.5f * x[(size_t)b * L + (L - ysz) + j]) : s;   // x == NULL: plain Synthesis.forward (cls_fe_dft.py:112-113)
        if (y_hat) y_hat[(size_t)b * ysz + j] = out;
        if (y_true) {
            const float dlt = y_true[(size_t)b * ysz + j] - out;
            const float a = fabsf(dlt);
            // log(cosh(d)) = log1p(2 sinh^2(d/2)): no cancellation for the small residuals of a trained model (the
            // a + log1p(e^-2a) - ln2 form loses ~1e-7 absolute per sample, i.e. 1e-3 of a 1e-4 mean); large |d|: overflow-free form
            if (a < 8.0f) { const float u = expm1f(0.5f * a); const float sh = 0.5f * (u + u / (u + 1.0f)); lc = log1pf(2.0f * sh * sh); }
            else lc = a + log1pf(__expf(-2.0f * a)) - 0.69314718056f;
            const float ds = -2.0f * tanhf(dlt) * inv_count;
            if (dsyn16) dsyn16[(size_t)b * (ysz + 2 * dsyn_pad) + dsyn_pad + j] = st_to_h16(ds, ht);
            else if (dsyn) dsyn[(size_t)b * (ysz + 2 * dsyn_pad) + dsyn_pad + j] = ds;   // dsyn_pad > 0: padded layout for the framed loaders
        }
    }
    if ((dsyn || dsyn16) && dsyn_pad > 0) {      // zero margins of the padded gradient signal (2*pad floats per window)
        const size_t ro = (size_t)b * (ysz + 2 * dsyn_pad);
        for (int m = blockIdx.x * 256 + threadIdx.x; m < 2 * dsyn_pad; m += gridDim.x * 256) {
            if (dsyn16) dsyn16[ro + (m < dsyn_pad ? m : ysz + m)] = 0;
            else dsyn[ro + (m < dsyn_pad ? m : ysz + m)] = 0.f;
        }
    }
    if (loss_partial) {      // the canonical tree of a slot (described at ola_loss4_kernel below): quads in sample order, then a balanced tree over the 64 quad sums
        const int lane = threadIdx.x & 63, g = lane & ~3;
        const float a0 = __shfl(lc, g), a1 = __shfl(lc, g + 1), a2 = __shfl(lc, g + 2), a3 = __shfl(lc, g + 3);
        float q = ((a0 + a1) + a2) + a3;
#pragma unroll
        for (int o = 4; o < 64; o <<= 1) q += __shfl_xor(q, o);      // the 16 quads of this wave: levels 1-4 of the tree
        if (lane == 0) red[threadIdx.x >> 6] = q;
        __syncthreads();
        if (threadIdx.x == 0) loss_partial[blockIdx.y * gridDim.x + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);      // levels 5 and 6
    }
}

// Round 5: the same, FOUR output samples per thread.  H, N, y (and with them every frame offset N + j - H t) are multiples of 4 for the reference's geometries, so the
// (at most three) frames x (at most three) slabs a sample quartet needs are 16-byte loads, all nine issued before the first add; same order of additions per sample as
// ola_loss_kernel (frames outer, slabs inner): y_hat is bit-identical.  The scalar kernel issued 6 dword loads per frame (3 of them dummies) inside a run-time-bounded
// loop -- 2-3 dependent round trips per thread, 3.1 TB/s at B = 256 and 2.0 TB/s at the 65536-sample window.  Loss partials keep their slot count (ceil(y / 256) per
// window, st_ola_loss_partials) AND their bits (the first version summed 1024 samples into one slot and zeroed three, so the loss depended on which kernel the
// pointers' alignment selected): slot s = samples [256 s, 256 s + 256) of the window, summed by ONE tree in both kernels -- quads ((l0 + l1) + l2) + l3 in sample order, then the
// balanced binary tree over the 64 quad sums by quad index (xor butterfly, offsets 1, 2, 4, ..., 32; a + b == b + a, so every lane holds the same bits).  Here a wave's 64 lanes
// hold exactly the 64 quads of one slot; the scalar kernel gathers a quad from four lanes, runs levels 1-4 inside each wave and levels 5-6 over its four waves.
template <int NS>
__global__ void __launch_bounds__(256)
ola_loss4_kernel(const float* __restrict__ frs, const float* __restrict__ x, const float* __restrict__ y_true,
                 float* __restrict__ y_hat, float* __restrict__ dsyn, float* __restrict__ loss_partial,
                 int L, int N, int H, int OT, int ysz, float inv_count, size_t slab, int dsyn_pad, int nslot,
                 unsigned short* __restrict__ dsyn16 = nullptr, int ht = 0)
{
    const int b = blockIdx.y;
    const int j = 4 * (blockIdx.x * 256 + threadIdx.x);
    float lc = 0.f;
    if (j < ysz) {
        const int t0 = j / H + 1;                     // first frame with N + j - H t < N  (the same for j .. j + 3: H is a multiple of 4)
        int t1 = (N + j) / H;                         // last frame with N + j - H t >= 0
        if (t1 > OT - 1) t1 = OT - 1;
        const float* fb = frs + (size_t)b * OT * N;
        float4 v[3][NS];
#pragma unroll
        for (int u = 0; u < 3; ++u) {                 // all loads first (a frame past t1 re-reads frame t1: valid address, value dropped)
            const int t = t0 + u <= t1 ? t0 + u : t1;
            const size_t o = (size_t)t * N + (size_t)(N + j - H * t);
#pragma unroll
            for (int z = 0; z < NS; ++z) v[u][z] = *reinterpret_cast<const float4*>(fb + z * slab + o);
        }
        float4 xv = make_float4(0.f, 0.f, 0.f, 0.f), yt = xv;
        if (x) xv = *reinterpret_cast<const float4*>(x + (size_t)b * L + (L - ysz) + j);
        if (y_true) yt = *reinterpret_cast<const float4*>(y_true + (size_t)b * ysz + j);
        float s[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const bool live = t0 + u <= t1;
#pragma unroll
            for (int z = 0; z < NS; ++z) {
                s[0] += live ? v[u][z].x : 0.f; s[1] += live ? v[u][z].y : 0.f; s[2] += live ? v[u][z].z : 0.f; s[3] += live ? v[u][z].w : 0.f;
            }
        }
        const float xs[4] = {xv.x, xv.y, xv.z, xv.w}, ys[4] = {yt.x, yt.y, yt.z, yt.w};
        float out[4], ds[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            out[q] = x ? 2.0f * (s[q] + 0.5f * xs[q]) : s[q];
            const float dlt = ys[q] - out[q];
            const float a = fabsf(dlt);
            float l1;
            if (a < 8.0f) { const float u = expm1f(0.5f * a); const float sh = 0.5f * (u + u / (u + 1.0f)); l1 = log1pf(2.0f * sh * sh); }
            else l1 = a + log1pf(__expf(-2.0f * a)) - 0.69314718056f;
            lc += y_true ? l1 : 0.f;
            ds[q] = -2.0f * tanhf(dlt) * inv_count;
        }
        if (y_hat) *reinterpret_cast<float4*>(y_hat + (size_t)b * ysz + j) = make_float4(out[0], out[1], out[2], out[3]);
        if (y_true) {
            const size_t o = (size_t)b * (ysz + 2 * dsyn_pad) + dsyn_pad + j;
            if (dsyn16) *reinterpret_cast<uint2*>(dsyn16 + o) = st_to_h16x4(make_float4(ds[0], ds[1], ds[2], ds[3]), ht);
            else if (dsyn) *reinterpret_cast<float4*>(dsyn + o) = make_float4(ds[0], ds[1], ds[2], ds[3]);
        }
    }
    if ((dsyn || dsyn16) && dsyn_pad > 0) {      // zero margins of the padded gradient signal (2*pad floats per window)
        const size_t ro = (size_t)b * (ysz + 2 * dsyn_pad);
        for (int m = blockIdx.x * 256 + threadIdx.x; m < 2 * dsyn_pad; m += gridDim.x * 256) {
            if (dsyn16) dsyn16[ro + (m < dsyn_pad ? m : ysz + m)] = 0;
            else dsyn[ro + (m < dsyn_pad ? m : ysz + m)] = 0.f;
        }
    }
    if (loss_partial) {      // a wave's 64 quads ARE one slot (256 samples): the same tree as the scalar kernel's, no LDS, no barrier
        float q = lc;        // ((l0 + l1) + l2) + l3, formed sequentially above
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) q += __shfl_xor(q, o);
        const int slot = 4 * blockIdx.x + (threadIdx.x >> 6);
        if ((threadIdx.x & 63) == 0 && slot < nslot) loss_partial[blockIdx.y * nslot + slot] = q;
    }
}

// out[b][pad + Ls + pad] = zero margins | s * in[b][Ls]: the padded, pre-scaled signal the framed GEMM loaders read
// (x/2 of nn_proc.py:307 with the Conv1d padding of cls_fe_dft.py:28-31 materialised once per step, 10 MB at B=256).
__device__ __forceinline__ void pad_scale_block(const float* __restrict__ in, float* __restrict__ out, const int Ls, const int pad, const float s,
                                                const int bx, const int nbx, const int b, unsigned short* __restrict__ out16 = nullptr, const int ht = 0)
{
    const int Lp4 = (Ls + 2 * pad) / 4;
    const float4* src = reinterpret_cast<const float4*>(in + (size_t)b * Ls);
    float4* dst = reinterpret_cast<float4*>(out + (size_t)b * (Ls + 2 * pad));
    uint2* dst16 = reinterpret_cast<uint2*>(out16 + (size_t)b * (Ls + 2 * pad));
    for (int i = bx * 256 + threadIdx.x; i < Lp4; i += nbx * 256) {
        const int j = i - pad / 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (j >= 0 && j < Ls / 4) { v = src[j]; v.x *= s; v.y *= s; v.z *= s; v.w *= s; }
        if (out16) dst16[i] = st_to_h16x4(v, ht);      // 16-bit configurations: the GEMM operand, rounded once here (x/2 is exact in either type's exponent range)
        else dst[i] = v;
    }
}
__global__ void __launch_bounds__(256)
pad_scale_kernel(const float* __restrict__ in, float* __restrict__ out, int Ls, int pad, float s)
{
    pad_scale_block(in, out, Ls, pad, s, blockIdx.x, gridDim.x, blockIdx.y);
}

// Everything the fused forward needs before its first GEMM, in ONE launch (three independent HBM-bound jobs that were three
// launches of 6-8 us each, mostly launch ramp): [0, n_pad) the padded, pre-scaled signal copy; [n_pad, n_pad + n_fold) the Hermitian
// fold of the synthesis bases (they change only in the optimizer, but the workspace is the caller's and may be re-carved between
// steps, so the fold is rebuilt per step -- 12 MB of traffic, as 32 x 32 tiles written in both orientations); the rest zeroes the frames that lie wholly in the Conv1d padding.
// The zero-padded copies of W_1 ([64][Tp]) and W_5 ([16][32]) of both autoencoders for the wide autoencoder path (st_ae_wide.h): up to four row-padding jobs
struct PadJobs { const float* src[4]; float* dst[4]; int rows[4], cols[4], pitch[4], blk0[5]; };
__device__ __forceinline__ void pad_rows4_block(const PadJobs& j, const int blk)
{
    int q = 0;
#pragma unroll
    for (int k = 1; k < 4; ++k) q += blk >= j.blk0[k];
    const int i = (blk - j.blk0[q]) * 256 + threadIdx.x;
    if (i >= j.rows[q] * j.pitch[q]) return;
    const int r = i / j.pitch[q], c = i - r * j.pitch[q];
    j.dst[q][i] = c < j.cols[q] ? j.src[q][r * j.cols[q] + c] : 0.f;
}
// Round 4, wide geometries (T > 32 or OT > 16): the analysis GEMM's polar epilogue writes mag / phs STRAIGHT into the feature-major layout the wide
// autoencoder path reads (V[a][t][b * FP + f], st_ae_wide.h) -- the transposing copy kernel in between (wide_in_kernel: 91 MB, 18 us at the
// 65536-sample window) is gone from the fused step.  What that kernel did on the side moves here: the pad columns f in [F, FP) and the all-padding
// frames of V as zeros, the knob rows of the layer-5 input, the zero-padded weight copies.
struct PrepWide {
    float *Vm, *Vp; int FP, B; unsigned R;            // R = B * FP columns
    float *H4Km, *H4Kp; const float* knobs; int K;     // knob rows 16 + k of the layer-5 input of both nets
    PadJobs pj;
    int n_vpad, n_kn, n_pj;                            // block counts of the three jobs (0: not a wide geometry)
};
struct PrepArgs {
    const float* x; float* xp; int Ls, pad; float scale; int nbx, n_pad;
    const float* Sr; const float* Si; float* Sfold; float* SfoldT; int N, F, KP, n_fold;
    float *re, *im, *mag, *phs; int T, t_lo, Tv;
    // 16-bit configurations (st_gemm16.h): ht = 1 bf16 / 2 fp16 -- the padded waveform and both folds go out in 16 bits INSTEAD of fp32,
    // and n_w16 more blocks write the F used rows of the analysis bases as rows (bin, re | im) interleaved: W16[2 bin + part][N]
    int ht; unsigned short *xp16, *Sfold16, *SfoldT16, *W16; const float* Wr; const float* Wi; int n_w16, n_dead;
    PrepWide wd;
};
// 32 x 32 tile of the folded synthesis bases, written twice: Sfold [KP][N] (rows k: the K-contiguous operand of the synthesis data-gradient
// GEMM) and its transpose SfoldT [N][KP] (rows n: the K-contiguous operand of the synthesis FRAMES GEMM, which otherwise has to take
// Sfold as an M/N-contiguous operand with scalar LDS fragment reads -- 52 % of the fp32 MFMA peak against 69 % for the NT x NT form).
__device__ __forceinline__ void fold_tile(const float* __restrict__ Sr, const float* __restrict__ Si, float* __restrict__ Sfold,
                                          float* __restrict__ SfoldT, const int N, const int F, const int KP, const int tile,
                                          unsigned short* __restrict__ Sfold16 = nullptr, unsigned short* __restrict__ SfoldT16 = nullptr, const int ht = 0)
{
    __shared__ float tl[32][33];
    const int ntn = N / 32, tk = tile / ntn, tn = tile - tk * ntn;
    const int half = KP / 2;
    const int cx = threadIdx.x & 31, ry = threadIdx.x >> 5;          // 32 columns x 8 row-lanes
    // all eight loads of a thread (row k and its mirror N - k for four rows) are requested before the first is used, from clamped -- always valid -- rows:
    // inside the row conditions each one was its own round trip
    float a0[4], a1[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int row = tk * 32 + ry + 8 * j, n = tn * 32 + cx;
        const bool is_im = row >= half;
        const int k = is_im ? row - half : row;
        const float* S = is_im ? Si : Sr;
        const int kc = k < F ? k : F - 1, km = (k >= 1 && k <= F - 2) ? N - k : N - 1;
        a0[j] = S[(size_t)kc * N + n]; a1[j] = S[(size_t)km * N + n];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int row = tk * 32 + ry + 8 * j, n = tn * 32 + cx;
        const bool is_im = row >= half;
        const int k = is_im ? row - half : row;
        float v = 0.f;
        if (k < F) {
            v = a0[j];
            if (k >= 1 && k <= F - 2) v += is_im ? -a1[j] : a1[j];
        }
        if (Sfold16) Sfold16[(size_t)row * N + n] = st_to_h16(v, ht); else Sfold[(size_t)row * N + n] = v;
        tl[ry + 8 * j][cx] = v;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int n = tn * 32 + ry + 8 * j, row = tk * 32 + cx;
        if (SfoldT16) SfoldT16[(size_t)n * KP + row] = st_to_h16(tl[cx][ry + 8 * j], ht); else SfoldT[(size_t)n * KP + row] = tl[cx][ry + 8 * j];
    }
}
__global__ void __launch_bounds__(256)
prep_kernel(const PrepArgs a)
{
    const int blk = blockIdx.x;
    if (blk < a.n_pad) { const int b = blk / a.nbx; pad_scale_block(a.x, a.xp, a.Ls, a.pad, a.scale, blk - b * a.nbx, a.nbx, b, a.ht ? a.xp16 : nullptr, a.ht); }
    else if (blk < a.n_pad + a.n_fold) fold_tile(a.Sr, a.Si, a.Sfold, a.SfoldT, a.N, a.F, a.KP, blk - a.n_pad, a.ht ? a.Sfold16 : nullptr, a.ht ? a.SfoldT16 : nullptr, a.ht);
    else if (blk < a.n_pad + a.n_fold + a.n_dead) {
        zero_dead_frame(a.re, a.im, a.mag, a.phs, a.T, a.F, a.t_lo, a.Tv, blk - a.n_pad - a.n_fold);
        if (a.wd.n_vpad) {                           // ... and the same frame of the feature-major copy, pad columns included
            const int nd = a.T - a.Tv, j0 = blk - a.n_pad - a.n_fold, b = j0 / nd, j = j0 - b * nd, t = j < a.t_lo ? j : j + a.Tv;
            const size_t o = (size_t)t * a.wd.R + (size_t)b * a.wd.FP;
            for (int f = threadIdx.x; f < a.wd.FP; f += 256) { a.wd.Vm[o + f] = 0.f; a.wd.Vp[o + f] = 0.f; }
        }
    }
    else if (blk < a.n_pad + a.n_fold + a.n_dead + a.wd.n_vpad) {          // pad columns f in [F, FP) of every frame of V
        const int np = a.wd.FP - a.F;
        const unsigned idx = (unsigned)(blk - a.n_pad - a.n_fold - a.n_dead) * 256u + threadIdx.x;
        if (np > 0 && idx < (unsigned)a.wd.B * (unsigned)a.T * (unsigned)np) {
            const unsigned bt = idx / (unsigned)np, pcol = idx - bt * (unsigned)np, b = bt / (unsigned)a.T, t = bt - b * (unsigned)a.T;
            const size_t o = (size_t)t * a.wd.R + (size_t)b * a.wd.FP + a.F + pcol;
            a.wd.Vm[o] = 0.f; a.wd.Vp[o] = 0.f;
        }
    }
    else if (blk < a.n_pad + a.n_fold + a.n_dead + a.wd.n_vpad + a.wd.n_kn) {   // knob rows: H4K[16 + k][b * FP + f] = knobs[b][k] (0 in the pad columns)
        const unsigned Q = (unsigned)a.wd.FP / 4, idx = (unsigned)(blk - a.n_pad - a.n_fold - a.n_dead - a.wd.n_vpad) * 256u + threadIdx.x;
        if (idx < (unsigned)a.wd.K * (unsigned)a.wd.B * Q) {
            const unsigned kb = idx / Q, j = idx - kb * Q, k = kb / (unsigned)a.wd.B, b = kb - k * (unsigned)a.wd.B;
            const int f0 = 4 * (int)j;
            const float v = a.wd.knobs[b * a.wd.K + k];
            const float4 q = make_float4(f0 < a.F ? v : 0.f, f0 + 1 < a.F ? v : 0.f, f0 + 2 < a.F ? v : 0.f, f0 + 3 < a.F ? v : 0.f);
            const size_t o = (size_t)(16 + k) * a.wd.R + (size_t)b * a.wd.FP;
            reinterpret_cast<float4*>(a.wd.H4Km + o)[j] = q; reinterpret_cast<float4*>(a.wd.H4Kp + o)[j] = q;
        }
    }
    else if (blk < a.n_pad + a.n_fold + a.n_dead + a.wd.n_vpad + a.wd.n_kn + a.wd.n_pj)
        pad_rows4_block(a.wd.pj, blk - a.n_pad - a.n_fold - a.n_dead - a.wd.n_vpad - a.wd.n_kn);
    else {                                           // analysis bases, 4 taps per thread
        const size_t i4 = (size_t)(blk - a.n_pad - a.n_fold - a.n_dead - a.wd.n_vpad - a.wd.n_kn - a.wd.n_pj) * 256 + threadIdx.x, n4 = a.N / 4;
        if (i4 < (size_t)2 * a.F * n4) {
            const int jrow = (int)(i4 / n4), c = (int)(i4 - (size_t)jrow * n4);
            const float4 v = reinterpret_cast<const float4*>(((jrow & 1) ? a.Wi : a.Wr) + (size_t)(jrow >> 1) * a.N)[c];
            reinterpret_cast<uint2*>(a.W16 + (size_t)jrow * a.N)[c] = st_to_h16x4(v, a.ht);
        }
    }
}

// dsyn = 2 * g_y_hat  (y_hat = 2*(syn + x/2), nn_proc.py:332,340) -- generic autograd entry
__global__ void __launch_bounds__(256)
scale_kernel(const float* __restrict__ in, float* __restrict__ out, int64_t n, float s)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) out[i] = s * in[i];
}

// ---------------------------------------------------------------- polar backward (nn_proc.py:309-310)
// dre = dmag*re/mag [0 at mag==0] - dphs*im/((re+eps)^2+im^2);  dim = dmag*im/mag + dphs*(re+eps)/(...)
// Output dG[R,KP]: d re at [0,F), d im at [KP/2,KP/2+F), pads written as zeros.
__device__ __forceinline__ void polar_bwd_block(const float* __restrict__ re, const float* __restrict__ im, const float* __restrict__ dmag,
                 const float* __restrict__ dphs, const float* __restrict__ g_mag, float* __restrict__ dG, const int F, const int KP,
                 const float sat,      // > 0: saturate the result to +-sat (the consumer GEMM narrows it to fp16, see below)
                 const int bx, const int r, unsigned short* __restrict__ dG16 = nullptr, const int ht = 0)
{
    const int half = KP / 2;
    const int c = bx * 256 + threadIdx.x;     // column in [0, half)
    if (c >= half) return;
    float gre = 0.f, gim = 0.f;
    if (c < F) {
        const size_t i = (size_t)r * F + c;
        const float a = re[i], bb = im[i], dm = dmag[i] + (g_mag ? g_mag[i] : 0.f), dp = dphs[i];
        const float mg = sqrtf(a * a + bb * bb);
        const float inv = mg > 0.f ? 1.0f / mg : 0.f;
        const float rp = a + 1e-7f;
        const float den = rp * rp + bb * bb;
        gre = dm * a * inv - dp * bb / den;
        gim = dm * bb * inv + dp * rp / den;
        // fp16 configurations: d atan2 is ~1e7 x dphs on (near-)silent frames; times the loss scale that leaves the
        // fp16 range, and inf x 0 (the silent frame itself, the other GEMM operand) would poison the weight gradient with NaN.
        // The sub-gradient is computed in fp32 as always and saturated HERE, before the consumer narrows it.
        if (sat > 0.f) { gre = __builtin_amdgcn_fmed3f(gre, -sat, sat); gim = __builtin_amdgcn_fmed3f(gim, -sat, sat); }
    }
    if (dG16) { dG16[(size_t)r * KP + c] = st_to_h16(gre, ht); dG16[(size_t)r * KP + half + c] = st_to_h16(gim, ht); }      // the operand of the analysis weight-gradient GEMM (st_gemm16.h)
    if (dG) { dG[(size_t)r * KP + c] = gre; dG[(size_t)r * KP + half + c] = gim; }
}
// The same over the flattened (row, column) space, FOUR elements per thread with all sixteen loads issued before the first use (post_ae_kernel):
// one element per thread -- three blocks per 528-column row, the third with 16 live lanes -- ran at half the HBM rate (71 us for 264 MB at B = 1024).
// e = r * half + c;  r = umulhi(e, magic) with magic = ceil(2^32 / half), exact for the e the host admits (polar_flat_ok).
constexpr int POLAR_EPT = 4;
__host__ __device__ static inline unsigned polar_magic(int half) { return (unsigned)((0x100000000ull + (unsigned)half - 1) / (unsigned)half); }
__host__ static inline bool polar_flat_ok(long long R, int half)
{
    const unsigned long long M = polar_magic(half), slack = M * (unsigned)half - 0x100000000ull;      // q is exact while e * slack < 2^32
    const unsigned long long total = (unsigned long long)R * (unsigned)half;
    return total < 0x7fffffffull && (slack == 0 || total < 0x100000000ull / slack);
}
__device__ __forceinline__ void polar_bwd_flat(const float* __restrict__ re, const float* __restrict__ im, const float* __restrict__ dmag,
                 const float* __restrict__ dphs, const float* __restrict__ g_mag, float* __restrict__ dG, const int F, const int KP, const float sat,
                 const unsigned bx, const unsigned total, const unsigned magic, unsigned short* __restrict__ dG16, const int ht)
{
    const unsigned half = (unsigned)KP / 2;
    unsigned r[POLAR_EPT], c[POLAR_EPT]; bool on[POLAR_EPT], live[POLAR_EPT];
    float a[POLAR_EPT], bb[POLAR_EPT], dm[POLAR_EPT], dp[POLAR_EPT];
#pragma unroll
    for (int u = 0; u < POLAR_EPT; ++u) {
        const unsigned e = (bx * POLAR_EPT + u) * 256 + threadIdx.x;
        on[u] = e < total;
        r[u] = __umulhi(on[u] ? e : 0u, magic); c[u] = (on[u] ? e : 0u) - r[u] * half;
        live[u] = on[u] && c[u] < (unsigned)F;
        const size_t i = live[u] ? (size_t)r[u] * F + c[u] : 0;
        a[u] = re[i]; bb[u] = im[i]; dm[u] = dmag[i] + (g_mag ? g_mag[i] : 0.f); dp[u] = dphs[i];
    }
#pragma unroll
    for (int u = 0; u < POLAR_EPT; ++u) {
        float gre = 0.f, gim = 0.f;
        if (live[u]) {
            const float mg = sqrtf(a[u] * a[u] + bb[u] * bb[u]);
            const float inv = mg > 0.f ? 1.0f / mg : 0.f;
            const float rp = a[u] + 1e-7f;
            const float den = rp * rp + bb[u] * bb[u];
            gre = dm[u] * a[u] * inv - dp[u] * bb[u] / den;
            gim = dm[u] * bb[u] * inv + dp[u] * rp / den;
            if (sat > 0.f) { gre = __builtin_amdgcn_fmed3f(gre, -sat, sat); gim = __builtin_amdgcn_fmed3f(gim, -sat, sat); }
        }
        if (on[u]) {
            const size_t o = (size_t)r[u] * KP + c[u];
            if (dG16) { dG16[o] = st_to_h16(gre, ht); dG16[o + half] = st_to_h16(gim, ht); }
            if (dG) { dG[o] = gre; dG[o + half] = gim; }
        }
    }
}
__global__ void __launch_bounds__(256)
polar_bwd_kernel(const float* __restrict__ re, const float* __restrict__ im, const float* __restrict__ dmag,
                 const float* __restrict__ dphs, const float* __restrict__ g_mag, float* __restrict__ dG, int R, int F, int KP, const float sat,
                 unsigned short* __restrict__ dG16 = nullptr, const int ht = 0)
{
    polar_bwd_block(re, im, dmag, dphs, g_mag, dG, F, KP, sat, blockIdx.x, blockIdx.y, dG16, ht);
}

// ---------------------------------------------------------------- split-K slab reduce (+ unfold, + |g| sums)
// ws[nz][KP][N] -> gradient tensors [N,N].  mode 0 (analysis): row k<F -> gRe[k], row half+k -> gIm[k].
// mode 1 (synthesis): additionally mirror to row N-k with sign +1 (real) / -1 (imag)  (SURVEY.md 8a' "unfold").
//
// Round 3: the 128 x 128-tile weight-gradient GEMM (st_gemm_tn.h) covers the 2 (F - 1) = N rows that tile exactly and leaves the
// two NYQUIST rows (bin F - 1 of the real and of the imaginary basis) to plain FMAs: g[n] = sum_k A[k][col] * B[k][n] is formed as P
// partial sums over groups of windows by one extra z-slice of workgroups of the GEMM launch itself (st_gemm_tn.h nyq_partial: light
// vector work beside the MFMA workgroups) and the P partials are added here, in a fixed order.  A ninth tile row for two rows of
// output would cost the GEMM 11 % more MFMA work; a single-level dot product in this kernel measured 37 us (latency-bound).
struct NyqJob {
    const float* part; int P;        // partials [P][2][N] (real row, imaginary row) written by the GEMM launch's extra workgroups
    int on;                          // 0: the Nyquist rows come from the slabs like every other row (gemm_kernel<3, ...> wrote them)
};
// The packed copy of the live analysis rows for the data-parallel exchange: fp32, or (st_dp_train_step exchange flag 4, the 16-bit configurations)
// bfloat16 in the first half of the same buffer -- the exposed collective then moves half the bytes
__device__ __forceinline__ void stage_store4(float* __restrict__ stage, const size_t elem, const float4 v, const int bf16)
{
    if (bf16) {
        typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
        typedef float f32x2_t __attribute__((ext_vector_type(2)));
        union { bf16x2_t h; unsigned u; } lo, hi;
        lo.h = __builtin_convertvector((f32x2_t){v.x, v.y}, bf16x2_t); hi.h = __builtin_convertvector((f32x2_t){v.z, v.w}, bf16x2_t);
        *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(stage) + elem) = make_uint2(lo.u, hi.u);
    } else *reinterpret_cast<float4*>(stage + elem) = v;
}
constexpr int NYQ_CW = 256;                              // output columns per Nyquist block (64 float4 lanes x 4 partial groups)
__host__ __device__ static inline int nyq_blocks(int N) { return 2 * ((N + NYQ_CW - 1) / NYQ_CW); }
__host__ __device__ static inline int norm_partial_count(int F, int N) { return 2 * F + nyq_blocks(N); }
__device__ __forceinline__ void
nyquist_chunk(const NyqJob& q, float* __restrict__ gRe, float* __restrict__ gIm, float* __restrict__ norm_partial,
              const int N, const int F, const int blk, const int slot, float* __restrict__ stage, const int stage_bf16 = 0)
{
    __shared__ float4 part[4][64];
    __shared__ float red[4];
    const int per = (N + NYQ_CW - 1) / NYQ_CW;
    const bool is_im = blk >= per;
    const int n4 = (is_im ? blk - per : blk) * (NYQ_CW / 4) + (threadIdx.x & 63), grp = threadIdx.x >> 6;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (4 * n4 < N)
        for (int p0 = grp; p0 < q.P; p0 += 32) {            // eight partials in flight per trip (same order of additions)
            float4 u[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) { const int p = p0 + 4 * j < q.P ? p0 + 4 * j : grp; u[j] = *reinterpret_cast<const float4*>(q.part + ((size_t)p * 2 + (is_im ? 1 : 0)) * N + 4 * n4); }
#pragma unroll
            for (int j = 0; j < 8; ++j) if (p0 + 4 * j < q.P) { v.x += u[j].x; v.y += u[j].y; v.z += u[j].z; v.w += u[j].w; }
        }
    part[grp][threadIdx.x & 63] = v;
    __syncthreads();
    float na = 0.f;
    if (threadIdx.x < 64 && 4 * n4 < N) {
        const float4 a = part[0][threadIdx.x], b = part[1][threadIdx.x], c = part[2][threadIdx.x], d = part[3][threadIdx.x];
        v = make_float4((a.x + b.x) + (c.x + d.x), (a.y + b.y) + (c.y + d.y), (a.z + b.z) + (c.z + d.z), (a.w + b.w) + (c.w + d.w));
        *reinterpret_cast<float4*>((is_im ? gIm : gRe) + (size_t)(F - 1) * N + 4 * n4) = v;
        if (stage) stage_store4(stage, (size_t)((is_im ? F : 0) + F - 1) * N + 4 * n4, v, stage_bf16);
        na = fabsf(v.x) + fabsf(v.y) + fabsf(v.z) + fabsf(v.w);
    }
    const float tot = block_sum<4>(na, red);
    if (threadIdx.x == 0) norm_partial[slot] = tot;
}
__device__ __forceinline__ void
wgrad_reduce_block(const float* __restrict__ ws, int nz, float* __restrict__ gRe, float* __restrict__ gIm,
                   float* __restrict__ norm_partial, int N, int F, int KP, int mode, const int row, float* __restrict__ stage, const NyqJob& nyq,
                   const int stage_bf16 = 0)
{
    __shared__ float red[4];                          // row: 0 .. 2F-1 : [0,F) real rows, [F,2F) imag rows; >= 2F: Nyquist blocks
    if (row >= 2 * F) {
        if (nyq.on) nyquist_chunk(nyq, gRe, gIm, norm_partial, N, F, row - 2 * F, row, stage, stage_bf16);
        else if (threadIdx.x == 0) norm_partial[row] = 0.f;      // the partial count is fixed (norm_partial_count): unused slots read as 0
        return;
    }
    const bool is_im = row >= F;
    const int k = is_im ? row - F : row;
    if (nyq.on && k == F - 1) { if (threadIdx.x == 0) norm_partial[row] = 0.f; return; }
    const int src = is_im ? KP / 2 + k : k;
    float* g = is_im ? gIm : gRe;
    const bool mirror = mode == 1 && k >= 1 && k <= F - 2;
    const float sgn = is_im ? -1.f : 1.f;
    const size_t slab = (size_t)KP * N;
    float na = 0.f;
    for (int n4 = threadIdx.x; n4 < N / 4; n4 += blockDim.x) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        // eight slabs requested before the first is added (same summation order, same bits): as one load per trip of a run-time loop every slab was its
        // own memory round trip -- 7 of them at B = 256 -- and this 4 MB kernel took 9 us
        for (int z0 = 0; z0 < nz; z0 += 8) {
            float4 u[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) { const int z = z0 + j < nz ? z0 + j : nz - 1; u[j] = reinterpret_cast<const float4*>(ws + z * slab + (size_t)src * N)[n4]; }
#pragma unroll
            for (int j = 0; j < 8; ++j) if (z0 + j < nz) { v.x += u[j].x; v.y += u[j].y; v.z += u[j].z; v.w += u[j].w; }
        }
        reinterpret_cast<float4*>(g + (size_t)k * N)[n4] = v;
        if (stage) stage_store4(stage, (size_t)row * N + 4 * n4, v, stage_bf16);      // packed copy [2F][N] of the live rows (data-parallel all-reduce buffer)
        float a = fabsf(v.x) + fabsf(v.y) + fabsf(v.z) + fabsf(v.w);
        if (mirror) {
            reinterpret_cast<float4*>(g + (size_t)(N - k) * N)[n4] = make_float4(sgn * v.x, sgn * v.y, sgn * v.z, sgn * v.w);
            a *= 2.f;
        }
        na += a;
    }
    const float tot = block_sum<4>(na, red);
    if (threadIdx.x == 0) norm_partial[row] = tot;
}
// grid: norm_partial_count(F, N) blocks (row0 = 0), or F + nyq_blocks(N) for one half (row0 = 0 / F: the staged data-parallel schedule,
// never with the Nyquist form)
__global__ void __launch_bounds__(256)
wgrad_reduce_kernel(const float* __restrict__ ws, int nz, float* __restrict__ gRe, float* __restrict__ gIm,
                    float* __restrict__ norm_partial, int N, int F, int KP, int mode, int row0, int nrows, float* __restrict__ stage, const NyqJob nyq,
                    const int extra0 = 0, const int stage_bf16 = 0)
{
    // blocks [0, nrows): gradient rows row0 ..; the rest: the Nyquist / unused partial slots 2F + extra0 .. (extra0: the one-basis launches of the split exchange)
    const int row = (int)blockIdx.x < nrows ? (int)blockIdx.x + row0 : 2 * F + extra0 + ((int)blockIdx.x - nrows);
    wgrad_reduce_block(ws, nz, gRe, gIm, norm_partial, N, F, KP, mode, row, stage, nyq, stage_bf16);
}

// L1 norm partials of an arbitrary flat range (data-parallel path: norm of the *reduced* gradient).
__global__ void __launch_bounds__(256)
l1_partial_kernel(const float* __restrict__ g, int64_t n, float scale, float* __restrict__ partial)
{
    __shared__ float red[4];
    float a = 0.f;
    const int64_t n4 = n / 4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const float4 v = reinterpret_cast<const float4*>(g)[i];
        a += fabsf(v.x * scale) + fabsf(v.y * scale) + fabsf(v.z * scale) + fabsf(v.w * scale);
    }
    const float tot = block_sum<4>(a, red);
    if (threadIdx.x == 0) partial[blockIdx.x] = tot;
}

// Data-parallel tail: the all-reduced live analysis rows [2F][N] go back into the two gradient tensors AND their scaled |g| sums are formed in the
// same pass (one partial per row; slots >= 2F of the fixed-count partial array are zeroed by the last block) -- was two device copies + a norm
// pass over the whole STFT range, all of it exposed behind the last collective.
__global__ void __launch_bounds__(256)
unstage_l1_kernel(const float* __restrict__ stage, float* __restrict__ gRe, float* __restrict__ gIm, const int F, const int N, const float scale,
                  float* __restrict__ partial, const int n_partial, const int stage_bf16 = 0)
{
    __shared__ float red[4];
    const int row = blockIdx.x;                       // 0 .. 2F-1
    const bool is_im = row >= F;
    float* dst = (is_im ? gIm : gRe) + (size_t)(is_im ? row - F : row) * N;
    const float* src = stage + (size_t)row * N;
    float a = 0.f;
    for (int i = threadIdx.x; i < N / 4; i += 256) {
        float4 v;
        if (stage_bf16) {          // the exchange ran on bfloat16 values (first half of the buffer): widen
            const uint2 u = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(stage) + (size_t)row * N + 4 * i);
            v = make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u));
        } else v = reinterpret_cast<const float4*>(src)[i];
        reinterpret_cast<float4*>(dst)[i] = v;
        a += fabsf(v.x * scale) + fabsf(v.y * scale) + fabsf(v.z * scale) + fabsf(v.w * scale);
    }
    const float tot = block_sum<4>(a, red);
    if (threadIdx.x == 0) partial[row] = tot;
    if (row == 2 * F - 1) for (int j = 2 * F + threadIdx.x; j < n_partial; j += 256) partial[j] = 0.f;
}

// ---------------------------------------------------------------- per-wave AE gradient partial reduce
// ws[nparts][2][PG] -> g_m[PG], g_p[PG].  Block = 64 columns x 4 partial-lanes; 8 loads in flight per thread.
// norm_out != NULL: the block also publishes sum |g| of the 64 gradient values it produced (slot `slot`): the clip over ALL parameters (st_dims.clip_all)
// needs the L1 norm of the autoencoder range, which was a 5 us launch of its own over 77 KB.
__device__ __forceinline__ void ae_grad_reduce_block(const float* __restrict__ ws, const int nparts, const int PG,
                                                     float* __restrict__ g_m, float* __restrict__ g_p, const int bx, const int ae,
                                                     float* __restrict__ norm_out = nullptr, const int slot = 0)
{
    __shared__ float red[4][64];
    const int col = threadIdx.x & 63, pl = threadIdx.x >> 6;
    const int i = bx * 64 + col;
    float s = 0.f;
    if (i < PG) {
        const float* base = ws + (size_t)ae * PG + i;
        const size_t stride = (size_t)2 * PG;
        for (int p0 = pl; p0 < nparts; p0 += 32) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { const int p = p0 + 4 * u; v[u] = p < nparts ? base[(size_t)p * stride] : 0.f; }
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
    }
    red[pl][col] = s;
    __syncthreads();
    if (pl != 0) return;                                   // wave 0 holds the 64 results
    const float r = i < PG ? (red[0][col] + red[1][col]) + (red[2][col] + red[3][col]) : 0.f;
    if (i < PG) (ae ? g_p : g_m)[i] = r;
    if (norm_out) { const float t = wave_sum(fabsf(r)); if (col == 0) norm_out[slot] = t; }
}
__global__ void __launch_bounds__(256)
ae_grad_reduce_kernel(const float* __restrict__ ws, int nparts, int PG, float* __restrict__ g_m, float* __restrict__ g_p)
{
    ae_grad_reduce_block(ws, nparts, PG, g_m, g_p, blockIdx.x, blockIdx.y);
}
// The two consumers of ae_bwd_kernel's outputs in ONE launch (they are independent of each other): blocks [0, n_red) sum the
// per-workgroup autoencoder gradient partials, the rest is the polar backward (blocks enumerate (row, column chunk)).
struct PostAeArgs {
    const float* ws; int nparts, PG; float* g_m; float* g_p; int n_red_x, n_red;
    const float* re; const float* im; const float* dmag; const float* dphs; const float* g_mag; float* dG; int F, KP, gx; float sat;
    // third role (fused step): the split-K slabs of the synthesis weight gradient, written before the autoencoder backward, are summed,
    // un-folded and normed here instead of in a launch of their own
    int n_polar; const float* wg; int wg_nz; float* gSr; float* gSi; float* norm_s; int N; NyqJob nyq;
    unsigned short* dG16; int ht;          // 16-bit configurations: d G rounded to the GEMM operand type (dG itself may then be NULL)
    unsigned polar_total, polar_magic;     // polar_total > 0: the polar blocks walk the flattened element space (polar_bwd_flat)
    float* norm_e;                         // n_red |g| partials of the autoencoder gradients (st_dims.clip_all), or NULL
};
__global__ void __launch_bounds__(256)
post_ae_kernel(const PostAeArgs a)
{
    const int blk = blockIdx.x;
    if (blk < a.n_red) { const int ae = blk / a.n_red_x; ae_grad_reduce_block(a.ws, a.nparts, a.PG, a.g_m, a.g_p, blk - ae * a.n_red_x, ae, a.norm_e, blk); }
    else if (blk < a.n_red + a.n_polar && a.polar_total) polar_bwd_flat(a.re, a.im, a.dmag, a.dphs, a.g_mag, a.dG, a.F, a.KP, a.sat, (unsigned)(blk - a.n_red), a.polar_total, a.polar_magic, a.dG16, a.ht);
    else if (blk < a.n_red + a.n_polar) { const int q = blk - a.n_red, r = q / a.gx; polar_bwd_block(a.re, a.im, a.dmag, a.dphs, a.g_mag, a.dG, a.F, a.KP, a.sat, q - r * a.gx, r, a.dG16, a.ht); }
    else wgrad_reduce_block(a.wg, a.wg_nz, a.gSr, a.gSi, a.norm_s, a.N, a.F, a.KP, 1, blk - a.n_red - a.n_polar, nullptr, a.nyq);
}

// ---------------------------------------------------------------- scalars
// scalars: [0]=loss [1]=mean logcosh [2]=reg term [3]=L1 norm of STFT grads [4]=clip coef
// (torch.nn.utils.clip_grad_norm_: coef = min(1, max_norm/(norm+1e-6)), nn_proc.py:299-302)
struct FinArgs {
    const float* loss_partial; int n_loss; const float* reg_partial; int n_reg;
    const float* norm_a; int n_na; const float* norm_s; int n_ns;
    float inv_ycount, reg_scale, norm_scale;
    const float* norm_e; int n_ne;        // clip over ALL parameters (st_dims.clip_all): |g| partials of the autoencoder range
};
// Block-wide: the loss terms (if want_loss) and the clip coefficient from the partial sums; every thread returns the
// coefficient, thread 0 also gets the loss terms.  One summation order, so any block that evaluates this gets the same bits.
__device__ __forceinline__ float finalize_block(const FinArgs& f, const bool want_loss, float* red, float& lc, float& rg, float& nrm)
{
    // a thread's share of a partial array: float4 loads, four of them in flight per trip (the arrays grow with the batch -- 8 partials per window
    // from ola_loss: as one dependent 4-byte load per trip, block 0 of the optimizer kernel spent 32 round trips = 20 us here at B = 1024 while
    // every other block waited for nothing)
    auto psum = [](const float* __restrict__ p, const int n) {
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        if ((reinterpret_cast<uintptr_t>(p) & 15) == 0) {
            const int n4 = n >> 2;
            const float4* p4 = reinterpret_cast<const float4*>(p);
            int i = threadIdx.x;
            for (; i + 768 < n4; i += 1024) {
                const float4 u0 = p4[i], u1 = p4[i + 256], u2 = p4[i + 512], u3 = p4[i + 768];
                s0 += (u0.x + u0.y) + (u0.z + u0.w); s1 += (u1.x + u1.y) + (u1.z + u1.w);
                s2 += (u2.x + u2.y) + (u2.z + u2.w); s3 += (u3.x + u3.y) + (u3.z + u3.w);
            }
            for (; i < n4; i += 256) { const float4 u = p4[i]; s0 += (u.x + u.y) + (u.z + u.w); }
            for (int j = 4 * n4 + threadIdx.x; j < n; j += 256) s1 += p[j];
        } else for (int j = threadIdx.x; j < n; j += 256) s0 += p[j];
        return (s0 + s1) + (s2 + s3);
    };
    float a = 0.f, b = 0.f, c = 0.f;
    if (want_loss && f.loss_partial) a = psum(f.loss_partial, f.n_loss);
    if (want_loss && f.reg_partial) b = psum(f.reg_partial, f.n_reg);
    if (f.norm_a) c += psum(f.norm_a, f.n_na);
    if (f.norm_s) c += psum(f.norm_s, f.n_ns);
    if (f.norm_e) c += psum(f.norm_e, f.n_ne);
    if (want_loss) { a = block_sum<4>(a, red); b = block_sum<4>(b, red); }
    c = block_sum<4>(c, red);
    __shared__ float bc;
    if (threadIdx.x == 0) bc = c;
    __syncthreads();
    lc = a * f.inv_ycount; rg = b * f.reg_scale; nrm = bc * f.norm_scale;
    const float coef = 1.0f / (nrm + 1e-6f);
    return coef < 1.0f ? coef : 1.0f;
}
__global__ void __launch_bounds__(256)
finalize_kernel(const FinArgs f, float* __restrict__ scalars)
{
    __shared__ float red[4];
    float lc, rg, nrm;
    const float coef = finalize_block(f, true, red, lc, rg, nrm);
    if (threadIdx.x == 0) {
        if (f.loss_partial) { scalars[1] = lc; scalars[2] = rg; scalars[0] = lc + rg; }
        if (f.norm_a || f.norm_s || f.norm_e) { scalars[3] = nrm; scalars[4] = coef; }
    }
}

// ---------------------------------------------------------------- L1 clip + Adam (train.py:145-147)
// torch.optim.Adam (single-tensor path): m.lerp_(g, 1-b1); v = v*b2 + (1-b2)*g*g;
// denom = sqrt(v)/sqrt(bc2) + eps ; p += (-lr/bc1) * m / denom.   STFT range [0,n_stft) is scaled by
// the clip coefficient first; every gradient is pre-scaled by grad_scale (1/world in data parallel).
// FIN: the scalars are not ready yet -- every block derives the clip coefficient from the partial sums itself (2k floats out
// of L2) and block 0 publishes the loss scalars: saves the single-block finalize launch (9 us) in the fused train step.
// [0, n4_clip) is the clipped range: the 4 STFT tensors (nn_proc.py:299-302) or everything (train.py:136, st_dims.clip_all).
// Mixed precision: a non-finite norm (a gradient overflowed under the loss scale) SKIPS the update on every block -- parameters
// and moments stay untouched -- and scalars[5] counts the skipped steps for the host's loss-scale policy (what Apex's dynamic
// scaler does with its overflow flag, train.py:134-135).
// DEV (the HIP-graph form of the step, st_graph_*): the step number and the learning rate are read from device memory
// (scalars[6], scalars[7], written by step_tick_kernel at the head of the graph) and the bias corrections are formed here, in
// double like the host does, so that ONE captured graph replays every iteration of a run.
template <bool FIN, bool DEV = false>
__global__ void __launch_bounds__(256)
clip_adam_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                 int64_t n4_total, int64_t n4_clip, float* __restrict__ scalars, float grad_scale,
                 float neg_step_size, float w1, float b2, float w2, float bc2_sqrt, float eps, const FinArgs fin, const float b1 = 0.f)
{
    if constexpr (DEV) {
        __shared__ float hs[2];
        if (threadIdx.x == 0) {
            const double st = (double)scalars[6];
            hs[0] = (float)(-(double)scalars[7] / (1.0 - pow((double)b1, st)));
            hs[1] = (float)sqrt(1.0 - pow((double)b2, st));
        }
        __syncthreads();
        neg_step_size = hs[0]; bc2_sqrt = hs[1];
    }
    float coef, nrm;
    if constexpr (FIN) {
        __shared__ float red[4];
        float lc, rg;
        coef = finalize_block(fin, blockIdx.x == 0, red, lc, rg, nrm);
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            if (fin.loss_partial) { scalars[1] = lc; scalars[2] = rg; scalars[0] = lc + rg; }
            scalars[3] = nrm; scalars[4] = coef;
        }
    } else {
        coef = scalars[4]; nrm = scalars[3];
    }
    if (!(nrm <= 3.0e38f)) {                         // inf or NaN: overflow under the loss scale -> skip this step everywhere
        if (blockIdx.x == 0 && threadIdx.x == 0) scalars[5] = scalars[5] + 1.0f;
        return;
    }
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4_total; i += (int64_t)gridDim.x * 256) {
        const float sc = (i < n4_clip) ? grad_scale * coef : grad_scale;
        float4 G = reinterpret_cast<float4*>(g)[i];
        float4 M = reinterpret_cast<float4*>(m)[i];
        float4 V = reinterpret_cast<float4*>(v)[i];
        // g = m = v = 0 is a fixed point of the update (m, v stay 0, p += 0): the 2 x 511 structurally-zero rows of the analysis
        // bases -- a quarter of all parameters -- skip the parameter read and all four writes
        if (G.x == 0.f && G.y == 0.f && G.z == 0.f && G.w == 0.f && M.x == 0.f && M.y == 0.f && M.z == 0.f && M.w == 0.f &&
            V.x == 0.f && V.y == 0.f && V.z == 0.f && V.w == 0.f) continue;
        float4 P = reinterpret_cast<float4*>(p)[i];
#define ST_ADAM1(c)                                                        \
        {                                                                  \
            const float gg = (sc == 1.0f) ? G.c : G.c * sc;                \
            M.c = M.c + w1 * (gg - M.c);                                   \
            V.c = V.c * b2 + (w2 * gg) * gg;                               \
            const float den = sqrtf(V.c) / bc2_sqrt + eps;                 \
            P.c = P.c + (neg_step_size * M.c) / den;                       \
            G.c = gg;                                                      \
        }
        ST_ADAM1(x) ST_ADAM1(y) ST_ADAM1(z) ST_ADAM1(w)
#undef ST_ADAM1
        if (sc != 1.0f) reinterpret_cast<float4*>(g)[i] = G;       // the clipped / rescaled gradient is observable in the reference (p.grad); with sc == 1 (clip inactive, one rank,
                                                                   // no loss scale) the buffer already holds these very values: 12.6 MB of writes less per step
        reinterpret_cast<float4*>(m)[i] = M;
        reinterpret_cast<float4*>(v)[i] = V;
        reinterpret_cast<float4*>(p)[i] = P;
    }
}


// Head of the captured step (st_graph_*): advance the device-side step counter and look the learning rate up in the device copy
// of the 1-cycle table -- iteration i (0-based) runs with lr_sched[max(i - 1, 0)] (train.py:150 writes the rate AFTER the step),
// i.e. step t = i + 1 uses entry max(t - 2, 0).
__global__ void step_tick_kernel(float* __restrict__ scalars, const float* __restrict__ lr_table, const int n_lr)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const float t = scalars[6] + 1.0f;
    int idx = (int)t - 2; idx = idx < 0 ? 0 : (idx >= n_lr ? n_lr - 1 : idx);
    scalars[6] = t; scalars[7] = lr_table[idx];
}

// ------------------------------------------------------------------------------ device-side data feed
// audio.compressor_4controls (audio.py:380-426) for a batch of windows: static gain curve (parallel), switched one-pole
// attack/release smoother of the gain in dB (inherently sequential per window: the coefficient depends on the previous
// OUTPUT), dB -> linear and apply (parallel).  One workgroup per window; the recurrence runs in one lane over an LDS
// copy of the gain curve, in chunks of CH samples so any window length fits.  State is rounded to float32 every step
// exactly like the reference's float32 lin_A array; the step arithmetic is float64 (numpy float64 scalars alphaA/R).
// y receives the LAST ysz samples of each processed window (the training target, datasets.py:327-330).
// the static gain curve of one sample (audio.py:392-399), shared by the compressor kernels and the feed generator.  float32 arithmetic, which is what the
// reference's lines give under NUMPY's promotion rules (x_dB / gainChange_dB float32 arrays, threshold / ratio weak Python scalars: 20 * log10(|x| + 1e-8),
// the clip at -96 and thresh + (x_dB - thresh) / ratio - x_dB all stay float32) -- the way golden G9 was captured, with numba's @jit stubbed to the identity
// (tools/_ref_import.py; numba is not in this image).  Under the real @jit(nopython=True) a float32 array plus the float64 literal 1e-8 promotes to float64,
// so x_dB, the threshold compare and gainChange are float64 until the store into the float32 array: the two differ by ~1e-7 relative (one float32 rounding of
// x_dB), far below anything a training TARGET needs, and which of them "the reference" is depends on whether numba is installed.  Round 3
// evaluated this in float64 -- a software log10 of ~200 instructions per sample: most of the feed generator's 2.2 ms per 2048 windows at the 65536-sample window.
__device__ __forceinline__ float comp_gain_curve(const float xv, const double thresh, const double ratio)
{
    const float th = (float)thresh, ra = (float)ratio;
    float xdb = 20.0f * log10f(fabsf(xv) + 1e-8f);
    xdb = fmaxf(xdb, -96.0f);
    return xdb > th ? (th + (xdb - th) / ra) - xdb : 0.0f;
}
// dB -> linear (audio.py:421): np.power(10.0, lin_A / 20) on a float32 array
__device__ __forceinline__ float comp_db_to_lin(const float g) { return exp10f(g / 20.0f); }
constexpr int COMP_CH = 8192;
// one window; g: LDS float[COMP_CH], carry: LDS float.  Called by every thread of a 256-thread workgroup.
__device__ __forceinline__ void
compressor_window(const float* __restrict__ xb, float* __restrict__ yb, const double thresh, const double ratio, const double alphaA, const double alphaR,
                  const int L, const int ysz, float* __restrict__ g, float* __restrict__ carry_p)
{
    if (threadIdx.x == 0) *carry_p = 0.f;
    for (int c0 = 0; c0 < L; c0 += COMP_CH) {
        const int n = L - c0 < COMP_CH ? L - c0 : COMP_CH;
        for (int i = threadIdx.x; i < n; i += 256) g[i] = comp_gain_curve(xb[c0 + i], thresh, ratio);
        __syncthreads();
        if (threadIdx.x == 0) {
            float prev = *carry_p;
            if (c0 == 0) g[0] = 0.f;                      // lin_A[0] = 0: the loop of the reference starts at n = 1
            // one step: lin_A[n] = float32( (1-a) g + a prev ), a = attack coefficient while the gain is falling.
            // Evaluated as g + a (prev - g) in float64 (differs from the reference's expression by < 1e-16 relative
            // before the float32 rounding); the compare runs on the float32 values beside the convert.
            auto step = [&](float gi_f) {
                const double a = gi_f < prev ? alphaA : alphaR;
                const double gi = gi_f;
                prev = (float)__builtin_fma(a, (double)prev - gi, gi);
                return prev;
            };
            // 8 samples per trip through registers: the LDS round trip leaves the dependent chain (the chain is the
            // float32->float64 convert, compare, select, multiply-add, round of each step)
            int i = 0;
            if (c0 == 0) {                                // first block: sample 0 stays 0 and does not update the state
                for (i = 1; i < 8 && i < n; ++i) g[i] = step(g[i]);
            }
            for (; i + 8 <= n; i += 8) {
                float4 u = *reinterpret_cast<const float4*>(g + i), v = *reinterpret_cast<const float4*>(g + i + 4);
                u.x = step(u.x); u.y = step(u.y); u.z = step(u.z); u.w = step(u.w);
                v.x = step(v.x); v.y = step(v.y); v.z = step(v.z); v.w = step(v.w);
                *reinterpret_cast<float4*>(g + i) = u; *reinterpret_cast<float4*>(g + i + 4) = v;
            }
            for (; i < n; ++i) g[i] = step(g[i]);
            *carry_p = prev;
        }
        __syncthreads();
        for (int i = threadIdx.x; i < n; i += 256) {
            const int j = c0 + i - (L - ysz);
            if (j >= 0) yb[j] = comp_db_to_lin(g[i]) * xb[c0 + i];
        }
        __syncthreads();
    }
}
// The same effect with the sequential stage turned sideways: ONE LANE PER WINDOW, 64 windows per wave.  In compressor_window the attack /
// release recurrence runs in one lane while the other 255 threads of the workgroup (and its LDS) wait -- 8192 dependent float64 steps, ~240 us
// per window; as part of the training feed that held every CU for a millisecond per 2048 windows.  Here the gain curve arrives precomputed
// (gc [B][L], written by the parallel generator kernel), a wave stages [64 windows][64 samples] tiles through LDS (coalesced float4 rows in,
// one row per lane out), every lane advances ITS window by 64 steps and the smoothed gain goes back in place; comp_apply_kernel then applies it to
// the last ysz samples.  Same step arithmetic as compressor_window (bit-identical results); 32 waves for 2048 windows, so the kernel runs beside the
// training step on a side stream instead of in front of it.  Requires L % 64 == 0.
__global__ void __launch_bounds__(64)
comp_smooth_kernel(float* __restrict__ gc, const float* __restrict__ kw, const float sr, const int B, const int L)
{
    __shared__ float tile[64][65];
    const int lane = threadIdx.x, b0 = blockIdx.x * 64, b = b0 + lane;
    const int bc = b < B ? b : B - 1;
    const double alphaA = exp(-log(9.0) / ((double)sr * (double)kw[4 * bc + 2]));
    const double alphaR = exp(-log(9.0) / ((double)sr * (double)kw[4 * bc + 3]));
    const int q = lane >> 4, c4 = 4 * (lane & 15);
    float prev = 0.f;
    float4 nx[16];                                         // the next tile, loaded while this one's chain runs
    auto load = [&](const int c0) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int r = 4 * i + q;
            nx[i] = *reinterpret_cast<const float4*>(gc + (size_t)(b0 + r < B ? b0 + r : B - 1) * L + c0 + c4);
        }
    };
    load(0);
    for (int c0 = 0; c0 < L; c0 += 64) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int r = 4 * i + q;
            tile[r][c4] = nx[i].x; tile[r][c4 + 1] = nx[i].y; tile[r][c4 + 2] = nx[i].z; tile[r][c4 + 3] = nx[i].w;
        }
        __syncthreads();
        load(c0 + 64 < L ? c0 + 64 : c0);
        // the lane's 64 samples through REGISTERS: reading tile[lane][k + 1] behind the store of tile[lane][k] serialises an LDS round trip
        // into every step of the dependent chain
        float gv[64];
#pragma unroll
        for (int k = 0; k < 64; ++k) gv[k] = tile[lane][k];
#pragma unroll
        for (int k = 0; k < 64; ++k) {
            if (k == 0 && c0 == 0) { gv[0] = 0.f; continue; }        // lin_A[0] = 0: the loop of the reference starts at n = 1
            const float gi_f = gv[k];
            const double a = gi_f < prev ? alphaA : alphaR;
            const double gi = gi_f;
            prev = (float)__builtin_fma(a, (double)prev - gi, gi);
            gv[k] = prev;
        }
#pragma unroll
        for (int k = 0; k < 64; ++k) tile[lane][k] = gv[k];
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 16; ++i) {                      // smoothed gain back in place, coalesced rows
            const int r = 4 * i + q;
            if (b0 + r < B) *reinterpret_cast<float4*>(gc + (size_t)(b0 + r) * L + c0 + c4) = make_float4(tile[r][c4], tile[r][c4 + 1], tile[r][c4 + 2], tile[r][c4 + 3]);
        }
        __syncthreads();
    }
}
// dB -> linear and apply (audio.py:421-425) on the last ysz samples: fully parallel (the float64 pow is ~300 instructions; inside the
// lane-per-window kernel it was 2/3 of its 1.8 ms)
__global__ void __launch_bounds__(256)
comp_apply_kernel(const float* __restrict__ x, const float* __restrict__ g, const int L, const int ysz, float* __restrict__ y)
{
    const int b = blockIdx.y, j = blockIdx.x * 256 + threadIdx.x;
    if (j >= ysz) return;
    const size_t i = (size_t)b * L + (L - ysz) + j;
    y[(size_t)b * ysz + j] = comp_db_to_lin(g[i]) * x[i];
}
__global__ void __launch_bounds__(256)
compressor_4c_kernel(const float* __restrict__ x, const float* __restrict__ knobs_wc, float sr, int L, int ysz, float* __restrict__ y)
{
    __shared__ __attribute__((aligned(16))) float g[COMP_CH];
    __shared__ float carry;
    const int b = blockIdx.x;
    const double thresh = knobs_wc[4 * b + 0], ratio = knobs_wc[4 * b + 1];
    const double alphaA = exp(-log(9.0) / ((double)sr * (double)knobs_wc[4 * b + 2]));
    const double alphaR = exp(-log(9.0) / ((double)sr * (double)knobs_wc[4 * b + 3]));
    compressor_window(x + (size_t)b * L, y + (size_t)b * ysz, thresh, ratio, alphaA, alphaR, L, ysz, g, &carry);
}


// ------------------------------------------------------------------------------ layer activations of one autoencoder (diagnostics)
// nn_proc.py:77-126 with return_acts=True (what utils/viz.py:135 plots): the ten tensors the reference appends -- ELU outputs of the four
// encoder layers, the [z ; knobs] concatenation, ELU outputs of fnn_addknobs / fnn_dec4 / fnn_dec3 / fnn_dec2 and the final output (ELU of
// fnn_dec, times the last OT input frames in 'sf' mode) -- each in the reference's [B][F][width] layout.  One thread per (window, bin) row
// walks the nine layers with plain fp32 FMAs: a diagnostic kernel (the training kernels keep these activations in registers), also an
// MFMA-free cross-check of ae_fwd_kernel.  ae: the packed parameter block of ONE autoencoder (weights row-major [out][in], then bias).
struct AeActsArgs { const float* v; const float* knobs; const float* ae; int w_off[9], b_off[9]; float* out[10]; int B, T, OT, F, K, sf; };
__global__ void __launch_bounds__(64)
ae_acts_kernel(const AeActsArgs a)
{
    const int row = blockIdx.x * 64 + threadIdx.x;
    if (row >= a.B * a.F) return;
    const int b = row / a.F, f = row - b * a.F;
    float h[64 + 16], z[64 + 16];                      // widths <= 64 (+ K <= 16 knobs)
    const int outw[9] = {64, 32, 16, 16, 16, 16, 32, 64, a.OT};
    int inw = a.T;
    // layer 1 reads the T input frames of this row straight from [B][T][F]
    for (int l = 0; l < 9; ++l) {
        const float* W = a.ae + a.w_off[l]; const float* bias = a.ae + a.b_off[l];
        if (l == 4) {                                  // fnn_addknobs: input = [z ; knobs] (nn_proc.py:92-93), itself an activation entry
            for (int k = 0; k < a.K; ++k) h[16 + k] = a.knobs[b * a.K + k];
            for (int j = 0; j < 16 + a.K; ++j) a.out[4][(size_t)row * (16 + a.K) + j] = h[j];
            inw = 16 + a.K;
        }
        for (int o = 0; o < outw[l]; ++o) {
            float s = bias[o];
            for (int i = 0; i < inw; ++i) s = __builtin_fmaf(W[o * inw + i], l == 0 ? a.v[((size_t)b * a.T + i) * a.F + f] : h[i], s);
            z[o] = s > 0.f ? s : expm1f(s);            // ELU, alpha = 1
        }
        if (l == 8 && a.sf) for (int o = 0; o < a.OT; ++o) z[o] *= a.v[((size_t)b * a.T + (a.T - a.OT + o)) * a.F + f];      // skip-filter (nn_proc.py:115)
        const int slot = l < 4 ? l : l + 1;
        for (int o = 0; o < outw[l]; ++o) { h[o] = z[o]; a.out[slot][(size_t)row * outw[l] + o] = z[o]; }
        inw = outw[l];
    }
}

// ------------------------------------------------------------------------------ generic learned-basis front end (cls_fe_dct_bases.py)
// ConvTranspose1d(C -> 1, k = KW, stride = hop) after its GEMM: overlap-add of the frames [B*T][KW] and crop
// (cls_fe_dct_bases.py:174-179); also the input gradient of Conv1d(1 -> C) (crop = its padding).
// out[b][j] = sum_t frs[b, t, j + crop - hop t]
__global__ void __launch_bounds__(256)
ola_crop_kernel(const float* __restrict__ frs, float* __restrict__ out, int T, int KW, int hop, int crop, int len)
{
    const int b = blockIdx.y, j = blockIdx.x * 256 + threadIdx.x;
    if (j >= len) return;
    const int p = j + crop;
    int t1 = p / hop; if (t1 > T - 1) t1 = T - 1;
    int t0 = p - KW + 1 <= 0 ? 0 : (p - KW + hop) / hop;       // ceil((p - KW + 1) / hop)
    float s = 0.f;
    for (int t = t0; t <= t1; ++t) s += frs[((size_t)b * T + t) * KW + (p - hop * t)];
    out[(size_t)b * len + j] = s;
}

__global__ void sum_slabs_flat_kernel(const float* __restrict__ ws, int nslab, size_t n, float* __restrict__ out)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float s = 0.f;
    for (int z = 0; z < nslab; ++z) s += ws[(size_t)z * n + i];
    out[i] = s;
}

// column sums of a row-major [R][C] matrix (Conv1d bias gradient); one thread per column, fixed order
__global__ void col_sum_kernel(const float* __restrict__ X, int R, int C, float* __restrict__ out)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float s = 0.f;
    for (int r = 0; r < R; ++r) s += X[(size_t)r * C + c];
    out[c] = s;
}

}  // namespace stm
