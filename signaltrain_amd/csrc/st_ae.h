// st_ae.h -- fused knob-conditioned autoencoders (nn_proc.py:28-126) on fp32 MFMA (gfx950).
//
// Rows of the problem are (window b, frequency bin f); the features of a row are its T STFT frames.
// A wave processes groups of 16 rows with v_mfma_f32_16x16x4_f32 in the orientation
//     D[o][row] = sum_i W[o][i] * H[i][row]          (A = weights, B = activations)
// whose result layout (lane (g,c) = (l>>4, l&15), reg r  <->  o = 16*tile + 4g + r, row = c) is exactly
// the B-operand layout of the next layer when its k-steps enumerate features in the order
// i = 16*tile + 4g + r: activations never leave registers, no transposes, no LDS traffic for them.
// All nine weight matrices of an autoencoder sit in LDS (zero padded to 16-multiples, odd row pitch),
// the A operand of each MFMA is one ds_read_b32.  The two autoencoders (magnitude / phase) of the same
// rows run as two interleaved chains in one wave (independent accumulators hide the 40-cycle dependent
// MFMA latency) and meet in the epilogue (nn_proc.py:322-326: phase residual, polar -> rect).
//
// Row space is padded per window to FP = KP/2 = roundup(F,16) "virtual bins": groups never straddle
// windows (knobs are wave-uniform) and the pad columns of the AA matrix get written as zeros.
#pragma once
#include "st_common.h"

namespace sta {

constexpr int NL = 9;
constexpr int R64 = 64, R32 = 32, R16 = 16;

// Global-memory description of one autoencoder inside the flat parameter buffer (float offsets from
// the autoencoder base: weight l at w[l], bias at b[l]); same for the gradient buffer.
struct AEOffsets { int w[NL]; int b[NL]; };

// LDS layout of one autoencoder (floats).  Layer l: Wpad[OUTp][P] (P odd), then bias[OUTp].
struct AELds {
    int w[NL], b[NL], P[NL], OUT[NL], IN[NL], OUTp[NL];
    int total;
};

__host__ __device__ inline AELds ae_lds_layout(int T, int OT, int K)
{
    AELds L;
    const int out[NL] = {R64, R32, R16, R16, R16, R16, R32, R64, OT};
    const int in[NL]  = {T, R64, R32, R16, R16 + K, R16, R16, R32, R64};
    int off = 0;
    for (int l = 0; l < NL; ++l) {
        L.OUT[l] = out[l]; L.IN[l] = in[l];
        L.OUTp[l] = (out[l] + 15) / 16 * 16;
        L.P[l] = (in[l] + 15) / 16 * 16 + 1;
        L.w[l] = off; off += L.OUTp[l] * L.P[l];
        L.b[l] = off; off += L.OUTp[l];
    }
    L.total = (off + 3) / 4 * 4;
    return L;
}

// Cooperative load of one autoencoder's parameters into LDS (zero padded): zero-fill, then a coalesced copy of
// the packed global tensors with 8 independent loads in flight per thread (a dependent load->store loop costs
// ~40 serialized L2 round trips per workgroup, i.e. tens of microseconds before the first MFMA).
__device__ inline void ae_load_lds(float* lds, const AELds& L, const float* __restrict__ ae, const AEOffsets& go,
                                   int tid, int nthreads)
{
    for (int e = tid; e < L.total; e += nthreads) lds[e] = 0.f;
    __syncthreads();
    for (int l = 0; l < NL; ++l) {
        const int P = L.P[l], IN = L.IN[l], n = L.OUT[l] * IN;
        const float* src = ae + go.w[l];
        float* dst = lds + L.w[l];
        for (int e0 = tid; e0 < n; e0 += 8 * nthreads) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { const int e = e0 + u * nthreads; v[u] = src[e < n ? e : 0]; }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int e = e0 + u * nthreads;
                if (e < n) { const int o = e / IN; dst[o * P + (e - o * IN)] = v[u]; }
            }
        }
        if (tid < L.OUT[l]) lds[L.b[l] + tid] = ae[go.b[l] + tid];
    }
}

#define ST_MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

// Hidden layer for NC interleaved chains: hout[ch][ot] = ELU(W[ch] * hin[ch] + bias[ch]).
// W[ch]: LDS pointer to the padded matrix of this layer, pitch P; lane (g,c).
template <int NC, int OTL, int ITL>
__device__ __forceinline__ void layer_fwd(const float* const (&W)[NC], const float* const (&bias)[NC], const int P,
                                          const f32x4 (&hin)[NC][ITL], f32x4 (&hout)[NC][OTL], const int g, const int c)
{
#pragma unroll
    for (int ot = 0; ot < OTL; ++ot) {
        f32x4 acc[NC];
#pragma unroll
        for (int ch = 0; ch < NC; ++ch) acc[ch] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int it = 0; it < ITL; ++it)
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int ch = 0; ch < NC; ++ch) {
                    const float a = W[ch][(16 * ot + c) * P + 16 * it + 4 * g + r];
                    acc[ch] = ST_MFMA16(a, hin[ch][it][r], acc[ch]);
                }
#pragma unroll
        for (int ch = 0; ch < NC; ++ch)
#pragma unroll
            for (int r = 0; r < 4; ++r) hout[ch][ot][r] = elu_f(acc[ch][r] + bias[ch][16 * ot + 4 * g + r]);
    }
}

// ------------------------------------------------------------------------------------------ forward
// Per-group inputs of one lane: layer-1 B operands (t = 4*ks + g, ks < 8 -> T <= 32) and the skip/residual
// tails (t = T-OT + 4g + r, OT <= 16).  Loaded in one burst and prefetched one group ahead.
struct FwdIn { float v[2][8]; float tl[2][4]; };

__device__ __forceinline__ void fwd_load(FwdIn& in, const float* __restrict__ mag, const float* __restrict__ phs,
                                         const int b, const int f, const bool fv, const int T, const int OT, const int F, const int g)
{
    const size_t base = (size_t)b * T * F + (fv ? f : 0);
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
        const int t = 4 * ks + g;
        const bool ok = fv && t < T;
        const size_t o = base + (size_t)(ok ? t : 0) * F;
        const float a = mag[o], p = phs[o];
        in.v[0][ks] = ok ? a : 0.f; in.v[1][ks] = ok ? p : 0.f;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int to = 4 * g + r;
        const bool ok = fv && to < OT;
        const size_t o = base + (size_t)(ok ? T - OT + to : 0) * F;
        const float a = mag[o], p = phs[o];
        in.tl[0][r] = ok ? a : 0.f; in.tl[1][r] = ok ? p : 0.f;
    }
}

// grid.x workgroups of NW waves; each wave walks 16-row groups: group id = b*(FP/16) + fg.
// FAST = true requires T <= 32 and OT <= 16 (register-prefetched inputs); FAST = false is the generic path.
template <int NW, bool FAST>
__global__ void __launch_bounds__(NW * 64)
ae_fwd_kernel(const float* __restrict__ mag, const float* __restrict__ phs, const float* __restrict__ knobs,
              const float* __restrict__ ae_m, const float* __restrict__ ae_p, const AEOffsets go,
              float* __restrict__ mag_hat, float* __restrict__ phs_hat, float* __restrict__ AA,
              float* __restrict__ reg_partial,
              const int B, const int T, const int OT, const int F, const int K, const int KP, const float expfac)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const AELds L = ae_lds_layout(T, OT, K);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, c = lane & 15;
    float* lw[2] = {lds, lds + L.total};
    ae_load_lds(lw[0], L, ae_m, go, tid, NW * 64);
    ae_load_lds(lw[1], L, ae_p, go, tid, NW * 64);
    __syncthreads();

    const int FP = KP / 2, gpw = FP / 16;              // groups per window
    const int ngroups = B * gpw;
    const int KS1 = (T + 3) / 4;
    const int KQ = (K + 3) / 4;
    const int OT9 = (OT + 15) / 16;
    const int gstride = gridDim.x * NW;
    float reg = 0.f;

    FwdIn cur;
    int grp = blockIdx.x * NW + wave;
    if (FAST && grp < ngroups) {
        const int b = grp / gpw, f = (grp - b * gpw) * 16 + c;
        fwd_load(cur, mag, phs, b, f, f < F, T, OT, F, g);
    }
    for (; grp < ngroups; grp += gstride) {
        asm volatile("" ::: "memory");      // keep the (loop-invariant) LDS weight fetches inside the loop: hoisting them spills
        const int b = grp / gpw, f = (grp - b * gpw) * 16 + c;
        const bool fv = f < F;
        const float* src[2] = {mag + (size_t)b * T * F + f, phs + (size_t)b * T * F + f};
        FwdIn nxt;
        if (FAST) {
            const int gn = grp + gstride;
            if (gn < ngroups) {
                const int bn = gn / gpw, fn = (gn - bn * gpw) * 16 + c;
                fwd_load(nxt, mag, phs, bn, fn, fn < F, T, OT, F, g);
            }
        }

        // ---- layer 1 (IN = T; B operand: t = 4*ks + g)
        f32x4 h1[2][4];
        {
            f32x4 acc[2][4];
#pragma unroll
            for (int ch = 0; ch < 2; ++ch)
#pragma unroll
                for (int ot = 0; ot < 4; ++ot) acc[ch][ot] = (f32x4){0.f, 0.f, 0.f, 0.f};
            const int P1 = L.P[0];
            if (FAST) {
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) {
                    if (ks < KS1) {
                        const int t = 4 * ks + g;
#pragma unroll
                        for (int ot = 0; ot < 4; ++ot)
#pragma unroll
                            for (int ch = 0; ch < 2; ++ch)
                                acc[ch][ot] = ST_MFMA16(lw[ch][L.w[0] + (16 * ot + c) * P1 + t], cur.v[ch][ks], acc[ch][ot]);
                    }
                }
            } else {
                for (int ks = 0; ks < KS1; ++ks) {
                    const int t = 4 * ks + g;
                    const bool ok = fv && t < T;
                    float v[2];
                    v[0] = ok ? src[0][(size_t)t * F] : 0.f;
                    v[1] = ok ? src[1][(size_t)t * F] : 0.f;
#pragma unroll
                    for (int ot = 0; ot < 4; ++ot)
#pragma unroll
                        for (int ch = 0; ch < 2; ++ch)
                            acc[ch][ot] = ST_MFMA16(lw[ch][L.w[0] + (16 * ot + c) * P1 + t], v[ch], acc[ch][ot]);
                }
            }
#pragma unroll
            for (int ch = 0; ch < 2; ++ch)
#pragma unroll
                for (int ot = 0; ot < 4; ++ot)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        h1[ch][ot][r] = elu_f(acc[ch][ot][r] + lw[ch][L.b[0] + 16 * ot + 4 * g + r]);
        }
        // ---- layers 2..4
        f32x4 h2[2][2], h3[2][1], h4[2][1];
        {
            const float* const W[2] = {lw[0] + L.w[1], lw[1] + L.w[1]}; const float* const bb[2] = {lw[0] + L.b[1], lw[1] + L.b[1]};
            layer_fwd<2, 2, 4>(W, bb, L.P[1], h1, h2, g, c);
        }
        {
            const float* const W[2] = {lw[0] + L.w[2], lw[1] + L.w[2]}; const float* const bb[2] = {lw[0] + L.b[2], lw[1] + L.b[2]};
            layer_fwd<2, 1, 2>(W, bb, L.P[2], h2, h3, g, c);
        }
        {
            const float* const W[2] = {lw[0] + L.w[3], lw[1] + L.w[3]}; const float* const bb[2] = {lw[0] + L.b[3], lw[1] + L.b[3]};
            layer_fwd<2, 1, 1>(W, bb, L.P[3], h3, h4, g, c);
        }
        // ---- layer 5: [h4 ; knobs] (nn_proc.py:92-96); knob features 16 + 4q + g
        f32x4 h5[2][1];
        {
            f32x4 acc[2];
            const int P5 = L.P[4];
#pragma unroll
            for (int ch = 0; ch < 2; ++ch) acc[ch] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int ch = 0; ch < 2; ++ch)
                    acc[ch] = ST_MFMA16(lw[ch][L.w[4] + c * P5 + 4 * g + r], h4[ch][0][r], acc[ch]);
            for (int q = 0; q < KQ; ++q) {
                const int kn = 4 * q + g;
                const float kv = kn < K ? knobs[(size_t)b * K + kn] : 0.f;
#pragma unroll
                for (int ch = 0; ch < 2; ++ch)
                    acc[ch] = ST_MFMA16(lw[ch][L.w[4] + c * P5 + 16 + kn], kv, acc[ch]);
            }
#pragma unroll
            for (int ch = 0; ch < 2; ++ch)
#pragma unroll
                for (int r = 0; r < 4; ++r) h5[ch][0][r] = elu_f(acc[ch][r] + lw[ch][L.b[4] + 4 * g + r]);
        }
        // ---- layers 6..8
        f32x4 h6[2][1], h7[2][2], h8[2][4];
        {
            const float* const W[2] = {lw[0] + L.w[5], lw[1] + L.w[5]}; const float* const bb[2] = {lw[0] + L.b[5], lw[1] + L.b[5]};
            layer_fwd<2, 1, 1>(W, bb, L.P[5], h5, h6, g, c);
        }
        {
            const float* const W[2] = {lw[0] + L.w[6], lw[1] + L.w[6]}; const float* const bb[2] = {lw[0] + L.b[6], lw[1] + L.b[6]};
            layer_fwd<2, 2, 1>(W, bb, L.P[6], h6, h7, g, c);
        }
        {
            const float* const W[2] = {lw[0] + L.w[7], lw[1] + L.w[7]}; const float* const bb[2] = {lw[0] + L.b[7], lw[1] + L.b[7]};
            layer_fwd<2, 4, 2>(W, bb, L.P[7], h7, h8, g, c);
        }
        // ---- layer 9 (OUT = OT, runtime output tiles) + epilogue (nn_proc.py:115,117,322-326)
        const float wf = fv ? expf(expfac * (float)f) : 0.f;     // train.py:115-117 frequency weight
        for (int o9 = 0; o9 < OT9; ++o9) {
            f32x4 e9[2][1];
            const float* const W[2] = {lw[0] + L.w[8] + 16 * o9 * L.P[8], lw[1] + L.w[8] + 16 * o9 * L.P[8]};
            const float* const bb[2] = {lw[0] + L.b[8] + 16 * o9, lw[1] + L.b[8] + 16 * o9};
            layer_fwd<2, 1, 4>(W, bb, L.P[8], h8, e9, g, c);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int to = 16 * o9 + 4 * g + r;
                if (to < OT) {
                    const size_t ro = (size_t)b * OT + to;
                    float mh = 0.f, ph = 0.f, sn = 0.f, cs = 1.f;
                    if (fv) {
                        const size_t ti = (size_t)(T - OT + to) * F;
                        const float mt = FAST ? cur.tl[0][r] : src[0][ti];
                        const float pt = FAST ? cur.tl[1][r] : src[1][ti];
                        mh = e9[0][0][r] * mt;                         // 'sf' skip-filter
                        ph = e9[1][0][r] + pt;                         // phase residual
                        sincosf(ph, &sn, &cs);
                        mag_hat[ro * F + f] = mh;
                        phs_hat[ro * F + f] = ph;
                        reg += fabsf(mh * wf);
                    }
                    AA[ro * KP + f] = mh * cs;                          // f < FP always: pads get zeros
                    AA[ro * KP + FP + f] = mh * sn;
                }
            }
        }
        if (FAST) cur = nxt;
    }
    if (reg_partial) {
        reg = wave_sum(reg);
        if (lane == 0) reg_partial[blockIdx.x * NW + wave] = reg;
    }
}

// ========================================================================================== backward
// One workgroup = NW waves, 1 wave per SIMD (the wave owns up to 512 registers).  blockIdx.y selects the
// autoencoder (0 = magnitude 'sf', 1 = phase).  Each wave walks PAIRS of 16-row groups (two interleaved
// chains).  Per pair it
//   1. recomputes the forward chain keeping every post-ELU activation in registers (68 regs / chain);
//   2. forms d out (polar->rect backward of nn_proc.py:322-326 + the L1 term of loss_functions.py:36);
//   3. walks the layers backwards: da_l (D layout) is the B operand of the next dgrad MFMA as is; for the
//      weight gradient both operands need rows on the k index, i.e. the transposed layout, obtained by a
//      wave-private LDS round trip (write [feat][row] pitch 20, read A/B fragments); the 16x16 tiles of
//      dW_l = sum_rows da_l (x) h_{l-1} accumulate in 144 persistent registers for the whole kernel;
//   4. writes d input rows (dmag / dphs).
// At the end each wave stores its partial dW/db (packed like the parameters); ae_grad_reduce_kernel sums.
constexpr int SP = 20;                 // scratch pitch (floats): 16 rows + 4, keeps rows 16-B aligned

template <int OTL, int ITL>
struct Tiles { f32x4 t[OTL][ITL]; };

template <int NC, int OTL, int ITL>
__device__ __forceinline__ void wgrad_mfma(const float* const (&Ysc)[NC], const float* const (&Xsc)[NC],
                                           f32x4 (&dW)[OTL][ITL], const int g, const int c)
{
#pragma unroll
    for (int ch = 0; ch < NC; ++ch)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            float a[OTL], b[ITL];
#pragma unroll
            for (int ot = 0; ot < OTL; ++ot) a[ot] = Ysc[ch][(16 * ot + c) * SP + 4 * ks + g];
#pragma unroll
            for (int it = 0; it < ITL; ++it) b[it] = Xsc[ch][(16 * it + c) * SP + 4 * ks + g];
#pragma unroll
            for (int ot = 0; ot < OTL; ++ot)
#pragma unroll
                for (int it = 0; it < ITL; ++it) dW[ot][it] = ST_MFMA16(a[ot], b[it], dW[ot][it]);
        }
}

template <int TL>
__device__ __forceinline__ void scratch_put(float* sc, const f32x4 (&x)[TL], const int g, const int c)
{
#pragma unroll
    for (int t = 0; t < TL; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) sc[(16 * t + 4 * g + r) * SP + c] = x[t][r];
}

// sum over the 16 rows of scratch row `o` (bias gradient), lanes o < n
template <int NC>
__device__ __forceinline__ float scratch_rowsum(const float* const (&Ysc)[NC], const int o, const int n)
{
    float s = 0.f;
    if (o < n) {
#pragma unroll
        for (int ch = 0; ch < NC; ++ch) {
            const float4* q = reinterpret_cast<const float4*>(Ysc[ch] + o * SP);
#pragma unroll
            for (int j = 0; j < 4; ++j) { const float4 v = q[j]; s += (v.x + v.y) + (v.z + v.w); }
        }
    }
    return s;
}

// dh_in[it] = sum_o W[o][16 it + c] * da[o]  (A = W^T fragments read from the row-major padded matrix)
template <int NC, int OTL, int ITL>
__device__ __forceinline__ void layer_dgrad(const float* W, const int P, const f32x4 (&da)[NC][OTL],
                                            f32x4 (&dh)[NC][ITL], const int g, const int c)
{
#pragma unroll
    for (int it = 0; it < ITL; ++it) {
        f32x4 acc[NC];
#pragma unroll
        for (int ch = 0; ch < NC; ++ch) acc[ch] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ot = 0; ot < OTL; ++ot)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float a = W[(16 * ot + 4 * g + r) * P + 16 * it + c];
#pragma unroll
                for (int ch = 0; ch < NC; ++ch) acc[ch] = ST_MFMA16(a, da[ch][ot][r], acc[ch]);
            }
#pragma unroll
        for (int ch = 0; ch < NC; ++ch) dh[ch][it] = acc[ch];
    }
}

template <int NC, int TL>
__device__ __forceinline__ void apply_elu_grad(f32x4 (&d)[NC][TL], const f32x4 (&h)[NC][TL])
{
#pragma unroll
    for (int ch = 0; ch < NC; ++ch)
#pragma unroll
        for (int t = 0; t < TL; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) d[ch][t][r] *= elu_grad_from_out(h[ch][t][r]);
}

// store a D-layout accumulator tile set of one layer's dW into the packed partial buffer
template <int OTL, int ITL>
__device__ __forceinline__ void store_dw(float* base, const int woff, const int OUT, const int IN,
                                         const f32x4 (&dW)[OTL][ITL], const int g, const int c)
{
#pragma unroll
    for (int ot = 0; ot < OTL; ++ot)
#pragma unroll
        for (int it = 0; it < ITL; ++it)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int o = 16 * ot + 4 * g + r, i = 16 * it + c;
                if (o < OUT && i < IN) base[woff + o * IN + i] = dW[ot][it][r];
            }
}

template <int OTL, int ITL>
__device__ __forceinline__ void zero_tiles(f32x4 (&x)[OTL][ITL])
{
#pragma unroll
    for (int a = 0; a < OTL; ++a)
#pragma unroll
        for (int b = 0; b < ITL; ++b) x[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
}

// Supported geometry of this instantiation: T <= 32 (T16 = 2 input tiles), OT <= 16, K <= 16.
template <int NW>
__global__ void __launch_bounds__(NW * 64, 1)
ae_bwd_kernel(const float* __restrict__ mag, const float* __restrict__ phs, const float* __restrict__ knobs,
              const float* __restrict__ ae_m, const float* __restrict__ ae_p, const AEOffsets go, const int PG,
              const float* __restrict__ mag_hat, const float* __restrict__ phs_hat, const float* __restrict__ dAA,
              const float* __restrict__ g_mag_hat, const float reg_coef, const float expfac,
              float* __restrict__ dmag, float* __restrict__ dphs, float* __restrict__ ws,
              const int B, const int T, const int OT, const int F, const int K, const int KP,
              const int to_lo, const int to_hi)      // live synthesis frames: dAA rows outside are treated as zero
{
    constexpr int NC = 2;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int ae = blockIdx.y;
    const AELds L = ae_lds_layout(T, OT, K);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, c = lane & 15;
    float* lw = lds;
    ae_load_lds(lw, L, ae ? ae_p : ae_m, go, tid, NW * 64);
    // wave-private scratch: per chain  V[32*SP] | X[64*SP] | Y[64*SP] | TAIL[16*SP]
    constexpr int SCR = (32 + 64 + 64 + 16) * SP;
    float* scr = lds + L.total + wave * (NC * SCR);
    float* Vs[NC]; float* Xs[NC]; float* Ys[NC]; float* Ts[NC];
#pragma unroll
    for (int ch = 0; ch < NC; ++ch) {
        Vs[ch] = scr + ch * SCR; Xs[ch] = Vs[ch] + 32 * SP; Ys[ch] = Xs[ch] + 64 * SP; Ts[ch] = Ys[ch] + 64 * SP;
    }
    __syncthreads();

    const float* vin = ae ? phs : mag;
    float* dvout = ae ? dphs : dmag;
    const int FP = KP / 2, gpw = FP / 16;
    const int ngroups = B * gpw, npairs = (ngroups + 1) / 2;
    const int KS1 = (T + 3) / 4;

    // persistent weight-gradient accumulators (144 regs) + 9 bias-gradient registers
    f32x4 dW1[4][2], dW2[2][4], dW3[1][2], dW4[1][1], dW5[1][2], dW6[1][1], dW7[2][1], dW8[4][2], dW9[1][4];
    zero_tiles(dW1); zero_tiles(dW2); zero_tiles(dW3); zero_tiles(dW4); zero_tiles(dW5);
    zero_tiles(dW6); zero_tiles(dW7); zero_tiles(dW8); zero_tiles(dW9);
    float db[NL];
#pragma unroll
    for (int l = 0; l < NL; ++l) db[l] = 0.f;

    const float* const Wl[NL] = {lw + L.w[0], lw + L.w[1], lw + L.w[2], lw + L.w[3], lw + L.w[4],
                                 lw + L.w[5], lw + L.w[6], lw + L.w[7], lw + L.w[8]};
    const float* const Bl[NL] = {lw + L.b[0], lw + L.b[1], lw + L.b[2], lw + L.b[3], lw + L.b[4],
                                 lw + L.b[5], lw + L.b[6], lw + L.b[7], lw + L.b[8]};

    for (int pr = blockIdx.x * NW + wave; pr < npairs; pr += gridDim.x * NW) {
        asm volatile("" ::: "memory");      // see ae_fwd_kernel
        int bb[NC], ff[NC]; bool fv[NC];
#pragma unroll
        for (int ch = 0; ch < NC; ++ch) {
            const int grp = 2 * pr + ch;
            const bool gv = grp < ngroups;
            const int gg = gv ? grp : ngroups - 1;
            bb[ch] = gg / gpw; ff[ch] = (gg - bb[ch] * gpw) * 16 + c;
            fv[ch] = gv && ff[ch] < F;
        }
        // ------------------------------------------------------------------ forward recompute
        f32x4 h1[NC][4], h2[NC][2], h3[NC][1], h4[NC][1], h5[NC][1], h6[NC][1], h7[NC][2], h8[NC][4], e9[NC][1];
        f32x4 kn[NC][1];                                  // knob "tile": feature 16 + 4g + r of layer 5's input
        {
            f32x4 acc[NC][4];
#pragma unroll
            for (int ch = 0; ch < NC; ++ch)
#pragma unroll
                for (int ot = 0; ot < 4; ++ot) acc[ch][ot] = (f32x4){0.f, 0.f, 0.f, 0.f};
            const int P1 = L.P[0];
            float vr[NC][8];                              // burst-load all inputs first (8 k-steps = 32 padded features)
#pragma unroll
            for (int ch = 0; ch < NC; ++ch)
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) {
                    const int t = 4 * ks + g;
                    const bool ok = fv[ch] && t < T;
                    const float x = vin[((size_t)bb[ch] * T + (ok ? t : 0)) * F + (fv[ch] ? ff[ch] : 0)];
                    vr[ch][ks] = ok ? x : 0.f;
                }
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                const int t = 4 * ks + g;
#pragma unroll
                for (int ch = 0; ch < NC; ++ch) Vs[ch][t * SP + c] = vr[ch][ks];   // [feat = t][row = c] for the layer-1 weight gradient
                if (ks < KS1) {
#pragma unroll
                    for (int ot = 0; ot < 4; ++ot)
#pragma unroll
                        for (int ch = 0; ch < NC; ++ch)
                            acc[ch][ot] = ST_MFMA16(Wl[0][(16 * ot + c) * P1 + t], vr[ch][ks], acc[ch][ot]);
                }
            }
#pragma unroll
            for (int ch = 0; ch < NC; ++ch)
#pragma unroll
                for (int ot = 0; ot < 4; ++ot)
#pragma unroll
                    for (int r = 0; r < 4; ++r) h1[ch][ot][r] = elu_f(acc[ch][ot][r] + Bl[0][16 * ot + 4 * g + r]);
        }
        {
            const float* const W[NC] = {Wl[1], Wl[1]}; const float* const bs[NC] = {Bl[1], Bl[1]};
            layer_fwd<NC, 2, 4>(W, bs, L.P[1], h1, h2, g, c);
        }
        {
            const float* const W[NC] = {Wl[2], Wl[2]}; const float* const bs[NC] = {Bl[2], Bl[2]};
            layer_fwd<NC, 1, 2>(W, bs, L.P[2], h2, h3, g, c);
        }
        {
            const float* const W[NC] = {Wl[3], Wl[3]}; const float* const bs[NC] = {Bl[3], Bl[3]};
            layer_fwd<NC, 1, 1>(W, bs, L.P[3], h3, h4, g, c);
        }
        // d-out inputs (needed after the forward recompute): issue the loads now so they land under layers 5..9
        float q_gre[NC][4], q_gim[NC][4], q_ph[NC][4], q_mh[NC][4], q_mt[NC][4];
#pragma unroll
        for (int ch = 0; ch < NC; ++ch)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int to = 4 * g + r;
                const bool ok = fv[ch] && to < OT;
                const bool lv = ok && to >= to_lo && to <= to_hi;
                const size_t ro = (size_t)bb[ch] * OT + (ok ? to : 0);
                const int fq = fv[ch] ? ff[ch] : 0;
                const float a0 = dAA[(lv ? ro : 0) * KP + fq], a1 = dAA[(lv ? ro : 0) * KP + FP + fq];
                const float a2 = phs_hat[ro * F + fq], a3 = mag_hat[ro * F + fq];
                const float a4 = vin[((size_t)bb[ch] * T + (ok ? T - OT + to : 0)) * F + fq];
                q_gre[ch][r] = lv ? a0 : 0.f; q_gim[ch][r] = lv ? a1 : 0.f;
                q_ph[ch][r] = a2; q_mh[ch][r] = a3; q_mt[ch][r] = a4;
            }
        {
            f32x4 acc[NC];
            const int P5 = L.P[4];
#pragma unroll
            for (int ch = 0; ch < NC; ++ch) {
                acc[ch] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int kidx = 4 * g + r;
                    kn[ch][0][r] = kidx < K ? knobs[(size_t)bb[ch] * K + kidx] : 0.f;
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int ch = 0; ch < NC; ++ch)
                    acc[ch] = ST_MFMA16(Wl[4][c * P5 + 4 * g + r], h4[ch][0][r], acc[ch]);
#pragma unroll
            for (int r = 0; r < 4; ++r)                    // knob features 16 + 4g + r (K <= 16)
#pragma unroll
                for (int ch = 0; ch < NC; ++ch)
                    acc[ch] = ST_MFMA16(Wl[4][c * P5 + 16 + 4 * g + r], kn[ch][0][r], acc[ch]);
#pragma unroll
            for (int ch = 0; ch < NC; ++ch)
#pragma unroll
                for (int r = 0; r < 4; ++r) h5[ch][0][r] = elu_f(acc[ch][r] + Bl[4][4 * g + r]);
        }
        {
            const float* const W[NC] = {Wl[5], Wl[5]}; const float* const bs[NC] = {Bl[5], Bl[5]};
            layer_fwd<NC, 1, 1>(W, bs, L.P[5], h5, h6, g, c);
        }
        {
            const float* const W[NC] = {Wl[6], Wl[6]}; const float* const bs[NC] = {Bl[6], Bl[6]};
            layer_fwd<NC, 2, 1>(W, bs, L.P[6], h6, h7, g, c);
        }
        {
            const float* const W[NC] = {Wl[7], Wl[7]}; const float* const bs[NC] = {Bl[7], Bl[7]};
            layer_fwd<NC, 4, 2>(W, bs, L.P[7], h7, h8, g, c);
        }
        {
            const float* const W[NC] = {Wl[8], Wl[8]}; const float* const bs[NC] = {Bl[8], Bl[8]};
            layer_fwd<NC, 1, 4>(W, bs, L.P[8], h8, e9, g, c);
        }
        // ------------------------------------------------------------------ d out  (D layout: t' = 4g + r)
        f32x4 da9[NC][1];
#pragma unroll
        for (int ch = 0; ch < NC; ++ch) {
            const float wf = fv[ch] ? expf(expfac * (float)ff[ch]) : 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int to = 4 * g + r;
                float d9 = 0.f, tail = 0.f;
                if (fv[ch] && to < OT) {
                    const size_t ro = (size_t)bb[ch] * OT + to;
                    const float gre = q_gre[ch][r], gim = q_gim[ch][r];
                    const float ph = q_ph[ch][r], mh = q_mh[ch][r];
                    float sn, cs; sincosf(ph, &sn, &cs);
                    if (ae == 0) {
                        const float sg = mh > 0.f ? 1.f : (mh < 0.f ? -1.f : 0.f);
                        float dmh = gre * cs + gim * sn + reg_coef * sg * wf;
                        if (g_mag_hat) dmh += g_mag_hat[ro * F + ff[ch]];   // generic upstream gradient (autograd path)
                        const float mt = q_mt[ch][r];
                        d9 = dmh * mt * elu_grad_from_out(e9[ch][0][r]);
                        tail = dmh * e9[ch][0][r];
                    } else {
                        const float dph = mh * (gim * cs - gre * sn);
                        d9 = dph * elu_grad_from_out(e9[ch][0][r]);
                        tail = dph;
                    }
                }
                da9[ch][0][r] = d9;
                Ts[ch][to * SP + c] = tail;
            }
        }
        // ------------------------------------------------------------------ backward through the layers
        const float* const Yc[NC] = {Ys[0], Ys[1]};
        const float* const Xc[NC] = {Xs[0], Xs[1]};
        const float* const Vc[NC] = {Vs[0], Vs[1]};
        // layer 9 (64 -> OT)
        f32x4 da8[NC][4];
#pragma unroll
        for (int ch = 0; ch < NC; ++ch) { scratch_put<1>(Ys[ch], da9[ch], g, c); scratch_put<4>(Xs[ch], h8[ch], g, c); }
        wgrad_mfma<NC, 1, 4>(Yc, Xc, dW9, g, c);
        db[8] += scratch_rowsum<NC>(Yc, lane, 16);
        layer_dgrad<NC, 1, 4>(Wl[8], L.P[8], da9, da8, g, c);
        apply_elu_grad<NC, 4>(da8, h8);
        // layer 8 (32 -> 64)
        f32x4 da7[NC][2];
#pragma unroll
        for (int ch = 0; ch < NC; ++ch) { scratch_put<4>(Ys[ch], da8[ch], g, c); scratch_put<2>(Xs[ch], h7[ch], g, c); }
        wgrad_mfma<NC, 4, 2>(Yc, Xc, dW8, g, c);
        db[7] += scratch_rowsum<NC>(Yc, lane, 64);
        layer_dgrad<NC, 4, 2>(Wl[7], L.P[7], da8, da7, g, c);
        apply_elu_grad<NC, 2>(da7, h7);
        // layer 7 (16 -> 32)
        f32x4 da6[NC][1];
#pragma unroll
        for (int ch = 0; ch < NC; ++ch) { scratch_put<2>(Ys[ch], da7[ch], g, c); scratch_put<1>(Xs[ch], h6[ch], g, c); }
        wgrad_mfma<NC, 2, 1>(Yc, Xc, dW7, g, c);
        db[6] += scratch_rowsum<NC>(Yc, lane, 32);
        layer_dgrad<NC, 2, 1>(Wl[6], L.P[6], da7, da6, g, c);
        apply_elu_grad<NC, 1>(da6, h6);
        // layer 6 (16 -> 16)
        f32x4 da5[NC][1];
#pragma unroll
        for (int ch = 0; ch < NC; ++ch) { scratch_put<1>(Ys[ch], da6[ch], g, c); scratch_put<1>(Xs[ch], h5[ch], g, c); }
        wgrad_mfma<NC, 1, 1>(Yc, Xc, dW6, g, c);
        db[5] += scratch_rowsum<NC>(Yc, lane, 16);
        layer_dgrad<NC, 1, 1>(Wl[5], L.P[5], da6, da5, g, c);
        apply_elu_grad<NC, 1>(da5, h5);
        // layer 5 ([h4 ; knobs] -> 16): weight gradient over both input tiles, data gradient to h4 only
        f32x4 da4[NC][1];
#pragma unroll
        for (int ch = 0; ch < NC; ++ch) {
            scratch_put<1>(Ys[ch], da5[ch], g, c);
            scratch_put<1>(Xs[ch], h4[ch], g, c);
            scratch_put<1>(Xs[ch] + 16 * SP, kn[ch], g, c);
        }
        wgrad_mfma<NC, 1, 2>(Yc, Xc, dW5, g, c);
        db[4] += scratch_rowsum<NC>(Yc, lane, 16);
        layer_dgrad<NC, 1, 1>(Wl[4], L.P[4], da5, da4, g, c);
        apply_elu_grad<NC, 1>(da4, h4);
        // layer 4 (16 -> 16)
        f32x4 da3[NC][1];
#pragma unroll
        for (int ch = 0; ch < NC; ++ch) { scratch_put<1>(Ys[ch], da4[ch], g, c); scratch_put<1>(Xs[ch], h3[ch], g, c); }
        wgrad_mfma<NC, 1, 1>(Yc, Xc, dW4, g, c);
        db[3] += scratch_rowsum<NC>(Yc, lane, 16);
        layer_dgrad<NC, 1, 1>(Wl[3], L.P[3], da4, da3, g, c);
        apply_elu_grad<NC, 1>(da3, h3);
        // layer 3 (32 -> 16)
        f32x4 da2[NC][2];
#pragma unroll
        for (int ch = 0; ch < NC; ++ch) { scratch_put<1>(Ys[ch], da3[ch], g, c); scratch_put<2>(Xs[ch], h2[ch], g, c); }
        wgrad_mfma<NC, 1, 2>(Yc, Xc, dW3, g, c);
        db[2] += scratch_rowsum<NC>(Yc, lane, 16);
        layer_dgrad<NC, 1, 2>(Wl[2], L.P[2], da3, da2, g, c);
        apply_elu_grad<NC, 2>(da2, h2);
        // layer 2 (64 -> 32)
        f32x4 da1[NC][4];
#pragma unroll
        for (int ch = 0; ch < NC; ++ch) { scratch_put<2>(Ys[ch], da2[ch], g, c); scratch_put<4>(Xs[ch], h1[ch], g, c); }
        wgrad_mfma<NC, 2, 4>(Yc, Xc, dW2, g, c);
        db[1] += scratch_rowsum<NC>(Yc, lane, 32);
        layer_dgrad<NC, 2, 4>(Wl[1], L.P[1], da2, da1, g, c);
        apply_elu_grad<NC, 4>(da1, h1);
        // layer 1 (T -> 64): input rows were staged in Vs during the forward pass
        f32x4 dv[NC][2];
#pragma unroll
        for (int ch = 0; ch < NC; ++ch) scratch_put<4>(Ys[ch], da1[ch], g, c);
        wgrad_mfma<NC, 4, 2>(Yc, Vc, dW1, g, c);
        db[0] += scratch_rowsum<NC>(Yc, lane, 64);
        layer_dgrad<NC, 4, 2>(Wl[0], L.P[0], da1, dv, g, c);
        // ------------------------------------------------------------------ d input rows (+ skip / residual tails)
#pragma unroll
        for (int ch = 0; ch < NC; ++ch)
#pragma unroll
            for (int it = 0; it < 2; ++it)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int t = 16 * it + 4 * g + r;
                    if (fv[ch] && t < T) {
                        float v = dv[ch][it][r];
                        if (t >= T - OT) v += Ts[ch][(t - (T - OT)) * SP + c];
                        dvout[((size_t)bb[ch] * T + t) * F + ff[ch]] = v;
                    }
                }
    }
    // ---------------------------------------------------------------------- per-wave partial gradients
    float* base = ws + ((size_t)(blockIdx.x * NW + wave) * 2 + ae) * PG;
    for (int i = lane; i < PG; i += 64) base[i] = 0.f;        // alignment pads and never-touched entries
    __builtin_amdgcn_s_waitcnt(0);                             // (stores below overwrite; same lane order not guaranteed across lanes)
    __builtin_amdgcn_wave_barrier();
    store_dw(base, go.w[0], L.OUT[0], L.IN[0], dW1, g, c);
    store_dw(base, go.w[1], L.OUT[1], L.IN[1], dW2, g, c);
    store_dw(base, go.w[2], L.OUT[2], L.IN[2], dW3, g, c);
    store_dw(base, go.w[3], L.OUT[3], L.IN[3], dW4, g, c);
    store_dw(base, go.w[4], L.OUT[4], L.IN[4], dW5, g, c);
    store_dw(base, go.w[5], L.OUT[5], L.IN[5], dW6, g, c);
    store_dw(base, go.w[6], L.OUT[6], L.IN[6], dW7, g, c);
    store_dw(base, go.w[7], L.OUT[7], L.IN[7], dW8, g, c);
    store_dw(base, go.w[8], L.OUT[8], L.IN[8], dW9, g, c);
#pragma unroll
    for (int l = 0; l < NL; ++l)
        if (lane < L.OUT[l]) base[go.b[l] + lane] = db[l];
}

}  // namespace sta
