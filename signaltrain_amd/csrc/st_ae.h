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
                                   int tid, int nthreads, int l0 = 0, int l1 = NL)     // layers [l0, l1): the wide path keeps 1..7 only
{
    for (int e = tid; e < L.total; e += nthreads) lds[e] = 0.f;
    __syncthreads();
    for (int l = l0; l < l1; ++l) {
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
        for (int ch = 0; ch < NC; ++ch)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[ch][r] = bias[ch][16 * ot + 4 * g + r];     // accumulator starts at the bias
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
            for (int r = 0; r < 4; ++r) hout[ch][ot][r] = elu_f(acc[ch][r]);
    }
}

// ------------------------------------------------------------------------------------------ forward
// Per-group inputs of one lane: layer-1 B operands (t = 4*ks + g, ks < 8 -> T <= 32) and the skip/residual
// tails (t = T-OT + 4g + r, OT <= 16).  Loaded in one burst and prefetched one group ahead.
struct FwdIn { float v[2][8]; float tl[2][4]; float kn[4]; };   // kn: knob 4q + g for the (up to 4) knob k-steps of layer 5

// RAW loads from clamped (always valid) addresses; fwd_mask() zeroes the padding rows / bins / knobs afterwards.  The
// selects must not sit right behind the loads, and the prefetch must not sit in a conditional block: either makes the
// compiler wait for the loads on the spot, turning the one-group-ahead prefetch into a synchronous load.
__device__ __forceinline__ void fwd_load(FwdIn& in, const float* __restrict__ mag, const float* __restrict__ phs,
                                         const float* __restrict__ knobs, const int K,
                                         const int b, const int f, const bool fv, const int T, const int OT, const int F, const int g)
{
#pragma unroll
    for (int q = 0; q < 4; ++q) { const int kn = 4 * q + g; in.kn[q] = knobs[(unsigned)b * K + (kn < K ? kn : 0)]; }
    const unsigned base = (unsigned)b * T * F + (fv ? f : 0);
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
        const int t = 4 * ks + g;
        const unsigned o = base + (unsigned)(t < T ? t : 0) * F;
        in.v[0][ks] = mag[o]; in.v[1][ks] = phs[o];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int to = 4 * g + r;
        const unsigned o = base + (unsigned)(to < OT ? T - OT + to : 0) * F;
        in.tl[0][r] = mag[o]; in.tl[1][r] = phs[o];
    }
}
__device__ __forceinline__ void fwd_mask(FwdIn& in, const int K, const bool fv, const int T, const int OT, const int g)
{
#pragma unroll
    for (int q = 0; q < 4; ++q) in.kn[q] = (4 * q + g) < K ? in.kn[q] : 0.f;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) { const bool ok = fv && 4 * ks + g < T; in.v[0][ks] = ok ? in.v[0][ks] : 0.f; in.v[1][ks] = ok ? in.v[1][ks] : 0.f; }
#pragma unroll
    for (int r = 0; r < 4; ++r) { const bool ok = fv && 4 * g + r < OT; in.tl[0][r] = ok ? in.tl[0][r] : 0.f; in.tl[1][r] = ok ? in.tl[1][r] : 0.f; }
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
    // block-fastest group numbering: the partial last round (ngroups is rarely a multiple of the wave count) then puts ONE
    // extra group on every workgroup instead of a full extra round on the first few workgroups while the rest idle
    int grp = wave * gridDim.x + blockIdx.x;
    if (FAST && grp < ngroups) {
        const int b = grp / gpw, f = (grp - b * gpw) * 16 + c;
        fwd_load(cur, mag, phs, knobs, K, b, f, f < F, T, OT, F, g);
        fwd_mask(cur, K, f < F, T, OT, g);
    }
    for (; grp < ngroups; grp += gstride) {
        asm volatile("" ::: "memory");      // keep the (loop-invariant) LDS weight fetches inside the loop: hoisting them spills
        const int b = grp / gpw, f = (grp - b * gpw) * 16 + c;
        const bool fv = f < F;
        const float* src[2] = {mag + (size_t)b * T * F + f, phs + (size_t)b * T * F + f};
        FwdIn nxt;
        const int gn = grp + gstride < ngroups ? grp + gstride : grp;      // last iteration: harmless reload of this group
        const int bn = gn / gpw, fn = (gn - bn * gpw) * 16 + c;
        if (FAST) fwd_load(nxt, mag, phs, knobs, K, bn, fn, fn < F, T, OT, F, g);

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
                const float kv = (FAST && q < 4) ? cur.kn[q] : (kn < K ? knobs[(size_t)b * K + kn] : 0.f);   // prefetched with the inputs
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
                        st_sincos(ph, sn, cs);
                        mag_hat[ro * F + f] = mh;
                        phs_hat[ro * F + f] = ph;
                        reg += fabsf(mh * wf);
                    }
                    AA[ro * KP + f] = mh * cs;                          // f < FP always: pads get zeros
                    AA[ro * KP + FP + f] = mh * sn;
                }
            }
        }
        if (FAST) { fwd_mask(nxt, K, fn < F, T, OT, g); cur = nxt; }
    }
    if (reg_partial) {
        reg = wave_sum(reg);
        if (lane == 0) reg_partial[blockIdx.x * NW + wave] = reg;
    }
}

// ------------------------------------------------------------------------------------------ forward, inner layers only
// Wide geometries (st_ae_wide.h): layers 1 and 9 are GEMMs over feature-major activations X[feature][R] (R = B*FP columns,
// column = b*FP + f); this kernel fuses layers 2..8 (64 -> 32 -> 16 -> 16 -> [+knobs] 16 -> 16 -> 32 -> 64) for both
// autoencoders: H1 [64][R] -> H8 [64][R], activations in registers exactly as in ae_fwd_kernel.  A 16-column group never
// straddles windows (FP % 16 == 0), so the knobs stay wave-uniform.  Pad columns (f >= F) are written as zeros.
template <int NW>
__global__ void __launch_bounds__(NW * 64)
ae_inner_fwd_kernel(const float* __restrict__ H1m, const float* __restrict__ H1p, const float* __restrict__ knobs,
                    const float* __restrict__ ae_m, const float* __restrict__ ae_p, const AEOffsets go,
                    float* __restrict__ H8m, float* __restrict__ H8p, const int B, const int F, const int K, const int KP)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const AELds L = ae_lds_layout(16, 16, K);          // layers 0 and 8 get (unused) minimal regions
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, c = lane & 15;
    float* lw[2] = {lds, lds + L.total};
    ae_load_lds(lw[0], L, ae_m, go, tid, NW * 64, 1, 8);
    ae_load_lds(lw[1], L, ae_p, go, tid, NW * 64, 1, 8);
    __syncthreads();
    const int FP = KP / 2, gpw = FP / 16, ngroups = B * gpw;
    const size_t R = (size_t)B * FP;
    const int KQ = (K + 3) / 4;
    const float* const Hin[2] = {H1m, H1p};
    float* const Hout[2] = {H8m, H8p};
    for (int grp = blockIdx.x * NW + wave; grp < ngroups; grp += gridDim.x * NW) {
        asm volatile("" ::: "memory");
        const int b = grp / gpw, f = (grp - b * gpw) * 16 + c;
        const size_t col = (size_t)b * FP + f;
        f32x4 h1[2][4];
#pragma unroll
        for (int ch = 0; ch < 2; ++ch)
#pragma unroll
            for (int ot = 0; ot < 4; ++ot)
#pragma unroll
                for (int r = 0; r < 4; ++r) h1[ch][ot][r] = Hin[ch][(size_t)(16 * ot + 4 * g + r) * R + col];
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
        const bool fv = f < F;
#pragma unroll
        for (int ch = 0; ch < 2; ++ch)
#pragma unroll
            for (int ot = 0; ot < 4; ++ot)
#pragma unroll
                for (int r = 0; r < 4; ++r) Hout[ch][(size_t)(16 * ot + 4 * g + r) * R + col] = fv ? h8[ch][ot][r] : 0.f;
    }
}

// ========================================================================================== backward
// blockIdx.y selects the autoencoder (0 = magnitude 'sf', 1 = phase); one workgroup = NW waves, two waves per SIMD.
// Per 16-row group a wave
//   1. recomputes the forward chain, keeping every post-ELU activation in registers in D layout (68 regs);
//   2. forms d out (polar->rect backward of nn_proc.py:322-326 + the L1 term of loss_functions.py:36);
//   3. walks the layers backwards.  The data gradient continues the D-layout chain (A = W^T fragments).  The
//      weight gradient dW_l = sum_rows da_l (x) h_{l-1} needs both operands with ROWS on the MFMA k index, i.e.
//      transposed.  Instead of LDS round trips the transposed copies come from the SECOND MFMA ORIENTATION
//          D[row][o] = sum_i H[row][i] * W^T[i][o]      (A = activations in D layout, B = the SAME weight fragment)
//      whose result layout (lane (g,c), reg r <-> row 4g+r, feature 16*tile+c) is exactly the wgrad operand layout;
//      likewise da^T_{l-1} = (da_l W_l)^T.  ~1.6x the MFMAs, zero transposes, activations stay in registers.
//   4. the 16x16 tiles of dW_l are added into an LDS copy of the gradient with ds_add_f32 (one per workgroup) --
//      no persistent accumulator registers, so two waves per SIMD hide each other's latencies.  (The summation
//      order of these LDS atomics is not deterministic: last-bit run-to-run differences in the AE gradients.)
// At the end the workgroup stores its partial dW/db (packed like the parameters); ae_grad_reduce_kernel sums them.
constexpr int SP = 20;                 // scratch pitch (floats): 16 rows + 4, keeps rows 16-B aligned

// Diagnostics only (st_set_debug bit 8): wave 0 of workgroup (0,0) accumulates s_memtime deltas per kernel stage.
__device__ unsigned long long g_ae_stage_cycles[32];
// Stage timers exist only in the TIMED instantiation.  They must not be a run-time branch of the production kernel:
// a basic-block boundary between an MFMA chain and the first v_accvgpr_read of its result escapes the compiler's
// MFMA->VALU hazard padding (measured: the last k-step of layer 5 missing from accumulator element 3), hence the
// explicit wait states ahead of the branch in the TIMED build.
#define ST_T(i_) do { if constexpr (TIMED) { __builtin_amdgcn_sched_barrier(0); asm volatile("s_nop 15\n\ts_nop 15"); __builtin_amdgcn_sched_barrier(0); if (timing) { const unsigned long long t1_ = __builtin_amdgcn_s_memtime(); if (lane == 0) g_ae_stage_cycles[i_] += t1_ - t0_; t0_ = __builtin_amdgcn_s_memtime(); } } } while (0)

// ---------------------------------------------------------------------------------------------------------
// Weight fragments in registers.  Left to itself the compiler issues each MFMA's LDS weight fetch just before
// the MFMA (one s_waitcnt per MFMA: the AE kernels were LDS-latency-bound).  These helpers burst-load every
// fragment of a layer stage into a register array; the caller places a scheduling fence between the burst and
// the MFMAs, so a stage pays one LDS round trip instead of one per MFMA.
// forward-order fragment (ot, it, r):  W[o = 16 ot + c][i = 16 it + 4 g + r]
template <int OTL, int ITL>
__device__ __forceinline__ void frags_fwd(float (&fr)[OTL * ITL * 4], const float* W, const int P, const int g, const int c)
{
#pragma unroll
    for (int ot = 0; ot < OTL; ++ot)
#pragma unroll
        for (int it = 0; it < ITL; ++it)
#pragma unroll
            for (int r = 0; r < 4; ++r) fr[(ot * ITL + it) * 4 + r] = W[(16 * ot + c) * P + 16 * it + 4 * g + r];
}
// dgrad-order fragment (it, ot, r):  W[o = 16 ot + 4 g + r][i = 16 it + c]
template <int OTL, int ITL>
__device__ __forceinline__ void frags_dgrad(float (&fr)[OTL * ITL * 4], const float* W, const int P, const int g, const int c)
{
#pragma unroll
    for (int it = 0; it < ITL; ++it)
#pragma unroll
        for (int ot = 0; ot < OTL; ++ot)
#pragma unroll
            for (int r = 0; r < 4; ++r) fr[(it * OTL + ot) * 4 + r] = W[(16 * ot + 4 * g + r) * P + 16 * it + c];
}
#define ST_FENCE() __builtin_amdgcn_sched_barrier(0)

template <int OTL, int ITL>
__device__ __forceinline__ void fwdD_fr(const float (&fr)[OTL * ITL * 4], const float* bias, const f32x4 (&hin)[ITL],
                                        f32x4 (&hout)[OTL], const int g)
{
#pragma unroll
    for (int ot = 0; ot < OTL; ++ot) {
        const float4 bq = *reinterpret_cast<const float4*>(bias + 16 * ot + 4 * g);   // accumulator starts at the bias (D layout: o = 16 ot + 4 g + r)
        f32x4 acc = (f32x4){bq.x, bq.y, bq.z, bq.w};
#pragma unroll
        for (int it = 0; it < ITL; ++it)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc = ST_MFMA16(fr[(ot * ITL + it) * 4 + r], hin[it][r], acc);
#pragma unroll
        for (int r = 0; r < 4; ++r) hout[ot][r] = elu_f(acc[r]);
    }
}
template <int OTL, int ITL>
__device__ __forceinline__ void fwdT_fr(const float (&fr)[OTL * ITL * 4], const float* bias, const f32x4 (&hin)[ITL],
                                        f32x4 (&houtT)[OTL], const int c)
{
#pragma unroll
    for (int ot = 0; ot < OTL; ++ot) {
        const float bv = bias[16 * ot + c];                    // T layout: feature 16 ot + c in every register
        f32x4 acc = (f32x4){bv, bv, bv, bv};
#pragma unroll
        for (int it = 0; it < ITL; ++it)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc = ST_MFMA16(hin[it][r], fr[(ot * ITL + it) * 4 + r], acc);
#pragma unroll
        for (int r = 0; r < 4; ++r) houtT[ot][r] = elu_f(acc[r]);
    }
}
// both data-gradient orientations from one fragment set
template <int OTL, int ITL>
__device__ __forceinline__ void dgrad_fr(const float (&fr)[OTL * ITL * 4], const f32x4 (&da)[OTL], f32x4 (&dh)[ITL], f32x4 (&dhT)[ITL])
{
#pragma unroll
    for (int it = 0; it < ITL; ++it) {
        f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f}, accT = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ot = 0; ot < OTL; ++ot)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float w = fr[(it * OTL + ot) * 4 + r];
                acc = ST_MFMA16(w, da[ot][r], acc);
                accT = ST_MFMA16(da[ot][r], w, accT);
            }
        dh[it] = acc; dhT[it] = accT;
    }
}
// D layout -> T layout of TL 16x16 tiles through a wave-private LDS scratch ([tile][row 16][feature 16, pitch 20]): one
// ds_write_b128 + four ds_read_b32 per tile, no barrier (LDS operations of one wave execute in order).  The transposed
// operands of the weight gradients used to be RE-COMPUTED with the second MFMA orientation (280 MFMAs + their ELU /
// ELU' per group): in a kernel that is issue-bound, not MFMA-bound, the transpose is far cheaper.
template <int TL>
__device__ __forceinline__ void to_T(float* scr, const f32x4 (&d)[TL], f32x4 (&t)[TL], const int g, const int c)
{
#pragma unroll
    for (int k = 0; k < TL; ++k)
        *reinterpret_cast<float4*>(scr + k * 320 + c * 20 + 4 * g) = make_float4(d[k][0], d[k][1], d[k][2], d[k][3]);
#pragma unroll
    for (int k = 0; k < TL; ++k)
#pragma unroll
        for (int r = 0; r < 4; ++r) t[k][r] = scr[k * 320 + (4 * g + r) * 20 + c];
}

template <int OTL, int ITL>
__device__ __forceinline__ void dgradD_fr(const float (&fr)[OTL * ITL * 4], const f32x4 (&da)[OTL], f32x4 (&dh)[ITL])
{
#pragma unroll
    for (int it = 0; it < ITL; ++it) {
        f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ot = 0; ot < OTL; ++ot)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc = ST_MFMA16(fr[(it * OTL + ot) * 4 + r], da[ot][r], acc);
        dh[it] = acc;
    }
}

// T-layout forward of one layer from the D-layout activations of the previous one:
// houtT[ot][r] = ELU(a)[row 4g+r][feature 16 ot + c]
template <int OTL, int ITL>
__device__ __forceinline__ void layer_fwdT(const float* W, const float* bias, const int P, const f32x4 (&hin)[ITL],
                                           f32x4 (&houtT)[OTL], const int g, const int c)
{
#pragma unroll
    for (int ot = 0; ot < OTL; ++ot) {
        f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int it = 0; it < ITL; ++it)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                acc = ST_MFMA16(hin[it][r], W[(16 * ot + c) * P + 16 * it + 4 * g + r], acc);
        const float bv = bias[16 * ot + c];
#pragma unroll
        for (int r = 0; r < 4; ++r) houtT[ot][r] = elu_f(acc[r] + bv);
    }
}

// D-layout forward, single chain
template <int OTL, int ITL>
__device__ __forceinline__ void layer_fwdD(const float* W, const float* bias, const int P, const f32x4 (&hin)[ITL],
                                           f32x4 (&hout)[OTL], const int g, const int c)
{
#pragma unroll
    for (int ot = 0; ot < OTL; ++ot) {
        f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int it = 0; it < ITL; ++it)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                acc = ST_MFMA16(W[(16 * ot + c) * P + 16 * it + 4 * g + r], hin[it][r], acc);
#pragma unroll
        for (int r = 0; r < 4; ++r) hout[ot][r] = elu_f(acc[r] + bias[16 * ot + 4 * g + r]);
    }
}

// D-layout data gradient: dh[it] = sum_o W[o][16 it + c-th feature] * da[o]   (A = W^T fragments)
template <int OTL, int ITL>
__device__ __forceinline__ void layer_dgrad(const float* W, const int P, const f32x4 (&da)[OTL], f32x4 (&dh)[ITL],
                                            const int g, const int c)
{
#pragma unroll
    for (int it = 0; it < ITL; ++it) {
        f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ot = 0; ot < OTL; ++ot)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                acc = ST_MFMA16(W[(16 * ot + 4 * g + r) * P + 16 * it + c], da[ot][r], acc);
        dh[it] = acc;
    }
}
// T-layout data gradient: dhT[it][r] = (da W)[row 4g+r][feature 16 it + c]   (A = da in D layout, B = same fragments)
template <int OTL, int ITL>
__device__ __forceinline__ void layer_dgradT(const float* W, const int P, const f32x4 (&da)[OTL], f32x4 (&dhT)[ITL],
                                             const int g, const int c)
{
#pragma unroll
    for (int it = 0; it < ITL; ++it) {
        f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ot = 0; ot < OTL; ++ot)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                acc = ST_MFMA16(da[ot][r], W[(16 * ot + 4 * g + r) * P + 16 * it + c], acc);
        dhT[it] = acc;
    }
}

template <int TL>
__device__ __forceinline__ void mul_elu_grad(f32x4 (&d)[TL], const f32x4 (&h)[TL])
{
#pragma unroll
    for (int t = 0; t < TL; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) d[t][r] = __builtin_fmaf(d[t][r], fminf(h[t][r], 0.f), d[t][r]);   // d * ELU'(a) = d * (1 + min(h, 0)): v_min + v_fma
}

// dW_l tile(ot,it) += sum_rows daT[ot] (x) hT[it]; result (D layout: o = 16ot+4g+r, i = 16it+c) -> LDS atomics.
// db_l (lane c <-> o = 16 ot + c) accumulates the row sums of daT in registers.
template <int OTL, int ITL>
__device__ __forceinline__ void wgrad_lds(float* dW, const int P, const f32x4 (&daT)[OTL], const f32x4 (&hT)[ITL],
                                          float* db, const int g, const int c)
{
#pragma unroll
    for (int ot = 0; ot < OTL; ++ot) {
#pragma unroll
        for (int it = 0; it < ITL; ++it) {
            f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int r = 0; r < 4; ++r) acc = ST_MFMA16(daT[ot][r], hT[it][r], acc);
#pragma unroll
            for (int r = 0; r < 4; ++r) atomicAdd(dW + (16 * ot + 4 * g + r) * P + 16 * it + c, acc[r]);
        }
        atomicAdd(db + 16 * ot + c, (daT[ot][0] + daT[ot][1]) + (daT[ot][2] + daT[ot][3]));   // bias gradient: row sums (4 lane groups hit the same word)
    }
}

// Register-accumulator form: dW tiles and the bias row sums persist in registers for the whole kernel.
template <int OTL, int ITL>
__device__ __forceinline__ void wgrad_reg(f32x4 (&dW)[OTL][ITL], float (&db)[OTL], const f32x4 (&daT)[OTL], const f32x4 (&hT)[ITL])
{
#pragma unroll
    for (int ot = 0; ot < OTL; ++ot) {
#pragma unroll
        for (int it = 0; it < ITL; ++it)
#pragma unroll
            for (int r = 0; r < 4; ++r) dW[ot][it] = ST_MFMA16(daT[ot][r], hT[it][r], dW[ot][it]);
        db[ot] += (daT[ot][0] + daT[ot][1]) + (daT[ot][2] + daT[ot][3]);
    }
}
// Flush of one wave's persistent accumulators into the workgroup's LDS gradient copy: plain read-modify-write -- the
// caller serialises the waves (wave 0, barrier, wave 1, ...) so the summation order, hence every bit, is fixed.
template <int OTL, int ITL>
__device__ __forceinline__ void dw_flush(float* dW, const int P, const f32x4 (&acc)[OTL][ITL], const int g, const int c)
{
#pragma unroll
    for (int ot = 0; ot < OTL; ++ot)
#pragma unroll
        for (int it = 0; it < ITL; ++it)
#pragma unroll
            for (int r = 0; r < 4; ++r) dW[(16 * ot + 4 * g + r) * P + 16 * it + c] += acc[ot][it][r];
}

template <int OTL>
__device__ __forceinline__ void db_flush(float* dst, float (&db)[OTL], const int g, const int c)
{
#pragma unroll
    for (int ot = 0; ot < OTL; ++ot) {
        float v = db[ot];
        v += __shfl_xor(v, 16); v += __shfl_xor(v, 32);
        if (g == 0) dst[16 * ot + c] += v;
    }
}

// Compile-time LDS layout of the (T <= 32, OT <= 16, K <= 16) instantiation: every padded dimension is fixed
// (IN_1 -> 32, OUT_9 -> 16, IN_5 -> 32), so all offsets / pitches fold into instruction immediates.
struct CL {
    static constexpr int P0 = 33, P1 = 65, P2 = 33, P3 = 17, P4 = 33, P5 = 17, P6 = 17, P7 = 33, P8 = 65;
    static constexpr int W0 = 0,            B0 = W0 + 64 * P0;
    static constexpr int W1 = B0 + 64,      B1 = W1 + 32 * P1;
    static constexpr int W2 = B1 + 32,      B2 = W2 + 16 * P2;
    static constexpr int W3 = B2 + 16,      B3 = W3 + 16 * P3;
    static constexpr int W4 = B3 + 16,      B4 = W4 + 16 * P4;
    static constexpr int W5 = B4 + 16,      B5 = W5 + 16 * P5;
    static constexpr int W6 = B5 + 16,      B6 = W6 + 32 * P6;
    static constexpr int W7 = B6 + 32,      B7 = W7 + 64 * P7;
    static constexpr int W8 = B7 + 64,      B8 = W8 + 16 * P8;
    static constexpr int TOTAL = (B8 + 16 + 3) / 4 * 4;
};
#define ST_SCHED_FENCE() do { if constexpr (!REG) __builtin_amdgcn_sched_barrier(0); } while (0)

// Supported geometry of this instantiation: T <= 32, OT <= 16, K <= 16.
// INNER (wide geometries, st_ae_wide.h): only layers 2..8 -- layers 1 and 9 are feature-major GEMMs.  Pointer roles then:
//   mag / phs         -> H1 [64][R] of the two nets (layer-1 outputs, R = B*FP columns)
//   mag_hat / phs_hat -> dH8 [64][R] = W9^T dA9 (the kernel applies ELU'(h8) itself: h8 is recomputed)
//   dmag / dphs       -> dA1 [64][R] = (W2^T dA2) * ELU'(h1), consumed by the layer-1 weight/data-gradient GEMMs
// and the partial gradients of layers 1 and 9 stay zero.
template <int NW, bool REG, bool TIMED, bool INNER = false>   // REG: persistent register accumulators (1 wave/SIMD); else per-group LDS atomics (2 waves/SIMD)
__global__ void __launch_bounds__(NW * 64, REG ? 1 : 2)
ae_bwd_kernel(const float* __restrict__ mag, const float* __restrict__ phs, const float* __restrict__ knobs,
              const float* __restrict__ ae_m, const float* __restrict__ ae_p, const AEOffsets go, const int PG,
              const float* __restrict__ mag_hat, const float* __restrict__ phs_hat, const float* __restrict__ dAA,
              const float* __restrict__ g_mag_hat, const float reg_coef, const float expfac,
              float* __restrict__ dmag, float* __restrict__ dphs, float* __restrict__ ws,
              const int B, const int T, const int OT, const int F, const int K, const int KP,
              const int to_lo, const int to_hi,      // live synthesis frames: dAA rows outside are treated as zero
              const int nslab, const size_t slab,     // dAA arrives as split-K slabs of the synthesis dgrad GEMM
              const int dbg)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int ae = blockIdx.y;
    const bool timing = TIMED && (dbg & 256) && blockIdx.x == 0 && blockIdx.y == 0 && (threadIdx.x >> 6) == 0;
    unsigned long long t0_ = timing ? __builtin_amdgcn_s_memtime() : 0ull;
    const AELds L = INNER ? ae_lds_layout(32, 16, K) : ae_lds_layout(T, OT, K);       // INNER: the CL layout with layers 1 / 9 left empty
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, c = lane & 15;
    float* lw = lds;                                   // weights (+bias), padded
    float* dwl = lds + CL::TOTAL;                      // gradient accumulator, same layout
    // wave-private scratch: V[32*SP] (input rows, transposed), Y[16*SP] (d a9 transposed), TAIL[16*SP]
    constexpr int SCR = (32 + 16 + 16) * SP + 2 * 4 * 320;      // + two 4-tile transpose scratches (to_T)
    float* Vs = lds + 2 * CL::TOTAL + wave * SCR;
    float* Ys = Vs + 32 * SP;
    float* Ts = Ys + 16 * SP;
    float* XH = Ts + 16 * SP;                          // transposes of activations
    float* XD = XH + 4 * 320;                          // transposes of activation gradients
    for (int e = tid; e < CL::TOTAL; e += NW * 64) dwl[e] = 0.f;
    ae_load_lds(lw, L, ae ? ae_p : ae_m, go, tid, NW * 64, INNER ? 1 : 0, INNER ? 8 : NL);
    __syncthreads();

    const float* vin = ae ? phs : mag;
    const float* dh8in = ae ? phs_hat : mag_hat;       // INNER only
    float* dvout = ae ? dphs : dmag;
    const int FP = KP / 2, gpw = FP / 16;
    const int ngroups = B * gpw;
    const int KS1 = (T + 3) / 4;
    const int gstride = gridDim.x * NW;

    const float* const Wl[NL] = {lw + CL::W0, lw + CL::W1, lw + CL::W2, lw + CL::W3, lw + CL::W4, lw + CL::W5, lw + CL::W6, lw + CL::W7, lw + CL::W8};
    const float* const Bl[NL] = {lw + CL::B0, lw + CL::B1, lw + CL::B2, lw + CL::B3, lw + CL::B4, lw + CL::B5, lw + CL::B6, lw + CL::B7, lw + CL::B8};
    float* const Dl[NL] = {dwl + CL::W0, dwl + CL::W1, dwl + CL::W2, dwl + CL::W3, dwl + CL::W4, dwl + CL::W5, dwl + CL::W6, dwl + CL::W7, dwl + CL::W8};
    float* const db1 = dwl + CL::B0; float* const db2 = dwl + CL::B1; float* const db3 = dwl + CL::B2;
    float* const db4 = dwl + CL::B3; float* const db5 = dwl + CL::B4; float* const db6 = dwl + CL::B5;
    float* const db7 = dwl + CL::B6; float* const db8 = dwl + CL::B7; float* const db9 = dwl + CL::B8;
    // REG mode: 36 persistent 16x16 dW tiles (144 registers) + 17 bias-gradient registers
    f32x4 rW1[4][2], rW2[2][4], rW3[1][2], rW4[1][1], rW5[1][2], rW6[1][1], rW7[2][1], rW8[4][2], rW9[1][4];
    float rb1[4], rb2[2], rb3[1], rb4[1], rb5[1], rb6[1], rb7[2], rb8[4], rb9[1];
#define ST_ZT(x, A, Bq) { _Pragma("unroll") for (int a_ = 0; a_ < A; ++a_) _Pragma("unroll") for (int b_ = 0; b_ < Bq; ++b_) x[a_][b_] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
#define ST_ZB(x, A) { _Pragma("unroll") for (int a_ = 0; a_ < A; ++a_) x[a_] = 0.f; }
    ST_ZT(rW1, 4, 2) ST_ZT(rW2, 2, 4) ST_ZT(rW3, 1, 2) ST_ZT(rW4, 1, 1) ST_ZT(rW5, 1, 2) ST_ZT(rW6, 1, 1) ST_ZT(rW7, 2, 1) ST_ZT(rW8, 4, 2) ST_ZT(rW9, 1, 4)
    ST_ZB(rb1, 4) ST_ZB(rb2, 2) ST_ZB(rb3, 1) ST_ZB(rb4, 1) ST_ZB(rb5, 1) ST_ZB(rb6, 1) ST_ZB(rb7, 2) ST_ZB(rb8, 4) ST_ZB(rb9, 1)
#undef ST_ZT
#undef ST_ZB

    // input rows of the first group (prefetched one group ahead afterwards): t = 4 ks + g, row c
    float vr[8];
    int grp = blockIdx.x * NW + wave;
    // Input rows of group gq as RAW loads from clamped (always valid) addresses; mask_v() zeroes the padding
    // rows/bins afterwards.  Keeping the select out of the load sequence (and the whole prefetch out of a
    // conditional block) matters: the `if (next < ngroups) { x = load; dst = ok ? x : 0; }` form compiled to
    // eight load / s_waitcnt vmcnt(0) pairs, i.e. eight serialized memory round trips per group.
    auto load_v = [&](int gq, float (&dst)[8]) {
        const int bq = gq / gpw, fq = (gq - bq * gpw) * 16 + c;
        const int fc = fq < F ? fq : 0;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            const int t = 4 * ks + g;
            dst[ks] = vin[((unsigned)bq * T + (t < T ? t : 0)) * F + fc];
        }
    };
    auto mask_v = [&](int gq, float (&dst)[8]) {
        const int bq = gq / gpw, fq = (gq - bq * gpw) * 16 + c;
        const bool ok0 = fq < F;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) dst[ks] = (ok0 && 4 * ks + g < T) ? dst[ks] : 0.f;
    };
    if constexpr (!INNER) { if (grp < ngroups) { load_v(grp, vr); mask_v(grp, vr); } }
    const unsigned Rw = (unsigned)B * FP;              // INNER: columns of the feature-major buffers

    for (; grp < ngroups; grp += gstride) {
        asm volatile("" ::: "memory");      // keep the (loop-invariant) LDS weight fetches inside the loop
        const int b = grp / gpw, f = (grp - b * gpw) * 16 + c;
        const bool fv = f < F;
        const int fq = fv ? f : 0;
        // ---- d-out inputs for this group (D layout: t' = 4g + r), issued early.  (Slicing these 53 loads between the
        // forward stages to overlap their issue with MFMA execution was measured: no gain.)
        float q_gre[4], q_gim[4], q_ph[4], q_mh[4], q_mt[4], q_gm[4];
        const float* gmp = g_mag_hat ? g_mag_hat : mag_hat;
        if constexpr (!INNER) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int to = 4 * g + r;
            const bool ok = fv && to < OT;
            const bool lv = ok && to >= to_lo && to <= to_hi;
            const unsigned ro = (unsigned)b * OT + (ok ? to : 0);      // 32-bit offsets throughout (host checks the sizes): saddr + voffset loads
            // up to 3 split-K slabs: all six loads are issued together, then summed (a runtime-trip-count loop here
            // serialised ~12 memory round trips per group)
            const unsigned p0 = (lv ? ro : 0u) * KP + fq;
            const unsigned o1 = nslab > 1 ? (unsigned)slab : 0u, o2 = nslab > 2 ? 2u * (unsigned)slab : 0u;
            const float x0 = dAA[p0], x1 = dAA[p0 + o1], x2 = dAA[p0 + o2];
            const float y0 = dAA[p0 + FP], y1 = dAA[p0 + o1 + FP], y2 = dAA[p0 + o2 + FP];
            const float a0 = x0 + (nslab > 1 ? x1 : 0.f) + (nslab > 2 ? x2 : 0.f);
            const float a1 = y0 + (nslab > 1 ? y1 : 0.f) + (nslab > 2 ? y2 : 0.f);
            q_gre[r] = lv ? a0 : 0.f; q_gim[r] = lv ? a1 : 0.f;
            q_ph[r] = phs_hat[ro * F + fq]; q_mh[r] = mag_hat[ro * F + fq];
            q_mt[r] = vin[((unsigned)b * T + (ok ? T - OT + to : 0)) * F + fq];
            { const float x = gmp[ro * F + fq]; q_gm[r] = (g_mag_hat && ok) ? x : 0.f; }      // branch-free (see load_v)
        }
        }
        // INNER: layer-1 outputs in both layouts and the gradient entering layer 8's output, straight from the
        // feature-major buffers (D layout: feature 16 tile + 4g + r at column col0 + c; T layout: feature 16 tile + c at
        // columns col0 + 4g .. + 3 = one aligned float4)
        f32x4 h1in[4], dh8[4];
        if constexpr (INNER) {
            const unsigned col0 = (unsigned)b * FP + (unsigned)(grp - b * gpw) * 16;
#pragma unroll
            for (int ot = 0; ot < 4; ++ot) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    h1in[ot][r] = vin[(unsigned)(16 * ot + 4 * g + r) * Rw + col0 + c];
                    dh8[ot][r] = dh8in[(unsigned)(16 * ot + 4 * g + r) * Rw + col0 + c];
                }
            }
        }
        // knob values: D-layout feature tile (16 + 4g + r) and T-layout feature lane (16 + c)
        f32x4 kn[1]; float knT;
#pragma unroll
        for (int r = 0; r < 4; ++r) { const int kidx = 4 * g + r; const float x = knobs[(unsigned)b * K + (kidx < K ? kidx : 0)]; kn[0][r] = kidx < K ? x : 0.f; }
        { const float x = knobs[(unsigned)b * K + (c < K ? c : 0)]; knT = c < K ? x : 0.f; }
        float vn[8];
        const int gnext = grp + gstride < ngroups ? grp + gstride : grp;      // last iteration: harmless reload of this group
        if constexpr (!INNER) load_v(gnext, vn);
        ST_T(0);

        // ------------------------------------------------------------------ forward recompute (D layout)
        f32x4 h1[4], h2[2], h3[1], h4[1], h5[1], h6[1], h7[2], h8[4], e9[1];
        if constexpr (INNER) {
#pragma unroll
            for (int ot = 0; ot < 4; ++ot) h1[ot] = h1in[ot];
        } else {
            f32x4 acc[4];
#pragma unroll
            for (int ot = 0; ot < 4; ++ot) acc[ot] = (f32x4){0.f, 0.f, 0.f, 0.f};
            const int P1 = CL::P0;
            float ff[4 * 8];
#pragma unroll
            for (int ot = 0; ot < 4; ++ot)
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) ff[ot * 8 + ks] = Wl[0][(16 * ot + c) * P1 + 4 * ks + g];
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) Vs[(4 * ks + g) * SP + c] = vr[ks];   // [feature t][row c]: read back transposed for the layer-1 wgrad
            ST_FENCE();
#pragma unroll
            for (int ks = 0; ks < 8; ++ks)
                if (ks < KS1) {
#pragma unroll
                    for (int ot = 0; ot < 4; ++ot) acc[ot] = ST_MFMA16(ff[ot * 8 + ks], vr[ks], acc[ot]);
                }
#pragma unroll
            for (int ot = 0; ot < 4; ++ot)
#pragma unroll
                for (int r = 0; r < 4; ++r) h1[ot][r] = elu_f(acc[ot][r] + Bl[0][16 * ot + 4 * g + r]);
        }
        ST_T(1);
        { float fr[2 * 4 * 4]; frags_fwd<2, 4>(fr, Wl[1], CL::P1, g, c); ST_FENCE(); fwdD_fr<2, 4>(fr, Bl[1], h1, h2, g); }
        ST_T(2);
        { float fr[1 * 2 * 4]; frags_fwd<1, 2>(fr, Wl[2], CL::P2, g, c); ST_FENCE(); fwdD_fr<1, 2>(fr, Bl[2], h2, h3, g); }
        { float fr[1 * 1 * 4]; frags_fwd<1, 1>(fr, Wl[3], CL::P3, g, c); ST_FENCE(); fwdD_fr<1, 1>(fr, Bl[3], h3, h4, g); }
        {
            f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
            const int P5 = CL::P4;
            float fa[4], fb[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) { fa[r] = Wl[4][c * P5 + 4 * g + r]; fb[r] = Wl[4][c * P5 + 16 + 4 * g + r]; }
            ST_FENCE();
#pragma unroll
            for (int r = 0; r < 4; ++r) acc = ST_MFMA16(fa[r], h4[0][r], acc);
#pragma unroll
            for (int r = 0; r < 4; ++r) acc = ST_MFMA16(fb[r], kn[0][r], acc);   // knob features 16 + 4g + r
#pragma unroll
            for (int r = 0; r < 4; ++r) h5[0][r] = elu_f(acc[r] + Bl[4][4 * g + r]);
        }
        ST_T(3);
        { float fr[1 * 1 * 4]; frags_fwd<1, 1>(fr, Wl[5], CL::P5, g, c); ST_FENCE(); fwdD_fr<1, 1>(fr, Bl[5], h5, h6, g); }
        { float fr[2 * 1 * 4]; frags_fwd<2, 1>(fr, Wl[6], CL::P6, g, c); ST_FENCE(); fwdD_fr<2, 1>(fr, Bl[6], h6, h7, g); }
        ST_T(4);
        { float fr[4 * 2 * 4]; frags_fwd<4, 2>(fr, Wl[7], CL::P7, g, c); ST_FENCE(); fwdD_fr<4, 2>(fr, Bl[7], h7, h8, g); }
        ST_T(5);
        if constexpr (!INNER) { float fr[1 * 4 * 4]; frags_fwd<1, 4>(fr, Wl[8], CL::P8, g, c); ST_FENCE(); fwdD_fr<1, 4>(fr, Bl[8], h8, e9, g); }
        ST_T(6);
        ST_SCHED_FENCE();
        // ------------------------------------------------------------------ d out  (D layout: t' = 4g + r)
        f32x4 da9[1];
        if constexpr (!INNER) {
            const float wf = fv ? expf(expfac * (float)f) : 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int to = 4 * g + r;
                float d9 = 0.f, tail = 0.f;
                if (fv && to < OT) {
                    const float gre = q_gre[r], gim = q_gim[r], ph = q_ph[r], mh = q_mh[r];
                    float sn, cs; st_sincos(ph, sn, cs);
                    if (ae == 0) {
                        const float sg = mh > 0.f ? 1.f : (mh < 0.f ? -1.f : 0.f);
                        const float dmh = gre * cs + gim * sn + reg_coef * sg * wf + q_gm[r];
                        d9 = dmh * q_mt[r] * elu_grad_from_out(e9[0][r]);
                        tail = dmh * e9[0][r];
                    } else {
                        const float dph = mh * (gim * cs - gre * sn);
                        d9 = dph * elu_grad_from_out(e9[0][r]);
                        tail = dph;
                    }
                }
                da9[0][r] = d9;
                Ts[to * SP + c] = tail;
                Ys[to * SP + c] = d9;                      // [feature t'][row c] -> read back transposed below
            }
        }
        // ------------------------------------------------------------------ backward through the layers
        // T layout: lane (g,c), reg r  <->  row 4g + r, feature 16*tile + c
        f32x4 daT9[1];
        if constexpr (!INNER) { const float4 v = *reinterpret_cast<const float4*>(Ys + c * SP + 4 * g); daT9[0] = (f32x4){v.x, v.y, v.z, v.w}; }
#define ST_WG(O_, I_, D_, P_, RW_, RB_, DB_, DAT_, HT_) \
        if constexpr (REG) wgrad_reg<O_, I_>(RW_, RB_, DAT_, HT_); else wgrad_lds<O_, I_>(D_, P_, DAT_, HT_, DB_, g, c);
        ST_T(7);
        // layer 9 (64 -> OT): needs h8^T (layer-8 forward fragments) and W9 in dgrad order
        f32x4 hT8[4], da8[4], daT8[4];
        if constexpr (INNER) {                     // dH8 arrives from the layer-9 data-gradient GEMM; h8^T is still needed for ELU' and dW8
            to_T<4>(XH, h8, hT8, g, c);
#pragma unroll
            for (int ot = 0; ot < 4; ++ot) da8[ot] = dh8[ot];
            mul_elu_grad<4>(da8, h8); to_T<4>(XD, da8, daT8, g, c);
        } else {
            float fd[1 * 4 * 4];
            frags_dgrad<1, 4>(fd, Wl[8], CL::P8, g, c); ST_FENCE();
            to_T<4>(XH, h8, hT8, g, c);
            ST_WG(1, 4, Dl[8], CL::P8, rW9, rb9, db9, daT9, hT8)
            dgradD_fr<1, 4>(fd, da9, da8); mul_elu_grad<4>(da8, h8); to_T<4>(XD, da8, daT8, g, c);
        }
        ST_T(8);
        // layer 8 (32 -> 64)
        f32x4 hT7[2], da7[2], daT7[2];
        {
            float fd[4 * 2 * 4];
            frags_dgrad<4, 2>(fd, Wl[7], CL::P7, g, c); ST_FENCE();
            to_T<2>(XH, h7, hT7, g, c);
            ST_WG(4, 2, Dl[7], CL::P7, rW8, rb8, db8, daT8, hT7)
            dgradD_fr<4, 2>(fd, da8, da7); mul_elu_grad<2>(da7, h7); to_T<2>(XD, da7, daT7, g, c);
        }
        ST_T(9);
        // layer 7 (16 -> 32)
        f32x4 hT6[1], da6[1], daT6[1];
        {
            float fd[2 * 1 * 4];
            frags_dgrad<2, 1>(fd, Wl[6], CL::P6, g, c); ST_FENCE();
            to_T<1>(XH, h6, hT6, g, c);
            ST_WG(2, 1, Dl[6], CL::P6, rW7, rb7, db7, daT7, hT6)
            dgradD_fr<2, 1>(fd, da7, da6); mul_elu_grad<1>(da6, h6); to_T<1>(XD, da6, daT6, g, c);
        }
        ST_T(10);
        // layer 6 (16 -> 16)
        f32x4 hT5[1], da5[1], daT5[1];
        {
            float fd[1 * 1 * 4];
            frags_dgrad<1, 1>(fd, Wl[5], CL::P5, g, c); ST_FENCE();
            to_T<1>(XH, h5, hT5, g, c);
            ST_WG(1, 1, Dl[5], CL::P5, rW6, rb6, db6, daT6, hT5)
            dgradD_fr<1, 1>(fd, da6, da5); mul_elu_grad<1>(da5, h5); to_T<1>(XD, da5, daT5, g, c);
        }
        // layer 5 ([h4 ; knobs] -> 16): weight gradient over both input tiles, data gradient to h4 only
        f32x4 hT4[1], hT4k[2], da4[1], daT4[1];
        {
            float fd[1 * 1 * 4];
            frags_dgrad<1, 1>(fd, Wl[4], CL::P4, g, c); ST_FENCE();
            to_T<1>(XH, h4, hT4, g, c);
            hT4k[0] = hT4[0];
            hT4k[1] = (f32x4){knT, knT, knT, knT};                   // features 16 + c = knob c, every row
            ST_WG(1, 2, Dl[4], CL::P4, rW5, rb5, db5, daT5, hT4k)
            dgradD_fr<1, 1>(fd, da5, da4); mul_elu_grad<1>(da4, h4); to_T<1>(XD, da4, daT4, g, c);
        }
        // layer 4 (16 -> 16)
        f32x4 hT3[1], da3[1], daT3[1];
        {
            float fd[1 * 1 * 4];
            frags_dgrad<1, 1>(fd, Wl[3], CL::P3, g, c); ST_FENCE();
            to_T<1>(XH, h3, hT3, g, c);
            ST_WG(1, 1, Dl[3], CL::P3, rW4, rb4, db4, daT4, hT3)
            dgradD_fr<1, 1>(fd, da4, da3); mul_elu_grad<1>(da3, h3); to_T<1>(XD, da3, daT3, g, c);
        }
        ST_T(11);
        // layer 3 (32 -> 16)
        f32x4 hT2[2], da2[2], daT2[2];
        {
            float fd[1 * 2 * 4];
            frags_dgrad<1, 2>(fd, Wl[2], CL::P2, g, c); ST_FENCE();
            to_T<2>(XH, h2, hT2, g, c);
            ST_WG(1, 2, Dl[2], CL::P2, rW3, rb3, db3, daT3, hT2)
            dgradD_fr<1, 2>(fd, da3, da2); mul_elu_grad<2>(da2, h2); to_T<2>(XD, da2, daT2, g, c);
        }
        ST_T(12);
        // layer 2 (64 -> 32); h1^T from the input rows
        f32x4 hT1[4], da1[4], daT1[4];
        if constexpr (INNER) {                     // dA1 goes back to memory for the layer-1 GEMMs
            float fd[2 * 4 * 4];
            frags_dgrad<2, 4>(fd, Wl[1], CL::P1, g, c); ST_FENCE();
            to_T<4>(XH, h1, hT1, g, c);
            ST_WG(2, 4, Dl[1], CL::P1, rW2, rb2, db2, daT2, hT1)
            dgradD_fr<2, 4>(fd, da2, da1); mul_elu_grad<4>(da1, h1);
            const unsigned col0 = (unsigned)b * FP + (unsigned)(grp - b * gpw) * 16;
#pragma unroll
            for (int ot = 0; ot < 4; ++ot)
#pragma unroll
                for (int r = 0; r < 4; ++r) dvout[(unsigned)(16 * ot + 4 * g + r) * Rw + col0 + c] = da1[ot][r];
        } else {
            float fd[2 * 4 * 4];
            frags_dgrad<2, 4>(fd, Wl[1], CL::P1, g, c); ST_FENCE();
            to_T<4>(XH, h1, hT1, g, c);
            ST_WG(2, 4, Dl[1], CL::P1, rW2, rb2, db2, daT2, hT1)
            dgradD_fr<2, 4>(fd, da2, da1); mul_elu_grad<4>(da1, h1); to_T<4>(XD, da1, daT1, g, c);
        }
        ST_T(13);
        // layer 1 (T -> 64): input rows transposed through the wave's scratch
        f32x4 vT[2], dv[2];
        if constexpr (!INNER) {
            float fd[4 * 2 * 4];
            frags_dgrad<4, 2>(fd, Wl[0], CL::P0, g, c);
#pragma unroll
            for (int it = 0; it < 2; ++it) { const float4 v = *reinterpret_cast<const float4*>(Vs + (16 * it + c) * SP + 4 * g); vT[it] = (f32x4){v.x, v.y, v.z, v.w}; }
            ST_FENCE();
            ST_WG(4, 2, Dl[0], CL::P0, rW1, rb1, db1, daT1, vT)
            dgradD_fr<4, 2>(fd, da1, dv);
        }
#undef ST_WG
        ST_T(14);
        // ------------------------------------------------------------------ d input rows (+ skip / residual tails)
        if constexpr (!INNER) {
        // Materialise the accumulators in VGPRs HERE, in the block of the MFMAs that produce them: the stores below sit in
        // conditional blocks, and an accumulator first read behind a skipped block would be read before the MFMA has
        // finished (the hazard class tools/check_mfma_hazards.py scans for).
        float dvs[2][4];
#pragma unroll
        for (int it = 0; it < 2; ++it)
#pragma unroll
            for (int r = 0; r < 4; ++r) { dvs[it][r] = dv[it][r]; asm volatile("" : "+v"(dvs[it][r])); }
#pragma unroll
        for (int it = 0; it < 2; ++it)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int t = 16 * it + 4 * g + r;
                if (fv && t < T) {
                    float v = dvs[it][r];
                    if (t >= T - OT) v += Ts[(t - (T - OT)) * SP + c];
                    dvout[((unsigned)b * T + t) * F + f] = v;
                }
            }
        }
        ST_T(15);
        if constexpr (!INNER) {
            mask_v(gnext, vn);
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) vr[ks] = vn[ks];
        }
    }
    // ---------------------------------------------------------------------- workgroup partial gradients
    if constexpr (REG) {
        for (int wv = 0; wv < NW; ++wv) {          // ordered, non-atomic: run-to-run identical bits
            if (wave == wv) {
                if constexpr (!INNER) { dw_flush<4, 2>(Dl[0], CL::P0, rW1, g, c); dw_flush<1, 4>(Dl[8], CL::P8, rW9, g, c); db_flush<4>(db1, rb1, g, c); db_flush<1>(db9, rb9, g, c); }
                dw_flush<2, 4>(Dl[1], CL::P1, rW2, g, c); dw_flush<1, 2>(Dl[2], CL::P2, rW3, g, c);
                dw_flush<1, 1>(Dl[3], CL::P3, rW4, g, c); dw_flush<1, 2>(Dl[4], CL::P4, rW5, g, c); dw_flush<1, 1>(Dl[5], CL::P5, rW6, g, c);
                dw_flush<2, 1>(Dl[6], CL::P6, rW7, g, c); dw_flush<4, 2>(Dl[7], CL::P7, rW8, g, c);
                db_flush<2>(db2, rb2, g, c); db_flush<1>(db3, rb3, g, c); db_flush<1>(db4, rb4, g, c);
                db_flush<1>(db5, rb5, g, c); db_flush<1>(db6, rb6, g, c); db_flush<2>(db7, rb7, g, c); db_flush<4>(db8, rb8, g, c);
            }
            __syncthreads();
        }
    }
    __syncthreads();
    float* base = ws + ((size_t)blockIdx.x * 2 + ae) * PG;
    for (int i = tid; i < PG; i += NW * 64) base[i] = 0.f;                 // alignment pads
    __syncthreads();
    for (int l = INNER ? 1 : 0; l < (INNER ? 8 : NL); ++l) {       // INNER: layers 1 and 9 come from the GEMM path
        const int P = L.P[l], IN = L.IN[l], n = L.OUT[l] * IN;
        for (int e = tid; e < n; e += NW * 64) { const int o = e / IN; base[go.w[l] + e] = dwl[L.w[l] + o * P + (e - o * IN)]; }
        if (tid < L.OUT[l]) base[go.b[l] + tid] = dwl[L.b[l] + tid];
    }
}

}  // namespace sta
