// st_ae32.h -- the fused autoencoder FORWARD of the 16-bit configurations on 32-row groups (gfx950), round 4.
//
// ae_fwd_kernel (st_ae.h) walks 16-row groups on v_mfma_f32_16x16x16_{bf16,f16}: per group pair (both nets) 785 vector, 82 LDS and 78 matrix
// instructions, and the kernel is ISSUE-bound (SQ_ACTIVE_INST_ANY ~ 80 % of the SIMD time at two waves per SIMD, profiles/r03_*): its time is its
// instruction count.  Per row, the activation arithmetic (ELU 3 + conversion 0.5 instructions per element) cannot shrink, everything else can:
// here a wave owns 32 rows (bins 32 j .. 32 j + 31 of one window) on v_mfma_f32_32x32x16_*:
//     D[o][row] = sum_i W[o][i] H[i][row]      A = weights (lane (h, m): row m, eight k), B = activations (lane (h, n): column n, eight k)
//   * the result layout -- lane (h, n) holds features 8 q + 4 h + r (acc[4 q + r]) of row n -- packed eight at a time IS the B operand of the next
//     layer if its k-step s enumerates the features 16 s + {4 h + j (j < 4), 8 + 4 h + (j - 4)} in lane half h: the weight tiles are stored in that
//     order, so the activations never leave registers (the D-layout chain of st_ae.h carried over to the 32 x 32 shape);
//   * a weight fragment is ONE 16-byte LDS read for 32 outputs x 16 inputs x 32 rows (st_ae.h: one 8-byte read for 16 x 16 x 16): 46 fragment
//     reads per 32 rows and both nets against 2 x 70; 46 matrix instructions against 2 x 70 (the narrow layers waste half a 32-wide tile, the
//     matrix pipe has the room); the weight images are 23 tiles of 1 KB per net (48 KB for both, was 76 KB);
//   * per-group work (index arithmetic, knob handling, masks, waits) is paid once per 32 rows.
// Outputs, their layouts and the code h4 handed to the split backward ([net][16-row group][lane] float4 in the 16 x 16 D layout) are exactly those
// of ae_fwd_kernel; the arithmetic is the same (operands rounded to 16 bits, fp32 accumulation, fp32 bias / ELU / epilogue), sums re-associate.
// Row space: 17 groups of 32 virtual bins per window (544 >= FP = 528; the last group holds the Nyquist bin and the pad columns).
#pragma once
#include "st_ae.h"
#include "st_gemm.h"

namespace sta {

struct CL32 {      // tiles (1 KB each: 64 lanes x 8 values of 2 bytes) of one net, layer by layer: [m-tile][k-step]
    static constexpr int MT0 = 2, KS0 = 2, MT1 = 1, KS1 = 4, MT2 = 1, KS2 = 2, MT3 = 1, KS3 = 1, MT4 = 1, KS4 = 2, MT5 = 1, KS5 = 1, MT6 = 1, KS6 = 1,
                         MT7 = 2, KS7 = 2, MT8 = 1, KS8 = 4;
    static constexpr int T0 = 0, T1 = T0 + MT0 * KS0, T2 = T1 + MT1 * KS1, T3 = T2 + MT2 * KS2, T4 = T3 + MT3 * KS3, T5 = T4 + MT4 * KS4, T6 = T5 + MT5 * KS5,
                         T7 = T6 + MT6 * KS6, T8 = T7 + MT7 * KS7, NTILES = T8 + MT8 * KS8;      // 23
    static constexpr int BIAS = NTILES * 256;            // float offset of the biases behind the tiles (a tile = 256 floats of storage)
    static constexpr int B0 = BIAS, B1 = B0 + 64, B2 = B1 + 32, B3 = B2 + 32, B4 = B3 + 32, B5 = B4 + 32, B6 = B5 + 32, B7 = B6 + 32, B8 = B7 + 64, TOTAL = B8 + 32;
    // (biases padded to whole 32-feature m-tiles)
};
__host__ __device__ constexpr int ae32_lds_floats(int FP) { return 2 * CL32::TOTAL + FP; }      // two nets + the per-bin frequency weights

// feature (within a 16-feature k-step) that slot j of lane half h carries
__device__ __forceinline__ int slot_feat(const int h, const int j) { return j < 4 ? 4 * h + j : 8 + 4 * h + (j - 4); }

// Cooperative build of one net's tiles + biases in LDS from the packed parameter block (gather form: every thread fills whole 2-byte slots, zero
// padding included; all global loads of a thread are issued before the first store).
template <int NT, int BF>
__device__ __forceinline__ void ae32_load(float* lds, const float* __restrict__ ae, const AEOffsets& go, const int T, const int OT, const int K, const int tid)
{
    static_assert(NT == 512, "one tile = 512 values = one value per thread");
    const int out[NL] = {64, 32, 16, 16, 16, 16, 32, 64, OT};
    const int in[NL] = {T, 64, 32, 16, 16 + K, 16, 16, 32, 64};
    const int mt_[NL] = {CL32::MT0, CL32::MT1, CL32::MT2, CL32::MT3, CL32::MT4, CL32::MT5, CL32::MT6, CL32::MT7, CL32::MT8};
    const int ks_[NL] = {CL32::KS0, CL32::KS1, CL32::KS2, CL32::KS3, CL32::KS4, CL32::KS5, CL32::KS6, CL32::KS7, CL32::KS8};
    const int lane = tid >> 3, j = tid & 7, h = lane >> 5, m = lane & 31;
    const int sf = slot_feat(h, j);
    float v[CL32::NTILES];
    int t = 0;
#pragma unroll
    for (int l = 0; l < NL; ++l)
#pragma unroll
        for (int mt = 0; mt < mt_[l]; ++mt)
#pragma unroll
            for (int ks = 0; ks < ks_[l]; ++ks, ++t) {
                const int o = 32 * mt + m, i = 16 * ks + sf;
                const bool ok = o < out[l] && i < in[l];
                v[t] = ae[go.w[l] + (ok ? o * in[l] + i : 0)];
                v[t] = ok ? v[t] : 0.f;
            }
    float bv[NL];
#pragma unroll
    for (int l = 0; l < NL; ++l) bv[l] = ae[go.b[l] + (tid < out[l] ? tid : 0)];
    unsigned short* tl = reinterpret_cast<unsigned short*>(lds);
#pragma unroll
    for (t = 0; t < CL32::NTILES; ++t) tl[t * 512 + tid] = st_half_bits<BF>(v[t]);
    const int bo[NL] = {CL32::B0, CL32::B1, CL32::B2, CL32::B3, CL32::B4, CL32::B5, CL32::B6, CL32::B7, CL32::B8};
    const int bw[NL] = {64, 32, 32, 32, 32, 32, 32, 64, 32};
#pragma unroll
    for (int l = 0; l < NL; ++l) if (tid < bw[l]) lds[bo[l] + tid] = tid < out[l] ? bv[l] : 0.f;
}

template <int BF> struct frag32 { typedef stg::st_bf16x8 type; };
template <> struct frag32<2> { typedef stg::st_f16x8 type; };
template <int BF>
__device__ __forceinline__ f32x16 mfma32h(const typename frag32<BF>::type a, const typename frag32<BF>::type b, const f32x16 c)
{
    if constexpr (BF == 2) return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
// eight fp32 values (two D-layout quads) -> one B / A operand
template <int BF>
__device__ __forceinline__ typename frag32<BF>::type pack8(const f32x4 lo, const f32x4 hi)
{
    union { s16x4 s[2]; typename frag32<BF>::type f; } u;
    u.s[0] = pack_h4<BF>(lo); u.s[1] = pack_h4<BF>(hi);
    return u.f;
}
__device__ __forceinline__ f32x4 quad(const f32x16& a, const int q) { return (f32x4){a[4 * q], a[4 * q + 1], a[4 * q + 2], a[4 * q + 3]}; }
__device__ __forceinline__ void set_quad(f32x16& a, const int q, const f32x4 v) { a[4 * q] = v[0]; a[4 * q + 1] = v[1]; a[4 * q + 2] = v[2]; a[4 * q + 3] = v[3]; }

// One layer for the two nets: MT output tiles of 32 features from KS k-steps of 16 input features.  LIVEQ = live quads of an output tile (4: all 32
// features; 2: a 16-wide layer -- the upper half of the tile is never read).  in[net][ks]: packed B operands; out[net][2 mt + s]: packed ELU outputs.
template <int MT, int KS, int LIVEQ, int BF>
__device__ __forceinline__ void layer32(const float* const (&lw)[2], const int tile0, const int bias0, const typename frag32<BF>::type (&in)[2][KS],
                                        typename frag32<BF>::type (&out)[2][MT * (LIVEQ / 2)], const int lane, const int h)
{
    typedef typename frag32<BF>::type frag_t;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        f32x16 acc[2];
#pragma unroll
        for (int ch = 0; ch < 2; ++ch)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                set_quad(acc[ch], q, *reinterpret_cast<const f32x4*>(lw[ch] + bias0 + 32 * mt + 8 * q + 4 * h));      // biases are zero-padded to whole tiles: the dead quads of a 16-wide layer read zeros (a v_mov per register otherwise)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            frag_t w[2];
#pragma unroll
            for (int ch = 0; ch < 2; ++ch) w[ch] = *reinterpret_cast<const frag_t*>(reinterpret_cast<const char*>(lw[ch]) + (size_t)(tile0 + mt * KS + ks) * 1024 + lane * 16);
#pragma unroll
            for (int ch = 0; ch < 2; ++ch) acc[ch] = mfma32h<BF>(w[ch], in[ch][ks], acc[ch]);
        }
#pragma unroll
        for (int ch = 0; ch < 2; ++ch)
#pragma unroll
            for (int s = 0; s < LIVEQ / 2; ++s) out[ch][mt * (LIVEQ / 2) + s] = pack8<BF>(elu4(quad(acc[ch], 2 * s)), elu4(quad(acc[ch], 2 * s + 1)));
    }
}

// grid.x workgroups of 8 waves; a wave walks 32-row groups: id = b * 17 + j (bins 32 j ..).  Requires T <= 32, OT <= 16, K <= 16, FP <= 544.
template <int NW, int BF>
__global__ void __launch_bounds__(NW * 64)
ae_fwd32_kernel(const float* __restrict__ mag, const float* __restrict__ phs, const float* __restrict__ knobs,
                const float* __restrict__ ae_m, const float* __restrict__ ae_p, const AEOffsets go,
                float* __restrict__ mag_hat, float* __restrict__ phs_hat, float* __restrict__ AA, float* __restrict__ reg_partial,
                const int B, const int T, const int OT, const int F, const int K, const int KP, const float expfac,
                float* __restrict__ h4x, unsigned short* __restrict__ AA16, const int aa_ht)
{
    typedef typename frag32<BF>::type frag_t;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, n = lane & 31;
    const float* const lw[2] = {lds, lds + CL32::TOTAL};
    ae32_load<NW * 64, BF>(lds, ae_m, go, T, OT, K, tid);
    ae32_load<NW * 64, BF>(lds + CL32::TOTAL, ae_p, go, T, OT, K, tid);
    const int FP = KP / 2;
    float* const wtab = lds + 2 * CL32::TOTAL;
    for (int i = tid; i < FP; i += NW * 64) wtab[i] = i < F ? expf(expfac * (float)i) : 0.f;
    __syncthreads();

    constexpr int GPW = 17;
    const int ngroups = B * GPW;
    const int ng16 = B * (FP / 16);                     // 16-row groups of the h4 exchange buffer
    float reg = 0.f;
    int sfeat[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) sfeat[j] = slot_feat(h, j);
    // per-lane constants of the input burst: frame offsets t * F of the layer-1 operands and of the tails, knob indices, validity bits
    unsigned xoff[2][8], toff[2][4], koff[8], okx = 0u, okt = 0u, okk = 0u;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int j = 0; j < 8; ++j) { const int t = 16 * ks + sfeat[j]; xoff[ks][j] = ST_MUL24(t < T ? t : 0, F); okx |= (t < T ? 1u : 0u) << (8 * ks + j); }
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int r = 0; r < 4; ++r) { const int to = 8 * q + 4 * h + r; toff[q][r] = ST_MUL24(to < OT ? T - OT + to : 0, F); okt |= (to < OT ? 1u : 0u) << (4 * q + r); }
#pragma unroll
    for (int j = 0; j < 8; ++j) { koff[j] = (unsigned)(sfeat[j] < K ? sfeat[j] : 0); okk |= (sfeat[j] < K ? 1u : 0u) << j; }

    const GroupWalk gw = fwd_group_walk(ngroups, NW, wave);      // an XCD owns a contiguous eighth of the groups (st_ae.h): shared 128-byte lines stay in one L2
    for (int grp = gw.first; grp < gw.end; grp += gw.stride) {
        asm volatile("" ::: "memory");
        const int b = grp / GPW, jg = grp - b * GPW, f = 32 * jg + n;
        const bool fv = f < F;
        // ---- inputs: layer-1 operands (features t = 16 ks + slot), the skip / residual tails (t = T - OT + 8 q + 4 h + r) and the knobs, one burst
        float xin[2][2][8], tl[2][2][4], kn[8];
        const unsigned base = ST_MUL24(ST_MUL24(b, T), F) + (unsigned)(fv ? f : 0);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int j = 0; j < 8; ++j) { const unsigned o = base + xoff[ks][j]; xin[0][ks][j] = ldg32(mag, o); xin[1][ks][j] = ldg32(phs, o); }
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int r = 0; r < 4; ++r) { const unsigned o = base + toff[q][r]; tl[0][q][r] = ldg32(mag, o); tl[1][q][r] = ldg32(phs, o); }
        const unsigned kb = ST_MUL24(b, K);
#pragma unroll
        for (int j = 0; j < 8; ++j) kn[j] = ldg32(knobs, kb + koff[j]);
        frag_t x0[2][2], knf[2][1];
#pragma unroll
        for (int ch = 0; ch < 2; ++ch)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                f32x4 lo, hi;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    lo[j] = (fv && ((okx >> (8 * ks + j)) & 1u)) ? xin[ch][ks][j] : 0.f;
                    hi[j] = (fv && ((okx >> (8 * ks + 4 + j)) & 1u)) ? xin[ch][ks][4 + j] : 0.f;
                }
                x0[ch][ks] = pack8<BF>(lo, hi);
            }
        {
            f32x4 lo, hi;
#pragma unroll
            for (int j = 0; j < 4; ++j) { lo[j] = ((okk >> j) & 1u) ? kn[j] : 0.f; hi[j] = ((okk >> (4 + j)) & 1u) ? kn[4 + j] : 0.f; }
            knf[0][0] = pack8<BF>(lo, hi); knf[1][0] = knf[0][0];
        }
        // ---- the nine layers
        frag_t h1[2][4], h2[2][2], h3[2][1], h4p[2][1], h5[2][1], h6[2][1], h7[2][2], h8[2][4];
        layer32<CL32::MT0, CL32::KS0, 4, BF>(lw, CL32::T0, CL32::B0, x0, h1, lane, h);
        layer32<CL32::MT1, CL32::KS1, 4, BF>(lw, CL32::T1, CL32::B1, h1, h2, lane, h);
        layer32<CL32::MT2, CL32::KS2, 2, BF>(lw, CL32::T2, CL32::B2, h2, h3, lane, h);
        // layer 4 (the code): its fp32 ELU outputs also go to the exchange buffer of the split backward, in the 16 x 16 D layout of st_ae.h
        {
            f32x16 acc[2];
#pragma unroll
            for (int ch = 0; ch < 2; ++ch) {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    set_quad(acc[ch], q, *reinterpret_cast<const f32x4*>(lw[ch] + CL32::B3 + 8 * q + 4 * h));
                const frag_t w = *reinterpret_cast<const frag_t*>(reinterpret_cast<const char*>(lw[ch]) + (size_t)CL32::T3 * 1024 + lane * 16);
                acc[ch] = mfma32h<BF>(w, h3[ch][0], acc[ch]);
            }
            f32x4 e[2][2];
#pragma unroll
            for (int ch = 0; ch < 2; ++ch) { e[ch][0] = elu4(quad(acc[ch], 0)); e[ch][1] = elu4(quad(acc[ch], 1)); h4p[ch][0] = pack8<BF>(e[ch][0], e[ch][1]); }
            if (h4x) {
                const int g16 = 2 * jg + (n >> 4);                  // the 16-row group of this lane's row
                if (g16 < FP / 16) {
                    float4* hv = reinterpret_cast<float4*>(h4x);
                    const unsigned gi = (unsigned)(b * (FP / 16) + g16);      // 2 * ng16 * 64 float4 < 2^31 (the host checks the batch)
#pragma unroll
                    for (int ch = 0; ch < 2; ++ch)
#pragma unroll
                        for (int q = 0; q < 2; ++q)      // quad q of lane half h = features 8 q + 4 h .. + 3 = lane group g = 2 q + h of the 16 x 16 layout
                            hv[((unsigned)ch * (unsigned)ng16 + gi) * 64u + (unsigned)((2 * q + h) * 16 + (n & 15))] = make_float4(e[ch][q][0], e[ch][q][1], e[ch][q][2], e[ch][q][3]);
                }
            }
        }
        if (!mag_hat) continue;                                   // h4-only pass (wave-uniform)
        {   // layer 5: [h4 ; knobs]
            frag_t in5[2][2];
#pragma unroll
            for (int ch = 0; ch < 2; ++ch) { in5[ch][0] = h4p[ch][0]; in5[ch][1] = knf[ch][0]; }
            layer32<CL32::MT4, CL32::KS4, 2, BF>(lw, CL32::T4, CL32::B4, in5, h5, lane, h);
        }
        layer32<CL32::MT5, CL32::KS5, 2, BF>(lw, CL32::T5, CL32::B5, h5, h6, lane, h);
        layer32<CL32::MT6, CL32::KS6, 4, BF>(lw, CL32::T6, CL32::B6, h6, h7, lane, h);
        layer32<CL32::MT7, CL32::KS7, 4, BF>(lw, CL32::T7, CL32::B7, h7, h8, lane, h);
        // layer 9 + epilogue (nn_proc.py:115,117,322-326): the fp32 ELU outputs of the OT <= 16 output frames
        f32x16 acc9[2];
#pragma unroll
        for (int ch = 0; ch < 2; ++ch) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
                set_quad(acc9[ch], q, *reinterpret_cast<const f32x4*>(lw[ch] + CL32::B8 + 8 * q + 4 * h));
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const frag_t w = *reinterpret_cast<const frag_t*>(reinterpret_cast<const char*>(lw[ch]) + (size_t)(CL32::T8 + ks) * 1024 + lane * 16);
                acc9[ch] = mfma32h<BF>(w, h8[ch][ks], acc9[ch]);
            }
        }
        const float wf = fv ? wtab[f < FP ? f : 0] : 0.f;
        const unsigned boF = ST_MUL24(ST_MUL24(b, OT), F) + (unsigned)f, boK = ST_MUL24(ST_MUL24(b, OT), KP) + (unsigned)f;
        const int ht = BF ? BF : aa_ht;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const f32x4 em = elu4(quad(acc9[0], q)), ep = elu4(quad(acc9[1], q));
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int to = 8 * q + 4 * h + r;
                if (((okt >> (4 * q + r)) & 1u) && f < FP) {
                    float mh = 0.f, ph = 0.f, sn = 0.f, cs = 1.f;
                    if (fv) {
                        mh = em[r] * tl[0][q][r];                      // 'sf' skip-filter
                        ph = ep[r] + tl[1][q][r];                      // phase residual
                        st_sincos(ph, sn, cs);
                        stg32(mag_hat, boF + ST_MUL24(to, F), mh);
                        stg32(phs_hat, boF + ST_MUL24(to, F), ph);
                        reg += fabsf(mh * wf);
                    }
                    if (AA16) {
                        AA16[boK + ST_MUL24(to, KP)] = st_to_h16(mh * cs, ht);
                        AA16[boK + ST_MUL24(to, KP) + (unsigned)FP] = st_to_h16(mh * sn, ht);
                    } else {
                        stg32(AA, boK + ST_MUL24(to, KP), mh * cs);
                        stg32(AA, boK + ST_MUL24(to, KP) + (unsigned)FP, mh * sn);
                    }
                }
            }
        }
    }
    if (reg_partial) {
        reg = wave_sum(reg);
        if (lane == 0) reg_partial[blockIdx.x * NW + wave] = reg;
    }
}

}  // namespace sta
