// st_ae_split.h -- the autoencoder backward as TWO kernels at two waves per SIMD (fused geometries: T <= 32, OT <= 16).
//
// ae_bwd_kernel (st_ae.h) keeps the 36 weight-gradient tiles of a net -- 144 accumulator registers -- beside the 68 activation
// registers of the recomputed forward chain, fills the 512-register budget and so runs ONE wave per SIMD; its timing-only
// ablation (tools/ae_ablate.sh) shows every component fully exposed: MFMA time + instruction-issue / latency skeleton ADD UP,
// nothing overlaps, because a single in-order wave has nothing else to issue while it waits.  Here the net is cut at its 16-wide
// code h4 (the output of fnn_enc4, nn_proc.py:89) into
//     PART 1  "decoder half":  h4 (saved by the forward kernel, 16 floats per row) -> recompute layers 5..9 -> d out -> backward
//                              9..5 -> d a4; 17 weight-gradient tiles (68 registers), 40 activation registers
//     PART 2  "encoder half":  input rows -> recompute layers 1..3 -> backward 4..1 from d a4 -> d input rows (+ skip / residual
//                              tails, handed over through the output buffer); 19 tiles (76 registers), 36 activation registers
// each of which fits 256 registers, i.e. TWO waves per SIMD (8-wave workgroups, one per CU): the partner wave's MFMAs and VALU
// fill the other's stalls.  No MFMA is executed twice (the split point costs 2 x 17 MB of h4 / d a4 traffic instead).  Per-wave
// arithmetic, summation orders inside a wave and the fixed-order reduction of the per-wave gradient images are as in
// ae_bwd_kernel, so the result is run-to-run bit-identical as before (the order in which rows reach a given accumulator changes,
// hence results differ from the single-kernel form by fp32 reassociation only).
#pragma once
#include "st_ae.h"

namespace sta {

// Compact LDS layouts: forward images, biases and dgrad images of the layers of one half only.
struct AETab { int ao[NL], bo[NL], gi[NL]; };
template <int PART> struct CP;
template <> struct CP<1> {                 // layers l = 4..8  (fnn_addknobs, fnn_dec4, fnn_dec3, fnn_dec2, fnn_dec)
    static constexpr int L0 = 4, L1 = 9;
    static constexpr int A4 = 0, A5 = A4 + CL::O4 * CL::I4, A6 = A5 + CL::O5 * CL::I5, A7 = A6 + CL::O6 * CL::I6, A8 = A7 + CL::O7 * CL::I7, AEND = A8 + CL::O8 * CL::I8;
    static constexpr int B4 = AEND, B5 = B4 + CL::O4, B6 = B5 + CL::O5, B7 = B6 + CL::O6, B8 = B7 + CL::O7, FWD_END = B8 + CL::O8;
    static constexpr int G4 = FWD_END, G5 = G4 + CL::O4 * CL::I4, G6 = G5 + CL::O5 * CL::I5, G7 = G6 + CL::O6 * CL::I6, G8 = G7 + CL::O7 * CL::I7, TOTAL = G8 + CL::O8 * CL::I8;
    static constexpr int SCR = 16 * SP + 2 * 4 * 320;          // per wave: Y rows + two 4-tile transpose scratches
    __device__ static AETab tab() { return AETab{{0, 0, 0, 0, A4, A5, A6, A7, A8}, {0, 0, 0, 0, B4, B5, B6, B7, B8}, {0, 0, 0, 0, G4, G5, G6, G7, G8}}; }
};
template <> struct CP<2> {                 // layers l = 0..3  (fnn_enc, fnn_enc2, fnn_enc3, fnn_enc4)
    static constexpr int L0 = 0, L1 = 4;
    static constexpr int A0 = 0, A1 = A0 + CL::O0 * CL::I0, A2 = A1 + CL::O1 * CL::I1, A3 = A2 + CL::O2 * CL::I2, AEND = A3 + CL::O3 * CL::I3;
    static constexpr int B0 = AEND, B1 = B0 + CL::O0, B2 = B1 + CL::O1, B3 = B2 + CL::O2, FWD_END = B3 + CL::O3;
    static constexpr int G0 = FWD_END, G1 = G0 + CL::O0 * CL::I0, G2 = G1 + CL::O1 * CL::I1, G3 = G2 + CL::O2 * CL::I2, TOTAL = G3 + CL::O3 * CL::I3;
    static constexpr int SCR = 32 * SP + 2 * 4 * 320;          // per wave: V rows + two 4-tile transpose scratches
    __device__ static AETab tab() { return AETab{{A0, A1, A2, A3, 0, 0, 0, 0, 0}, {B0, B1, B2, B3, 0, 0, 0, 0, 0}, {G0, G1, G2, G3, 0, 0, 0, 0, 0}}; }
};
// dynamic LDS of a split kernel (floats): images + per-wave scratch during the loop; four gradient images (waves w and w + 4 share one) at the end
template <int PART> constexpr int ae_split_lds_floats(int nw)
{
    return (CP<PART>::TOTAL + nw * CP<PART>::SCR) > 4 * CP<PART>::FWD_END ? (CP<PART>::TOTAL + nw * CP<PART>::SCR) : 4 * CP<PART>::FWD_END;
}

// Parameters of layers [l0, l1) into the compact images of `tab` (cf. ae_load_lds).
template <int NT, int BF = 0>
__device__ inline void ae_load_lds_tab(float* lds, const int total, const AETab tab, const float* __restrict__ ae, const AEOffsets& go,
                                       const int T, const int OT, const int K, const int tid, const int l0, const int l1)
{
    AEParamRegs<NT> r;
    ae_params_issue<NT>(r, ae, go, T, OT, K, tid, l0, l1);
    for (int e = tid; e < total; e += NT) lds[e] = 0.f;
    __syncthreads();
    const int out[NL] = {64, 32, 16, 16, 16, 16, 32, 64, OT};
    const int in[NL] = {T, 64, 32, 16, 16 + K, 16, 16, 32, 64};
    const int outp[NL] = {CL::O0, CL::O1, CL::O2, CL::O3, CL::O4, CL::O5, CL::O6, CL::O7, CL::O8};
    const int inp[NL] = {CL::I0, CL::I1, CL::I2, CL::I3, CL::I4, CL::I5, CL::I6, CL::I7, CL::I8};
#pragma unroll
    for (int l = 0; l < NL; ++l) {
        const bool on = l >= l0 && l < l1;
        const int IN = in[l], n = on ? out[l] * IN : 0, OP = outp[l], IP = inp[l];
#pragma unroll
        for (int u = 0; u < (ae_max_elems(l) + NT - 1) / NT; ++u) {
            const int e = tid + u * NT;
            if (e < n) {
                const int o = e / IN, i = e - o * IN;
                if constexpr (BF) {                       // 16-bit images, packed into the first half of each region (ae_params_scatter)
                    const unsigned short hb = st_half_bits<BF>(r.v[l][u]);
                    reinterpret_cast<unsigned short*>(lds + tab.ao[l])[(((i >> 2) * OP + o) << 2) + (i & 3)] = hb;
                    reinterpret_cast<unsigned short*>(lds + tab.gi[l])[(((o >> 2) * IP + i) << 2) + (o & 3)] = hb;
                } else {
                lds[tab.ao[l] + (((i >> 2) * OP + o) << 2) + (i & 3)] = r.v[l][u];
                lds[tab.gi[l] + (((o >> 2) * IP + i) << 2) + (o & 3)] = r.v[l][u];
                }
            }
        }
        if (on && tid < out[l]) lds[tab.bo[l] + tid] = r.bv[l];
    }
}

// Scheduling recipe of one backward stage (see ae_bwd_kernel): data-gradient MFMAs first, then one weight-gradient MFMA per pair of
// VALU / LDS instructions.
#define ST_PIPE2(N_) do { \
        __builtin_amdgcn_sched_group_barrier(0x008, (N_) + 2, 0); \
        _Pragma("unroll") for (int p_ = 0; p_ < (N_) - 2; ++p_) { \
            __builtin_amdgcn_sched_group_barrier(0x002, 2, 0); __builtin_amdgcn_sched_group_barrier(0x300, 1, 0); \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); } \
        ST_FENCE(); } while (0)
#define ST_BWD_STAGE2(O_, I_, FD_, DA_, DAT_, HP_, HTP_, DAP_, DATP_, RW_, RB_, NEXT_) \
        to_Th<I_, BF>(XH, HP_, HTP_, g, c); ST_FENCE(); \
        dgradD_fr<O_, I_, BF>(FD_, DA_, DAP_); mul_elu_grad<I_>(DAP_, HP_); \
        to_T<I_>(XD, DAP_, DATP_, g, c); NEXT_; \
        wgrad_regh<O_, I_, BF>(RW_, RB_, DAT_, HTP_); \
        ST_PIPE2(BF ? O_ * I_ : O_ * I_ * 4);

// grid (x: workgroups, y: net 0 = magnitude 'sf' / 1 = phase); NW = 8 waves = two per SIMD.  h4x / da4x: [net][group][lane] float4
// exchange buffers (h4 in D layout as the forward kernel left it; d a4 likewise).  GM: an upstream d/d mag_hat arrives.
template <int NW, int PART, int BF = 0, bool GM = false>
__global__ void __launch_bounds__(NW * 64)
ae_bwd_part_kernel(const float* __restrict__ mag, const float* __restrict__ phs, const float* __restrict__ knobs,
                   const float* __restrict__ ae_m, const float* __restrict__ ae_p, const AEOffsets go, const int PG,
                   const float* __restrict__ mag_hat, const float* __restrict__ phs_hat, const float* __restrict__ dAA,
                   const float* __restrict__ g_mag_hat, const float reg_coef, const float expfac,
                   float* __restrict__ dmag, float* __restrict__ dphs, float* __restrict__ ws,
                   const float* __restrict__ h4x, float* __restrict__ da4x,
                   const int B, const int T, const int OT, const int F, const int K, const int KP,
                   const int to_lo, const int to_hi, const int nslab, const size_t slab)
{
    static_assert(NW == 8, "the end-of-kernel reduction pairs waves w and w + 4");
    using P = CP<PART>;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int ae = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, c = lane & 15;
    float* lw = lds;
    float* scr = lds + P::TOTAL + wave * P::SCR;
    float* S0 = scr;                                          // PART 1: Y rows (d a9 transposed); PART 2: V rows (input rows transposed)
    float* XH = scr + (PART == 1 ? 16 : 32) * SP;
    float* XD = XH + 4 * 320;
    ae_load_lds_tab<NW * 64, BF>(lw, P::TOTAL, P::tab(), ae ? ae_p : ae_m, go, T, OT, K, tid, P::L0, P::L1);
    __syncthreads();

    const float* vin = ae ? phs : mag;
    float* dvout = ae ? dphs : dmag;
    const int FP = KP / 2, gpw = FP / 16;
    const int ngroups = B * gpw;
    const int gstride = gridDim.x * NW;
    const float4* h4v = reinterpret_cast<const float4*>(h4x) + (size_t)ae * ngroups * 64;
    float4* da4v = reinterpret_cast<float4*>(da4x) + (size_t)ae * ngroups * 64;

#define ST_ZT(x, A, Bq) { _Pragma("unroll") for (int a_ = 0; a_ < A; ++a_) _Pragma("unroll") for (int b_ = 0; b_ < Bq; ++b_) x[a_][b_] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
#define ST_ZB(x, A) { _Pragma("unroll") for (int a_ = 0; a_ < A; ++a_) x[a_] = 0.f; }

    int grp = blockIdx.x * NW + wave;
    if constexpr (PART == 1) {
        // ================================================================================= decoder half: layers 5..9
        f32x4 rW5[1][2], rW6[1][1], rW7[2][1], rW8[4][2], rW9[1][4];
        float rb5[1], rb6[1], rb7[2], rb8[4], rb9[1];
        ST_ZT(rW5, 1, 2) ST_ZT(rW6, 1, 1) ST_ZT(rW7, 2, 1) ST_ZT(rW8, 4, 2) ST_ZT(rW9, 1, 4)
        ST_ZB(rb5, 1) ST_ZB(rb6, 1) ST_ZB(rb7, 2) ST_ZB(rb8, 4) ST_ZB(rb9, 1)
        // per-group inputs prefetched one group ahead: h4 (one 16-byte load) and the knob values (D-layout tile + T-layout lane)
        auto load_in = [&](int gq, f32x4& h4q, f32x4& k4, float& kT) {
            const float4 v = h4v[(size_t)gq * 64 + lane];
            h4q = (f32x4){v.x, v.y, v.z, v.w};
            const unsigned kb = ST_MUL24(gq / gpw, K);
#pragma unroll
            for (int r = 0; r < 4; ++r) { const int kidx = 4 * g + r; k4[r] = ldg32(knobs, kb + (unsigned)(kidx < K ? kidx : 0)); }
            kT = ldg32(knobs, kb + (unsigned)(c < K ? c : 0));
        };
        auto mask_kn = [&](f32x4& d4, float& dT) {
#pragma unroll
            for (int r = 0; r < 4; ++r) d4[r] = (4 * g + r) < K ? d4[r] : 0.f;
            dT = c < K ? dT : 0.f;
        };
        f32x4 h4c = (f32x4){0.f, 0.f, 0.f, 0.f}, kn = h4c; float knT = 0.f;
        if (grp < ngroups) { load_in(grp, h4c, kn, knT); mask_kn(kn, knT); }
        for (; grp < ngroups; grp += gstride) {
            asm volatile("" ::: "memory");
            const int b = grp / gpw, f = (grp - b * gpw) * 16 + c;
            const bool fv = f < F;
            const int fq = fv ? f : 0;
            // ---- d-out inputs (D layout: t' = 4g + r): raw loads, sums and masks formed later (see ae_bwd_kernel)
            float q_x[4][3], q_y[4][3], q_ph[4], q_mh[4], q_mt[4], q_gm[4];
            {
                const size_t o1 = nslab > 1 ? slab : 0, o2 = nslab > 2 ? 2 * slab : 0;
                const float* const dA0 = dAA, * const dA1 = dAA + o1, * const dA2 = dAA + o2;
                const float* const dB0 = dA0 + FP, * const dB1 = dA1 + FP, * const dB2 = dA2 + FP;
                const unsigned bOT = ST_MUL24(b, OT), btF = ST_MUL24(ST_MUL24(b, T), F) + (unsigned)fq;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int to = 4 * g + r;
                    const bool ok = fv && to < OT;
                    const bool lv = ok && to >= to_lo && to <= to_hi;
                    const unsigned ro = bOT + (unsigned)(ok ? to : 0);
                    const unsigned p0 = ST_MUL24(lv ? ro : 0u, KP) + (unsigned)fq;
                    const unsigned pF = ST_MUL24(ro, F) + (unsigned)fq;
                    q_x[r][0] = ldg32(dA0, p0); q_x[r][1] = ldg32(dA1, p0); q_x[r][2] = ldg32(dA2, p0);
                    q_y[r][0] = ldg32(dB0, p0); q_y[r][1] = ldg32(dB1, p0); q_y[r][2] = ldg32(dB2, p0);
                    q_ph[r] = ldg32(phs_hat, pF); q_mh[r] = ldg32(mag_hat, pF);
                    if constexpr (GM) q_gm[r] = ldg32(g_mag_hat, pF); else q_gm[r] = 0.f;
                    q_mt[r] = ldg32(vin, btF + ST_MUL24(ok ? T - OT + to : 0, F));
                }
            }
            f32x4 h4n, knn; float knTn;
            const int gnext = grp + gstride < ngroups ? grp + gstride : grp;
            load_in(gnext, h4n, knn, knTn);
            // ---- forward recompute, layers 5..9 (rolling fragment prefetch)
            f32x4 h4[1] = {h4c}, h5[1], h6[1], h7[2], h8[4], e9[1];
            f32x4 fr5[1 * 2]; frags_fwd<1, 2, CL::O4, BF>(fr5, lw + P::A4, g, c);
            f32x4 fr6[1 * 1]; frags_fwd<1, 1, CL::O5, BF>(fr6, lw + P::A5, g, c); ST_FENCE();
            { const f32x4 hk[2] = {h4[0], kn}; fwdD_fr<1, 2, BF>(fr5, lw + P::B4, hk, h5, g); }
            f32x4 fr7[2 * 1]; frags_fwd<2, 1, CL::O6, BF>(fr7, lw + P::A6, g, c); ST_FENCE(); fwdD_fr<1, 1, BF>(fr6, lw + P::B5, h5, h6, g);
            f32x4 fr8[4 * 2]; frags_fwd<4, 2, CL::O7, BF>(fr8, lw + P::A7, g, c); ST_FENCE(); fwdD_fr<2, 1, BF>(fr7, lw + P::B6, h6, h7, g);
            f32x4 fr9[1 * 4]; frags_fwd<1, 4, CL::O8, BF>(fr9, lw + P::A8, g, c); ST_FENCE(); fwdD_fr<4, 2, BF>(fr8, lw + P::B7, h7, h8, g);
            // ---- d out, part A under the layer-9 MFMAs (nn_proc.py:322-326 backward + the L1 term of loss_functions.py:36)
            f32x4 da9[1];
            float dxA[4], mtA[4];
            ST_FENCE();
            {
                const float wf = fv ? expf(expfac * (float)f) : 0.f;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int to = 4 * g + r;
                    const bool ok = fv && to < OT;
                    const bool lv = ok && to >= to_lo && to <= to_hi;
                    const float gre = lv ? q_x[r][0] + (nslab > 1 ? q_x[r][1] : 0.f) + (nslab > 2 ? q_x[r][2] : 0.f) : 0.f;
                    const float gim = lv ? q_y[r][0] + (nslab > 1 ? q_y[r][1] : 0.f) + (nslab > 2 ? q_y[r][2] : 0.f) : 0.f;
                    const float ph = q_ph[r], mh = q_mh[r];
                    asm volatile("" :: "v"(q_mt[r]));
                    float sn, cs; st_sincos(ph, sn, cs);
                    const float sg = mh > 0.f ? 1.f : (mh < 0.f ? -1.f : 0.f);
                    const float dmh = gre * cs + gim * sn + reg_coef * sg * wf + (GM ? q_gm[r] : 0.f);
                    const float dph = mh * (gim * cs - gre * sn);
                    dxA[r] = ok ? (ae == 0 ? dmh : dph) : 0.f;
                    mtA[r] = ae == 0 ? q_mt[r] : 1.f;
                }
            }
            fwdD_fr<1, 4, BF>(fr9, lw + P::B8, h8, e9, g);
            if constexpr (BF == 0) {
#pragma unroll
                for (int p_ = 0; p_ < 16; ++p_) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, 7, 0); }
            }
            ST_FENCE();
            // ---- d out, part B; the skip / residual tails go to the output rows t >= T - OT, where the encoder half picks them up
            float tails[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int to = 4 * g + r;
                const float eg = elu_grad_from_out(e9[0][r]);
                const float e1 = ae == 0 ? e9[0][r] : 1.f;
                const float d9 = dxA[r] * mtA[r] * eg;
                tails[r] = dxA[r] * e1;
                da9[0][r] = d9;
                S0[to * SP + c] = d9;                          // [feature t'][row c] -> read back transposed below
            }
            // ---- backward 9..5
            typedef typename TTy<BF>::type tt_t;
            tt_t hT8[4]; f32x4 da8[4], daT8[4], fd8[2 * 4];
            {
                f32x4 daT9[1], fd9[4 * 1];
                frags_dgrad<1, 4, CL::I8, BF>(fd9, lw + P::G8, g, c);
                daT9[0] = *reinterpret_cast<const f32x4*>(S0 + c * SP + 4 * g);
                ST_BWD_STAGE2(1, 4, fd9, da9, daT9, h8, hT8, da8, daT8, rW9, rb9, (frags_dgrad<4, 2, CL::I7, BF>(fd8, lw + P::G7, g, c)))
            }
            tt_t hT7[2]; f32x4 da7[2], daT7[2], fd7[1 * 2];
            ST_BWD_STAGE2(4, 2, fd8, da8, daT8, h7, hT7, da7, daT7, rW8, rb8, (frags_dgrad<2, 1, CL::I6, BF>(fd7, lw + P::G6, g, c)))
            tt_t hT6[1]; f32x4 da6[1], daT6[1], fd6[1];
            ST_BWD_STAGE2(2, 1, fd7, da7, daT7, h6, hT6, da6, daT6, rW7, rb7, (frags_dgrad<1, 1, CL::I5, BF>(fd6, lw + P::G5, g, c)))
            tt_t hT5[1]; f32x4 da5[1], daT5[1], fd5[1];
            ST_BWD_STAGE2(1, 1, fd6, da6, daT6, h5, hT5, da5, daT5, rW6, rb6, (frags_dgrad<1, 1, CL::I4, BF>(fd5, lw + P::G4, g, c)))
            // layer 5 ([h4 ; knobs] -> 16): weight gradient over both input tiles, data gradient to h4 only -> d a4 leaves the kernel
            tt_t hT4[1], hT4k[2]; f32x4 da4[1];
            {
                to_Th<1, BF>(XH, h4, hT4, g, c); ST_FENCE();
                dgradD_fr<1, 1, BF>(fd5, da5, da4); mul_elu_grad<1>(da4, h4);
                hT4k[0] = hT4[0];
                hT4k[1] = tt_splat<BF>(knT);
                wgrad_regh<1, 2, BF>(rW5, rb5, daT5, hT4k);
            }
            {
                float d0 = da4[0][0], d1 = da4[0][1], d2 = da4[0][2], d3 = da4[0][3];
                asm volatile("" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3));          // accumulators materialised in the block of their MFMAs
                da4v[(size_t)grp * 64 + lane] = make_float4(d0, d1, d2, d3);
                const unsigned t0 = ST_MUL24(ST_MUL24(b, T), F) + (unsigned)f;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int to = 4 * g + r;
                    if (fv && to < OT) stg32(dvout, t0 + ST_MUL24(T - OT + to, F), tails[r]);
                }
            }
            mask_kn(knn, knTn); kn = knn; knT = knTn; h4c = h4n;
        }
        // ---- workgroup partial gradients of layers 5..9: waves 0..3 store their images, waves 4..7 add theirs, one pass sums four
        __syncthreads();
        float* dwl = lds + (wave & 3) * P::FWD_END;
        if (wave < 4) {
            dw_flush<1, 2, CL::I4, true>(dwl + P::A4, rW5, g, c); dw_flush<1, 1, CL::I5, true>(dwl + P::A5, rW6, g, c); dw_flush<2, 1, CL::I6, true>(dwl + P::A6, rW7, g, c);
            dw_flush<4, 2, CL::I7, true>(dwl + P::A7, rW8, g, c); dw_flush<1, 4, CL::I8, true>(dwl + P::A8, rW9, g, c);
            db_flush<1, true>(dwl + P::B4, rb5, g, c); db_flush<1, true>(dwl + P::B5, rb6, g, c); db_flush<2, true>(dwl + P::B6, rb7, g, c);
            db_flush<4, true>(dwl + P::B7, rb8, g, c); db_flush<1, true>(dwl + P::B8, rb9, g, c);
        }
        __syncthreads();
        if (wave >= 4) {
            dw_flush<1, 2, CL::I4, false>(dwl + P::A4, rW5, g, c); dw_flush<1, 1, CL::I5, false>(dwl + P::A5, rW6, g, c); dw_flush<2, 1, CL::I6, false>(dwl + P::A6, rW7, g, c);
            dw_flush<4, 2, CL::I7, false>(dwl + P::A7, rW8, g, c); dw_flush<1, 4, CL::I8, false>(dwl + P::A8, rW9, g, c);
            db_flush<1, false>(dwl + P::B4, rb5, g, c); db_flush<1, false>(dwl + P::B5, rb6, g, c); db_flush<2, false>(dwl + P::B6, rb7, g, c);
            db_flush<4, false>(dwl + P::B7, rb8, g, c); db_flush<1, false>(dwl + P::B8, rb9, g, c);
        }
        __syncthreads();
    } else {
        // ================================================================================= encoder half: layers 1..4
        f32x4 rW1[4][2], rW2[2][4], rW3[1][2], rW4[1][1];
        float rb1[4], rb2[2], rb3[1], rb4[1];
        ST_ZT(rW1, 4, 2) ST_ZT(rW2, 2, 4) ST_ZT(rW3, 1, 2) ST_ZT(rW4, 1, 1)
        ST_ZB(rb1, 4) ST_ZB(rb2, 2) ST_ZB(rb3, 1) ST_ZB(rb4, 1)
        // per-group inputs, prefetched one group ahead: the input rows (D layout: t = 16 it + 4g + r) and d a4 (one 16-byte load); the
        // skip / residual tails the decoder half left in the output rows t >= T - OT are fetched under the layer-1 MFMAs of the same
        // group (eight fewer registers carried through the iteration: the kernel sits at the 256-register line)
        struct EncIn { f32x4 v[2]; f32x4 d4; };
        auto load_in = [&](int gq, EncIn& in) {
            const int bq = gq / gpw, fq = (gq - bq * gpw) * 16 + c;
            const unsigned base = ST_MUL24(ST_MUL24(bq, T), F) + (unsigned)(fq < F ? fq : 0);
#pragma unroll
            for (int it = 0; it < 2; ++it)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int t = 16 * it + 4 * g + r;
                    const unsigned o = base + ST_MUL24(t < T ? t : 0, F);
                    in.v[it][r] = ldg32(vin, o);
                }
            const float4 v = reinterpret_cast<const float4*>(da4x)[((size_t)ae * ngroups + gq) * 64 + lane];
            in.d4 = (f32x4){v.x, v.y, v.z, v.w};
        };
        auto mask_in = [&](int gq, EncIn& in) {
            const int bq = gq / gpw, fq = (gq - bq * gpw) * 16 + c;
            const bool ok0 = fq < F;
#pragma unroll
            for (int it = 0; it < 2; ++it)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int t = 16 * it + 4 * g + r;
                    in.v[it][r] = (ok0 && t < T) ? in.v[it][r] : 0.f;
                }
        };
        EncIn cur;
        if (grp < ngroups) { load_in(grp, cur); mask_in(grp, cur); }
        for (; grp < ngroups; grp += gstride) {
            asm volatile("" ::: "memory");
            const int b = grp / gpw, f = (grp - b * gpw) * 16 + c;
            const bool fv = f < F;
            EncIn nxt;
            const int gnext = grp + gstride < ngroups ? grp + gstride : grp;
            load_in(gnext, nxt);
            // ---- forward recompute, layers 1..3
            f32x4 h1[4], h2[2], h3[1];
            f32x4 fr1[4 * 2]; frags_fwd<4, 2, CL::O0, BF>(fr1, lw + P::A0, g, c);
            f32x4 fr2[2 * 4]; frags_fwd<2, 4, CL::O1, BF>(fr2, lw + P::A1, g, c);
            typedef typename TTy<BF>::type tt_t;
            tt_t vT[2];
            if constexpr (BF) to_Th<2, BF>(S0, cur.v, vT, g, c);          // 16-bit layers: the input rows transposed in 16 bits right away (one write + one transpose read per tile)
            else {
#pragma unroll
            for (int it = 0; it < 2; ++it)
#pragma unroll
                for (int r = 0; r < 4; ++r) S0[(16 * it + 4 * g + r) * SP + c] = cur.v[it][r];   // [feature t][row c]: read back transposed for the layer-1 wgrad
            }
            ST_FENCE();
            fwdD_fr<4, 2, BF>(fr1, lw + P::B0, cur.v, h1, g);
            f32x4 fr3[1 * 2]; frags_fwd<1, 2, CL::O2, BF>(fr3, lw + P::A2, g, c); ST_FENCE(); fwdD_fr<2, 4, BF>(fr2, lw + P::B1, h1, h2, g);
            f32x4 fd4[1]; frags_dgrad<1, 1, CL::I3, BF>(fd4, lw + P::G3, g, c);
            f32x4 da4[1] = {cur.d4}, daT4[1];
            to_T<1>(XD, da4, daT4, g, c);
            ST_FENCE(); fwdD_fr<1, 2, BF>(fr3, lw + P::B2, h2, h3, g);
            // ---- backward 4..1
            tt_t hT3[1]; f32x4 da3[1], daT3[1], fd3[2 * 1];
            ST_BWD_STAGE2(1, 1, fd4, da4, daT4, h3, hT3, da3, daT3, rW4, rb4, (frags_dgrad<1, 2, CL::I2, BF>(fd3, lw + P::G2, g, c)))
            tt_t hT2[2]; f32x4 da2[2], daT2[2], fd2[4 * 2];
            ST_BWD_STAGE2(1, 2, fd3, da3, daT3, h2, hT2, da2, daT2, rW3, rb3, (frags_dgrad<2, 4, CL::I1, BF>(fd2, lw + P::G1, g, c)))
            tt_t hT1[4]; f32x4 da1[4], daT1[4], fd1[2 * 4];
            ST_BWD_STAGE2(2, 4, fd2, da2, daT2, h1, hT1, da1, daT1, rW2, rb2, (frags_dgrad<4, 2, CL::I0, BF>(fd1, lw + P::G0, g, c)))
            f32x4 dv[2];
            if constexpr (BF == 0) {
#pragma unroll
                for (int it = 0; it < 2; ++it) vT[it] = *reinterpret_cast<const f32x4*>(S0 + (16 * it + c) * SP + 4 * g);
            }
            float tl[2][4];
            {
                const unsigned tb = ST_MUL24(ST_MUL24(b, T), F) + (unsigned)(fv ? f : 0);
#pragma unroll
                for (int it = 0; it < 2; ++it)
#pragma unroll
                    for (int r = 0; r < 4; ++r) { const int t = 16 * it + 4 * g + r; tl[it][r] = ldg32(dvout, tb + ST_MUL24((t >= T - OT && t < T) ? t : T - 1, F)); }
            }
            ST_FENCE();
            dgradD_fr<4, 2, BF>(fd1, da1, dv);
            wgrad_regh<4, 2, BF>(rW1, rb1, daT1, vT);
            // ---- d input rows (+ tails)
            float dvs[2][4];
#pragma unroll
            for (int it = 0; it < 2; ++it)
#pragma unroll
                for (int r = 0; r < 4; ++r) { dvs[it][r] = dv[it][r]; asm volatile("" : "+v"(dvs[it][r])); }
            const unsigned dv0 = ST_MUL24(ST_MUL24(b, T), F) + (unsigned)f;
#pragma unroll
            for (int it = 0; it < 2; ++it)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int t = 16 * it + 4 * g + r;
                    if (fv && t < T) stg32(dvout, dv0 + ST_MUL24(t, F), dvs[it][r] + ((t >= T - OT) ? tl[it][r] : 0.f));
                }
            mask_in(gnext, nxt); cur = nxt;
        }
        __syncthreads();
        float* dwl = lds + (wave & 3) * P::FWD_END;
        if (wave < 4) {
            dw_flush<4, 2, CL::I0, true>(dwl + P::A0, rW1, g, c); dw_flush<2, 4, CL::I1, true>(dwl + P::A1, rW2, g, c);
            dw_flush<1, 2, CL::I2, true>(dwl + P::A2, rW3, g, c); dw_flush<1, 1, CL::I3, true>(dwl + P::A3, rW4, g, c);
            db_flush<4, true>(dwl + P::B0, rb1, g, c); db_flush<2, true>(dwl + P::B1, rb2, g, c); db_flush<1, true>(dwl + P::B2, rb3, g, c); db_flush<1, true>(dwl + P::B3, rb4, g, c);
        }
        __syncthreads();
        if (wave >= 4) {
            dw_flush<4, 2, CL::I0, false>(dwl + P::A0, rW1, g, c); dw_flush<2, 4, CL::I1, false>(dwl + P::A1, rW2, g, c);
            dw_flush<1, 2, CL::I2, false>(dwl + P::A2, rW3, g, c); dw_flush<1, 1, CL::I3, false>(dwl + P::A3, rW4, g, c);
            db_flush<4, false>(dwl + P::B0, rb1, g, c); db_flush<2, false>(dwl + P::B1, rb2, g, c); db_flush<1, false>(dwl + P::B2, rb3, g, c); db_flush<1, false>(dwl + P::B3, rb4, g, c);
        }
        __syncthreads();
    }
#undef ST_ZT
#undef ST_ZB
    // ---- packed partial gradient of this workgroup for the layers of this half (layout of the parameter block; cf. ae_bwd_kernel)
    auto sum4 = [&](int idx) { return (lds[idx] + lds[P::FWD_END + idx]) + (lds[2 * P::FWD_END + idx] + lds[3 * P::FWD_END + idx]); };
    float* base = ws + ((size_t)blockIdx.x * 2 + ae) * PG;
    const int out[NL] = {64, 32, 16, 16, 16, 16, 32, 64, OT};
    const int in[NL] = {T, 64, 32, 16, 16 + K, 16, 16, 32, 64};
    const int inp[NL] = {CL::I0, CL::I1, CL::I2, CL::I3, CL::I4, CL::I5, CL::I6, CL::I7, CL::I8};
    const AETab tab = P::tab();
    constexpr int NT = NW * 64;
#pragma unroll
    for (int l = 0; l < NL; ++l) {
        if (l < P::L0 || l >= P::L1) continue;
        const int IN = in[l], n = out[l] * IN;
#pragma unroll
        for (int u = 0; u < (ae_max_elems(l) + NT - 1) / NT; ++u) {
            const int e = tid + u * NT;
            if (e < n) { const int o = e / IN, i = e - o * IN; base[go.w[l] + e] = sum4(tab.ao[l] + o * inp[l] + i); }
        }
        if (tid < out[l]) base[go.b[l] + tid] = sum4(tab.bo[l] + tid);
        { const int p0 = go.w[l] + n, np = go.b[l] - p0; if (tid < np) base[p0 + tid] = 0.f; }
        { const int p0 = go.b[l] + out[l], np = (l + 1 < NL ? go.w[l + 1] : PG) - p0; if (tid < np) base[p0 + tid] = 0.f; }
    }
}
#undef ST_BWD_STAGE2
#undef ST_PIPE2

}  // namespace sta
