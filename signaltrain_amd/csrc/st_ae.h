// st_ae.h -- fused knob-conditioned autoencoders (nn_proc.py:28-126) on fp32 MFMA (gfx950).
//
// Rows of the problem are (window b, frequency bin f); the features of a row are its T STFT frames.
// A wave processes groups of 16 rows with v_mfma_f32_16x16x4_f32 in the orientation
//     D[o][row] = sum_i W[o][i] * H[i][row]          (A = weights, B = activations)
// whose result layout (lane (g,c) = (l>>4, l&15), reg r  <->  o = 16*tile + 4g + r, row = c; "D layout") is exactly
// the B-operand layout of the next layer when its k-steps enumerate features in the order
// i = 16*tile + 4g + r: activations never leave registers, no transposes, no LDS traffic for them.
//
// Weights sit in LDS as *fragment images*: the four A operands a lane needs for the k-steps r = 0..3 of one 16x16 tile
// are 16 contiguous bytes, so a tile costs ONE ds_read_b128 per lane instead of four ds_read_b32, and the image is
// k-block-major so the 16-lane service groups of ds_read_b128 (which mix g = 0/1 lanes) hit 16 distinct 16-B slots:
//     forward image  A_l[(i >> 2)][o][i & 3]   lane (g,c), tile (ot,it) reads slot (4 it + g) * OUTp + 16 ot + c
//     dgrad image    G_l[(o >> 2)][i][o & 3]   lane (g,c), tile (it,ot) reads slot (4 ot + g) * INp  + 16 it + c
// (the first version kept one row-major copy with an odd pitch: every fragment read was a 2-way bank conflict in both
// orientations -- a third of the LDS cycles of the backward kernel were conflict cycles).
//
// Forward: the two autoencoders (magnitude / phase) of the same rows run as two interleaved chains in one wave
// (independent accumulators hide the dependent-MFMA latency) and meet in the epilogue (nn_proc.py:322-326: phase
// residual, polar -> rect).  Backward: see ae_bwd_kernel.
//
// Row space is padded per window to FP = KP/2 = roundup(F,16) "virtual bins": groups never straddle
// windows (knobs are wave-uniform) and the pad columns of the AA matrix get written as zeros.
// These kernels cover T <= 32, OT <= 16, K <= 16 (every padded dimension fixed at compile time); wider geometries run
// layers 1 and 9 as GEMMs (st_ae_wide.h) around the INNER forms below.
#pragma once
#include "st_common.h"

namespace sta {

constexpr int NL = 9;
// Timing-only ablation build of ae_bwd_kernel (tools/ae_ablate.sh; results are INVALID when non-zero; never set in the product build):
// 1 no d-out global loads | 2 no dv stores | 4 no weight-gradient MFMAs | 8 ELU without the transcendental | 16 no LDS transposes |
// 32 no forward-recompute MFMAs | 64 no next-group prefetch loads
#ifndef ST_AE_ABLATE
#define ST_AE_ABLATE 0
#endif

// Global-memory description of one autoencoder inside the flat parameter buffer (float offsets from
// the autoencoder base: weight l at w[l], bias at b[l]); same for the gradient buffer.
struct AEOffsets { int w[NL]; int b[NL]; };

// Group enumeration of the forward kernels (round 4).  Neighbouring row groups share the 128-byte lines of the [B][T][F] arrays (a 16-row group reads 64 bytes of
// every row, a 32-row group 128 bytes at an arbitrary alignment: rows are F * 4 = 2052 bytes): with "workgroup-fastest" numbering neighbours ran on neighbouring
// WORKGROUPS, i.e. on different XCDs, and every shared line crossed the fabric twice (FETCH_SIZE of the 32-row forward: 2 x 26.0 MB for 26.3 MB of inputs).
// Workgroups go round-robin over the 8 XCDs, so XCD x = blockIdx.x & 7 now owns a contiguous eighth of the groups and walks it workgroup-fastest WITHIN the XCD
// (wave * S + slot: the partial last round still puts one extra group on every workgroup instead of a full round on a few).  Grids that are not a multiple of 8
// keep the old numbering.
struct GroupWalk { int first, end, stride; };
__device__ __forceinline__ GroupWalk fwd_group_walk(const int ngroups, const int NW, const int wave)
{
    const int G = (int)gridDim.x, b = (int)blockIdx.x;
    if ((G & 7) == 0) {
        const int S = G >> 3, x = b & 7, slot = b >> 3;
        const int per = (ngroups + 7) >> 3;
        const int lo = x * per, hi = lo + per < ngroups ? lo + per : ngroups;
        return GroupWalk{lo + wave * S + slot, hi, S * NW};
    }
    return GroupWalk{wave * G + b, ngroups, G * NW};
}

// Compile-time LDS layout (floats).  Padded shapes OUTp x INp per layer:
//   l = 0: 64 x 32 (IN = T)   1: 32 x 64   2: 16 x 32   3: 16 x 16   4: 16 x 32 (IN = 16 + K)   5: 16 x 16   6: 32 x 16
//   7: 64 x 32   8: 16 x 64 (OUT = OT)
struct CL {
    static constexpr int O0 = 64, O1 = 32, O2 = 16, O3 = 16, O4 = 16, O5 = 16, O6 = 32, O7 = 64, O8 = 16;
    static constexpr int I0 = 32, I1 = 64, I2 = 32, I3 = 16, I4 = 32, I5 = 16, I6 = 16, I7 = 32, I8 = 64;
    // forward images
    static constexpr int A0 = 0, A1 = A0 + O0 * I0, A2 = A1 + O1 * I1, A3 = A2 + O2 * I2, A4 = A3 + O3 * I3, A5 = A4 + O4 * I4,
                         A6 = A5 + O5 * I5, A7 = A6 + O6 * I6, A8 = A7 + O7 * I7, AEND = A8 + O8 * I8;
    // biases
    static constexpr int B0 = AEND, B1 = B0 + O0, B2 = B1 + O1, B3 = B2 + O2, B4 = B3 + O3, B5 = B4 + O4, B6 = B5 + O5,
                         B7 = B6 + O6, B8 = B7 + O7, FWD_TOTAL = B8 + O8;          // what the forward kernels keep per autoencoder
    // dgrad images (backward kernel only)
    static constexpr int G0 = FWD_TOTAL, G1 = G0 + O0 * I0, G2 = G1 + O1 * I1, G3 = G2 + O2 * I2, G4 = G3 + O3 * I3, G5 = G4 + O4 * I4,
                         G6 = G5 + O5 * I5, G7 = G6 + O6 * I6, G8 = G7 + O7 * I7, BWD_TOTAL = G8 + O8 * I8;
    static_assert(FWD_TOTAL % 4 == 0 && BWD_TOTAL % 4 == 0, "16-byte aligned regions");
};

// Cooperative load of one autoencoder's parameters into its LDS images (zero padded).  ALL global loads of a thread -- every
// layer's weights (coalesced reads of the packed tensors) and biases -- are issued back to back before anything is consumed:
// one memory round trip for the whole parameter block.  (Layer by layer, as a run-time loop, it was nine dependent round
// trips -- ~15 us of a workgroup's life before its first MFMA, at one wave per SIMD nothing hides that.)  The LDS zero-fill
// runs while the loads are in flight, then the values are scattered to their image positions.  Layers [l0, l1) only (the
// wide path keeps 1..7).  The caller synchronises afterwards.
constexpr int ae_max_elems(int l) { return l == 0 ? 64 * 32 : l == 1 ? 32 * 64 : l == 2 ? 16 * 32 : l == 3 ? 16 * 16 : l == 4 ? 16 * 32 :
                                           l == 5 ? 16 * 16 : l == 6 ? 32 * 16 : l == 7 ? 64 * 32 : 16 * 64; }
template <int NT>
struct AEParamRegs { float v[NL][(64 * 32 + NT - 1) / NT]; float bv[NL]; };      // one thread's share of an autoencoder's parameters

template <int NT>
__device__ __forceinline__ void ae_params_issue(AEParamRegs<NT>& r, const float* __restrict__ ae, const AEOffsets& go,
                                                const int T, const int OT, const int K, const int tid, const int l0, const int l1)
{
    const int out[NL] = {64, 32, 16, 16, 16, 16, 32, 64, OT};
    const int in[NL] = {T, 64, 32, 16, 16 + K, 16, 16, 32, 64};
#pragma unroll
    for (int l = 0; l < NL; ++l) {
        const bool on = l >= l0 && l < l1;
        const int n = on ? out[l] * in[l] : 0;
        const float* src = ae + go.w[l];
#pragma unroll
        for (int u = 0; u < (ae_max_elems(l) + NT - 1) / NT; ++u) { const int e = tid + u * NT; r.v[l][u] = src[e < n ? e : 0]; }
        r.bv[l] = ae[go.b[l] + (tid < out[l] ? tid : 0)];
    }
}
// BF != 0 (16-bit operand kernels): the weight images are written ALREADY ROUNDED to bfloat16 / float16, packed into the first half of
// each layer's region (same element indexing, 2-byte elements): a fragment is then one 8-byte read and needs no conversion at its use
// (the 16-bit backward kernel spent 144 of its 285 conversions per row group on weight fragments).  Biases stay fp32.
template <int BF> __device__ __forceinline__ unsigned short st_half_bits(const float v)
{
    if constexpr (BF == 2) { union { _Float16 h; unsigned short u; } p; p.h = (_Float16)v; return p.u; }
    else { union { __bf16 h; unsigned short u; } p; p.h = (__bf16)v; return p.u; }
}
template <int BF> __device__ __forceinline__ float st_half_to_float(const unsigned short b)
{
    if constexpr (BF == 2) { union { unsigned short u; _Float16 h; } p; p.u = b; return (float)p.h; }
    else return __uint_as_float((unsigned)b << 16);
}
template <int NT, int BF = 0>
__device__ __forceinline__ void ae_params_scatter(float* lds, const AEParamRegs<NT>& r, const int T, const int OT, const int K,
                                                  const int tid, const int l0, const int l1, const bool dgrad_images)
{
    const int out[NL] = {64, 32, 16, 16, 16, 16, 32, 64, OT};
    const int in[NL] = {T, 64, 32, 16, 16 + K, 16, 16, 32, 64};
    const int outp[NL] = {CL::O0, CL::O1, CL::O2, CL::O3, CL::O4, CL::O5, CL::O6, CL::O7, CL::O8};
    const int inp[NL] = {CL::I0, CL::I1, CL::I2, CL::I3, CL::I4, CL::I5, CL::I6, CL::I7, CL::I8};
    const int ao[NL] = {CL::A0, CL::A1, CL::A2, CL::A3, CL::A4, CL::A5, CL::A6, CL::A7, CL::A8};
    const int bo[NL] = {CL::B0, CL::B1, CL::B2, CL::B3, CL::B4, CL::B5, CL::B6, CL::B7, CL::B8};
    const int gi[NL] = {CL::G0, CL::G1, CL::G2, CL::G3, CL::G4, CL::G5, CL::G6, CL::G7, CL::G8};
#pragma unroll
    for (int l = 0; l < NL; ++l) {
        const bool on = l >= l0 && l < l1;
        const int IN = in[l], n = on ? out[l] * IN : 0, OP = outp[l], IP = inp[l];
#pragma unroll
        for (int u = 0; u < (ae_max_elems(l) + NT - 1) / NT; ++u) {
            const int e = tid + u * NT;
            if (e < n) {
                const int o = e / IN, i = e - o * IN;
                if constexpr (BF) {
                    const unsigned short hb = st_half_bits<BF>(r.v[l][u]);
                    reinterpret_cast<unsigned short*>(lds + ao[l])[(((i >> 2) * OP + o) << 2) + (i & 3)] = hb;
                    if (dgrad_images) reinterpret_cast<unsigned short*>(lds + gi[l])[(((o >> 2) * IP + i) << 2) + (o & 3)] = hb;
                } else {
                lds[ao[l] + (((i >> 2) * OP + o) << 2) + (i & 3)] = r.v[l][u];
                if (dgrad_images) lds[gi[l] + (((o >> 2) * IP + i) << 2) + (o & 3)] = r.v[l][u];
                }
            }
        }
        if (on && tid < out[l]) lds[bo[l] + tid] = r.bv[l];
    }
}
// One autoencoder (backward kernel: forward + dgrad images) ...
template <int NT, int BF = 0>
__device__ inline void ae_load_lds(float* lds, const float* __restrict__ ae, const AEOffsets& go, const int T, const int OT, const int K,
                                   const int tid, const int l0, const int l1, const bool dgrad_images)
{
    AEParamRegs<NT> r;
    ae_params_issue<NT>(r, ae, go, T, OT, K, tid, l0, l1);
    const int total = dgrad_images ? CL::BWD_TOTAL : CL::FWD_TOTAL;
    for (int e = tid; e < total; e += NT) lds[e] = 0.f;
    __syncthreads();
    ae_params_scatter<NT, BF>(lds, r, T, OT, K, tid, l0, l1, dgrad_images);
}
// ... or both (forward kernels: two forward images CL::FWD_TOTAL floats apart), still one round trip.
template <int NT, int BF = 0>
__device__ inline void ae_load_lds2(float* lds, const float* __restrict__ ae_m, const float* __restrict__ ae_p, const AEOffsets& go,
                                    const int T, const int OT, const int K, const int tid, const int l0, const int l1)
{
    AEParamRegs<NT> rm, rp;
    ae_params_issue<NT>(rm, ae_m, go, T, OT, K, tid, l0, l1);
    ae_params_issue<NT>(rp, ae_p, go, T, OT, K, tid, l0, l1);
    for (int e = tid; e < 2 * CL::FWD_TOTAL; e += NT) lds[e] = 0.f;
    __syncthreads();
    ae_params_scatter<NT, BF>(lds, rm, T, OT, K, tid, l0, l1, false);
    ae_params_scatter<NT, BF>(lds + CL::FWD_TOTAL, rp, T, OT, K, tid, l0, l1, false);
}

#define ST_MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

// Global accesses by 32-bit ELEMENT offset from a wave-uniform base (the host checks that every buffer is < 2^30 elements):
// they compile to the saddr + 32-bit voffset form, i.e. no 64-bit address arithmetic per access (3-4 VALU instructions
// each, ~60 accesses per row group in the backward kernel).  ST_MUL24: offsets are products of small indices.
__device__ __forceinline__ float ldg32(const float* __restrict__ base, const unsigned elem)
{
    return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + (size_t)(elem << 2));
}
__device__ __forceinline__ void stg32(float* __restrict__ base, const unsigned elem, const float v)
{
    *reinterpret_cast<float*>(reinterpret_cast<char*>(base) + (size_t)(elem << 2)) = v;
}
#define ST_MUL24(a, b) __umul24((unsigned)(a), (unsigned)(b))

// bf16-operand arithmetic of the forward kernels (st_set_precision(2)): weights and layer inputs rounded to bfloat16 (RNE),
// one v_mfma_f32_16x16x16_bf16 per 16x16 tile (k = the 16 features 4g + r of a tile, i.e. a packed D-layout register set IS
// the B operand) instead of four v_mfma_f32_16x16x4_f32; accumulation, bias, ELU and epilogue stay fp32.
// BF template parameter of everything below: 0 = fp32 MFMA, 1 = bfloat16 operands, 2 = float16 operands (BASELINE configs[4];
// IEEE conversion: out-of-range values become inf and the optimizer kernel skips the step, as under Apex amp).
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
typedef _Float16 f16x4_t __attribute__((ext_vector_type(4)));
template <int BF>
__device__ __forceinline__ f32x4 mfma16h(const s16x4 a, const s16x4 b, const f32x4 c)
{
    if constexpr (BF == 2) {
        union { s16x4 s; f16x4_t h; } pa, pb; pa.s = a; pb.s = b;
        return __builtin_amdgcn_mfma_f32_16x16x16f16(pa.h, pb.h, c, 0, 0, 0);
    } else return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c, 0, 0, 0);
}
#define ST_MFMA16B(a, b, c) mfma16h<BF>((a), (b), (c))
template <int BF>
__device__ __forceinline__ s16x4 pack_h4(f32x4 v)
{
    if constexpr (BF == 2) {
        union { f16x4_t h; s16x4 s; } p; p.h = __builtin_convertvector(v, f16x4_t); return p.s;
    } else {
        // two 2-element conversions = two v_cvt_pk_bf16_f32: the 4-element form was legalised element by element inside these
        // kernels (569 single conversions + 285 v_perm_b32 per loop body against 285 v_cvt_pk_f16_f32 for fp16)
        typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
        typedef float f32x2_t __attribute__((ext_vector_type(2)));
        union { bf16x2_t h; unsigned u; } lo, hi;
        lo.h = __builtin_convertvector((f32x2_t){v[0], v[1]}, bf16x2_t);
        hi.h = __builtin_convertvector((f32x2_t){v[2], v[3]}, bf16x2_t);
        union { unsigned u[2]; s16x4 s; } r; r.u[0] = lo.u; r.u[1] = hi.u; return r.s;
    }
}
#define pack_bf16x4(v) pack_h4<BF>(v)
template <int BF>
__device__ __forceinline__ float round_h(const float v)
{
    if constexpr (BF == 2) return (float)(_Float16)v;
    else {
        union { __bf16 h; unsigned short u; } p; p.h = (__bf16)v;
        union { unsigned u; float f; } q; q.u = (unsigned)p.u << 16; return q.f;
    }
}
#define bf16_rne(v) round_h<BF>(v)

// One 16x16 tile of A operands (k-steps r = 0..3) from a forward image: W[o = 16 ot + c][i = 16 it + 4 g + r].
// BF != 0: the image holds 16-bit values (ae_params_scatter); the four of a fragment arrive as ONE 8-byte read in the low half of the
// returned vector (frag_bits() hands them to the MFMA, the upper half is never touched).
template <int BF>
__device__ __forceinline__ f32x4 frag_read(const float* img, const int idx)
{
    if constexpr (BF) {
        typedef float f32x2_t __attribute__((ext_vector_type(2)));
        const f32x2_t lo = *reinterpret_cast<const f32x2_t*>(reinterpret_cast<const unsigned short*>(img) + idx);
        return __builtin_shufflevector(lo, lo, 0, 1, -1, -1);
    } else return *reinterpret_cast<const f32x4*>(img + idx);
}
__device__ __forceinline__ s16x4 frag_bits(const f32x4 v)
{
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    union { f32x2_t f; s16x4 s; } p; p.f = __builtin_shufflevector(v, v, 0, 1); return p.s;
}
template <int OUTP, int BF = 0>
__device__ __forceinline__ f32x4 frag_fwd(const float* img, const int ot, const int it, const int g, const int c)
{
    return frag_read<BF>(img, ((4 * it + g) * OUTP + 16 * ot + c) << 2);
}
// ... from a dgrad image: W[o = 16 ot + 4 g + r][i = 16 it + c].
template <int INP, int BF = 0>
__device__ __forceinline__ f32x4 frag_dgrad(const float* img, const int ot, const int it, const int g, const int c)
{
    return frag_read<BF>(img, ((4 * ot + g) * INP + 16 * it + c) << 2);
}

// Hidden layer for NC interleaved chains: hout[ch][ot] = ELU(W[ch] * hin[ch] + bias[ch]), weights fetched just in time
// (the forward kernels run two waves per SIMD, which hides the LDS latency).
template <int NC, int OTL, int ITL, int OUTP, int BF = 0>
__device__ __forceinline__ void layer_fwd(const float* const (&W)[NC], const float* const (&bias)[NC],
                                          const f32x4 (&hin)[NC][ITL], f32x4 (&hout)[NC][OTL], const int g, const int c)
{
    s16x4 ph[NC][ITL];
    if constexpr (BF) {
#pragma unroll
        for (int ch = 0; ch < NC; ++ch)
#pragma unroll
            for (int it = 0; it < ITL; ++it) ph[ch][it] = pack_bf16x4(hin[ch][it]);
    }
#pragma unroll
    for (int ot = 0; ot < OTL; ++ot) {
        f32x4 acc[NC];
#pragma unroll
        for (int ch = 0; ch < NC; ++ch) acc[ch] = *reinterpret_cast<const f32x4*>(bias[ch] + 16 * ot + 4 * g);   // accumulator starts at the bias
#pragma unroll
        for (int it = 0; it < ITL; ++it) {
            f32x4 w[NC];
#pragma unroll
            for (int ch = 0; ch < NC; ++ch) w[ch] = frag_fwd<OUTP, BF>(W[ch], ot, it, g, c);
            if constexpr (BF) {
#pragma unroll
                for (int ch = 0; ch < NC; ++ch) acc[ch] = ST_MFMA16B(frag_bits(w[ch]), ph[ch][it], acc[ch]);
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int ch = 0; ch < NC; ++ch) acc[ch] = ST_MFMA16(w[ch][r], hin[ch][it][r], acc[ch]);
            }
        }
#pragma unroll
        for (int ch = 0; ch < NC; ++ch) hout[ch][ot] = elu4(acc[ch]);
    }
}

// Layer 5 of NC chains: [h4 ; knobs] -> 16 (nn_proc.py:92-96).  The knob block uses its own k-step order (step q: lane
// group g carries knob 4q + g), so K <= 4 knobs cost one MFMA per chain instead of four.
template <int NC, int BF = 0>
__device__ __forceinline__ void layer5_fwd(const float* const (&lw)[NC], const f32x4 (&h4)[NC][1], const float (&kn)[4], const int KQ,
                                           f32x4 (&h5)[NC][1], const int g, const int c)
{
    f32x4 acc[NC], w[NC];
#pragma unroll
    for (int ch = 0; ch < NC; ++ch) { acc[ch] = *reinterpret_cast<const f32x4*>(lw[ch] + CL::B4 + 4 * g); w[ch] = frag_fwd<CL::O4, BF>(lw[ch] + CL::A4, 0, 0, g, c); }
    if constexpr (BF) {
#pragma unroll
        for (int ch = 0; ch < NC; ++ch) acc[ch] = ST_MFMA16B(frag_bits(w[ch]), pack_bf16x4(h4[ch][0]), acc[ch]);
    } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int ch = 0; ch < NC; ++ch) acc[ch] = ST_MFMA16(w[ch][r], h4[ch][0][r], acc[ch]);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q)
        if (q < KQ) {
#pragma unroll
            for (int ch = 0; ch < NC; ++ch) {
                // W5[o = c][i = 16 + 4q + g].  BF: the knob block keeps its own k-step order on the fp32 MFMA, with both operands rounded
                // to 16 bits first (the image already is; products of such values are exact in fp32, so this IS the 16-bit arithmetic)
                float wk;
                if constexpr (BF) wk = st_half_to_float<BF>(reinterpret_cast<const unsigned short*>(lw[ch] + CL::A4)[(((4 + q) * CL::O4 + c) << 2) + g]);
                else wk = lw[ch][CL::A4 + (((4 + q) * CL::O4 + c) << 2) + g];
                acc[ch] = ST_MFMA16(wk, BF ? bf16_rne(kn[q]) : kn[q], acc[ch]);
            }
        }
#pragma unroll
    for (int ch = 0; ch < NC; ++ch) h5[ch][0] = elu4(acc[ch]);
}

#define ST_W2(off_) {lw[0] + (off_), lw[1] + (off_)}

// ------------------------------------------------------------------------------------------ forward
// Per-group inputs of one lane: layer-1 B operands (D layout: t = 16 it + 4g + r, T <= 32), the skip/residual
// tails (t = T-OT + 4g + r, OT <= 16) and the knob values -- as RAW loads (fwd_load_* / fwd_mask_* below).
struct FwdIn { f32x4 v[2][2]; float tl[2][4]; float kn[4]; };   // kn: knob 4q + g for the (up to 4) knob k-steps of layer 5

// RAW loads from clamped (always valid) addresses; the fwd_mask_*() zero the padding rows / bins / knobs afterwards.  The
// selects must not sit right behind the loads, and the prefetch must not sit in a conditional block: either makes the
// compiler wait for the loads on the spot, turning the one-group-ahead prefetch into a synchronous load.
// Round 6: three parts.  Only the input rows (consumed by layer 1 at the very top) are prefetched one group ahead; the knobs (layer 5) and the tails (epilogue)
// of a group are loaded at ITS top and masked where they are consumed -- 12 fewer loop-carried registers (the kernel with the kept activations spilled at three
// waves per SIMD, and a spill reload in the epilogue waits for every store issued before it).  In-place refills of loop-carried registers were tried and
// removed: a refill that meets its older value at the loop latch is COPIED there, and the copy waits for the load on the spot.
__device__ __forceinline__ void fwd_load_v(FwdIn& in, const float* __restrict__ mag, const float* __restrict__ phs,
                                           const int b, const int f, const bool fv, const int T, const int F, const int g)
{
    const unsigned base = ST_MUL24(ST_MUL24(b, T), F) + (unsigned)(fv ? f : 0);
#pragma unroll
    for (int it = 0; it < 2; ++it)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int t = 16 * it + 4 * g + r;
            const unsigned o = base + ST_MUL24(t < T ? t : 0, F);
            in.v[0][it][r] = ldg32(mag, o); in.v[1][it][r] = ldg32(phs, o);
        }
}
__device__ __forceinline__ void fwd_load_kn(FwdIn& in, const float* __restrict__ knobs, const int K, const int b, const int g)
{
#pragma unroll
    for (int q = 0; q < 4; ++q) { const int kn = 4 * q + g; in.kn[q] = ldg32(knobs, ST_MUL24(b, K) + (unsigned)(kn < K ? kn : 0)); }
}
__device__ __forceinline__ void fwd_load_tl(FwdIn& in, const float* __restrict__ mag, const float* __restrict__ phs,
                                            const int b, const int f, const bool fv, const int T, const int OT, const int F, const int g)
{
    const unsigned base = ST_MUL24(ST_MUL24(b, T), F) + (unsigned)(fv ? f : 0);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int to = 4 * g + r;
        const unsigned o = base + ST_MUL24(to < OT ? T - OT + to : 0, F);
        in.tl[0][r] = ldg32(mag, o); in.tl[1][r] = ldg32(phs, o);
    }
}
__device__ __forceinline__ void fwd_mask_v(f32x4 (&v)[2][2], const FwdIn& in, const bool fv, const int T, const int g)
{
#pragma unroll
    for (int it = 0; it < 2; ++it)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const bool ok = fv && 16 * it + 4 * g + r < T;
            v[0][it][r] = ok ? in.v[0][it][r] : 0.f; v[1][it][r] = ok ? in.v[1][it][r] : 0.f;
        }
}
__device__ __forceinline__ void fwd_mask_kn(float (&kn)[4], const FwdIn& in, const int K, const int g)
{
#pragma unroll
    for (int q = 0; q < 4; ++q) kn[q] = (4 * q + g) < K ? in.kn[q] : 0.f;
}
__device__ __forceinline__ void fwd_mask_tl(float (&tl)[2][4], const FwdIn& in, const bool fv, const int OT, const int g)
{
#pragma unroll
    for (int r = 0; r < 4; ++r) { const bool ok = fv && 4 * g + r < OT; tl[0][r] = ok ? in.tl[0][r] : 0.f; tl[1][r] = ok ? in.tl[1][r] : 0.f; }
}

// Round 6: the post-ELU activations of both nets KEPT for the backward (what the reference's autograd keeps, nn_proc.py:77-126) instead of being recomputed there.
// Per (net, 16-row group) 17 tiles of 16 x 16 in D layout, [net][group][tile][lane] float4: h1 tiles 0-3, h2 4-5, h3 6, h4 7, h5 8, h6 9, h7 10-11, h8 12-15,
// ELU(a9) 16 -- every store / load is one fully coalesced 1 KB access per wave, 17 KB per (net, group), 2 * groups * 17 KB per step (294 MB at B = 256).
constexpr int AE_SV_TILES = 17;
// ST_SV_NT: bit 0 = the stores, bit 1 = the loads carry the non-temporal hint.  Measured (B = 256, lab notebook): stores without it 81 us (ae_fwd), with it 76;
// the loads make no difference.  ST_SV_DIAG=<mask> (timing only, results INVALID): every (net, group) block lands on block (group & mask), i.e. the stores
// stay in the L2 (mask 15) or the Infinity Cache (mask 1023) -- how the store cost was split into issue / fabric / HBM parts.
#ifndef ST_SV_NT
#define ST_SV_NT 1
#endif
// p: WAVE-UNIFORM base of the (net, group) block (the callers pass the group through readfirstlane), lane16 = lane * 16: the accesses compile to the
// saddr + 32-bit voffset form with the tile as an immediate / scalar offset -- no 64-bit vector address per access.
template <int NTL>
__device__ __forceinline__ void sv_store(float* __restrict__ p, const unsigned lane16, const int tile0, const f32x4 (&h)[NTL])
{
#pragma unroll
    for (int t = 0; t < NTL; ++t) {
        f32x4* q = reinterpret_cast<f32x4*>(reinterpret_cast<char*>(p + (tile0 + t) * 256) + (size_t)lane16);
#if ST_SV_NT & 1
        __builtin_nontemporal_store(h[t], q);
#else
        *q = h[t];
#endif
    }
}
template <int NTL>
__device__ __forceinline__ void sv_load(const float* __restrict__ p, const unsigned lane16, const int tile0, f32x4 (&h)[NTL])
{
#pragma unroll
    for (int t = 0; t < NTL; ++t) {
        const f32x4* q = reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(p + (tile0 + t) * 256) + (size_t)lane16);
#if ST_SV_NT & 2
        h[t] = __builtin_nontemporal_load(q);
#else
        h[t] = *q;
#endif
    }
}

// grid.x workgroups of NW waves; each wave walks 16-row groups: group id = b*(FP/16) + fg.  Requires T <= 32, OT <= 16, K <= 16.
template <int NW, int BF = 0, bool SV = false>      // BF: bf16 operands in the nine Linear layers (st_set_precision(2)); SV: keep the activations (sv != NULL)
__global__ void __launch_bounds__(NW * 64)
ae_fwd_kernel(const float* __restrict__ mag, const float* __restrict__ phs, const float* __restrict__ knobs,
              const float* __restrict__ ae_m, const float* __restrict__ ae_p, const AEOffsets go,
              float* __restrict__ mag_hat, float* __restrict__ phs_hat, float* __restrict__ AA,
              float* __restrict__ reg_partial,
              const int B, const int T, const int OT, const int F, const int K, const int KP, const float expfac,
              float* __restrict__ h4x = nullptr,      // optional: the 16-wide code h4 of both nets, [net][group][lane] float4 in D layout, for
                                                      // the split backward (st_ae_split.h); mag_hat == NULL: h4 only (no other output is written)
              unsigned short* __restrict__ AA16 = nullptr, const int aa_ht = 0,      // 16-bit GEMM configurations (st_gemm16.h): the spectra go out rounded to
                                                      // the operand type (1 bf16 / 2 fp16) INSTEAD of fp32 -- their only consumers are the two synthesis GEMMs
              float* __restrict__ sv = nullptr)       // SV: the kept activations of both nets (layout above)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, c = lane & 15;
    const float* const lw[2] = {lds, lds + CL::FWD_TOTAL};
    ae_load_lds2<NW * 64, BF>(lds, ae_m, ae_p, go, T, OT, K, tid, 0, NL);
    __syncthreads();

    const int FP = KP / 2, gpw = FP / 16;              // groups per window
    const int ngroups = B * gpw;
    const int KQ = (K + 3) / 4;
    float reg = 0.f;
    // frequency weights of the L1 term, one per bin (the table sits behind the two forward images; the host sizes the LDS request for it)
    float* const wtab = lds + 2 * CL::FWD_TOTAL;
    for (int i = tid; i < FP; i += NW * 64) wtab[i] = i < F ? expf(expfac * (float)i) : 0.f;
    unsigned toF[4], toK[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) { toF[r] = ST_MUL24(4 * g + r, F); toK[r] = ST_MUL24(4 * g + r, KP); }
    __syncthreads();

    f32x4 vin[2][2];      // masked input rows of the current group (loop-carried: taken over from the prefetch at the end of the previous iteration)
    // block-fastest group numbering: the partial last round (ngroups is rarely a multiple of the wave count) then puts ONE
    // extra group on every workgroup instead of a full extra round on the first few workgroups while the rest idle
    const GroupWalk gw = fwd_group_walk(ngroups, NW, wave);      // XCD-contiguous since round 4 (above)
    int grp = gw.first;
    if (grp < gw.end) {
        const int b = grp / gpw, f = (grp - b * gpw) * 16 + c;
        FwdIn first; fwd_load_v(first, mag, phs, b, f, f < F, T, F, g); fwd_mask_v(vin, first, f < F, T, g);
    }
    for (; grp < gw.end; grp += gw.stride) {
        asm volatile("" ::: "memory");      // keep the (loop-invariant) LDS weight fetches inside the loop: hoisting them spills
        const int b = grp / gpw, f = (grp - b * gpw) * 16 + c;
        const bool fv = f < F;
        const int gn = grp + gw.stride < gw.end ? grp + gw.stride : grp;      // last iteration: harmless reload of this group
        const int bn = gn / gpw, fn = (gn - bn * gpw) * 16 + c;
        FwdIn cur;      // RAW loads: the NEXT group's input rows, THIS group's knobs and tails
        fwd_load_v(cur, mag, phs, bn, fn, fn < F, T, F, g); fwd_load_kn(cur, knobs, K, b, g); fwd_load_tl(cur, mag, phs, b, f, fv, T, OT, F, g);

        f32x4 h1[2][4], h2[2][2], h3[2][1], h4[2][1], h5[2][1], h6[2][1], h7[2][2], h8[2][4], e9[2][1];
        // kept activations: wave-uniform bases of (net 0, grp) and (net 1, grp) -- net 1 sits ngroups * 17 KB further on
        const int grp_u = SV ? __builtin_amdgcn_readfirstlane(grp) : 0;
#ifdef ST_SV_DIAG
        float* const sv0 = SV ? sv + (size_t)(grp_u & ST_SV_DIAG) * (AE_SV_TILES * 256) : nullptr;      // timing only: every store stays in the L2 / MALL
#else
        float* const sv0 = SV ? sv + (size_t)grp_u * (AE_SV_TILES * 256) : nullptr;
#endif
        float* const sv1 = SV ? sv0 + (size_t)ngroups * (AE_SV_TILES * 256) : nullptr;
        const unsigned lane16 = (unsigned)lane << 4;
#define ST_SV(t0_, h_) do { if constexpr (SV) { sv_store(sv0, lane16, t0_, h_[0]); sv_store(sv1, lane16, t0_, h_[1]); } } while (0)
        { const float* const W[2] = ST_W2(CL::A0); const float* const bb[2] = ST_W2(CL::B0); layer_fwd<2, 4, 2, CL::O0, BF>(W, bb, vin, h1, g, c); } ST_SV(0, h1);
        { const float* const W[2] = ST_W2(CL::A1); const float* const bb[2] = ST_W2(CL::B1); layer_fwd<2, 2, 4, CL::O1, BF>(W, bb, h1, h2, g, c); } ST_SV(4, h2);
        { const float* const W[2] = ST_W2(CL::A2); const float* const bb[2] = ST_W2(CL::B2); layer_fwd<2, 1, 2, CL::O2, BF>(W, bb, h2, h3, g, c); } ST_SV(6, h3);
        { const float* const W[2] = ST_W2(CL::A3); const float* const bb[2] = ST_W2(CL::B3); layer_fwd<2, 1, 1, CL::O3, BF>(W, bb, h3, h4, g, c); } ST_SV(7, h4);
        if (h4x) {
            float4* hv = reinterpret_cast<float4*>(h4x);
            hv[(size_t)grp * 64 + lane] = make_float4(h4[0][0][0], h4[0][0][1], h4[0][0][2], h4[0][0][3]);
            hv[((size_t)ngroups + grp) * 64 + lane] = make_float4(h4[1][0][0], h4[1][0][1], h4[1][0][2], h4[1][0][3]);
        }
        if (mag_hat) {                                                           // else: h4-only pass (wave-uniform)
        float knv[4]; fwd_mask_kn(knv, cur, K, g);
        layer5_fwd<2, BF>(lw, h4, knv, KQ, h5, g, c); ST_SV(8, h5);
        { const float* const W[2] = ST_W2(CL::A5); const float* const bb[2] = ST_W2(CL::B5); layer_fwd<2, 1, 1, CL::O5, BF>(W, bb, h5, h6, g, c); } ST_SV(9, h6);
        { const float* const W[2] = ST_W2(CL::A6); const float* const bb[2] = ST_W2(CL::B6); layer_fwd<2, 2, 1, CL::O6, BF>(W, bb, h6, h7, g, c); } ST_SV(10, h7);
        { const float* const W[2] = ST_W2(CL::A7); const float* const bb[2] = ST_W2(CL::B7); layer_fwd<2, 4, 2, CL::O7, BF>(W, bb, h7, h8, g, c); } ST_SV(12, h8);
        { const float* const W[2] = ST_W2(CL::A8); const float* const bb[2] = ST_W2(CL::B8); layer_fwd<2, 1, 4, CL::O8, BF>(W, bb, h8, e9, g, c); } ST_SV(16, e9);
#undef ST_SV
        // ---- epilogue (nn_proc.py:115,117,322-326).  Round 4: the frequency weight comes from a per-workgroup LDS table (was a full-precision expf per
        // group), the row offsets of a lane's four output frames are formed once before the loop, and the 16-bit type of the spectra is a compile-time
        // constant in the 16-bit instantiations (it was converted BOTH ways and selected at run time): 785 -> ~700 vector instructions per row-group pair
        const float wf = fv ? wtab[f] : 0.f;                     // train.py:115-117 frequency weight exp(expfac * f)
        float tlv[2][4]; fwd_mask_tl(tlv, cur, fv, OT, g);
        const unsigned boF = ST_MUL24(ST_MUL24(b, OT), F) + (unsigned)f, boK = ST_MUL24(ST_MUL24(b, OT), KP) + (unsigned)f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int to = 4 * g + r;
            if (to < OT) {
                float mh = 0.f, ph = 0.f, sn = 0.f, cs = 1.f;
                if (fv) {
                    mh = e9[0][0][r] * tlv[0][r];               // 'sf' skip-filter
                    ph = e9[1][0][r] + tlv[1][r];               // phase residual
                    st_sincos(ph, sn, cs);
                    stg32(mag_hat, boF + toF[r], mh);
                    stg32(phs_hat, boF + toF[r], ph);
                    reg += fabsf(mh * wf);
                }
                if (AA16) {                                                   // wave-uniform
                    const int ht = BF ? BF : aa_ht;                           // *_all modes: the spectra's type IS the layers' type
                    AA16[boK + toK[r]] = st_to_h16(mh * cs, ht);
                    AA16[boK + toK[r] + (unsigned)FP] = st_to_h16(mh * sn, ht);
                } else {
                    stg32(AA, boK + toK[r], mh * cs);       // f < FP always: pads get zeros
                    stg32(AA, boK + toK[r] + (unsigned)FP, mh * sn);
                }
            }
        }
        }
        fwd_mask_v(vin, cur, fn < F, T, g);                                     // take over the next group's input rows
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
template <int NW, int BF = 0>
__global__ void __launch_bounds__(NW * 64)
ae_inner_fwd_kernel(const float* __restrict__ H1m, const float* __restrict__ H1p, const float* __restrict__ knobs,
                    const float* __restrict__ ae_m, const float* __restrict__ ae_p, const AEOffsets go,
                    float* __restrict__ H8m, float* __restrict__ H8p, const int B, const int F, const int K, const int KP)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, c = lane & 15;
    const float* const lw[2] = {lds, lds + CL::FWD_TOTAL};
    ae_load_lds2<NW * 64, BF>(lds, ae_m, ae_p, go, 16, 16, K, tid, 1, 8);
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
        float kn[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) { const int k = 4 * q + g; const float x = knobs[(size_t)b * K + (k < K ? k : 0)]; kn[q] = k < K ? x : 0.f; }
        f32x4 h2[2][2], h3[2][1], h4[2][1], h5[2][1], h6[2][1], h7[2][2], h8[2][4];
        { const float* const W[2] = ST_W2(CL::A1); const float* const bb[2] = ST_W2(CL::B1); layer_fwd<2, 2, 4, CL::O1, BF>(W, bb, h1, h2, g, c); }
        { const float* const W[2] = ST_W2(CL::A2); const float* const bb[2] = ST_W2(CL::B2); layer_fwd<2, 1, 2, CL::O2, BF>(W, bb, h2, h3, g, c); }
        { const float* const W[2] = ST_W2(CL::A3); const float* const bb[2] = ST_W2(CL::B3); layer_fwd<2, 1, 1, CL::O3, BF>(W, bb, h3, h4, g, c); }
        layer5_fwd<2, BF>(lw, h4, kn, KQ, h5, g, c);
        { const float* const W[2] = ST_W2(CL::A5); const float* const bb[2] = ST_W2(CL::B5); layer_fwd<2, 1, 1, CL::O5, BF>(W, bb, h5, h6, g, c); }
        { const float* const W[2] = ST_W2(CL::A6); const float* const bb[2] = ST_W2(CL::B6); layer_fwd<2, 2, 1, CL::O6, BF>(W, bb, h6, h7, g, c); }
        { const float* const W[2] = ST_W2(CL::A7); const float* const bb[2] = ST_W2(CL::B7); layer_fwd<2, 4, 2, CL::O7, BF>(W, bb, h7, h8, g, c); }
        const bool fv = f < F;
#pragma unroll
        for (int ch = 0; ch < 2; ++ch)
#pragma unroll
            for (int ot = 0; ot < 4; ++ot)
#pragma unroll
                for (int r = 0; r < 4; ++r) Hout[ch][(size_t)(16 * ot + 4 * g + r) * R + col] = fv ? h8[ch][ot][r] : 0.f;
    }
}
#undef ST_W2

// ========================================================================================== backward
// blockIdx.y selects the autoencoder (0 = magnitude 'sf', 1 = phase); one workgroup = NW waves, ONE wave per SIMD (the 144
// persistent weight-gradient accumulators + the activations fill the 512-register budget).  Per 16-row group a wave
//   1. recomputes the forward chain, keeping every post-ELU activation in registers in D layout (68 regs);
//   2. forms d out (polar->rect backward of nn_proc.py:322-326 + the L1 term of loss_functions.py:36);
//   3. walks the layers backwards.  The data gradient continues the D-layout chain (A = W^T fragments from the dgrad
//      images).  The weight gradient dW_l = sum_rows da_l (x) h_{l-1} needs both operands with ROWS on the MFMA k index
//      ("T layout": lane (g,c), reg r <-> row 4g + r, feature 16*tile + c): they come from wave-private LDS transposes
//      (to_T: one ds_write_b128 + four ds_read_b32 per tile, no barrier);
//   4. accumulates the 36 dW tiles and the bias row sums in registers for the whole kernel.
// At the end every wave stores its accumulators into its own LDS gradient image (all four in parallel, over the no-longer
// needed weight images and scratch) and one unrolled pass sums the four images in a fixed order into the workgroup's partial
// dW/db (packed like the parameters): run-to-run identical bits; ae_grad_reduce_kernel sums the workgroups.
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
// the MFMA (one s_waitcnt per MFMA: with one wave per SIMD the kernel was LDS-latency-bound).  These helpers burst-load
// every fragment of a layer stage into a register array; the caller places a scheduling fence between the burst and
// the MFMAs, so a stage pays one LDS round trip instead of one per MFMA.
template <int OTL, int ITL, int OUTP, int BF = 0>
__device__ __forceinline__ void frags_fwd(f32x4 (&fr)[OTL * ITL], const float* img, const int g, const int c)
{
#pragma unroll
    for (int ot = 0; ot < OTL; ++ot)
#pragma unroll
        for (int it = 0; it < ITL; ++it) fr[ot * ITL + it] = frag_fwd<OUTP, BF>(img, ot, it, g, c);
}
template <int OTL, int ITL, int INP, int BF = 0>
__device__ __forceinline__ void frags_dgrad(f32x4 (&fr)[ITL * OTL], const float* img, const int g, const int c)
{
#pragma unroll
    for (int it = 0; it < ITL; ++it)
#pragma unroll
        for (int ot = 0; ot < OTL; ++ot) fr[it * OTL + ot] = frag_dgrad<INP, BF>(img, ot, it, g, c);
}
#define ST_FENCE() __builtin_amdgcn_sched_barrier(0)

template <int OTL, int ITL, int BF = 0>
__device__ __forceinline__ void fwdD_fr(const f32x4 (&fr)[OTL * ITL], const float* bias, const f32x4 (&hin)[ITL],
                                        f32x4 (&hout)[OTL], const int g)
{
    s16x4 ph[ITL];
    if constexpr (BF) {
#pragma unroll
        for (int it = 0; it < ITL; ++it) ph[it] = pack_bf16x4(hin[it]);
    }
#pragma unroll
    for (int ot = 0; ot < OTL; ++ot) {
        f32x4 acc = *reinterpret_cast<const f32x4*>(bias + 16 * ot + 4 * g);   // accumulator starts at the bias (D layout: o = 16 ot + 4 g + r)
#pragma unroll
        for (int it = 0; it < ITL; ++it) {
#if ST_AE_ABLATE & 32
            acc[0] += fr[ot * ITL + it][0] * hin[it][0];
#else
            if constexpr (BF) acc = ST_MFMA16B(frag_bits(fr[ot * ITL + it]), ph[it], acc);
            else {
#pragma unroll
                for (int r = 0; r < 4; ++r) acc = ST_MFMA16(fr[ot * ITL + it][r], hin[it][r], acc);
            }
#endif
        }
#if ST_AE_ABLATE & 8
#pragma unroll
        for (int r = 0; r < 4; ++r) hout[ot][r] = fmaxf(acc[r], 0.1f * acc[r]);
#else
        hout[ot] = elu4(acc);
#endif
    }
}
// D layout -> T layout of TL 16x16 tiles through a wave-private LDS scratch ([tile][row 16][feature 16, pitch 20]): one
// ds_write_b128 + four ds_read_b32 per tile, no barrier (LDS operations of one wave execute in order).  The transposed
// operands of the weight gradients used to be RE-COMPUTED with the second MFMA orientation (280 MFMAs + their ELU /
// ELU' per group): in a kernel that is issue-bound, not MFMA-bound, the transpose is far cheaper.
template <int TL>
__device__ __forceinline__ void to_T(float* scr, const f32x4 (&d)[TL], f32x4 (&t)[TL], const int g, const int c)
{
#if ST_AE_ABLATE & 16
#pragma unroll
    for (int k = 0; k < TL; ++k) t[k] = d[k];
    return;
#endif
#pragma unroll
    for (int k = 0; k < TL; ++k) *reinterpret_cast<f32x4*>(scr + k * 320 + c * 20 + 4 * g) = d[k];
#pragma unroll
    for (int k = 0; k < TL; ++k)
#pragma unroll
        for (int r = 0; r < 4; ++r) t[k][r] = scr[k * 320 + (4 * g + r) * 20 + c];
}

// Round 4, 16-bit Linear layers: the ACTIVATION operand of a weight gradient (h^T: rows on the k index) is only ever used rounded to 16 bits -- so it is
// transposed IN 16 bits: the lane's packed quad (features 4g .. 4g + 3 of row c: the very value the forward MFMA chain consumes) goes to a [row][16 features]
// half-precision scratch with one ds_write_b64, and the LDS transpose read of gfx950 (ds_read_b64_tr_b16: lane i of a 16-lane group supplies the address of k row
// i >> 2, columns 4 (i & 3) .. + 3 of a [4 k][16] block and receives the four k of column i) hands lane (g, c) the rows 4g .. 4g + 3 of feature c, packed:
// ONE write + ONE read per 16 x 16 tile instead of one ds_write_b128 + four ds_read_b32 + two conversions.  (d A^T stays fp32: the bias gradient sums it unrounded.)
template <int BF> struct TTy { typedef f32x4 type; };
template <> struct TTy<1> { typedef s16x4 type; };
template <> struct TTy<2> { typedef s16x4 type; };
template <int TL, int BF>
__device__ __forceinline__ void to_Th(float* scr, const f32x4 (&d)[TL], typename TTy<BF>::type (&t)[TL], const int g, const int c)
{
    if constexpr (BF == 0) to_T<TL>(scr, d, t, g, c);
    else {
        short* hs = reinterpret_cast<short*>(scr);
#pragma unroll
        for (int k = 0; k < TL; ++k) *reinterpret_cast<s16x4*>(hs + k * 256 + c * 16 + 4 * g) = pack_h4<BF>(d[k]);
#pragma unroll
        for (int k = 0; k < TL; ++k)
            t[k] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(hs + k * 256 + (4 * g + (c >> 2)) * 16 + 4 * (c & 3)));
    }
}
template <int BF>
__device__ __forceinline__ typename TTy<BF>::type tt_splat(const float v)      // the same value in every row (a knob feature)
{
    if constexpr (BF == 0) return (f32x4){v, v, v, v};
    else return pack_h4<BF>((f32x4){v, v, v, v});
}
// wgrad_reg with the activation operand in its transposed type (fp32 quads, or packed 16-bit quads from to_Th)
template <int OTL, int ITL, int BF>
__device__ __forceinline__ void wgrad_regh(f32x4 (&dW)[OTL][ITL], float (&db)[OTL], const f32x4 (&daT)[OTL], const typename TTy<BF>::type (&hT)[ITL]);

template <int OTL, int ITL, int BF = 0>
__device__ __forceinline__ void dgradD_fr(const f32x4 (&fr)[ITL * OTL], const f32x4 (&da)[OTL], f32x4 (&dh)[ITL])
{
    s16x4 pd[OTL];
    if constexpr (BF) {
#pragma unroll
        for (int ot = 0; ot < OTL; ++ot) pd[ot] = pack_bf16x4(da[ot]);
    }
#pragma unroll
    for (int it = 0; it < ITL; ++it) {
        f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ot = 0; ot < OTL; ++ot) {
            if constexpr (BF) acc = ST_MFMA16B(frag_bits(fr[it * OTL + ot]), pd[ot], acc);
            else {
#pragma unroll
                for (int r = 0; r < 4; ++r) acc = ST_MFMA16(fr[it * OTL + ot][r], da[ot][r], acc);
            }
        }
        dh[it] = acc;
    }
}

template <int TL>
__device__ __forceinline__ void mul_elu_grad(f32x4 (&d)[TL], const f32x4 (&h)[TL])
{
#pragma unroll
    for (int t = 0; t < TL; ++t)
#pragma unroll
        // d * ELU'(a) = d * (1 + min(h, 0)).  min as v_med3_f32(h, 0, -FLT_MAX): fminf() is IEEE and costs a v_max (quieting) in front of
        // every v_min -- 68 extra VALU per 16-row group in kernels whose time is instruction count (h >= -1 always, no NaN by construction)
        for (int r = 0; r < 4; ++r) d[t][r] = __builtin_fmaf(d[t][r], __builtin_amdgcn_fmed3f(h[t][r], 0.f, -3.0e38f), d[t][r]);
}

// dW_l tile(ot,it) += sum_rows daT[ot] (x) hT[it] (result in D layout: o = 16ot+4g+r, i = 16it+c); db_l (lane c <-> o = 16 ot + c)
// accumulates the row sums of daT.  Tiles and sums persist in registers for the whole kernel.
template <int OTL, int ITL, int BF = 0>
__device__ __forceinline__ void wgrad_reg(f32x4 (&dW)[OTL][ITL], float (&db)[OTL], const f32x4 (&daT)[OTL], const f32x4 (&hT)[ITL])
{
    s16x4 pht[ITL];
    if constexpr (BF) {
#pragma unroll
        for (int it = 0; it < ITL; ++it) pht[it] = pack_bf16x4(hT[it]);      // k = the four rows 4g + r of this lane group
    }
#pragma unroll
    for (int ot = 0; ot < OTL; ++ot) {
        s16x4 pdt;
        if constexpr (BF) pdt = pack_bf16x4(daT[ot]);
#pragma unroll
        for (int it = 0; it < ITL; ++it) {
#if ST_AE_ABLATE & 4
            dW[ot][it][0] += daT[ot][0] * hT[it][0];
#else
            if constexpr (BF) dW[ot][it] = ST_MFMA16B(pdt, pht[it], dW[ot][it]);
            else {
#pragma unroll
                for (int r = 0; r < 4; ++r) dW[ot][it] = ST_MFMA16(daT[ot][r], hT[it][r], dW[ot][it]);
            }
#endif
        }
        db[ot] += (daT[ot][0] + daT[ot][1]) + (daT[ot][2] + daT[ot][3]);
    }
}
template <int OTL, int ITL, int BF>
__device__ __forceinline__ void wgrad_regh(f32x4 (&dW)[OTL][ITL], float (&db)[OTL], const f32x4 (&daT)[OTL], const typename TTy<BF>::type (&hT)[ITL])
{
    if constexpr (BF == 0) wgrad_reg<OTL, ITL, 0>(dW, db, daT, hT);
    else {
#pragma unroll
        for (int ot = 0; ot < OTL; ++ot) {
            const s16x4 pdt = pack_bf16x4(daT[ot]);
#pragma unroll
            for (int it = 0; it < ITL; ++it) dW[ot][it] = ST_MFMA16B(pdt, hT[it], dW[ot][it]);
            db[ot] += (daT[ot][0] + daT[ot][1]) + (daT[ot][2] + daT[ot][3]);
        }
    }
}
// Flush of one wave's persistent accumulators into the workgroup's LDS gradient image ([o][INP] row-major): the caller
// serialises the waves (wave 0 stores, barrier, wave 1 adds, ...) so the summation order, hence every bit, is fixed.
template <int OTL, int ITL, int INP, bool FIRST>
__device__ __forceinline__ void dw_flush(float* dW, const f32x4 (&acc)[OTL][ITL], const int g, const int c)
{
#pragma unroll
    for (int ot = 0; ot < OTL; ++ot)
#pragma unroll
        for (int it = 0; it < ITL; ++it)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float* p = dW + (16 * ot + 4 * g + r) * INP + 16 * it + c;
                if constexpr (FIRST) *p = acc[ot][it][r]; else *p += acc[ot][it][r];
            }
}
template <int OTL, bool FIRST>
__device__ __forceinline__ void db_flush(float* dst, float (&db)[OTL], const int g, const int c)
{
#pragma unroll
    for (int ot = 0; ot < OTL; ++ot) {
        float v = db[ot];
        v += __shfl_xor(v, 16); v += __shfl_xor(v, 32);
        if (g == 0) { if constexpr (FIRST) dst[16 * ot + c] = v; else dst[16 * ot + c] += v; }
    }
}

// INNER (wide geometries, st_ae_wide.h): only layers 2..8 -- layers 1 and 9 are feature-major GEMMs.  Pointer roles then:
//   mag / phs         -> H1 [64][R] of the two nets (layer-1 outputs, R = B*FP columns)
//   mag_hat / phs_hat -> dH8 [64][R] = W9^T dA9 (the kernel applies ELU'(h8) itself: h8 is recomputed)
//   dmag / dphs       -> dA1 [64][R] = (W2^T dA2) * ELU'(h1), consumed by the layer-1 weight/data-gradient GEMMs
// and the partial gradients of layers 1 and 9 stay zero.
constexpr int AE_BWD_SCR = (32 + 16 + 16) * SP + 2 * 4 * 320;      // per wave: V, Y, TAIL rows + two 4-tile transpose scratches (to_T)
// LDS of the backward kernel (floats): images + per-wave scratch during the loop, four per-wave gradient images at the end
constexpr int ae_bwd_lds_floats(int nw) { return (CL::BWD_TOTAL + nw * AE_BWD_SCR) > nw * CL::FWD_TOTAL ? (CL::BWD_TOTAL + nw * AE_BWD_SCR) : nw * CL::FWD_TOTAL; }
// VAR (fused geometries only): bit 0 = an upstream gradient w.r.t. mag_hat arrives (g_mag_hat, the autograd entry);
// bit 1 = T - OT == 16 (the default geometry): the skip-filter tails mag[b, T-OT+t', f] ARE the second input tile already in
// registers.  Both exist to cut global-load INSTRUCTIONS: the four waves of a workgroup issue their ~50 scattered dword loads
// per group at the same moment and queue at the CU's one address path (the "loads issue" stage was 11 % of a group).
// SAVED (round 6; fused geometries, fp32): no forward recompute -- the post-ELU activations come from the buffer the forward kernel kept (ae_fwd_kernel<.., SV>;
// ELU' needs only the OUTPUT of ELU).  A third of the kernel's MFMAs, the ELU transcendentals and the forward fragment reads go; with nothing left at the top
// of a group to hide a memory round trip behind, EVERY per-group input is loaded one full group ahead and IN PLACE: right behind the last use of a register
// set in this group, the same registers are refilled for the next one (no second buffer, no copies).
template <int NW, bool TIMED, bool INNER = false, int BF = 0, int VAR = 1, bool SAVED = false>      // BF: 16-bit operands in all Linear-layer products (ST_PREC_*_ALL)
__global__ void __launch_bounds__(NW * 64, 1)
ae_bwd_kernel(const float* __restrict__ mag, const float* __restrict__ phs, const float* __restrict__ knobs,
              const float* __restrict__ ae_m, const float* __restrict__ ae_p, const AEOffsets go, const int PG,
              const float* __restrict__ mag_hat, const float* __restrict__ phs_hat, const float* __restrict__ dAA,
              const float* __restrict__ g_mag_hat, const float reg_coef, const float expfac,
              float* __restrict__ dmag, float* __restrict__ dphs, float* __restrict__ ws,
              const int B, const int T, const int OT, const int F, const int K, const int KP,
              const int to_lo, const int to_hi,      // live synthesis frames: dAA rows outside are treated as zero
              const int nslab, const size_t slab,     // dAA arrives as split-K slabs of the synthesis dgrad GEMM
              const int dbg, const float* __restrict__ sv = nullptr)      // SAVED: the kept activations ([net][group][17 tiles][lane] float4)
{
    static_assert(!SAVED || (!INNER && BF == 0), "the kept-activation backward exists for the fused fp32 geometries");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int ae = blockIdx.y;
    const bool timing = TIMED && (dbg & 256) && blockIdx.x == 0 && blockIdx.y == 0 && (threadIdx.x >> 6) == 0;
    unsigned long long t0_ = timing ? __builtin_amdgcn_s_memtime() : 0ull;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, c = lane & 15;
    float* lw = lds;                                   // forward images + biases, then the dgrad images (CL)
    // wave-private scratch: V[32*SP] (input rows, transposed), Y[16*SP] (d a9 transposed), TAIL[16*SP]
    float* Vs = lds + CL::BWD_TOTAL + wave * AE_BWD_SCR;
    float* Ys = Vs + 32 * SP;
    float* Ts = Ys + 16 * SP;
    float* XH = Ts + 16 * SP;                          // transposes of activations
    float* XD = XH + 4 * 320;                          // transposes of activation gradients
    ae_load_lds<NW * 64, BF>(lw, ae ? ae_p : ae_m, go, INNER ? 16 : T, INNER ? 16 : OT, K, tid, INNER ? 1 : 0, INNER ? 8 : NL, true);
    __syncthreads();

    const float* vin = ae ? phs : mag;
    const float* dh8in = ae ? phs_hat : mag_hat;       // INNER only
    float* dvout = ae ? dphs : dmag;
    const int FP = KP / 2, gpw = FP / 16;
    const int ngroups = B * gpw;
    const int gstride = gridDim.x * NW;

    // 36 persistent 16x16 dW tiles (144 registers) + 17 bias-gradient registers
    f32x4 rW1[4][2], rW2[2][4], rW3[1][2], rW4[1][1], rW5[1][2], rW6[1][1], rW7[2][1], rW8[4][2], rW9[1][4];
    float rb1[4], rb2[2], rb3[1], rb4[1], rb5[1], rb6[1], rb7[2], rb8[4], rb9[1];
#define ST_ZT(x, A, Bq) { _Pragma("unroll") for (int a_ = 0; a_ < A; ++a_) _Pragma("unroll") for (int b_ = 0; b_ < Bq; ++b_) x[a_][b_] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
#define ST_ZB(x, A) { _Pragma("unroll") for (int a_ = 0; a_ < A; ++a_) x[a_] = 0.f; }
    ST_ZT(rW1, 4, 2) ST_ZT(rW2, 2, 4) ST_ZT(rW3, 1, 2) ST_ZT(rW4, 1, 1) ST_ZT(rW5, 1, 2) ST_ZT(rW6, 1, 1) ST_ZT(rW7, 2, 1) ST_ZT(rW8, 4, 2) ST_ZT(rW9, 1, 4)
    ST_ZB(rb1, 4) ST_ZB(rb2, 2) ST_ZB(rb3, 1) ST_ZB(rb4, 1) ST_ZB(rb5, 1) ST_ZB(rb6, 1) ST_ZB(rb7, 2) ST_ZB(rb8, 4) ST_ZB(rb9, 1)
#undef ST_ZT
#undef ST_ZB

    // input rows of the first group (prefetched one group ahead afterwards), D layout: t = 16 it + 4g + r, row c
    f32x4 vr[2];
    int grp = blockIdx.x * NW + wave;
    // Input rows of group gq as RAW loads from clamped (always valid) addresses; mask_v() zeroes the padding
    // rows/bins afterwards.  Keeping the select out of the load sequence (and the whole prefetch out of a
    // conditional block) matters: the `if (next < ngroups) { x = load; dst = ok ? x : 0; }` form compiled to
    // eight load / s_waitcnt vmcnt(0) pairs, i.e. eight serialized memory round trips per group.
    auto load_v = [&](int gq, f32x4 (&dst)[2]) {
        const int bq = gq / gpw, fq = (gq - bq * gpw) * 16 + c;
        const unsigned base = ST_MUL24(ST_MUL24(bq, T), F) + (unsigned)(fq < F ? fq : 0);
#pragma unroll
        for (int it = 0; it < 2; ++it)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int t = 16 * it + 4 * g + r;
                dst[it][r] = ldg32(vin, base + ST_MUL24(t < T ? t : 0, F));
            }
    };
    auto mask_v = [&](int gq, f32x4 (&dst)[2]) {
        const int bq = gq / gpw, fq = (gq - bq * gpw) * 16 + c;
        const bool ok0 = fq < F;
#pragma unroll
        for (int it = 0; it < 2; ++it)
#pragma unroll
            for (int r = 0; r < 4; ++r) dst[it][r] = (ok0 && 16 * it + 4 * g + r < T) ? dst[it][r] : 0.f;
    };
    if constexpr (!INNER) { if (grp < ngroups) { load_v(grp, vr); mask_v(grp, vr); } }
    // knob values of a group's window, also one group ahead (a select right behind their loads at the top of the loop body
    // made the wave wait there for ALL the loads just issued -- the memory counter is in-order -- i.e. one full memory
    // latency per group): D-layout feature tile (16 + 4g + r) and T-layout feature lane (16 + c)
    f32x4 kn = (f32x4){0.f, 0.f, 0.f, 0.f}; float knT = 0.f;
    auto load_kn = [&](int gq, f32x4& d4, float& dT) {
        const unsigned kb = ST_MUL24(gq / gpw, K);
#pragma unroll
        for (int r = 0; r < 4; ++r) { const int kidx = 4 * g + r; d4[r] = ldg32(knobs, kb + (unsigned)(kidx < K ? kidx : 0)); }
        dT = ldg32(knobs, kb + (unsigned)(c < K ? c : 0));
    };
    auto mask_kn = [&](f32x4& d4, float& dT) {
#pragma unroll
        for (int r = 0; r < 4; ++r) d4[r] = (4 * g + r) < K ? d4[r] : 0.f;
        dT = c < K ? dT : 0.f;
    };
    if (grp < ngroups) { load_kn(grp, kn, knT); mask_kn(kn, knT); }
    const unsigned Rw = (unsigned)B * FP;              // INNER: columns of the feature-major buffers

    constexpr bool GM = (VAR & 1) != 0, TAIL16 = (VAR & 2) != 0;
    // d-out inputs of a group (D layout: t' = 4g + r): RAW loads from clamped addresses, nothing consumed here
    float q_x[4][3], q_y[4][3], q_ph[4], q_mh[4], q_mt[4], q_gm[4];
    auto load_q = [&](const int gq) {
        const int bq = gq / gpw, fq0 = (gq - bq * gpw) * 16 + c;
        const bool fvq = fq0 < F;
        const int fq = fvq ? fq0 : 0;
        // wave-uniform base pointers for the slabs and the imaginary half: the six dAA loads of a row share ONE offset register
        const size_t o1 = nslab > 1 ? slab : 0, o2 = nslab > 2 ? 2 * slab : 0;
        const float* const dA0 = dAA, * const dA1 = dAA + o1, * const dA2 = dAA + o2;
        const float* const dB0 = dA0 + FP, * const dB1 = dA1 + FP, * const dB2 = dA2 + FP;
        const unsigned bOT = ST_MUL24(bq, OT), btF = ST_MUL24(ST_MUL24(bq, T), F) + (unsigned)fq;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int to = 4 * g + r;
            const bool ok = fvq && to < OT;
            const bool lv = ok && to >= to_lo && to <= to_hi;
            const unsigned ro = bOT + (unsigned)(ok ? to : 0);
            // up to 3 split-K slabs: all six loads are issued together (a runtime-trip-count loop here serialised ~12
            // memory round trips per group)
            const unsigned p0 = ST_MUL24(lv ? ro : 0u, KP) + (unsigned)fq;
            const unsigned pF = ST_MUL24(ro, F) + (unsigned)fq;
#if ST_AE_ABLATE & 1
            q_x[r][0] = q_x[r][1] = q_x[r][2] = q_y[r][0] = q_y[r][1] = q_y[r][2] = 1e-3f * (float)(p0 & 7); q_ph[r] = 0.3f; q_mh[r] = 0.2f + 1e-3f * (float)(pF & 3);
#else
            q_x[r][0] = ldg32(dA0, p0); q_x[r][1] = ldg32(dA1, p0); q_x[r][2] = ldg32(dA2, p0);
            q_y[r][0] = ldg32(dB0, p0); q_y[r][1] = ldg32(dB1, p0); q_y[r][2] = ldg32(dB2, p0);
            q_ph[r] = ldg32(phs_hat, pF); q_mh[r] = ldg32(mag_hat, pF);
#endif
            if constexpr (GM) q_gm[r] = ldg32(g_mag_hat, pF); else q_gm[r] = 0.f;
            if constexpr (!TAIL16) q_mt[r] = ldg32(vin, btF + ST_MUL24(ok ? T - OT + to : 0, F));
        }
    };
    // SAVED: the kept activations of this net, loop-carried -- tiles of the NEXT group replace a set right behind its last use
    f32x4 s_h1[4], s_h2[2], s_h3[1], s_h4[1], s_h5[1], s_h6[1], s_h7[2], s_h8[4], s_e9[1];
    const float* const svn = SAVED ? sv + (size_t)ae * ngroups * (AE_SV_TILES * 256) : nullptr;
    const unsigned lane16 = (unsigned)lane << 4;
#ifdef ST_SV_DIAG
    auto svp = [&](const int gq) { return svn + (size_t)(__builtin_amdgcn_readfirstlane(gq) & ST_SV_DIAG) * (AE_SV_TILES * 256); };
#else
    auto svp = [&](const int gq) { return svn + (size_t)__builtin_amdgcn_readfirstlane(gq) * (AE_SV_TILES * 256); };      // wave-uniform
#endif
    if constexpr (SAVED) {
        if (grp < ngroups) {
            load_q(grp);
            const float* const p = svp(grp);
            sv_load(p, lane16, 16, s_e9); sv_load(p, lane16, 12, s_h8); sv_load(p, lane16, 10, s_h7); sv_load(p, lane16, 9, s_h6); sv_load(p, lane16, 8, s_h5);
            sv_load(p, lane16, 7, s_h4); sv_load(p, lane16, 6, s_h3); sv_load(p, lane16, 4, s_h2); sv_load(p, lane16, 0, s_h1);
        }
    }

    for (; grp < ngroups; grp += gstride) {
        ST_T(16);
        asm volatile("" ::: "memory");      // keep the (loop-invariant) LDS weight fetches inside the loop
        const int b = grp / gpw, f = (grp - b * gpw) * 16 + c;
        const bool fv = f < F;
        // ---- d-out inputs for this group (D layout: t' = 4g + r), issued early.  (Slicing these 53 loads between the
        // forward stages to overlap their issue with MFMA execution was measured: no gain.)
        // Raw values only: the slab sums and masks are formed in the d-out stage -- arithmetic on a loaded value up here
        // makes the wave wait in the middle of the burst (the memory counter is in-order).
        if constexpr (!INNER && !SAVED) load_q(grp);
        // INNER: layer-1 outputs in both layouts and the gradient entering layer 8's output, straight from the
        // feature-major buffers (D layout: feature 16 tile + 4g + r at column col0 + c)
        f32x4 h1in[4], dh8[4];
        if constexpr (INNER) {
            const unsigned col0 = (unsigned)b * FP + (unsigned)(grp - b * gpw) * 16;
#pragma unroll
            for (int ot = 0; ot < 4; ++ot) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    h1in[ot][r] = ldg32(vin, (unsigned)(16 * ot + 4 * g + r) * Rw + col0 + c);
                    dh8[ot][r] = ldg32(dh8in, (unsigned)(16 * ot + 4 * g + r) * Rw + col0 + c);
                }
            }
        }
        f32x4 vn[2], knn; float knTn;
        const int gnext = grp + gstride < ngroups ? grp + gstride : grp;      // last iteration: harmless reload of this group
#if ST_AE_ABLATE & 64
        vn[0] = vr[0]; vn[1] = vr[1]; knn = kn; knTn = knT;
#else
        if constexpr (!INNER) load_v(gnext, vn);
        load_kn(gnext, knn, knTn);
#endif
        ST_T(0);

        // ------------------------------------------------------------------ forward recompute (D layout)
        // Rolling fragment prefetch: each stage first issues the LDS reads of the NEXT layer's fragments, then runs its own
        // MFMA chain, so no stage starts by waiting for an LDS round trip (one wave per SIMD: nothing else would hide it).
        f32x4 h1[4], h2[2], h3[1], h4[1], h5[1], h6[1], h7[2], h8[4], e9[1];
        f32x4 fr9[1 * 4];
        if constexpr (SAVED) {
#pragma unroll
            for (int it = 0; it < 2; ++it)
#pragma unroll
                for (int r = 0; r < 4; ++r) Vs[(16 * it + 4 * g + r) * SP + c] = vr[it][r];   // [feature t][row c]: read back transposed for the layer-1 wgrad
#pragma unroll
            for (int k = 0; k < 4; ++k) { h1[k] = s_h1[k]; h8[k] = s_h8[k]; }
#pragma unroll
            for (int k = 0; k < 2; ++k) { h2[k] = s_h2[k]; h7[k] = s_h7[k]; }
            h3[0] = s_h3[0]; h4[0] = s_h4[0]; h5[0] = s_h5[0]; h6[0] = s_h6[0]; e9[0] = s_e9[0];
        } else {
        f32x4 fr2[2 * 4];
        if constexpr (INNER) {
#pragma unroll
            for (int ot = 0; ot < 4; ++ot) h1[ot] = h1in[ot];
            frags_fwd<2, 4, CL::O1, BF>(fr2, lw + CL::A1, g, c);
        } else {
            f32x4 fr1[4 * 2]; frags_fwd<4, 2, CL::O0, BF>(fr1, lw + CL::A0, g, c);
            frags_fwd<2, 4, CL::O1, BF>(fr2, lw + CL::A1, g, c);
#pragma unroll
            for (int it = 0; it < 2; ++it)
#pragma unroll
                for (int r = 0; r < 4; ++r) Vs[(16 * it + 4 * g + r) * SP + c] = vr[it][r];   // [feature t][row c]: read back transposed for the layer-1 wgrad
            ST_FENCE();
            fwdD_fr<4, 2, BF>(fr1, lw + CL::B0, vr, h1, g);
        }
        ST_T(1);
        f32x4 fr3[1 * 2]; frags_fwd<1, 2, CL::O2, BF>(fr3, lw + CL::A2, g, c); ST_FENCE(); fwdD_fr<2, 4, BF>(fr2, lw + CL::B1, h1, h2, g);
        ST_T(2);
        f32x4 fr4[1 * 1]; frags_fwd<1, 1, CL::O3, BF>(fr4, lw + CL::A3, g, c); ST_FENCE(); fwdD_fr<1, 2, BF>(fr3, lw + CL::B2, h2, h3, g);
        f32x4 fr5[1 * 2]; frags_fwd<1, 2, CL::O4, BF>(fr5, lw + CL::A4, g, c); ST_FENCE(); fwdD_fr<1, 1, BF>(fr4, lw + CL::B3, h3, h4, g);
        f32x4 fr6[1 * 1]; frags_fwd<1, 1, CL::O5, BF>(fr6, lw + CL::A5, g, c); ST_FENCE();
        {
            const f32x4 hk[2] = {h4[0], kn};                                 // knob features 16 + 4g + r
            fwdD_fr<1, 2, BF>(fr5, lw + CL::B4, hk, h5, g);
        }
        ST_T(3);
        f32x4 fr7[2 * 1]; frags_fwd<2, 1, CL::O6, BF>(fr7, lw + CL::A6, g, c); ST_FENCE(); fwdD_fr<1, 1, BF>(fr6, lw + CL::B5, h5, h6, g);
        f32x4 fr8[4 * 2]; frags_fwd<4, 2, CL::O7, BF>(fr8, lw + CL::A7, g, c); ST_FENCE(); fwdD_fr<2, 1, BF>(fr7, lw + CL::B6, h6, h7, g);
        ST_T(4);
        if constexpr (!INNER) frags_fwd<1, 4, CL::O8, BF>(fr9, lw + CL::A8, g, c);
        ST_FENCE(); fwdD_fr<4, 2, BF>(fr8, lw + CL::B7, h7, h8, g);
        ST_T(5);
        }
        // ---- d out (D layout: t' = 4g + r), part A: everything that does not need e9 -- polar->rect backward of nn_proc.py:322-326 and
        // the L1 term of loss_functions.py:36 -- sits in the SAME scheduling region as the layer-9 MFMAs (one dependent chain of 16,
        // 512 cycles of matrix pipe with nothing else to issue) and is paced into their shadows: one MFMA, then seven VALU.
        f32x4 da9[1];
        float dxA[4], mtA[4];
        if constexpr (!INNER) {
            ST_FENCE();
            const float wf = fv ? expf(expfac * (float)f) : 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                // Branch-free on purpose: with the loads' only uses inside an `if (fv && to < OT)` block the compiler SANK ten
                // of them from the top of the loop body into that block, where they became three serialized memory round trips.
                const int to = 4 * g + r;
                const bool ok = fv && to < OT;
                const bool lv = ok && to >= to_lo && to <= to_hi;
                const float gre = lv ? q_x[r][0] + (nslab > 1 ? q_x[r][1] : 0.f) + (nslab > 2 ? q_x[r][2] : 0.f) : 0.f;
                const float gim = lv ? q_y[r][0] + (nslab > 1 ? q_y[r][1] : 0.f) + (nslab > 2 ? q_y[r][2] : 0.f) : 0.f;
                const float ph = q_ph[r], mh = q_mh[r];
                if constexpr (!TAIL16) asm volatile("" :: "v"(q_mt[r]));       // second use: a single-use load feeding a select is turned into a branch with the load sunk into it
                float sn, cs; st_sincos(ph, sn, cs);
                const float sg = mh > 0.f ? 1.f : (mh < 0.f ? -1.f : 0.f);
                const float dmh = gre * cs + gim * sn + reg_coef * sg * wf + (GM ? q_gm[r] : 0.f);
                const float dph = mh * (gim * cs - gre * sn);
                // one formula for both nets (x * 1.0f is exact): magnitude d9 = dmh * mag_tail * ELU', tail = dmh * e9; phase d9 = dph * ELU', tail = dph
                dxA[r] = ok ? (ae == 0 ? dmh : dph) : 0.f;
                mtA[r] = ae == 0 ? (TAIL16 ? vr[1][r] : q_mt[r]) : 1.f;      // TAIL16: t = T - OT + 4g + r = 16 + 4g + r is input tile 1 of this lane (already masked)
            }
            if constexpr (!SAVED) fwdD_fr<1, 4, BF>(fr9, lw + CL::B8, h8, e9, g);
            if constexpr (BF == 0 && !SAVED) {
#pragma unroll
                for (int p_ = 0; p_ < 16; ++p_) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, 7, 0); }
            }
            ST_FENCE();
        }
        ST_T(6);
        // ---- d out, part B: ELU'(a9) and the skip / residual tails
        if constexpr (!INNER) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int to = 4 * g + r;
                const float eg = elu_grad_from_out(e9[0][r]);
                const float e1 = ae == 0 ? e9[0][r] : 1.f;
                const float d9 = dxA[r] * mtA[r] * eg;             // dxA is already zero outside the valid rows / frames
                const float tail = dxA[r] * e1;
                da9[0][r] = d9;
                Ts[to * SP + c] = tail;
                Ys[to * SP + c] = d9;                      // [feature t'][row c] -> read back transposed below
            }
        }
        // SAVED: in-place refills for the NEXT group, each right behind the last use of its registers in this one (every load gets a whole group to land)
#define ST_REFILL(...) do { if constexpr (SAVED) { ST_FENCE(); __VA_ARGS__; ST_FENCE(); } } while (0)
        const float* const svq = SAVED ? svp(gnext) : nullptr;
        ST_REFILL(load_q(gnext); sv_load(svq, lane16, 16, s_e9));
        // ------------------------------------------------------------------ backward through the layers
        // T layout: lane (g,c), reg r  <->  row 4g + r, feature 16*tile + c.
        // Every stage l runs in the order   [hT_{l-1} = to_T(h_{l-1})]  ->  data gradient MFMAs (fragments fd_l were fetched
        // during the previous stage)  ->  ELU' and the to_T of da_{l-1} + the fragment fetch of stage l-1 ISSUED  ->  weight
        // gradient MFMAs of layer l.  The weight-gradient MFMAs depend on nothing issued in this stage except hT, so their
        // ~1k cycles in the MFMA pipe cover the LDS round trips of the transposes and the next fragments (with one wave per
        // SIMD nothing else would).
// Scheduling recipe of one backward stage (everything between its two fences): the N data-gradient MFMAs first, two
// weight-gradient MFMAs of slack for their results to land, then one weight-gradient MFMA per pair of VALU / LDS
// instructions (ELU', transposes, next fragments), so that the latter issue while the MFMA pipe is busy.
#define ST_PIPE(N_) do { \
        __builtin_amdgcn_sched_group_barrier(0x008, (N_) + 2, 0); \
        _Pragma("unroll") for (int p_ = 0; p_ < (N_) - 2; ++p_) { \
            __builtin_amdgcn_sched_group_barrier(0x002, 2, 0); __builtin_amdgcn_sched_group_barrier(0x300, 1, 0); \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); } \
        ST_FENCE(); } while (0)
#define ST_BWD_STAGE(O_, I_, FD_, DA_, DAT_, HP_, HTP_, DAP_, DATP_, RW_, RB_, NEXT_) \
        to_Th<I_, BF>(XH, HP_, HTP_, g, c); ST_FENCE(); \
        dgradD_fr<O_, I_, BF>(FD_, DA_, DAP_); mul_elu_grad<I_>(DAP_, HP_); \
        to_T<I_>(XD, DAP_, DATP_, g, c); NEXT_; \
        wgrad_regh<O_, I_, BF>(RW_, RB_, DAT_, HTP_); \
        ST_PIPE(BF ? O_ * I_ : O_ * I_ * 4);
        typedef typename TTy<BF>::type tt_t;      // transposed activation operands: fp32 quads, or packed 16-bit quads (to_Th)
        tt_t hT8[4]; f32x4 da8[4], daT8[4], fd8[2 * 4];
        if constexpr (INNER) {                     // dH8 arrives from the layer-9 data-gradient GEMM; h8^T is still needed for ELU' and dW8
            frags_dgrad<4, 2, CL::I7, BF>(fd8, lw + CL::G7, g, c);
            to_Th<4, BF>(XH, h8, hT8, g, c);
#pragma unroll
            for (int ot = 0; ot < 4; ++ot) da8[ot] = dh8[ot];
            mul_elu_grad<4>(da8, h8); to_T<4>(XD, da8, daT8, g, c);
            ST_T(7);
        } else {
            // layer 9 (64 -> OT): d a9 transposed through the wave's scratch
            f32x4 daT9[1], fd9[4 * 1];
            frags_dgrad<1, 4, CL::I8, BF>(fd9, lw + CL::G8, g, c);
            daT9[0] = *reinterpret_cast<const f32x4*>(Ys + c * SP + 4 * g);
            ST_T(7);
            ST_BWD_STAGE(1, 4, fd9, da9, daT9, h8, hT8, da8, daT8, rW9, rb9, (frags_dgrad<4, 2, CL::I7, BF>(fd8, lw + CL::G7, g, c)))
            ST_REFILL(sv_load(svq, lane16, 12, s_h8));
        }
        ST_T(8);
        // layer 8 (32 -> 64)
        tt_t hT7[2]; f32x4 da7[2], daT7[2], fd7[1 * 2];
        ST_BWD_STAGE(4, 2, fd8, da8, daT8, h7, hT7, da7, daT7, rW8, rb8, (frags_dgrad<2, 1, CL::I6, BF>(fd7, lw + CL::G6, g, c)))
        ST_REFILL(sv_load(svq, lane16, 10, s_h7));
        ST_T(9);
        // layer 7 (16 -> 32)
        tt_t hT6[1]; f32x4 da6[1], daT6[1], fd6[1];
        ST_BWD_STAGE(2, 1, fd7, da7, daT7, h6, hT6, da6, daT6, rW7, rb7, (frags_dgrad<1, 1, CL::I5, BF>(fd6, lw + CL::G5, g, c)))
        ST_REFILL(sv_load(svq, lane16, 9, s_h6));
        ST_T(10);
        // layer 6 (16 -> 16)
        tt_t hT5[1]; f32x4 da5[1], daT5[1], fd5[1];
        ST_BWD_STAGE(1, 1, fd6, da6, daT6, h5, hT5, da5, daT5, rW6, rb6, (frags_dgrad<1, 1, CL::I4, BF>(fd5, lw + CL::G4, g, c)))
        ST_REFILL(sv_load(svq, lane16, 8, s_h5));
        // layer 5 ([h4 ; knobs] -> 16): weight gradient over both input tiles, data gradient to h4 only
        tt_t hT4[1], hT4k[2]; f32x4 da4[1], daT4[1], fd4[1];
        {
            to_Th<1, BF>(XH, h4, hT4, g, c); ST_FENCE();
            dgradD_fr<1, 1, BF>(fd5, da5, da4); mul_elu_grad<1>(da4, h4);
            to_T<1>(XD, da4, daT4, g, c); frags_dgrad<1, 1, CL::I3, BF>(fd4, lw + CL::G3, g, c); ST_FENCE();
            hT4k[0] = hT4[0];
            hT4k[1] = tt_splat<BF>(knT);                               // features 16 + c = knob c, every row
            wgrad_regh<1, 2, BF>(rW5, rb5, daT5, hT4k);
        }
        ST_REFILL(sv_load(svq, lane16, 7, s_h4));
        // layer 4 (16 -> 16)
        tt_t hT3[1]; f32x4 da3[1], daT3[1], fd3[2 * 1];
        ST_BWD_STAGE(1, 1, fd4, da4, daT4, h3, hT3, da3, daT3, rW4, rb4, (frags_dgrad<1, 2, CL::I2, BF>(fd3, lw + CL::G2, g, c)))
        ST_REFILL(sv_load(svq, lane16, 6, s_h3));
        ST_T(11);
        // layer 3 (32 -> 16)
        tt_t hT2[2]; f32x4 da2[2], daT2[2], fd2[4 * 2];
        ST_BWD_STAGE(1, 2, fd3, da3, daT3, h2, hT2, da2, daT2, rW3, rb3, (frags_dgrad<2, 4, CL::I1, BF>(fd2, lw + CL::G1, g, c)))
        ST_REFILL(sv_load(svq, lane16, 4, s_h2));
        ST_T(12);
        // layer 2 (64 -> 32)
        tt_t hT1[4]; f32x4 da1[4], daT1[4], fd1[2 * 4];
        if constexpr (INNER) {                     // dA1 goes back to memory for the layer-1 GEMMs
            to_Th<4, BF>(XH, h1, hT1, g, c); ST_FENCE();
            dgradD_fr<2, 4, BF>(fd2, da2, da1); mul_elu_grad<4>(da1, h1);
            const unsigned col0 = (unsigned)b * FP + (unsigned)(grp - b * gpw) * 16;
#pragma unroll
            for (int ot = 0; ot < 4; ++ot)
#pragma unroll
                for (int r = 0; r < 4; ++r) stg32(dvout, (unsigned)(16 * ot + 4 * g + r) * Rw + col0 + c, da1[ot][r]);
            ST_FENCE();
            wgrad_regh<2, 4, BF>(rW2, rb2, daT2, hT1);
        } else {
            ST_BWD_STAGE(2, 4, fd2, da2, daT2, h1, hT1, da1, daT1, rW2, rb2, (frags_dgrad<4, 2, CL::I0, BF>(fd1, lw + CL::G0, g, c)))
            ST_REFILL(sv_load(svq, lane16, 0, s_h1));
        }
        ST_T(13);
        // layer 1 (T -> 64): input rows transposed through the wave's scratch
        f32x4 vT[2], dv[2];
        if constexpr (!INNER) {
#pragma unroll
            for (int it = 0; it < 2; ++it) vT[it] = *reinterpret_cast<const f32x4*>(Vs + (16 * it + c) * SP + 4 * g);
            ST_FENCE();
            dgradD_fr<4, 2, BF>(fd1, da1, dv);
            wgrad_reg<4, 2, BF>(rW1, rb1, daT1, vT);
        }
#undef ST_BWD_STAGE
#undef ST_PIPE
#undef ST_REFILL
        ST_T(14);
        // Round 6: the NEXT group's input rows / knobs are taken over HERE, ahead of this group's stores.  Behind them (as until round 5) the wait for these
        // loads -- the memory counter is in-order and counts stores too -- also waited for the write acknowledgements of the stores just issued: one exposed
        // round trip per group, the "loads issue" stage that stayed at ~10 % of a group through rounds 2-5 whatever was done to the loads themselves.
        if constexpr (!INNER) {
            mask_v(gnext, vn);
            vr[0] = vn[0]; vr[1] = vn[1];
        }
        mask_kn(knn, knTn); kn = knn; knT = knTn;
        ST_T(17);
        // ------------------------------------------------------------------ d input rows (+ skip / residual tails)
        if constexpr (!INNER) {
        // Materialise the accumulators in VGPRs HERE, in the block of the MFMAs that produce them: the stores below sit in
        // conditional blocks, and an accumulator first read behind a skipped block would be read before the MFMA has
        // finished (the hazard class tools/check_mfma_hazards.py scans for).
        float dvs[2][4], tls[2][4];
#pragma unroll
        for (int it = 0; it < 2; ++it)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                dvs[it][r] = dv[it][r]; asm volatile("" : "+v"(dvs[it][r]));
                const int ti = 16 * it + 4 * g + r - (T - OT);                 // all eight tail reads up front: one LDS round trip, not one per store
                tls[it][r] = Ts[(ti >= 0 && ti < OT ? ti : 0) * SP + c];
            }
        const unsigned dv0 = ST_MUL24(ST_MUL24(b, T), F) + (unsigned)f;
#pragma unroll
        for (int it = 0; it < 2; ++it)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int t = 16 * it + 4 * g + r;
                const float v = dvs[it][r] + (t >= T - OT ? tls[it][r] : 0.f);
#if ST_AE_ABLATE & 2
                asm volatile("" :: "v"(v));
#else
                if (fv && t < T) stg32(dvout, dv0 + ST_MUL24(t, F), v);
#endif
            }
        }
        ST_T(15);
    }
    // ---------------------------------------------------------------------- workgroup partial gradients
    // All LDS contents are dead now.  Every wave stores its accumulators into ITS OWN gradient image (dW_l as [o][INp] at
    // CL::A_l, db_l at CL::B_l, images CL::FWD_TOTAL floats apart), all four in parallel, and the store pass below sums the
    // images in a fixed order ((0 + 1) + (2 + 3)): run-to-run identical bits without the serialised wave-by-wave
    // read-modify-write flush the first versions used (three extra barrier-separated LDS passes).
    __syncthreads();
    static_assert(NW == 4, "the store pass sums exactly four per-wave images");
    float* dwl = lds + wave * CL::FWD_TOTAL;
    if constexpr (!INNER) { dw_flush<4, 2, CL::I0, true>(dwl + CL::A0, rW1, g, c); dw_flush<1, 4, CL::I8, true>(dwl + CL::A8, rW9, g, c);
                            db_flush<4, true>(dwl + CL::B0, rb1, g, c); db_flush<1, true>(dwl + CL::B8, rb9, g, c); }
    dw_flush<2, 4, CL::I1, true>(dwl + CL::A1, rW2, g, c); dw_flush<1, 2, CL::I2, true>(dwl + CL::A2, rW3, g, c);
    dw_flush<1, 1, CL::I3, true>(dwl + CL::A3, rW4, g, c); dw_flush<1, 2, CL::I4, true>(dwl + CL::A4, rW5, g, c);
    dw_flush<1, 1, CL::I5, true>(dwl + CL::A5, rW6, g, c); dw_flush<2, 1, CL::I6, true>(dwl + CL::A6, rW7, g, c);
    dw_flush<4, 2, CL::I7, true>(dwl + CL::A7, rW8, g, c);
    db_flush<2, true>(dwl + CL::B1, rb2, g, c); db_flush<1, true>(dwl + CL::B2, rb3, g, c); db_flush<1, true>(dwl + CL::B3, rb4, g, c);
    db_flush<1, true>(dwl + CL::B4, rb5, g, c); db_flush<1, true>(dwl + CL::B5, rb6, g, c); db_flush<2, true>(dwl + CL::B6, rb7, g, c);
    db_flush<4, true>(dwl + CL::B7, rb8, g, c);
    __syncthreads();
    auto sum4 = [&](int idx) { return (lds[idx] + lds[CL::FWD_TOTAL + idx]) + (lds[2 * CL::FWD_TOTAL + idx] + lds[3 * CL::FWD_TOTAL + idx]); };
    // Packed partial gradient of this workgroup (layout of the parameter block).  One unrolled pass: every layer's elements per
    // thread are compile-time bounded, so the LDS reads batch up and the index arithmetic folds (the first version zeroed the
    // whole block, synchronised, then ran nine run-time loops with a run-time division per element: 14-22 us per launch).
    float* base = ws + ((size_t)blockIdx.x * 2 + ae) * PG;
    const int out[NL] = {64, 32, 16, 16, 16, 16, 32, 64, OT};
    const int in[NL] = {T, 64, 32, 16, 16 + K, 16, 16, 32, 64};
    const int inp[NL] = {CL::I0, CL::I1, CL::I2, CL::I3, CL::I4, CL::I5, CL::I6, CL::I7, CL::I8};
    const int ao[NL] = {CL::A0, CL::A1, CL::A2, CL::A3, CL::A4, CL::A5, CL::A6, CL::A7, CL::A8};
    const int bo[NL] = {CL::B0, CL::B1, CL::B2, CL::B3, CL::B4, CL::B5, CL::B6, CL::B7, CL::B8};
    constexpr int NT = NW * 64;
#pragma unroll
    for (int l = 0; l < NL; ++l) {
        const bool have = !INNER || (l >= 1 && l < 8);                  // INNER: layers 1 and 9 come from the GEMM path, their partials stay zero
        const int IN = in[l], n = out[l] * IN;
#pragma unroll
        for (int u = 0; u < (ae_max_elems(l) + NT - 1) / NT; ++u) {
            const int e = tid + u * NT;
            if (e < n) { const int o = e / IN, i = e - o * IN; base[go.w[l] + e] = have ? sum4(ao[l] + o * inp[l] + i) : 0.f; }
        }
        if (tid < out[l]) base[go.b[l] + tid] = have ? sum4(bo[l] + tid) : 0.f;
        // alignment pads behind the weight and the bias tensor
        { const int p0 = go.w[l] + n, np = go.b[l] - p0; if (tid < np) base[p0 + tid] = 0.f; }
        { const int p0 = go.b[l] + out[l], np = (l + 1 < NL ? go.w[l + 1] : PG) - p0; if (tid < np) base[p0 + tid] = 0.f; }
    }
}

}  // namespace sta
