// st_ae_wide.h -- the knob-conditioned autoencoders (nn_proc.py:28-126) for WIDE geometries (T > 32 or OT > 16, e.g. the
// 65536-sample window of BASELINE configs[4]: T = 174, OT = 46), where the first/last layers no longer fit the fused
// register/LDS kernels of st_ae.h (64 x 174 weights + 44 persistent dW tiles per wave).
//
// Layout decision: activations are kept FEATURE-MAJOR over the whole batch, X[feature][R] with R = B * FP columns
// (column = b * FP + f, FP = roundup(F, 16) "virtual bins", pad columns hold zeros).  Then every layer of the MLP is a
// plain 2-D GEMM on the one MFMA GEMM family (st_gemm.h), with the fusions moved into epilogues:
//     forward   H_l  [OUT][R] = ELU( W_l [OUT][IN] . H_{l-1} [IN][R] + b_l )            (ActStore / OutStore for layer 9)
//     dgrad     dA_{l-1}[IN][R] = ( W_l^T . dA_l ) * ELU'(H_{l-1})                        (DgradStore / DvStore for layer 1)
//     wgrad     dW_l [OUT][IN] = dA_l [OUT][R] . H_{l-1}[IN][R]^T   (K = R, split-K slabs, summed in slab order)
// The knobs (nn_proc.py:92-93, concatenated after the 16-wide code) are K extra ROWS of the layer-5 input, so layer 5
// and its weight gradient need no special case.  Weight matrices whose row length is not a multiple of the k-tile
// (W_1: T, W_5: 16 + K) are re-packed zero-padded once per call (pad_rows_kernel).
#pragma once
#include "st_common.h"
#include "st_gemm.h"
#include <type_traits>
#include "st_ae.h"
#include "st_misc.h"

namespace stw {
using stg::NJ;
using stg::d_row;

// ------------------------------------------------------------------------------------------------ epilogues
// All epilogues: a lane owns 3 columns (col = n0 + 32 j + lane%32) and 16 rows of the 32 x 96 wave tile; everything that
// depends on the column only (validity, batch index b = col / FP, bin f) is computed once per column, not per element
// (the integer division is ~20 instructions).
struct ColInfo { int col, b, f; bool ok, real; };
__device__ __forceinline__ void col_info(ColInfo (&ci)[NJ], int n0, int lane, int R, int FP, int F)
{
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int col = n0 + 32 * j + (lane & 31);
        ci[j].col = col; ci[j].ok = col < R;
        ci[j].b = col / FP; ci[j].f = col - ci[j].b * FP;
        ci[j].real = ci[j].ok && ci[j].f < F;
    }
}

struct ActStore {          // layers 1..8 forward: out[row][col] = ELU(acc + bias[row]) on real bins, 0 on pad columns
    float* out; const float* bias; int M, R, FP, F;
    __device__ void operator()(int m0, int n0, const f32x16 (&acc)[NJ]) const {
        const int lane = threadIdx.x & 63;
        ColInfo ci[NJ]; col_info(ci, n0, lane, R, FP, F);
        float bvv[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) { const int row = m0 + d_row(i, lane); bvv[i] = bias[row < M ? row : M - 1]; }      // one batch of loads, not one wait per row
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int row = m0 + d_row(i, lane);
            if (row < M) {
                const float bv = bvv[i];
#pragma unroll
                for (int j = 0; j < NJ; ++j)
                    if (ci[j].ok) out[(size_t)row * R + ci[j].col] = ci[j].real ? elu_f(acc[j][i] + bv) : 0.f;
            }
        }
    }
};

struct OutStore {          // layer 9 forward (nn_proc.py:113-117, :322): e = ELU(a9); 'sf': e * input tail; phase: e + input tail
    float* e9; float* outp; const float* tail; const float* bias; int M, R, FP, F, mode;      // M = OT; outp [B][OT][F] or null
    __device__ void operator()(int m0, int n0, const f32x16 (&acc)[NJ]) const {
        const int lane = threadIdx.x & 63;
        ColInfo ci[NJ]; col_info(ci, n0, lane, R, FP, F);
        // all 48 tail values (and the 16 biases) are requested BEFORE the first is used, from clamped -- always valid -- addresses: inside the
        // predicated store loop each load was followed by its own wait (48 dependent round trips: this epilogue was most of a 32 us launch for 0.2 GFLOP)
        float tl[16][NJ], bv[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int row = m0 + d_row(i, lane), rc = row < M ? row : M - 1;
            bv[i] = bias[rc];
#pragma unroll
            for (int j = 0; j < NJ; ++j) tl[i][j] = outp ? tail[(size_t)rc * R + (ci[j].ok ? ci[j].col : 0)] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int row = m0 + d_row(i, lane);
            if (row < M) {
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    if (ci[j].ok) {
                        const float e = elu_f(acc[j][i] + bv[i]);
                        const size_t ix = (size_t)row * R + ci[j].col;
                        e9[ix] = ci[j].real ? e : 0.f;
                        if (outp && ci[j].real) outp[((size_t)ci[j].b * M + row) * F + ci[j].f] = mode ? e + tl[i][j] : e * tl[i][j];
                    }
                }
            }
        }
    }
};

struct DgradStore {        // dA_{l-1} = (W_l^T dA_l) * ELU'(h_{l-1}), ELU' from the stored output: h > 0 ? 1 : h + 1
    float* out; const float* H; int M, R;
    __device__ void operator()(int m0, int n0, const f32x16 (&acc)[NJ]) const {
        const int lane = threadIdx.x & 63;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int row = m0 + d_row(i, lane);
            if (row < M) {
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    const int col = n0 + 32 * j + (lane & 31);
                    if (col < R) { const size_t ix = (size_t)row * R + col; out[ix] = acc[j][i] * elu_grad_from_out(H[ix]); }
                }
            }
        }
    }
};

struct DvStore {           // gradient w.r.t. the AE input rows, written in the [B][T][F] layout of mag / phs (+ skip / residual tails)
    float* dv; const float* tail; int T, OT, R, FP, F;
    __device__ void operator()(int m0, int n0, const f32x16 (&acc)[NJ]) const {
        const int lane = threadIdx.x & 63;
        ColInfo ci[NJ]; col_info(ci, n0, lane, R, FP, F);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int row = m0 + d_row(i, lane);
            if (row < T) {
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    if (ci[j].real) {
                        float v = acc[j][i];
                        if (row >= T - OT) v += tail[(size_t)(row - (T - OT)) * R + ci[j].col];
                        dv[((size_t)ci[j].b * T + row) * F + ci[j].f] = v;
                    }
                }
            }
        }
    }
};

// ------------------------------------------------------------------------------------------------ layer-1 data gradient + polar backward
// Round 3.  At the 65536-sample window the gradient w.r.t. the autoencoder inputs was two generic GEMMs with a scatter epilogue (DvStore:
// 36 us each for 0.75 GFLOP) followed by the polar backward (29 us): dmag / dphs went out to HBM (46 MB) only to be read back.  One kernel now
// walks 16-row groups like the fused autoencoder kernels: d a1 of both nets in D layout (16 + 16 dwords per lane from the feature-major
// buffers), W1^T fragments from an LDS dgrad image, one 16-frame output tile at a time
//     dv[t][row] = sum_o W1[o][t] * d a1[o][row]  (+ the skip / residual tail for the last OT frames)
// and -- fused step -- straight through nn_proc.py:309-310's backward into d G in the weight-gradient GEMM's operand type.
// re == NULL (per-op st_ae_bwd): dmag / dphs only.  BF: 16-bit operands (both rounded, as gemm_half_kernel did).
struct DvPolarArgs {
    const float *DA1m, *DA1p, *TLm, *TLp, *W1m, *W1p;       // d a1 [64][R], tails [OT][R], layer-1 weights [64][T] of the two nets
    const float *re, *im, *g_mag;                           // forward state [B][T][F] (NULL: no polar stage); optional upstream d mag
    float *dmag, *dphs, *dG; unsigned short* dG16;          // any of them may be NULL
    int ht; float sat; int B, T, OT, F, FP, KP;
};
template <int BF>
__global__ void __launch_bounds__(512)
wide_dv_polar_kernel(const DvPolarArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float dvp_lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, c = lane & 15;
    const int T = a.T, TP = (T + 15) / 16 * 16, NIT = TP / 16;
    constexpr int RW = BF ? 32 : 64;                        // floats per image row block: 16-bit images take half the room
    float* const img[2] = {dvp_lds, dvp_lds + RW * TP};      // G[(o >> 2)][i][o & 3] per net
    const int gpw = a.FP / 16, ngroups = a.B * gpw;
    const unsigned R = (unsigned)a.B * a.FP;
    // The NWV waves of a workgroup take ADJACENT 16-bin groups and walk the frame tiles together: a group touches only 64 bytes of every
    // [B][T][F] row, eight neighbours touch 512 contiguous bytes at about the same time (DRAM page / L2 line locality: with the waves on
    // different frames of ONE group the kernel ran at 1.3 TB/s).
    // Round 4: the unit of work is (block of NWV adjacent groups, frame tile), and every workgroup takes an equal, contiguous share of the units.
    // Whole groups per workgroup left 124 of 256 CUs idle at the 65536-sample window (2112 groups = 264 blocks: 132 workgroups took two blocks
    // each, i.e. 22 dependent tile steps, the rest none); now every workgroup walks 11 or 12 tile steps and re-reads d a1 where its share
    // crosses a block boundary.
    constexpr int NWV = 8;
    const int nblk = (ngroups + NWV - 1) / NWV, nitems = nblk * NIT;
    const int i_lo = (int)((long long)blockIdx.x * nitems / (int)gridDim.x), i_hi = (int)((long long)(blockIdx.x + 1) * nitems / (int)gridDim.x);
    struct Geo { int it, b, f; bool gv, fv; unsigned col; };
    auto geo_of = [&](const int item) {
        Geo q;
        const int gb = item / NIT; q.it = item - gb * NIT;
        const int gw = gb * NWV + wave;
        const int grp = gw < ngroups ? gw : ngroups - 1;
        q.gv = gw < ngroups;
        q.b = grp / gpw; q.f = (grp - q.b * gpw) * 16 + c;
        q.fv = q.gv && q.f < a.F;
        q.col = (unsigned)q.b * a.FP + (unsigned)q.f;
        return q;
    };
    f32x4 da[2][4];
    sta::s16x4 pd[2][4];
    int cur_gb = -1;
    float xr[3][4], yr[3][4], tm[3][4], tp[3][4], gm[3][4];
    auto load_tile = [&](const int item, const int s_) {
        const Geo q = geo_of(item);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int t = 16 * q.it + 4 * g + r;
            const bool ok = q.fv && t < T;
            const unsigned o = ((unsigned)q.b * T + (unsigned)(ok ? t : 0)) * (unsigned)a.F + (unsigned)(q.fv ? q.f : 0);
            const bool tl = ok && t >= T - a.OT;
            const unsigned qq = (unsigned)(tl ? t - (T - a.OT) : 0) * R + q.col;
            xr[s_][r] = a.re ? sta::ldg32(a.re, o) : 0.f; yr[s_][r] = a.re ? sta::ldg32(a.im, o) : 0.f;
            gm[s_][r] = a.g_mag ? sta::ldg32(a.g_mag, o) : 0.f;
            const float u = sta::ldg32(a.TLm, qq), v = sta::ldg32(a.TLp, qq);
            tm[s_][r] = tl ? u : 0.f; tp[s_][r] = tl ? v : 0.f;
        }
    };
    // two frame tiles per trip so that the look-ahead buffers are indexed statically (a run-time buffer index put them in scratch)
    auto tile = [&](const int item, auto sb) {
        constexpr int SB = decltype(sb)::value;
        const Geo q = geo_of(item);
        const int it = q.it, b = q.b, f = q.f;
        const bool gv = q.gv, fv = q.fv;
        if (item / NIT != cur_gb) {                          // workgroup-uniform: a new block of groups -- its d a1 (D layout: 16 + 16 dwords per lane from the feature-major buffers)
            cur_gb = item / NIT;
#pragma unroll
            for (int ot = 0; ot < 4; ++ot)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    da[0][ot][r] = sta::ldg32(a.DA1m, (unsigned)(16 * ot + 4 * g + r) * R + q.col);
                    da[1][ot][r] = sta::ldg32(a.DA1p, (unsigned)(16 * ot + 4 * g + r) * R + q.col);
                }
            if constexpr (BF) {
#pragma unroll
                for (int n = 0; n < 2; ++n)
#pragma unroll
                    for (int ot = 0; ot < 4; ++ot) pd[n][ot] = sta::pack_h4<BF>(da[n][ot]);
            }
        }
        f32x4 acc[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int ot = 0; ot < 4; ++ot)
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                const f32x4 w = sta::frag_read<BF>(img[n], ((4 * ot + g) * TP + 16 * it + c) << 2);
                if constexpr (BF) acc[n] = sta::mfma16h<BF>(sta::frag_bits(w), pd[n][ot], acc[n]);
                else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[n] = ST_MFMA16(w[r], da[n][ot][r], acc[n]);
                }
            }
        // tile `it` of dv in D layout: lane (g, c), register r <-> frame t = 16 it + 4 g + r, row c
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int t = 16 * it + 4 * g + r;
            if (t >= T) continue;
            const unsigned o = ((unsigned)b * T + (unsigned)t) * (unsigned)a.F + (unsigned)(fv ? f : 0);
            const float dm = fv ? acc[0][r] + tm[SB][r] : 0.f, dp = fv ? acc[1][r] + tp[SB][r] : 0.f;
            if (a.dmag && fv) { sta::stg32(a.dmag, o, dm); sta::stg32(a.dphs, o, dp); }
            if (a.re) {
                float gre = 0.f, gim = 0.f;
                if (fv) {                      // stm::polar_bwd_block's formulas (v_rcp_f32 / v_sqrt_f32: 1 ulp, against a 1e-4 tolerance)
                    const float x = xr[SB][r], y = yr[SB][r];
                    const float dmt = dm + gm[SB][r];
                    const float m2 = x * x + y * y;
                    const float inv = m2 > 0.f ? __builtin_amdgcn_rcpf(__builtin_amdgcn_sqrtf(m2)) : 0.f;
                    const float rp = x + 1e-7f;
                    const float rden = __builtin_amdgcn_rcpf(rp * rp + y * y);
                    gre = dmt * x * inv - dp * y * rden;
                    gim = dmt * y * inv + dp * rp * rden;
                    if (a.sat > 0.f) { gre = __builtin_amdgcn_fmed3f(gre, -a.sat, a.sat); gim = __builtin_amdgcn_fmed3f(gim, -a.sat, a.sat); }
                }
                const size_t go = ((size_t)b * T + t) * a.KP + f;          // f < FP = KP / 2: pad columns get zeros
                if (gv && a.dG16) { a.dG16[go] = st_to_h16(gre, a.ht); a.dG16[go + a.FP] = st_to_h16(gim, a.ht); }
                if (gv && a.dG) { a.dG[go] = gre; a.dG[go + a.FP] = gim; }
            }
        }
    };
    // the first two items' inputs and the first block's d a1 are requested BEFORE the layer-1 weights are staged into LDS (dependent round trips, ~10 us
    // of this kernel): they travel while the staging runs
    if (i_lo < i_hi) {
        const Geo q0 = geo_of(i_lo);
        cur_gb = i_lo / NIT;
#pragma unroll
        for (int ot = 0; ot < 4; ++ot)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                da[0][ot][r] = sta::ldg32(a.DA1m, (unsigned)(16 * ot + 4 * g + r) * R + q0.col);
                da[1][ot][r] = sta::ldg32(a.DA1p, (unsigned)(16 * ot + 4 * g + r) * R + q0.col);
            }
        load_tile(i_lo, 0);
        load_tile(i_lo + 1 < i_hi ? i_lo + 1 : i_lo, 1);
    }
    for (int e = tid; e < 2 * RW * TP; e += 512) dvp_lds[e] = 0.f;
    __syncthreads();
    // thread i < T owns input column i of W1 [64][T] (coalesced across threads, NO integer division: e / T for 22 k elements was 6 k of this
    // kernel's 9 k vector instructions per wave), sixteen loads in flight
    // (threads [0, T) stage the first net, [T, 2 T) the second: four batches of loads per thread instead of eight -- the staging is dependent round
    // trips, ~10 us of a 55 us kernel in which every wave handles about one row group)
    if (tid < 2 * T) {
        const int n = tid >= T, ti = tid - n * T;
        {
            const float* W = n ? a.W1p : a.W1m;
            float* const im = n ? img[1] : img[0];
#pragma unroll
            for (int o0 = 0; o0 < 64; o0 += 16) {
                float v[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) v[u] = W[(o0 + u) * T + ti];
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    const int o = o0 + u;
                    const int idx = (((o >> 2) * TP + ti) << 2) + (o & 3);
                    if constexpr (BF) reinterpret_cast<unsigned short*>(im)[idx] = sta::st_half_bits<BF>(v[u]);
                    else im[idx] = v[u];
                }
            }
        }
    }
    __syncthreads();
    if constexpr (BF) {
        if (i_lo < i_hi) {
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int ot = 0; ot < 4; ++ot) pd[n][ot] = sta::pack_h4<BF>(da[n][ot]);
        }
    }
    // TWO tiles of look-ahead (three statically indexed slots; the balanced form of the kernel needs 118 registers, the whole-group form sat at the
    // 256-register line with one): the inputs of items i + 1 and i + 2 are in flight while item i is computed
    for (int item = i_lo; item < i_hi; item += 3) {
        const int last = i_hi - 1;
        load_tile(item + 2 < i_hi ? item + 2 : last, 2);
        tile(item, std::integral_constant<int, 0>{});
        if (item + 1 < i_hi) {                             // workgroup-uniform
            load_tile(item + 3 < i_hi ? item + 3 : last, 0);
            tile(item + 1, std::integral_constant<int, 1>{});
        }
        if (item + 2 < i_hi) {
            load_tile(item + 4 < i_hi ? item + 4 : last, 1);
            tile(item + 2, std::integral_constant<int, 2>{});
        }
    }
}

// ------------------------------------------------------------------------------------------------ small kernels
__global__ void pad_rows_kernel(const float* __restrict__ src, int rows, int cols, float* __restrict__ dst, int pitch)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * pitch) return;
    const int r = i / pitch, c = i - r * pitch;
    dst[i] = c < cols ? src[r * cols + c] : 0.f;
}

// The zero-padded copies of W_1 ([64][Tp]) and W_5 ([16][32]) of both autoencoders ride in this kernel (per-op entry) or in prep_kernel (fused step): stm::PadJobs
using stm::PadJobs;
using stm::pad_rows4_block;
// Inputs in the feature-major layout: V[a][t][b*FP + f] = (mag | phs)[b][t][f]; knob rows 16.. of the layer-5 input.
// Flattened over (row, window, quad of bins): one thread = four bins of both nets = eight 4-byte loads (the [B][T][F] rows are 4-byte aligned
// only) and two 16-byte stores (FP % 4 == 0).  Round 3: as one block per (row, window) with a 256-stride loop over 528 bins -- the third
// trip 16 lanes wide -- this copy took 23 us for 91 MB.  Blocks [n_copy, n_copy + n_pad): the weight padding jobs.
__global__ void __launch_bounds__(256)
wide_in_kernel(const float* __restrict__ mag, const float* __restrict__ phs, const float* __restrict__ knobs,
               float* __restrict__ Vm, float* __restrict__ Vp, float* __restrict__ H4Km, float* __restrict__ H4Kp,
               int B, int T, int F, int FP, int K, int n_copy, const PadJobs pj)
{
    if ((int)blockIdx.x >= n_copy) { pad_rows4_block(pj, (int)blockIdx.x - n_copy); return; }
    const unsigned Q = (unsigned)FP / 4, idx = blockIdx.x * 256u + threadIdx.x;
    if (idx >= (unsigned)(T + K) * (unsigned)B * Q) return;
    const unsigned rb = idx / Q, j = idx - rb * Q, row = rb / (unsigned)B, b = rb - row * (unsigned)B;
    const size_t R = (size_t)B * FP;
    const int f0 = 4 * (int)j;
    float4 vm, vp;
    float* dm; float* dp;
    if ((int)row < T) {
        const float* sm = mag + ((size_t)b * T + row) * F;
        const float* sp = phs + ((size_t)b * T + row) * F;
        float m4[4], p4[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) { const int fc = f0 + q < F ? f0 + q : F - 1; m4[q] = sm[fc]; p4[q] = sp[fc]; }      // eight unconditional loads (clamped), then the selects
        vm = make_float4(f0 < F ? m4[0] : 0.f, f0 + 1 < F ? m4[1] : 0.f, f0 + 2 < F ? m4[2] : 0.f, f0 + 3 < F ? m4[3] : 0.f);
        vp = make_float4(f0 < F ? p4[0] : 0.f, f0 + 1 < F ? p4[1] : 0.f, f0 + 2 < F ? p4[2] : 0.f, f0 + 3 < F ? p4[3] : 0.f);
        dm = Vm + (size_t)row * R + (size_t)b * FP; dp = Vp + (size_t)row * R + (size_t)b * FP;
    } else {
        const int k = (int)row - T;
        const float v = knobs[b * K + k];
        vm = make_float4(f0 < F ? v : 0.f, f0 + 1 < F ? v : 0.f, f0 + 2 < F ? v : 0.f, f0 + 3 < F ? v : 0.f); vp = vm;
        dm = H4Km + (size_t)(16 + k) * R + (size_t)b * FP; dp = H4Kp + (size_t)(16 + k) * R + (size_t)b * FP;
    }
    reinterpret_cast<float4*>(dm)[j] = vm;
    reinterpret_cast<float4*>(dp)[j] = vp;
}

// nn_proc.py:325-326 (polar -> rectangular) into the KP-pitched synthesis operand + partial sums of the L1 term
// (loss_functions.py:33-36).  One partial per workgroup (st_ae_fwd_partials() of them), fixed summation order.
__global__ void __launch_bounds__(256)
wide_polar_out_kernel(const float* __restrict__ mag_hat, const float* __restrict__ phs_hat, float* __restrict__ AA,
                      float* __restrict__ reg_partial, int B, int OT, int F, int FP, int KP, float expfac,
                      unsigned short* __restrict__ AA16 = nullptr, int aa_ht = 0)      // 16-bit GEMM configurations: see sta::ae_fwd_kernel
{
    __shared__ float red[256];
    const unsigned n = (unsigned)B * (unsigned)OT * (unsigned)FP;          // < 2^30 (host check): 32-bit index arithmetic (the size_t divisions were most of this kernel)
    float reg = 0.f;
    for (unsigned idx = blockIdx.x * 256u + threadIdx.x; idx < n; idx += gridDim.x * 256u) {
        const unsigned ro = idx / (unsigned)FP; const int f = (int)(idx - ro * (unsigned)FP);
        float re = 0.f, im = 0.f;
        if (f < F) {
            const float mh = mag_hat[(size_t)ro * F + f], ph = phs_hat[(size_t)ro * F + f];
            float sn, cs; st_sincos(ph, sn, cs);                             // as sta::ae_fwd_kernel
            re = mh * cs; im = mh * sn;
            reg += fabsf(mh * expf(expfac * (float)f));
        }
        if (AA16) { AA16[(size_t)ro * KP + f] = st_to_h16(re, aa_ht); AA16[(size_t)ro * KP + FP + f] = st_to_h16(im, aa_ht); }
        else { AA[(size_t)ro * KP + f] = re; AA[(size_t)ro * KP + FP + f] = im; }
    }
    red[threadIdx.x] = reg; __syncthreads();
    for (int s = 128; s > 0; s >>= 1) { if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s]; __syncthreads(); }
    if (reg_partial && threadIdx.x == 0) reg_partial[blockIdx.x] = red[0];
}

struct OnesRows { float* p[18]; };
// Gradient entering the two output layers (same algebra as the d-out stage of sta::ae_bwd_kernel): from d(AA) (split-K
// slabs of the synthesis dgrad, dead frames = 0), the L1 term and an optional upstream d(mag_hat).
__global__ void __launch_bounds__(256)
wide_dout_kernel(const float* __restrict__ dAA, int nslab, size_t slab, const float* __restrict__ mag_hat, const float* __restrict__ phs_hat,
                 const float* __restrict__ E9m, const float* __restrict__ E9p, const float* __restrict__ mag_tail,
                 const float* __restrict__ g_mag_hat, float reg_coef, float expfac,
                 float* __restrict__ DA9m, float* __restrict__ DA9p, float* __restrict__ TLm, float* __restrict__ TLp,
                 int B, int OT, int F, int FP, int KP, int to_lo, int to_hi, int n_main, const OnesRows ones)
{
    if ((int)blockIdx.x >= n_main) {             // the rows of ones the bias gradients ride on (was a launch of its own): block = (buffer, window)
        const int q = (int)blockIdx.x - n_main, buf = q / B, b = q - buf * B;
        float* d = ones.p[buf] + (size_t)b * FP;
        for (int f = threadIdx.x; f < FP; f += 256) d[f] = f < F ? 1.f : 0.f;
        return;
    }
    const unsigned n = (unsigned)B * (unsigned)OT * (unsigned)FP;          // < 2^30 (host check)
    const size_t R = (size_t)B * FP;
    for (unsigned idx = blockIdx.x * 256u + threadIdx.x; idx < n; idx += (unsigned)n_main * 256u) {
        const unsigned ro = idx / (unsigned)FP; const int f = (int)(idx - ro * (unsigned)FP);
        const int b = (int)(ro / (unsigned)OT), to = (int)(ro - (unsigned)b * (unsigned)OT);
        const size_t ix = (size_t)to * R + (size_t)b * FP + f;
        float d9m = 0.f, d9p = 0.f, tm = 0.f, tp = 0.f;
        if (f < F) {
            float gre = 0.f, gim = 0.f;
            if (to >= to_lo && to <= to_hi) {           // up to 8 slabs requested together (fixed order of additions)
                for (int z0 = 0; z0 < nslab; z0 += 8) {
                    float ur[8], ui[8];
#pragma unroll
                    for (int q = 0; q < 8; ++q) { const int z = z0 + q < nslab ? z0 + q : nslab - 1; ur[q] = dAA[z * slab + (size_t)ro * KP + f]; ui[q] = dAA[z * slab + (size_t)ro * KP + FP + f]; }
#pragma unroll
                    for (int q = 0; q < 8; ++q) if (z0 + q < nslab) { gre += ur[q]; gim += ui[q]; }
                }
            }
            const float mh = mag_hat[(size_t)ro * F + f], ph = phs_hat[(size_t)ro * F + f];
            float sn, cs; st_sincos(ph, sn, cs);                             // as the d-out stage of sta::ae_bwd_kernel
            const float wf = expf(expfac * (float)f);
            const float sg = mh > 0.f ? 1.f : (mh < 0.f ? -1.f : 0.f);
            const float dmh = gre * cs + gim * sn + reg_coef * sg * wf + (g_mag_hat ? g_mag_hat[(size_t)ro * F + f] : 0.f);
            const float em = E9m[ix], ep = E9p[ix];
            d9m = dmh * mag_tail[ix] * elu_grad_from_out(em);
            tm = dmh * em;
            const float dph = mh * (gim * cs - gre * sn);
            d9p = dph * elu_grad_from_out(ep);
            tp = dph;
        }
        DA9m[ix] = d9m; DA9p[ix] = d9p; TLm[ix] = tm; TLp[ix] = tp;
    }
}

// The bias gradient rides in the weight-gradient GEMM: every layer input buffer carries one extra row of ones (on real
// bins), so column IN of the [OUT][IN + 1] product is the row sum of dA.  grid (18, B): the 9 input buffers of both nets.

// Sum of the split-K slabs of all nine weight-gradient GEMMs of one autoencoder (fixed slab order) scattered into the
// packed gradient block: slab layout per layer [OUT][IN + 1] at so[l]; column IN is the bias gradient.
struct GradTab { int so[10]; int out[9]; int in[9]; int gw[9]; int gb[9]; };
struct SynReduce { const float* wg; int nz; float* gSr; float* gSi; float* norm_s; int N, F, KP; stm::NyqJob nyq; };      // wg == NULL: none
__global__ void __launch_bounds__(256)
wide_grad_finish_kernel(const float* __restrict__ slabs0, int nslab, size_t SL, const GradTab tab, float* __restrict__ g0, float* __restrict__ g1,
                        const int n_fin = 1 << 30, const float* __restrict__ red_ws = nullptr, const int red_parts = 0, const int PG = 0,
                        const SynReduce syn = SynReduce{}, float* __restrict__ norm_e = nullptr)      // norm_e: one |g| partial per block of the first two roles, slot = y * (n_fin + n_red) + x
{
    float* g = blockIdx.y ? g1 : g0;
    const int n_red = red_ws ? (PG + 63) / 64 : 0;
    if ((int)blockIdx.x >= n_fin + n_red) {
        // third role (fused step): the split-K slabs of the SYNTHESIS weight gradient -> the two synthesis gradient tensors (un-folded) + their |g|
        // partials, exactly stm::wgrad_reduce_kernel (which this replaces as a launch); the slabs were written before the autoencoder backward
        if (blockIdx.y == 0 && syn.wg) stm::wgrad_reduce_block(syn.wg, syn.nz, syn.gSr, syn.gSi, syn.norm_s, syn.N, syn.F, syn.KP, 1, (int)blockIdx.x - n_fin - n_red, nullptr, syn.nyq);
        return;
    }
    if ((int)blockIdx.x >= n_fin) {
        // second role (fused wide path): the workgroup partials of the inner-layer kernel ws[nparts][2][PG] -> g, as stm::ae_grad_reduce_block, but ONLY
        // the parameters of layers 2..8: the layer-1 / layer-9 entries of those partials are zeros and their gradients come from the slabs (first role)
        __shared__ float rr[4][64];
        const int col = threadIdx.x & 63, pl = threadIdx.x >> 6;
        const int i = ((int)blockIdx.x - n_fin) * 64 + col;
        auto in_range = [&](int lo, int n) { return i >= lo && i < lo + n; };
        const bool mine = i < PG && !in_range(tab.gw[0], tab.out[0] * tab.in[0]) && !in_range(tab.gb[0], tab.out[0]) &&
                          !in_range(tab.gw[8], tab.out[8] * tab.in[8]) && !in_range(tab.gb[8], tab.out[8]);
        float s = 0.f;
        if (mine) {
            const float* base = red_ws + (size_t)blockIdx.y * PG + i;
            const size_t stride = (size_t)2 * PG;
            for (int p0 = pl; p0 < red_parts; p0 += 32) {
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) { const int p = p0 + 4 * u; v[u] = p < red_parts ? base[(size_t)p * stride] : 0.f; }
#pragma unroll
                for (int u = 0; u < 8; ++u) s += v[u];
            }
        }
        rr[pl][col] = s;
        __syncthreads();
        if (pl != 0) return;
        const float r = mine ? (rr[0][col] + rr[1][col]) + (rr[2][col] + rr[3][col]) : 0.f;
        if (mine) g[i] = r;
        if (norm_e) { const float t = wave_sum(fabsf(r)); if (col == 0) norm_e[blockIdx.y * (n_fin + n_red) + blockIdx.x] = t; }
        return;
    }
    const float* slabs = slabs0 + (size_t)blockIdx.y * nslab * SL;      // net y: its slabs follow net 0's
    // block = 64 slab elements x 4 slab lanes, 8 loads in flight per thread (one thread walking 256 slabs four at a time was 30 us of pure
    // latency per net at the 65536-sample window); the four lanes are added in a fixed order
    __shared__ float red[4][64];
    const int col = threadIdx.x & 63, pl = threadIdx.x >> 6;
    const int idx = blockIdx.x * 64 + col;
    int l = 0;
#pragma unroll
    for (int q = 1; q < 9; ++q) l += idx >= tab.so[q];
    const int e = idx - tab.so[l], n1 = tab.in[l] + 1;
    const bool live = idx < tab.so[9] && e < tab.out[l] * n1;      // not an alignment pad between layers, not a layer the fused inner kernel owns (out = 0)
    float s = 0.f;
    if (live) {
        const float* base = slabs + idx;
        for (int z0 = pl; z0 < nslab; z0 += 32) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { const int z = z0 + 4 * u; v[u] = z < nslab ? base[(size_t)z * SL] : 0.f; }
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
    }
    red[pl][col] = s;
    __syncthreads();
    if (pl != 0) return;
    s = live ? (red[0][col] + red[1][col]) + (red[2][col] + red[3][col]) : 0.f;
    if (live) {
        const int o = e / n1, i = e - o * n1;
        if (i < tab.in[l]) g[tab.gw[l] + o * tab.in[l] + i] = s; else g[tab.gb[l] + o] = s;
    }
    if (norm_e) { const float t = wave_sum(fabsf(s)); if (col == 0) norm_e[blockIdx.y * (n_fin + n_red) + blockIdx.x] = t; }
}

}  // namespace stw
