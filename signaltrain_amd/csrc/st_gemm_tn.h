// st_gemm_tn.h -- the weight-gradient GEMMs of the STFT bases (fp32 MFMA, gfx950), round 3.
//
//   C[m][n] = sum_k A[k][m] * B[k][n]        k = compact live frame row (window b, frame t), both operands M/N-contiguous:
//     analysis  (autograd of cls_fe_dft.py:55-56):   A = dG  [B*T ][KP] (d re | d im),  B = frames of the padded waveform x/2
//     synthesis (autograd of cls_fe_dft.py:112):     A = AA  [B*OT][KP] (an_real | an_imag), B = frames of the padded d syn
//
// Why a second kernel beside st_gemm.h's family.  On gfx950 the fp32 MFMA shares the vector ALUs (tools/ubench): an fp32
// GEMM's time is its MFMA passes PLUS every other instruction it issues.  The 32 x 96 wave strips of gemm_kernel<3, ...> pay, per
// 24 MFMAs, 32 scalar ds_read_b32 fragment reads and ~100 address VALU (61 % of the fp32 peak at B = 256) and need 12 split-K slabs
// (50 MB written, 25 MB re-read) to fill the chip with 96 x 96 tiles.  Here:
//   * workgroup tile 128 x 128, four waves as 2 x 2, each wave a 64 x 64 tile = 2 x 2 accumulators of v_mfma_f32_32x32x2_f32;
//   * k-major LDS tiles [k][128] (a global float4 along m is ONE ds_write_b128, no transpose) and a ROW PERMUTATION that makes the
//     fragments wide reads: MFMA block mi, row r  <->  tile row 2 r + mi, so the two A operands of a lane are adjacent floats --
//     one ds_read_b64 per operand per k-step, 2 LDS reads per 4 MFMAs instead of 4 per 3.  Lanes 0-31 read 256 contiguous bytes
//     (64 banks, conflict-free), lanes 32-63 the next k row.  The permutation costs nothing: the epilogue stores float2 pairs of
//     adjacent columns (256 contiguous bytes per half-wave);
//   * M = 2 (F - 1) = N rows exactly: the two NYQUIST rows (bin F - 1 of d re / d im) would cost a ninth, 98 %-empty tile row; they
//     are two dot products per output column and are formed by the slab-reduce kernel with plain FMAs (st_misc.h nyquist_chunk);
//     with 64 tiles, FOUR k-slices fill the 256 CUs: 4 slabs (17 MB) instead of 12;
//   * addresses: both operands as 32-bit element offsets from ONE wave-uniform base (they live in the same workspace), rows past the
//     end of the reduction read a block of zeros that the padded signal provides (its Conv1d margin) -- no selects on loaded values.
#pragma once
#include "st_gemm.h"

namespace stg {

struct TNJob {
    const float* base;             // common base: every offset below is in ELEMENTS from it and < 2^30
    unsigned a0, b0;               // origin of operand A / B
    unsigned SA1, SA2, SB1, SB2;   // element (k, c) of A: a0 + b * SA1 + t * SA2 + c, of B: b0 + b * SB1 + t * SB2 + c,  (b, t) = split(k)
    unsigned magic; int Tv, t_lo;  // RowMap of the reduction index (shared by both operands)
    int K;                         // reduction length (compact rows)
    unsigned zero;                 // >= 128 consecutive zero floats (A rows past K)
    int mh; unsigned mstride;      // tile row tm -> first A column = output row: (tm % mh) * 128 + (tm / mh) * mstride
    // the rows the tiles do not cover (the Nyquist bin of each basis): per-window-group partial dot products, see nyq_partial
    float* nyq_out; unsigned nyq_c0, nyq_c1; int nyq_P, Nc, nsplit;
    // round 5 (frame-major reduction order only, see FrameTrim): tile column tx (taps [128 tx, 128 tx + 128) of operand B) is structurally zero outside
    // the reduction rows [fa[tx] * fB, fb[tx] * fB) -- the k-slices of that tile column divide THAT range
    int trim, fB; unsigned char fa[64], fb[64];
};
// Reduction rows a tile column needs.  Operand B of the weight-gradient GEMMs is the frame matrix of a zero-padded signal (x / 2 with its Conv1d padding,
// cls_fe_dft.py:28-31; d syn with the cropped margins of cls_fe_dft.py:113): tap n of frame t is a structural zero unless pad <= H t + n < pad + Ls.  With the
// reduction rows enumerated FRAME-major (all windows of frame t_lo, then t_lo + 1, ...) the frames that carry anything for taps [128 j, 128 j + 128) are a
// contiguous run [fa, fb) -- 21-22 of the 23 analysis frames, 5-6 of the 7 synthesis frames at the default geometry: 7.6 % / 25 % of the MACs skipped.
struct FrameTrim { int on, B; unsigned char fa[64], fb[64]; };
static inline FrameTrim frame_trim(const RowMap& live, int B, int H, int Ntaps, int pad, int Ls)
{
    FrameTrim f; f.on = (Ntaps % 128 == 0 && Ntaps / 128 <= 64 && live.Tv <= 255) ? 1 : 0; f.B = B;
    for (int j = 0; j < 64; ++j) { f.fa[j] = 0; f.fb[j] = (unsigned char)(live.Tv <= 255 ? live.Tv : 255); }
    if (!f.on) return f;
    for (int j = 0; j < Ntaps / 128; ++j) {
        const int n0 = 128 * j;
        int a = 0, b = live.Tv;
        while (a < b && H * (live.t_lo + a) + n0 + 127 < pad) ++a;
        while (b > a && H * (live.t_lo + b - 1) + n0 >= pad + Ls) --b;
        f.fa[j] = (unsigned char)a; f.fb[j] = (unsigned char)b;
    }
    return f;
}

// One extra z-slice of workgroups (blockIdx.z == nsplit) forms the two Nyquist rows C[c][n] = sum_k A[k][c] * B[k][n], c in {nyq_c0, nyq_c1},
// as nyq_P partial sums over groups of windows: out[p][0 | 1][n].  Light vector work (K / nyq_P rows of float4 FMAs per thread) that runs
// beside the MFMA workgroups as a second resident workgroup of its CU; the slab-reduce kernel adds the nyq_P partials in a fixed order.
__device__ __forceinline__ void nyq_partial(const TNJob& j, const int p)
{
    if (p >= j.nyq_P) return;
    // partial p = the reduction rows [k0, k1) (round 5: plain row ranges -- with the frame-major order "groups of windows" would be 7-23 long groups)
    const int per = (j.K + j.nyq_P - 1) / j.nyq_P, k0 = p * per, k1 = (k0 + per < j.K) ? k0 + per : j.K;
    for (int n4 = threadIdx.x; n4 < j.Nc / 4; n4 += 256) {
        float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0;
        // eight rows at a time with all their loads issued before the first FMA (the order of the sum stays k-ascending)
        for (int kb = k0; kb < k1; kb += 8) {
            float a0[8], a1[8]; float4 x[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int k = kb + u < k1 ? kb + u : k1 - 1;
                const unsigned w = __umulhi((unsigned)k, j.magic), t = (unsigned)j.t_lo + ((unsigned)k - w * (unsigned)j.Tv);
                const float* arow = j.base + j.a0 + (size_t)w * j.SA1 + (size_t)t * j.SA2;
                a0[u] = kb + u < k1 ? arow[j.nyq_c0] : 0.f; a1[u] = kb + u < k1 ? arow[j.nyq_c1] : 0.f;
                x[u] = *reinterpret_cast<const float4*>(j.base + j.b0 + (size_t)w * j.SB1 + (size_t)t * j.SB2 + 4 * n4);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                s0.x = __builtin_fmaf(a0[u], x[u].x, s0.x); s0.y = __builtin_fmaf(a0[u], x[u].y, s0.y); s0.z = __builtin_fmaf(a0[u], x[u].z, s0.z); s0.w = __builtin_fmaf(a0[u], x[u].w, s0.w);
                s1.x = __builtin_fmaf(a1[u], x[u].x, s1.x); s1.y = __builtin_fmaf(a1[u], x[u].y, s1.y); s1.z = __builtin_fmaf(a1[u], x[u].z, s1.z); s1.w = __builtin_fmaf(a1[u], x[u].w, s1.w);
            }
        }
        float* o = j.nyq_out + (size_t)p * 2 * j.Nc + 4 * n4;
        *reinterpret_cast<float4*>(o) = s0;
        *reinterpret_cast<float4*>(o + j.Nc) = s1;
    }
}

template <int BKT>
__global__ void __launch_bounds__(256)
gemm_tn128_kernel(const TNJob j, float* __restrict__ out, const int ldo, const size_t slab, const int ksplit)
{
    constexpr int TS = BKT * 128;                       // floats per operand tile
    constexpr int NP = BKT / 8;                         // load passes: 256 threads = 32 float4 columns x 8 k rows
    constexpr int KS = BKT / 2;                         // MFMA k-steps per tile
    extern __shared__ __attribute__((aligned(16))) float tn_lds[];       // As[2][BKT][128] | Bs[2][BKT][128]
    float* const As = tn_lds;
    float* const Bs = tn_lds + 2 * TS;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    // the Nyquist workgroups are the PHYSICAL last z-slice: dispatched after the GEMM workgroups, round-robin over the XCDs, and outside
    // the XCD-aware remapping of the GEMM tiles (inside it they would all land on the last XCD and push its GEMM tiles onto the others)
    if ((int)blockIdx.z >= j.nsplit) { nyq_partial(j, (int)(blockIdx.y * gridDim.x + blockIdx.x)); return; }       // workgroup-uniform
    int tbx, tby, tbz; xcd_tile(tbx, tby, tbz, j.nsplit);
    const unsigned rowA = (unsigned)(tby % j.mh) * 128u + (unsigned)(tby / j.mh) * j.mstride;
    const unsigned colB = (unsigned)tbx * 128u;
    int k_begin = tbz * ksplit;
    int k_end = (k_begin + ksplit < j.K) ? k_begin + ksplit : j.K;
    if (j.trim) {                                        // the rows that carry anything for this tile column, divided among the k-slices (workgroup-uniform)
        const int lo = (int)j.fa[tbx] * j.fB, hi = (int)j.fb[tbx] * j.fB;
        const int per = ((hi - lo + j.nsplit - 1) / j.nsplit + BKT - 1) / BKT * BKT;
        k_begin = lo + tbz * per;
        k_end = (k_begin + per < hi) ? k_begin + per : hi;      // a last k-tile may run past hi: those rows are structural zeros of B for this tile column (or past K: masked)
    }

    const int c4 = tid & 31, kr = tid >> 5;
    // BYTE offsets from j.base; every product below has 24-bit factors (host-checked): full-rate v_mad_u32_u24 instead of the
    // quarter-rate 32-bit multiplies (16 of them per k-tile were 9 % of the loop)
    const unsigned la = 4u * (j.a0 + rowA + 4u * (unsigned)c4), lb = 4u * (j.b0 + colB + 4u * (unsigned)c4), lz = 4u * (j.zero + 4u * (unsigned)c4);
    const unsigned sa1 = 4u * j.SA1, sa2 = 4u * j.SA2, sb1 = 4u * j.SB1, sb2 = 4u * j.SB2;
    float4 ra[NP], rb[NP];
    auto ld = [&](const unsigned byte_off) { return *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(j.base) + byte_off); };
    auto gload = [&](const int kt) {
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const int k = kt + kr + 8 * p;
            const unsigned kc = (unsigned)(k < j.K ? k : j.K - 1);
            const unsigned b = __umulhi(kc, j.magic);                                   // Tv >= 2 (host-checked): magic != 0
            const unsigned t = (unsigned)j.t_lo + kc - __umul24(b, (unsigned)j.Tv);
            const unsigned oa = __umul24(b, sa1) + __umul24(t, sa2) + la;
            const unsigned ob = __umul24(b, sb1) + __umul24(t, sb2) + lb;
            const unsigned live = (unsigned)((k - j.K) >> 31);       // all ones while k < K; a bit select (v_bfi), not a ?: the compiler turns into a branch
            ra[p] = ld((oa & live) | (lz & ~live));                  // rows past the end of the reduction: the block of zeros (B's row is clamped: finite x 0)
            rb[p] = ld(ob);
        }
    };
    auto lstore = [&](const int buf) {
        float* as = As + buf * TS + kr * 128 + 4 * c4;
        float* bs = Bs + buf * TS + kr * 128 + 4 * c4;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            *reinterpret_cast<float4*>(as + p * 8 * 128) = ra[p];
            *reinterpret_cast<float4*>(bs + p * 8 * 128) = rb[p];
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int nj = 0; nj < 2; ++nj)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[mi][nj][i] = 0.f;

    const int h = lane >> 5, l31 = lane & 31;
    if (k_begin < k_end) {
        gload(k_begin);
        lstore(0);
        __syncthreads();
        int cur = 0;
        const int a_off = h * 128 + wm * 64 + 2 * l31;          // k-step s reads row 2 s + h
        const int b_off = h * 128 + wn * 64 + 2 * l31;
        for (int kt = k_begin; kt < k_end; kt += BKT) {
            // one basic block per k-tile (the last iteration re-loads its own tile instead of branching around the prefetch)
            const bool more = kt + BKT < k_end;
            gload(more ? kt + BKT : kt);
            __builtin_amdgcn_sched_barrier(0);
            const float* as = As + cur * TS + a_off;
            const float* bs = Bs + cur * TS + b_off;
            // fragments are double-buffered in registers by PAIRS of k-steps (the compiler fuses the two ds_read_b64 of a pair into one
            // ds_read2st64_b64): the pair p + 1 is read before the 8 MFMAs of pair p issue.  Left to itself the compiler re-uses ONE
            // register set -- read / s_waitcnt lgkmcnt(0) / 8 MFMAs -- and at one wave per SIMD every LDS round trip is exposed.
            float2 fa[2][2], fb[2][2];
            auto frag = [&](const int pr, const int buf) {
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    fa[buf][u] = *reinterpret_cast<const float2*>(as + (2 * pr + u) * 256);
                    fb[buf][u] = *reinterpret_cast<const float2*>(bs + (2 * pr + u) * 256);
                }
            };
            frag(0, 0);
#pragma unroll
            for (int pr = 0; pr < KS / 2; ++pr) {
                if (pr + 1 < KS / 2) frag(pr + 1, (pr + 1) & 1);
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const float2 a = fa[pr & 1][u], b = fb[pr & 1][u];
                    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc[0][0], 0, 0, 0);
                    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.y, acc[0][1], 0, 0, 0);
                    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.x, acc[1][0], 0, 0, 0);
                    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc[1][1], 0, 0, 0);
                }
            }
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
#pragma unroll
            for (int pr = 0; pr < KS / 2; ++pr) {
                if (pr + 1 < KS / 2) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            lstore(cur ^ 1);
            __syncthreads();
            cur ^= 1;
        }
    }
    // epilogue: acc[mi][nj][i] = C[tile row wm*64 + 2*d_row(i) + mi][tile col wn*64 + 2*l31 + nj]
    float* o = out + (size_t)tbz * slab + (size_t)(rowA + wm * 64) * ldo + colB + wn * 64 + 2 * l31;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int r = 2 * d_row(i, lane);
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
            *reinterpret_cast<float2*>(o + (size_t)(r + mi) * ldo) = make_float2(acc[mi][0][i], acc[mi][1][i]);
    }
}

// Host side.  Returns false when the problem does not fit the kernel's assumptions (the caller then uses gemm_kernel<3, ...>).
struct TNOperand { const float* base; unsigned S1, S2; };     // element (k, c) at base[b * S1 + t * S2 + c]
static inline bool tn128_fits(const TNOperand& A, const TNOperand& B, const float* zeros, const RowMap& map, int M, int Nc, size_t extentA, size_t extentB)
{
    if (M % 128 || Nc % 128 || map.Tv < 2) return false;
    const unsigned lim24 = 1u << 22;                 // strides are used as BYTE strides in 24-bit multiplies
    if (A.S1 >= lim24 || A.S2 >= lim24 || B.S1 >= lim24 || B.S2 >= lim24) return false;
    const float* lo = A.base < B.base ? A.base : B.base; if (zeros < lo) lo = zeros;
    const size_t lim = (size_t)1 << 30;
    return (size_t)(A.base - lo) + extentA < lim && (size_t)(B.base - lo) + extentB < lim && (size_t)(zeros - lo) + 128 < lim;
}
template <int BKT>
static inline int launch_tn128(const TNOperand& A, const TNOperand& B, const float* zeros, const RowMap& map, int K,
                               int M, int mh, unsigned mstride, int Nc, float* out, int ldo, size_t slab, int nsplit, hipStream_t s,
                               float* nyq_out = nullptr, unsigned nyq_c0 = 0, unsigned nyq_c1 = 0, int* nyq_P = nullptr, const FrameTrim* trim = nullptr)
{
    const float* lo = A.base < B.base ? A.base : B.base; if (zeros < lo) lo = zeros;
    TNJob j;
    j.base = lo; j.a0 = (unsigned)(A.base - lo); j.b0 = (unsigned)(B.base - lo); j.zero = (unsigned)(zeros - lo);
    j.SA1 = A.S1; j.SA2 = A.S2; j.SB1 = B.S1; j.SB2 = B.S2;
    j.magic = map.magic; j.Tv = map.Tv; j.t_lo = map.t_lo; j.K = K; j.mh = mh; j.mstride = mstride;
    j.trim = (trim && trim->on) ? 1 : 0; j.fB = trim ? trim->B : 0;
    for (int i = 0; i < 64; ++i) { j.fa[i] = trim ? trim->fa[i] : 0; j.fb[i] = trim ? trim->fb[i] : 0; }
    int ksplit = K;
    if (nsplit > 1) ksplit = st_round_up((K + nsplit - 1) / nsplit, BKT);
    constexpr size_t lds = (size_t)4 * BKT * 128 * sizeof(float);
    if (lds > 65536) { const int rc = ::ensure_dyn_lds((const void*)gemm_tn128_kernel<BKT>, "gemm_tn128_kernel"); if (rc) return rc; }
    const int nz = nsplit > 1 ? nsplit : 1, tiles = (Nc / 128) * (M / 128);
    j.nsplit = nz; j.Nc = Nc; j.nyq_out = nyq_out; j.nyq_c0 = nyq_c0; j.nyq_c1 = nyq_c1;
    j.nyq_P = tiles < 64 ? tiles : 64; { const int W = K / 8 > 0 ? K / 8 : 1; if (j.nyq_P > W) j.nyq_P = W; }      // partials of >= 8 reduction rows
    if (nyq_P) *nyq_P = j.nyq_P;
    hipLaunchKernelGGL((gemm_tn128_kernel<BKT>), dim3(Nc / 128, M / 128, nz + (nyq_out ? 1 : 0)), dim3(256), lds, s, j, out, ldo, slab, ksplit);
    return 0;
}

}  // namespace stg

// ================================================================================================ NT x NT on the same 128 x 128 tiles (round 3; work list: round 5)
//   C[m][n] = sum_k A[m][k] * B[n][k],  both operands K-contiguous:  the synthesis FRAMES GEMM (cls_fe_dft.py:112 as a GEMM: A = spectra AA
//   [live frames][KP], B = transposed fold [N][KP]) and the synthesis DATA-GRADIENT GEMM (A = frames of the padded d syn, B = fold [KP][N]).
//   128 x 128 tiles, one workgroup per CU; row-major LDS tiles [row][BK + 4] (a global float4 along k = one ds_write_b128; a lane's 16 k of its
//   row = four ds_read_b128, conflict-free at pitch 36), lane half h takes k in [16 h, 16 h + 16) of the 32-deep tile as in gemm_kernel; row
//   offsets are per-thread constants and the k offset rides in the scalar base: NO address arithmetic in the loop.
//
//   Round 5 -- the STRUCTURAL ZEROS of the transposed convolution are not multiplied any more.  ConvTranspose1d(stride H) followed by the crop
//   wave_form[:, :, N:-N] (cls_fe_dft.py:112-113) keeps, of output frame t', only the taps n with N <= H t' + n < N + y: at the default geometry
//   frames 1, 2, 6, 7 keep 384 / 768 / 768 / 384 of their 1024 taps -- a quarter of the frames GEMM's outputs are cropped away and a quarter of the
//   data-gradient GEMM's reduction reads the zero margins of the padded d syn.  With the compact rows enumerated FRAME-major (RowMap::fm) a 128-row
//   tile has one frame index (B a multiple of 128; otherwise a short run of them, whose union is used), so the host can say per tile which taps
//   live: the kernel runs a WORK LIST (one entry per workgroup, in the kernel arguments) of (tile row, tile column, k range, slab) --
//     frames GEMM:   only the tile columns that hold live taps (84 of 112 tiles at B = 256), which leaves room for a third k-slice on 256 CUs;
//     data gradient: per tile row the live k range only, cut into as many slices as keep every slice <= 384 taps (3 / 2 / 1 slices for whole /
//                    three-quarter / three-eighth frames: 240 workgroups of equal length instead of 252 of length 512); slabs a tile row does not
//                    use are zero-filled by its first slice, so the consumers' slab count stays a function of the geometry alone.
//   KP = 2 * 528 columns are 8 tiles of 128 + 32: as in gemm_tn128_kernel the two Nyquist columns (bin F - 1 of d an_real / d an_imag) do not get a
//   ninth, 87 %-empty tile column -- the tile columns cover bins [0, F - 1) of each half (column map: GEMM column c -> (c % nh) + (c / nh) * stride) and
//   one light workgroup per tile row (kind 1) forms the two Nyquist columns as plain dot products.
namespace stg {

struct NTRows { const float* base; unsigned S1, S2, magic; int Tv, t_lo, R; };      // row r -> base[off(r) + k]; off = b * S1 + t * S2, (b, t) = split(min(r, R - 1)); magic == 0: b = r, t = 0
__device__ __forceinline__ unsigned ntrows_off(const NTRows& o, const int r)
{
    const unsigned rc = (unsigned)(r < o.R ? r : o.R - 1);
    const unsigned b = o.magic ? __umulhi(rc, o.magic) : rc;
    const unsigned t = o.magic ? (unsigned)o.t_lo + (rc - b * (unsigned)o.Tv) : 0u;
    return b * o.S1 + t * o.S2;
}
// the rows of a [windows][frames] array enumerated frame-major (RowMap::fm): row r -> frame t_lo + r / B, window r % B
static inline NTRows ntrows_frame_major(const float* base, unsigned s_window, unsigned s_frame, const RowMap& live, int B, int R)
{
    return NTRows{base + (size_t)live.t_lo * s_frame, s_frame, s_window, B > 1 ? rowmap_magic(B) : 0u, B, 0, R};      // "b" = r / B = the frame, "t" = r % B = the window
}

constexpr int NTW_MAX = 768;                               // work-list entries (one workgroup each): 3 KB of kernel arguments
struct NTWork {
    int n, nslabs;                                         // entries; slabs the consumer sums
    int col_h, col_stride;                                 // output column of GEMM column c: (c % col_h) + (c / col_h) * col_stride  (col_h = 0: c)
    int kunit, kt_total;                                   // k ranges are in units of kunit k-tiles (1 unless the reduction has more than 64 k-tiles), clamped to kt_total k-tiles
    unsigned nyq_b[2], nyq_col[2];                         // kind 1: element offsets (from rb.base) of the two B rows, and their output columns
    unsigned e[NTW_MAX];                                   // tile row (8 bits) | tile column (6) << 8 | slab (2) << 14 | first slab to zero-fill (2) << 16 | kind (1) << 18 | first k unit (6) << 19 | k units (7) << 25
};
static inline unsigned ntw_pack(int mt, int nt, int z, int zf, int kind, int k0, int kl)
{
    return (unsigned)mt | (unsigned)nt << 8 | (unsigned)z << 14 | (unsigned)zf << 16 | (unsigned)kind << 18 | (unsigned)k0 << 19 | (unsigned)kl << 25;
}

// slab z, 32 x 64 block of a wave: out[z][full_row][col]
struct StoreSlab {
    float* out; int M, Nc, ld; size_t slab; RowMap map;
    __device__ void operator()(const int z, const int m0, const int n0, const f32x16 (&acc)[2]) const {
        const int lane = threadIdx.x & 63;
        float* o = out + (size_t)z * slab;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int row = m0 + d_row(i, lane);
            if (row < M) {
                float* orow = o + (size_t)map.full(row) * ld;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int col = n0 + 32 * j + (lane & 31);
                    if (col < Nc) orow[col] = acc[j][i];
                }
            }
        }
    }
};

// kind 1: the two columns the tiles leave out, for the 128 rows of tile row mt: out[0][row][col_c] = sum_k A[row][k] * Brow_c[k] over the entry's k range,
// zeros in the other slabs.  A wave takes rows wave, wave + 4, ...; its 64 lanes read 1 KB of the row at a time (the first version gave every thread a
// row of its own: 64 cache lines per load instruction, 27 us of address-path time -- longer than the tiles it runs beside) and reduce by lane exchange.
__device__ __forceinline__ void nt128_nyquist(const NTRows& ra, const NTRows& rb, const StoreSlab& epi, const NTWork& wk, const int m_blk, const int k_begin, const int k_end)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* b0 = rb.base + wk.nyq_b[0];
    const float* b1 = rb.base + wk.nyq_b[1];
    for (int r0 = wave; r0 < 128; r0 += 16) {               // four rows in flight per wave
        float s0[4], s1[4];
        const float* a[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { s0[u] = 0.f; s1[u] = 0.f; a[u] = ra.base + ntrows_off(ra, m_blk + r0 + 4 * u); }
        for (int k = k_begin + 4 * lane; k < k_end; k += 256) {
            const f32x4 vb0 = *reinterpret_cast<const f32x4*>(b0 + k), vb1 = *reinterpret_cast<const f32x4*>(b1 + k);
            f32x4 va[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) va[u] = *reinterpret_cast<const f32x4*>(a[u] + k);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                s0[u] += (va[u][0] * vb0[0] + va[u][1] * vb0[1]) + (va[u][2] * vb0[2] + va[u][3] * vb0[3]);
                s1[u] += (va[u][0] * vb1[0] + va[u][1] * vb1[1]) + (va[u][2] * vb1[2] + va[u][3] * vb1[3]);
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) { s0[u] += __shfl_xor(s0[u], m); s1[u] += __shfl_xor(s1[u], m); }
            const int row = m_blk + r0 + 4 * u;
            if (lane < 2 && row < epi.M) {
                float* o = epi.out + (size_t)epi.map.full(row) * epi.ld + wk.nyq_col[lane];
                o[0] = lane ? s1[u] : s0[u];
                for (int z = 1; z < wk.nslabs; ++z) o[(size_t)z * epi.slab] = 0.f;
            }
        }
    }
}

__global__ void __launch_bounds__(256)
gemm_nt128_kernel(const NTRows ra, const NTRows rb, const StoreSlab epi, const NTWork wk)
{
    constexpr int BKT = 32, LD = BKT + 4, TS = 128 * LD;
    constexpr int NP = 4;                                  // 256 threads = 32 rows x 8 float4 per pass
    extern __shared__ __attribute__((aligned(16))) float nt_lds[];       // As[2][128][LD] | Bs[2][128][LD]
    float* const As = nt_lds;
    float* const Bs = nt_lds + 2 * TS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    const unsigned ent = wk.e[blockIdx.x];                 // workgroup-uniform (a scalar load from the kernel-argument segment)
    const int m_blk = (int)(ent & 255u) * 128, n_blk = (int)((ent >> 8) & 63u) * 128, tbz = (int)((ent >> 14) & 3u), zf = (int)((ent >> 16) & 3u);
    const int k_begin = (int)((ent >> 19) & 63u) * wk.kunit * BKT;
    int k_end = k_begin + (int)(ent >> 25) * wk.kunit * BKT;
    if (k_end > wk.kt_total * BKT) k_end = wk.kt_total * BKT;
    if ((ent >> 18) & 1u) { nt128_nyquist(ra, rb, epi, wk, m_blk, k_begin, k_end); return; }

    const int lr = tid >> 3, lk = (tid & 7) * 4;
    unsigned ao[NP], bo[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        ao[p] = 4u * (ntrows_off(ra, m_blk + lr + 32 * p) + (unsigned)lk);       // BYTE offsets
        bo[p] = 4u * (ntrows_off(rb, n_blk + lr + 32 * p) + (unsigned)lk);
    }
    f32x4 va[NP], vb[NP];
    auto gload = [&](const int kt) {
        const char* pa = reinterpret_cast<const char*>(ra.base) + 4 * (size_t)kt;
        const char* pb = reinterpret_cast<const char*>(rb.base) + 4 * (size_t)kt;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            va[p] = *reinterpret_cast<const f32x4*>(pa + ao[p]);
            vb[p] = *reinterpret_cast<const f32x4*>(pb + bo[p]);
        }
    };
    auto lstore = [&](const int buf) {
        float* as = As + buf * TS + lr * LD + lk;
        float* bs = Bs + buf * TS + lr * LD + lk;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            *reinterpret_cast<f32x4*>(as + p * 32 * LD) = va[p];
            *reinterpret_cast<f32x4*>(bs + p * 32 * LD) = vb[p];
        }
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int nj = 0; nj < 2; ++nj)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[mi][nj][i] = 0.f;

    if (k_begin < k_end) {
        gload(k_begin);
        lstore(0);
        __syncthreads();
        int cur = 0;
        const int h = lane >> 5, l31 = lane & 31;
        const int a_off = (wm * 64 + l31) * LD + 16 * h, b_off = (wn * 64 + l31) * LD + 16 * h;
        for (int kt = k_begin; kt < k_end; kt += BKT) {
            const bool more = kt + BKT < k_end;
            gload(more ? kt + BKT : kt);
            __builtin_amdgcn_sched_barrier(0);
            const float* as = As + cur * TS + a_off;
            const float* bs = Bs + cur * TS + b_off;
            // fragments in two halves of 8 k each: the second half is read while the first half's 32 MFMAs run
            f32x4 fa[2][2][2], fb[2][2][2];                // [half][block][quad] (native vectors: HIP's float4 struct indexed through a pointer went to scratch)
#pragma unroll
            for (int hf = 0; hf < 2; ++hf)
#pragma unroll
                for (int x = 0; x < 2; ++x)
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        fa[hf][x][q] = *reinterpret_cast<const f32x4*>(as + x * 32 * LD + 8 * hf + 4 * q);
                        fb[hf][x][q] = *reinterpret_cast<const f32x4*>(bs + x * 32 * LD + 8 * hf + 4 * q);
                    }
#pragma unroll
            for (int hf = 0; hf < 2; ++hf)
#pragma unroll
                for (int q = 0; q < 2; ++q)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float a0 = fa[hf][0][q][e], a1 = fa[hf][1][q][e];
                        const float b0 = fb[hf][0][q][e], b1 = fb[hf][1][q][e];
                        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
                        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
                        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
                        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
                    }
            lstore(cur ^ 1);
            // one wave per SIMD (see launch_nt128): nothing else hides LDS traffic, so it is placed by hand --
            // 8 reads | 32 MFMAs with the second half's 8 reads between them | 16 MFMAs | 16 MFMAs with the next tile's 8 ds_write_b128 between them
            // (the global loads behind those writes were issued ~48 MFMAs = 3000 cycles earlier)
            __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
#pragma unroll
            for (int i = 0; i < 8; ++i) { __builtin_amdgcn_sched_group_barrier(0x008, 4, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
            __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);
#pragma unroll
            for (int i = 0; i < 8; ++i) { __builtin_amdgcn_sched_group_barrier(0x008, 2, 0); __builtin_amdgcn_sched_group_barrier(0x200, 1, 0); }
            __builtin_amdgcn_sched_barrier(0);
            __syncthreads();
            cur ^= 1;
        }
    }
    // output columns of this tile (tile columns never straddle col_h: it is a multiple of 128)
    const int c_blk = wk.col_h ? (n_blk % wk.col_h) + (n_blk / wk.col_h) * wk.col_stride : n_blk;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) epi(tbz, m_blk + wm * 64 + 32 * mi, c_blk + wn * 64, acc[mi]);
    if (zf > 0 && zf < wk.nslabs) {                        // the slabs no k-slice of this tile computes (zf = the first of them, on the tile's slice 0 only): zeros
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int nj = 0; nj < 2; ++nj)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[mi][nj][i] = 0.f;
        for (int z = zf; z < wk.nslabs; ++z) {
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) epi(z, m_blk + wm * 64 + 32 * mi, c_blk + wn * 64, acc[mi]);
        }
    }
}

// ---- host side: the work list
// Frames [f0, f1] (absolute indices) of the rows [r0, r1) of a frame-major compact enumeration over B windows
static inline void ntw_tile_frames(const RowMap& live, int B, int R, int mt, int& f0, int& f1)
{
    const int r0 = mt * 128, r1 = (r0 + 128 < R ? r0 + 128 : R) - 1;
    f0 = live.t_lo + r0 / B; f1 = live.t_lo + r1 / B;
}
// Taps n of frames f0..f1 (union) that land in [pad, pad + Ls) at position H f + n:  [lo, hi), 0 <= lo < hi <= N
static inline void ntw_live_taps(int f0, int f1, int H, int N, int pad, int Ls, int& lo, int& hi)
{
    lo = pad - H * f1; if (lo < 0) lo = 0;                 // monotone in f: the union of the intervals is the interval of the extremes
    hi = pad + Ls - H * f0; if (hi > N) hi = N;
    if (hi <= lo) { lo = 0; hi = N; }                      // cannot happen for live frames; be safe
}
static inline void ntw_push(NTWork& w, int mt, int nt, int z, int zf, int kind, int k0, int kl)
{
    w.e[w.n++] = ntw_pack(mt, nt, z, zf, kind, k0, kl);
}
// entries are built in (slab, tile row, tile column) order; workgroup i runs on XCD i % 8, so XCD j gets the j-th contiguous eighth of that order
// (one band of A rows / one k-slice of B per L2, as xcd_tile does for the grid-shaped launches)
static inline void ntw_xcd_order(NTWork& w)
{
    unsigned tmp[NTW_MAX];
    const int T = w.n, q = T >> 3, r = T & 7;
    for (int L = 0; L < T; ++L) { const int j = L & 7; tmp[L] = w.e[j * q + (j < r ? j : r) + (L >> 3)]; }
    for (int L = 0; L < T; ++L) w.e[L] = tmp[L];
}
// Cost of a launch of `wgs` workgroups whose longest k-loop has `ktiles` k-tiles, one workgroup per CU: rounds x (fixed + per k-tile), in us (B = 256 measurements:
// 252 workgroups x 11 k-tiles 31.5 us, 224 x 16.5 43.4 us).  Only used to RANK slice counts.
static inline double ntw_cost(int wgs, int ktiles, int ncus) { return (double)((wgs + ncus - 1) / ncus) * (6.0 + 2.35 * (double)ktiles); }
// several rounds only if they are well filled (one workgroup per CU: a last round at 30 % costs a whole round -- there the smaller tiles of gemm_kernel<4 / 2, ...> balance better)
static inline bool ntw_rounds_ok(int wgs, int ncus) { const int r = (wgs + ncus - 1) / ncus; return r <= 1 || (double)wgs >= 0.85 * (double)r * (double)ncus; }
static inline bool ntw_header(NTWork& w, int nslabs, int ktiles)
{
    w.n = 0; w.nslabs = nslabs; w.col_h = 0; w.col_stride = 0; w.nyq_b[0] = w.nyq_b[1] = w.nyq_col[0] = w.nyq_col[1] = 0u;
    w.kunit = (ktiles + 63) / 64; w.kt_total = ktiles;
    return nslabs >= 1 && nslabs <= 3 && ktiles >= 1 && (ktiles + w.kunit - 1) / w.kunit <= 127;      // slab / zero-fill fields are 2 bits
}
// Frames GEMM: M = live frames (frame-major), columns = the N taps, reduction K (a multiple of 32).  false: does not fit the list.
// Round 5b: more workgroups than CUs are allowed (several rounds of one workgroup per CU: B = 512 runs 504 entries in two rounds, 64 us against 107 on
// gemm_kernel<2, ...>); the k-slice count is the cheapest of 1 .. slabs by ntw_cost.
static inline bool ntw_frames(NTWork& w, const RowMap& live, int B, int H, int N, int pad, int Ls, int K, int nslabs, int ncus)
{
    const int R = live.rows(B), MT = (R + 127) / 128;
    if (N % 128 || K % 32 || MT > 255 || N / 128 > 63 || !ntw_header(w, nslabs, K / 32)) return false;
    int tiles = 0, c0[256], c1[256];
    for (int mt = 0; mt < MT; ++mt) {
        int f0, f1, lo, hi; ntw_tile_frames(live, B, R, mt, f0, f1); ntw_live_taps(f0, f1, H, N, pad, Ls, lo, hi);
        c0[mt] = lo / 128; c1[mt] = (hi + 127) / 128; tiles += c1[mt] - c0[mt];
    }
    const int ku = (w.kt_total + w.kunit - 1) / w.kunit;   // reduction length in k units
    int nact = 0; double best = 0.0;
    for (int c = 1; c <= nslabs && c <= ku; ++c) {
        if (tiles * c > NTW_MAX) break;
        const double t = ntw_cost(tiles * c, ((ku + c - 1) / c) * w.kunit, ncus);
        if (!nact || t < best - 1e-9) { nact = c; best = t; }
    }
    if (!nact || !ntw_rounds_ok(tiles * nact, ncus)) return false;
    for (int z = 0; z < nact; ++z) {
        const int k0 = (int)((long long)ku * z / nact), k1 = (int)((long long)ku * (z + 1) / nact);
        for (int mt = 0; mt < MT; ++mt)
            for (int nt = c0[mt]; nt < c1[mt]; ++nt) ntw_push(w, mt, nt, z, (z == 0 && nact < nslabs) ? nact : 0, 0, k0, k1 - k0);
    }
    ntw_xcd_order(w);
    return true;
}
// Data gradient: M = live frames (frame-major), reduction = the N taps of a frame (per tile row: the live ones only), columns = KP = 2 * FP spectral
// columns; F - 1 a multiple of 128: 2 (F - 1) / 128 tile columns + the two Nyquist columns as kind-1 entries, else ceil(KP / 128) plain tile columns.
static inline bool ntw_dgrad(NTWork& w, const RowMap& live, int B, int H, int N, int pad, int Ls, int F, int KP, int nslabs, int ncus)
{
    const int R = live.rows(B), MT = (R + 127) / 128;
    const bool nyq = (F - 1) % 128 == 0 && F > 1;
    const int NT = nyq ? 2 * (F - 1) / 128 : (KP + 127) / 128;
    if (N % 32 || MT > 255 || NT > 63 || !ntw_header(w, nslabs, N / 32)) return false;
    const int U = w.kunit;
    int k0[256], k1[256];                                  // live range of a tile row, in k units (rounded outward: the extra taps are zeros of the padded d syn)
    for (int mt = 0; mt < MT; ++mt) {
        int f0, f1, lo, hi; ntw_tile_frames(live, B, R, mt, f0, f1); ntw_live_taps(f0, f1, H, N, pad, Ls, lo, hi);
        k0[mt] = lo / (32 * U); k1[mt] = (hi + 32 * U - 1) / (32 * U);
    }
    // slice length c (k units): every tile row is cut into ceil(len / c) <= slabs equal slices; the cheapest c by ntw_cost
    const int ku = (w.kt_total + U - 1) / U;
    int cbest = 0, wbest = 0; double best = 0.0;
    for (int c = 1; c <= ku; ++c) {
        int sum = 0, longest = 0; bool ok = true;
        for (int mt = 0; mt < MT; ++mt) {
            const int len = k1[mt] - k0[mt], sl = (len + c - 1) / c;
            if (sl > nslabs) { ok = false; break; }
            sum += sl; const int each = (len + sl - 1) / sl; if (each > longest) longest = each;
        }
        if (!ok) continue;
        const int wgs = sum * NT + (nyq ? MT : 0);
        if (wgs > NTW_MAX) continue;
        const double t = ntw_cost(wgs, longest * U, ncus);
        if (!cbest || t < best - 1e-9) { cbest = c; best = t; wbest = wgs; }
    }
    if (!cbest || !ntw_rounds_ok(wbest, ncus)) return false;
    w.col_h = nyq ? F - 1 : 0; w.col_stride = KP / 2;
    w.nyq_b[0] = (unsigned)(F - 1) * (unsigned)N; w.nyq_b[1] = (unsigned)(KP / 2 + F - 1) * (unsigned)N; w.nyq_col[0] = (unsigned)(F - 1); w.nyq_col[1] = (unsigned)(KP / 2 + F - 1);
    for (int z = 0; z < nslabs; ++z)
        for (int mt = 0; mt < MT; ++mt) {
            const int len = k1[mt] - k0[mt], sl = (len + cbest - 1) / cbest;
            if (z >= sl) continue;
            const int a = k0[mt] + (int)((long long)len * z / sl), b = k0[mt] + (int)((long long)len * (z + 1) / sl);
            for (int nt = 0; nt < NT; ++nt) ntw_push(w, mt, nt, z, (z == 0 && sl < nslabs) ? sl : 0, 0, a, b - a);
        }
    ntw_xcd_order(w);
    if (nyq) for (int mt = 0; mt < MT; ++mt) ntw_push(w, mt, 0, 0, 0, 1, k0[mt], k1[mt] - k0[mt]);      // behind the tiles: dispatched last, on the CUs the tiles leave free
    return true;
}

// Launched with MORE than half the LDS of a CU so that no two workgroups share one: with <= #CUs entries every workgroup then has a CU
// to itself -- measured without this, 224 + 112 workgroups at 2 per CU: the dispatcher doubled up heavy ones and the GEMM took 79 us (61 TFLOP/s).
static inline int launch_nt128(const NTRows& ra, const NTRows& rb, const StoreSlab& epi, const NTWork& wk, hipStream_t s)
{
    constexpr size_t lds = (size_t)84 * 1024;              // tiles: 4 * 128 * 36 * 4 = 72 KB; 84 KB > 160 / 2 keeps a CU to one workgroup
    const int rc = ::ensure_dyn_lds((const void*)gemm_nt128_kernel, "gemm_nt128_kernel"); if (rc) return rc;
    hipLaunchKernelGGL(gemm_nt128_kernel, dim3(wk.n), dim3(256), lds, s, ra, rb, epi, wk);
    return 0;
}

}  // namespace stg
