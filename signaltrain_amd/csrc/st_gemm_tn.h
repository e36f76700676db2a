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
};

// One extra z-slice of workgroups (blockIdx.z == nsplit) forms the two Nyquist rows C[c][n] = sum_k A[k][c] * B[k][n], c in {nyq_c0, nyq_c1},
// as nyq_P partial sums over groups of windows: out[p][0 | 1][n].  Light vector work (K / nyq_P rows of float4 FMAs per thread) that runs
// beside the MFMA workgroups as a second resident workgroup of its CU; the slab-reduce kernel adds the nyq_P partials in a fixed order.
__device__ __forceinline__ void nyq_partial(const TNJob& j, const int p)
{
    if (p >= j.nyq_P) return;
    const int W = j.K / j.Tv;                                   // windows
    const int per = (W + j.nyq_P - 1) / j.nyq_P, w0 = p * per, w1 = (w0 + per < W) ? w0 + per : W;
    for (int n4 = threadIdx.x; n4 < j.Nc / 4; n4 += 256) {
        float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0;
        // rows k = w * Tv + i, eight at a time with all their loads issued before the first FMA (the order of the sum stays k-ascending)
        const int k0 = w0 * j.Tv, k1 = w1 * j.Tv;
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
    const int k_begin = tbz * ksplit;
    const int k_end = (k_begin + ksplit < j.K) ? k_begin + ksplit : j.K;

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
                               float* nyq_out = nullptr, unsigned nyq_c0 = 0, unsigned nyq_c1 = 0, int* nyq_P = nullptr)
{
    const float* lo = A.base < B.base ? A.base : B.base; if (zeros < lo) lo = zeros;
    TNJob j;
    j.base = lo; j.a0 = (unsigned)(A.base - lo); j.b0 = (unsigned)(B.base - lo); j.zero = (unsigned)(zeros - lo);
    j.SA1 = A.S1; j.SA2 = A.S2; j.SB1 = B.S1; j.SB2 = B.S2;
    j.magic = map.magic; j.Tv = map.Tv; j.t_lo = map.t_lo; j.K = K; j.mh = mh; j.mstride = mstride;
    int ksplit = K;
    if (nsplit > 1) ksplit = st_round_up((K + nsplit - 1) / nsplit, BKT);
    constexpr size_t lds = (size_t)4 * BKT * 128 * sizeof(float);
    if (lds > 65536) { const int rc = ::ensure_dyn_lds((const void*)gemm_tn128_kernel<BKT>, "gemm_tn128_kernel"); if (rc) return rc; }
    const int nz = nsplit > 1 ? nsplit : 1, tiles = (Nc / 128) * (M / 128);
    j.nsplit = nz; j.Nc = Nc; j.nyq_out = nyq_out; j.nyq_c0 = nyq_c0; j.nyq_c1 = nyq_c1;
    j.nyq_P = tiles < 64 ? tiles : 64; { const int W = K / map.Tv; if (j.nyq_P > W) j.nyq_P = W; }
    if (nyq_P) *nyq_P = j.nyq_P;
    hipLaunchKernelGGL((gemm_tn128_kernel<BKT>), dim3(Nc / 128, M / 128, nz + (nyq_out ? 1 : 0)), dim3(256), lds, s, j, out, ldo, slab, ksplit);
    return 0;
}

}  // namespace stg

// ================================================================================================ NT x NT on the same 128 x 128 tiles (round 3)
//   C[m][n] = sum_k A[m][k] * B[n][k],  both operands K-contiguous:  the synthesis FRAMES GEMM (cls_fe_dft.py:112 as a GEMM: A = spectra AA
//   [live frames][KP], B = transposed fold [N][KP]) -- 58 us = 53 % of the fp32 peak on the 64 x 96 workgroup tiles of gemm_kernel<2, ...>
//   (M = 1792 live frames only: 19 x 11 small tiles x 3 k-slices).  Here: 14 x 8 tiles of 128 x 128 x 2 k-slices = 224 workgroups, one per CU;
//   row-major LDS tiles [row][BK + 4] (a global float4 along k = one ds_write_b128; a lane's 16 k of its row = four ds_read_b128, conflict-free at
//   pitch 36), lane half h takes k in [16 h, 16 h + 16) of the 32-deep tile as in gemm_kernel; row offsets are per-thread constants and the k
//   offset rides in the scalar base: NO address arithmetic in the loop.
namespace stg {

struct NTRows { const float* base; unsigned S1, S2, magic; int Tv, t_lo, R; };      // row r -> base[off(r) + k]; off = b * S1 + t * S2, (b, t) = split(min(r, R - 1)); magic == 0: b = r, t = 0
__device__ __forceinline__ unsigned ntrows_off(const NTRows& o, const int r)
{
    const unsigned rc = (unsigned)(r < o.R ? r : o.R - 1);
    const unsigned b = o.magic ? __umulhi(rc, o.magic) : rc;
    const unsigned t = o.magic ? (unsigned)o.t_lo + (rc - b * (unsigned)o.Tv) : 0u;
    return b * o.S1 + t * o.S2;
}

template <class EPI>
__global__ void __launch_bounds__(256)
gemm_nt128_kernel(const NTRows ra, const NTRows rb, const EPI epi, const int K, const int ksplit, const int nzero)
{
    constexpr int BKT = 32, LD = BKT + 4, TS = 128 * LD;
    constexpr int NP = 4;                                  // 256 threads = 32 rows x 8 float4 per pass
    extern __shared__ __attribute__((aligned(16))) float nt_lds[];       // As[2][128][LD] | Bs[2][128][LD]
    float* const As = nt_lds;
    float* const Bs = nt_lds + 2 * TS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    int tbx, tby, tbz; xcd_tile(tbx, tby, tbz);
    const int m_blk = tby * 128, n_blk = tbx * 128;
    const int k_begin = tbz * ksplit;
    const int k_end = (k_begin + ksplit < K) ? k_begin + ksplit : K;

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
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) epi(m_blk + wm * 64 + 32 * mi, n_blk + wn * 64, acc[mi]);
    if (nzero > 0 && tbz == 0) {                           // the slabs no k-slice computes: this tile of each, zeros
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int nj = 0; nj < 2; ++nj)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[mi][nj][i] = 0.f;
        for (int z = 0; z < nzero; ++z) {
            const EPI ez = epi.slab_shifted((int)gridDim.z + z);
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) ez(m_blk + wm * 64 + 32 * mi, n_blk + wn * 64, acc[mi]);
        }
    }
}

// nslabs: slabs the consumer sums (grid z); nactive <= nslabs: k-slices that carry work -- the others start past K and store zeros
// (the consumers' slab counts are a function of the geometry alone; this kernel fills the chip with fewer, longer slices).
// Launched with MORE than half the LDS of a CU so that no two workgroups share one: with tiles x nactive <= #CUs every workgroup then has a CU
// to itself -- measured without this, 224 + 112 workgroups at 2 per CU: the dispatcher doubled up heavy ones and the GEMM took 79 us (61 TFLOP/s).
template <class EPI>
static inline int launch_nt128(const NTRows& ra, const NTRows& rb, const EPI& epi, int M, int Nc, int K, int nslabs, int nactive, hipStream_t s)
{
    int ksplit = K;
    if (nactive > 1) ksplit = st_round_up((K + nactive - 1) / nactive, 32);
    constexpr size_t lds = (size_t)84 * 1024;              // tiles: 4 * 128 * 36 * 4 = 72 KB; 84 KB > 160 / 2 keeps a CU to one workgroup
    const int rc = ::ensure_dyn_lds((const void*)gemm_nt128_kernel<EPI>, "gemm_nt128_kernel"); if (rc) return rc;
    hipLaunchKernelGGL((gemm_nt128_kernel<EPI>), dim3((Nc + 127) / 128, (M + 127) / 128, nactive > 1 ? nactive : 1), dim3(256), lds, s, ra, rb, epi, K, ksplit, nslabs - (nactive > 1 ? nactive : 1));
    return 0;
}

}  // namespace stg
