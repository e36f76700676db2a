// st_gemm16.h -- the STFT GEMMs of the 16-bit configurations on PRE-ROUNDED 16-bit operands (gfx950), round 3.
//
// BASELINE.json configs[2], [3] (bf16) and [4] (fp16): every GEMM operand is stored in 16 bits by the kernel that produces it
// (prep_kernel: padded waveform, analysis bases, folded synthesis bases in both orientations; ae_fwd: the spectra AA; ola_loss: d syn;
// post_ae: d G) -- the SAME values the rounding oracle uses (round each GEMM operand to bf16 / fp16, multiply exactly, accumulate in
// fp32), so parity is that of gemm_half_kernel, which converted fp32 operands while staging them (SQ_INSTS_VALU / SQ_INSTS_MFMA = 13,
// analysis forward 42 us = 13 % of the bf16 peak at B = 256).  Here no conversion and no fp32 byte is left in any k-loop:
//   * workgroup tile 128 x 128, four waves as 2 x 2, wave tile 64 x 64 = 2 x 2 accumulators of v_mfma_f32_32x32x16_{bf16,f16}
//     (one 16-byte LDS read per operand block per MFMA k-step: 4 reads per 4 MFMAs; the 32 x 96 strips needed 4 per 3 and, at two
//     workgroups per CU, more LDS bandwidth than the CU has);
//   * K-contiguous ("NT") operands: row-major LDS tiles [row][BK + 8] (pitch 144 / 80 bytes: the 16 lanes of a ds_read_b128 service
//     group hit 16 distinct 16-byte slots), a global dwordx4 = 8 k is one ds_write_b128;
//   * M/N-contiguous ("TN") operands of the weight-gradient GEMMs: k-major tiles [k][128 + 32], a global dwordx4 = 8 rows at one k is
//     one ds_write_b128, and the MFMA fragments come out of ds_read_b64_tr_b16 -- the LDS transpose read of gfx950: a 16-lane group
//     reads a [4 k][16 rows] block, lane i gets the four k of row i -- two reads per operand block per k-step, no register transposes
//     (gemm_half_kernel transposed 4 x 4 fp32 micro-tiles with VALU moves);
//   * MEASURED and dropped (round 3): a second register set so that the loads of tile t + 2 are in flight during tile t (analysis forward 32.1 ->
//     35.6 us): these kernels are not waiting for their loads -- the analysis forward WRITES 48 MB of fp32 re / im / mag / phs (12 us at HBM speed)
//     through 4-byte scattered stores, the synthesis GEMMs write 22 MB of split-K slabs each;
//   * row offsets of NT operands are per-thread constants, the k offset rides in the scalar base; TN operands split k -> (window, frame)
//     once per load pass with 24-bit multiplies.
#pragma once
#include "st_gemm.h"

namespace stg {

typedef unsigned short h16_t;
typedef short st_s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned st_u32x4 __attribute__((ext_vector_type(4)));      // native vector (HIP's uint4 struct in a captured array ends up in scratch)

// NT-type operand: row r -> elements base[off(r) + k], k contiguous.  off = b * S1 + t * S2 with (b, t) = RowMap::split(min(r, R - 1))
// (magic == 0: b = r, t = 0).  Rows past R are clamped: their products only reach outputs the epilogue masks.
struct Rows16 { const h16_t* base; unsigned S1, S2, magic; int Tv, t_lo, R; };
__device__ __forceinline__ unsigned rows16_off(const Rows16& o, const int r)
{
    const unsigned rc = (unsigned)(r < o.R ? r : o.R - 1);
    const unsigned b = o.magic ? __umulhi(rc, o.magic) : rc;
    const unsigned t = o.magic ? (unsigned)o.t_lo + (rc - b * (unsigned)o.Tv) : 0u;
    return b * o.S1 + t * o.S2;
}
static inline Rows16 rows16(const h16_t* base, unsigned S1, unsigned S2, const RowMap& m, int R) { return Rows16{base, S1, S2, m.magic, m.Tv, m.t_lo, R}; }
static inline Rows16 rows16_plain(const h16_t* base, unsigned ld, int R) { return Rows16{base, ld, 0u, 0u, 1, 0, R}; }
// the rows of a [windows][frames] array enumerated FRAME-major (RowMap::fm, round 5): row r -> frame t_lo + r / B, window r % B
static inline Rows16 rows16_frame_major(const h16_t* base, unsigned s_window, unsigned s_frame, const RowMap& live, int B, int R)
{
    return Rows16{base + (size_t)live.t_lo * s_frame, s_frame, s_window, B > 1 ? rowmap_magic(B) : 0u, B, 0, R};
}
// Round 5: the structural zeros of the cropped transposed convolution (cls_fe_dft.py:112-113; see st_gemm_tn.h) in the 16-bit synthesis GEMMs.  Rows are the live
// frames enumerated frame-major; frames f0..f1 of a tile row keep the taps [lo, hi) = [pad - H f1, pad + Ls - H f0) n [0, N).
//   mode 1 (frames GEMM, columns = taps): a tile whose columns hold no live tap returns at once -- nothing reads it;
//   mode 2 (data gradient, reduction = taps): the k-slices of a tile row divide ITS live tap range (rounded to the k-tile: the extra taps are zeros of the padded d syn).
struct Crop16 { int mode, B, H, N, pad, Ls, t_lo, R, nsplit; };

template <int HT> struct frag16 { typedef st_bf16x8 type; };
template <> struct frag16<2> { typedef st_f16x8 type; };
template <int HT>
__device__ __forceinline__ f32x16 mfma16x(const typename frag16<HT>::type a, const typename frag16<HT>::type b, const f32x16 c)
{
    if constexpr (HT == 2) return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// ------------------------------------------------------------------------------ NT x NT:  C[m][n] = sum_k A[m][k] * B[n][k]
template <int HT, int BKH, class EPI>
__global__ void __launch_bounds__(256)
gemm16_nt_kernel(const Rows16 ra, const Rows16 rb, const EPI epi, const int K, const int ksplit, const Crop16 cr)
{
    typedef typename frag16<HT>::type frag_t;
    constexpr int LD = BKH + 8;                          // elements per LDS row
    constexpr int TS = 128 * LD;                         // elements per operand tile
    constexpr int TPR = BKH / 8, RP = 256 / TPR, NP = 128 / RP, KS = BKH / 16;
    extern __shared__ __attribute__((aligned(16))) h16_t g16_lds[];      // As[2][128][LD] | Bs[2][128][LD]
    h16_t* const As = g16_lds;
    h16_t* const Bs = g16_lds + 2 * TS;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    int tbx, tby, tbz;
    if constexpr (kPolarEpi<EPI>) xcd_tile_2d(tbx, tby, tbz); else xcd_tile(tbx, tby, tbz);      // analysis forward: 2-D blocks of tiles per XCD (st_gemm.h)
    const int m_blk = tby * 128, n_blk = tbx * 128;
    int k_begin = tbz * ksplit;
    int k_end = (k_begin + ksplit < K) ? k_begin + ksplit : K;
    if (cr.mode) {                                        // workgroup-uniform
        const int r1 = (m_blk + 128 < cr.R ? m_blk + 128 : cr.R) - 1;
        const int f0 = cr.t_lo + m_blk / cr.B, f1 = cr.t_lo + r1 / cr.B;
        int lo = cr.pad - cr.H * f1, hi = cr.pad + cr.Ls - cr.H * f0;
        lo = lo < 0 ? 0 : lo; hi = hi > cr.N ? cr.N : hi;
        if (cr.mode == 1) { if (n_blk + 128 <= lo || n_blk >= hi) return; }
        else {
            const int klo = lo / BKH * BKH, khi = (hi + BKH - 1) / BKH * BKH;
            const int per = ((khi - klo + cr.nsplit - 1) / cr.nsplit + BKH - 1) / BKH * BKH;
            k_begin = klo + tbz * per;
            k_end = (k_begin + per < khi) ? k_begin + per : khi;
        }
    }

    const int lr = tid / TPR, lk = (tid % TPR) * 8;
    unsigned ao[NP], bo[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        ao[p] = 2u * (rows16_off(ra, m_blk + lr + RP * p) + (unsigned)lk);       // BYTE offsets
        bo[p] = 2u * (rows16_off(rb, n_blk + lr + RP * p) + (unsigned)lk);
    }
    st_u32x4 va[NP], vb[NP];
    auto gload = [&](const int kt) {
        const char* pa = reinterpret_cast<const char*>(ra.base) + 2 * (size_t)kt;     // wave-uniform: the k offset rides in the scalar base
        const char* pb = reinterpret_cast<const char*>(rb.base) + 2 * (size_t)kt;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            va[p] = *reinterpret_cast<const st_u32x4*>(pa + ao[p]);
            vb[p] = *reinterpret_cast<const st_u32x4*>(pb + bo[p]);
        }
    };
    auto lstore = [&](const int buf) {
        h16_t* as = As + buf * TS + lr * LD + lk;
        h16_t* bs = Bs + buf * TS + lr * LD + lk;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            *reinterpret_cast<st_u32x4*>(as + p * RP * LD) = va[p];
            *reinterpret_cast<st_u32x4*>(bs + p * RP * LD) = vb[p];
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
        const int g = lane >> 5, l31 = lane & 31;
        const int a_off = (wm * 64 + l31) * LD + 8 * g;
        const int b_off = (wn * 64 + l31) * LD + 8 * g;
        for (int kt = k_begin; kt < k_end; kt += BKH) {
            const bool more = kt + BKH < k_end;
            gload(more ? kt + BKH : kt);                     // branch-free body (the last iteration re-loads its own tile)
            __builtin_amdgcn_sched_barrier(0);
            const h16_t* as = As + cur * TS + a_off;
            const h16_t* bs = Bs + cur * TS + b_off;
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const frag_t a0 = *reinterpret_cast<const frag_t*>(as + 16 * s), a1 = *reinterpret_cast<const frag_t*>(as + 32 * LD + 16 * s);
                const frag_t b0 = *reinterpret_cast<const frag_t*>(bs + 16 * s), b1 = *reinterpret_cast<const frag_t*>(bs + 32 * LD + 16 * s);
                acc[0][0] = mfma16x<HT>(a0, b0, acc[0][0]);
                acc[0][1] = mfma16x<HT>(a0, b1, acc[0][1]);
                acc[1][0] = mfma16x<HT>(a1, b0, acc[1][0]);
                acc[1][1] = mfma16x<HT>(a1, b1, acc[1][1]);
            }
            __builtin_amdgcn_sched_barrier(0);
            lstore(cur ^ 1);
            __syncthreads();
            cur ^= 1;
        }
    }
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) epi(m_blk + wm * 64 + 32 * mi, n_blk + wn * 64, acc[mi]);
}

// ------------------------------------------------------------------------------ NT x NT through LDS-DMA (round 3)
// gemm16_nt_kernel above re-fetches (128 + 128) x 64 operand elements per 128 x 128 x 64 of work through registers, one tile ahead: measured
// (bf16, B = 1024 analysis forward, epilogue ablated) 94 us = 590 TFLOP/s -- each k-tile costs a workgroup ~4100 cycles against 512 of MFMA.
// Here:  * workgroup tile 256 x 128, eight waves as 4 x 2 (wave tile 64 x 64 as before: two waves per SIMD, the partner fills the stalls), one
//          workgroup per CU: 25 % less L2 -> LDS traffic per unit of work;
//        * operands go global -> LDS by global_load_lds_dwordx4 (no staging registers, no ds_write pass), THREE 48 KB stages, the loads of tile
//          t + 2 issued while tile t is multiplied: the wait before tile t is vmcnt(6) (this thread's six loads of tile t + 1 may still fly),
//          never 0, and there is ONE barrier per tile (it both publishes tile t and retires the readers of the stage tile t + 2 overwrites);
//        * an LDS-DMA instruction writes its 64 x 16 bytes lane-linear, so the bank swizzle is applied to the SOURCE address: LDS row r
//          (128 bytes = 8 chunks of 8 k) holds chunk c at position c ^ ((r >> 1) & 7); the 16 rows x one chunk a ds_read_b128 service group
//          fetches then cover all 64 banks (same involution on the read side).
// K % 64 == 0 only (analysis forward: K = N; synthesis data gradient: K = N).
template <int HT, class EPI>
__global__ void __launch_bounds__(768)
gemm16_nt256_kernel(const Rows16 ra, const Rows16 rb, const EPI epi, const int K, const int ksplit, const int dbg)
{
    typedef typename frag16<HT>::type frag_t;
    constexpr int A_BYTES = 256 * 128, B_BYTES = 128 * 128, STAGE = A_BYTES + B_BYTES, NST = 3;
    extern __shared__ __attribute__((aligned(16))) h16_t g16_lds[];      // NST x [A 256 rows x 128 B | B 128 rows x 128 B]
    char* const lds = reinterpret_cast<char*>(g16_lds);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int tbx, tby, tbz; xcd_tile(tbx, tby, tbz);
    const int m_blk = tby * 256, n_blk = tbx * 128;
    const int k_begin = tbz * ksplit;
    const int k_end = (k_begin + ksplit < K) ? k_begin + ksplit : K;
    const int nt = (dbg & 8) ? 1 : (k_end - k_begin) / 64;
    if (nt <= 0) {                                                          // a k-slice past K: zeros (consumers only)
        if (wave < 8) {
            f32x16 z[2];
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int i = 0; i < 16; ++i) z[j][i] = 0.f;
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) epi(m_blk + (wave >> 1) * 64 + 32 * mi, n_blk + (wave & 1) * 64, z);
        }
        return;
    }

    // ---------------------------------------------------------------- PRODUCERS: waves 8..11 (one per SIMD) only move data
    // An LDS-DMA instruction holds its wave until the CU's one address path has taken it -- measured 100 .. 190 cycles each beside ds_reads: issued
    // by the multiplying waves themselves (6 per wave per tile) they made the READ phase 1150 cycles long against 512 of MFMAs.  A producer has
    // nothing else to do.  Producer p issues 8 instructions of A (LDS rows 64 p .. 64 p + 63) and 4 of B (rows 32 p ..) per tile; lane l fills
    // position l & 7 of row l >> 3 of its 8-row piece with source chunk (l & 7) ^ ((row >> 1) & 7).
    if (wave >= 8) {
        const int p = wave - 8, dr = lane >> 3, dp = lane & 7;
        unsigned ao[8], bo[4];
#pragma unroll
        for (int i = 0; i < 8; ++i) { const int r = 8 * (8 * p + i) + dr; ao[i] = 2u * (rows16_off(ra, m_blk + r) + 8u * (unsigned)(dp ^ ((r >> 1) & 7))); }
#pragma unroll
        for (int i = 0; i < 4; ++i) { const int r = 8 * (4 * p + i) + dr; bo[i] = 2u * (rows16_off(rb, n_blk + r) + 8u * (unsigned)(dp ^ ((r >> 1) & 7))); }
        typedef __attribute__((address_space(3))) void* lds_ptr_t;
        typedef const __attribute__((address_space(1))) void* glb_ptr_t;
        auto issue = [&](const int tile, const int stage) {
            const int tc = tile < nt ? tile : nt - 1;                      // past the end: reload the last tile into a free stage (uniform wait counts)
            const char* pa = reinterpret_cast<const char*>(ra.base) + 2 * (size_t)(k_begin + 64 * tc);
            const char* pb = reinterpret_cast<const char*>(rb.base) + 2 * (size_t)(k_begin + 64 * tc);
            char* la = lds + stage * STAGE + 1024 * (8 * p);
            char* lb = lds + stage * STAGE + A_BYTES + 1024 * (4 * p);
#pragma unroll
            for (int i = 0; i < 8; ++i) __builtin_amdgcn_global_load_lds((glb_ptr_t)(pa + ao[i]), (lds_ptr_t)(la + 1024 * i), 16, 0, 0);
#pragma unroll
            for (int i = 0; i < 4; ++i) __builtin_amdgcn_global_load_lds((glb_ptr_t)(pb + bo[i]), (lds_ptr_t)(lb + 1024 * i), 16, 0, 0);
        };
        issue(0, 0);
        issue(1, 1);
        asm volatile("s_waitcnt vmcnt(12)" ::: "memory");                  // tile 0 has landed
        __builtin_amdgcn_s_barrier();                                       // P0
        int stage = 0;
        for (int t = 0; t < nt; ++t) {
            // start of phase 2 t: group B fetched tile t - 1 during phase 2 t - 1 -> stage (t - 1) % 3 == (t + 2) % 3 is free
            if (!(dbg & 1)) issue(t + 2, stage >= 1 ? stage - 1 : NST - 1);
            asm volatile("s_waitcnt vmcnt(12)" ::: "memory");              // tile t + 1 (issued a whole tile ago) has landed: group A reads it in phase 2 t + 2
            __builtin_amdgcn_s_barrier();                                   // end of phase 2 t
            __builtin_amdgcn_s_barrier();                                   // end of phase 2 t + 1
            stage = stage == NST - 1 ? 0 : stage + 1;
        }
        __builtin_amdgcn_s_barrier();                                       // end of phase 2 nt (group B's last MULTIPLY phase)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                   // the tail reloads: nothing may still be writing LDS when the workgroup retires
        return;
    }

    // ---------------------------------------------------------------- CONSUMERS: waves 0..7 as 4 x 2, wave tile 64 x 64
    // PING-PONG: waves w and w + 4 share a SIMD; group A = waves 0..3, group B = waves 4..7 runs ONE PHASE behind.  A consumer alternates a READ phase
    // (the 16 fragments of tile t into registers) and a MULTIPLY phase (16 MFMAs), a barrier after each: while one wave of a SIMD multiplies, the
    // other one reads (all eight in step, a k-tile cost LDS time + MFMA time: 2700 .. 3400 cycles against 1024 of MFMAs per SIMD).
    const int wm = wave >> 1, wn = wave & 1;
    f32x16 acc[2][2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int nj = 0; nj < 2; ++nj)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[mi][nj][i] = 0.f;
    {
        const int g = lane >> 5, l31 = lane & 31;
        const int p0 = (g ^ (l31 >> 1)) & 7;                               // chunk position of k-step 0 (chunk g of row l31); k-step s (chunk 2 s + g): p0 ^ 2 s
        int a_off[4], b_off[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            a_off[s] = (wm * 64 + l31) * 128 + 16 * (p0 ^ (2 * s));
            b_off[s] = A_BYTES + (wn * 64 + l31) * 128 + 16 * (p0 ^ (2 * s));
        }
        const bool grp_b = wave >= 4;
        __builtin_amdgcn_s_barrier();                                       // P0: tile 0 is in LDS
        if (grp_b) __builtin_amdgcn_s_barrier();
        int stage = 0;
        for (int t = 0; t < nt; ++t) {
            __builtin_amdgcn_sched_barrier(0);
            const char* st = lds + stage * STAGE;
            frag_t fa[4][2], fb[4][2];                                      // [k-step][32-row block]
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                fa[s][0] = *reinterpret_cast<const frag_t*>(st + a_off[s]); fb[s][0] = *reinterpret_cast<const frag_t*>(st + b_off[s]);
                fa[s][1] = *reinterpret_cast<const frag_t*>(st + a_off[s] + 32 * 128); fb[s][1] = *reinterpret_cast<const frag_t*>(st + b_off[s] + 32 * 128);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");             // the stage may be overwritten once everyone is past the next barrier
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                acc[0][0] = mfma16x<HT>(fa[s][0], fb[s][0], acc[0][0]);
                acc[0][1] = mfma16x<HT>(fa[s][0], fb[s][1], acc[0][1]);
                acc[1][0] = mfma16x<HT>(fa[s][1], fb[s][0], acc[1][0]);
                acc[1][1] = mfma16x<HT>(fa[s][1], fb[s][1], acc[1][1]);
            }
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            stage = stage == NST - 1 ? 0 : stage + 1;
        }
        if (!grp_b) __builtin_amdgcn_s_barrier();
    }
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) epi(m_blk + wm * 64 + 32 * mi, n_blk + wn * 64, acc[mi]);
}

// ------------------------------------------------------------------------------ TN x TN:  C[m][n] = sum_k A[k][m] * B[k][n]
// (the weight-gradient GEMMs: k = compact live frame row; A = d G / AA, B = frames of the padded waveform / of d syn, all 16-bit)
struct TN16Job {
    const h16_t* base;             // common base of both operands and the zero block (element offsets below, < 2^30)
    unsigned a0, b0, zero;         // origins; >= 128 + 32 zero elements for A rows past K
    unsigned SA1, SA2, SB1, SB2;   // element (k, c): a0 + b * SA1 + t * SA2 + c  /  b0 + b * SB1 + t * SB2 + c,  (b, t) = split(k)
    unsigned magic; int Tv, t_lo, K;
    int trim, fB, nsplit; unsigned char fa[64], fb[64];      // round 5, frame-major reduction order: tile column tx needs the rows [fa[tx] * fB, fb[tx] * fB) only (st_gemm_tn.h FrameTrim)
};
template <int HT, int BKH>
__global__ void __launch_bounds__(256)
gemm16_tn_kernel(const TN16Job j, const StoreC epi, const int ksplit)
{
    typedef typename frag16<HT>::type frag_t;
    constexpr int LD = 128 + 32;                          // elements per k row: 320 bytes -- the 4 k rows x 2 row-halves a 32-lane half reads
                                                          // with ds_read_b64_tr_b16 land on 8 distinct 32-byte bank groups
    constexpr int TS = BKH * LD;
    constexpr int NP = BKH / 16, KS = BKH / 16;           // 16 threads x 8 elements per k row, 16 k rows per load pass
    extern __shared__ __attribute__((aligned(16))) h16_t g16_lds[];      // As[2][BKH][LD] | Bs[2][BKH][LD]
    h16_t* const As = g16_lds;
    h16_t* const Bs = g16_lds + 2 * TS;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    int tbx, tby, tbz; xcd_tile(tbx, tby, tbz);
    const int m_blk = tby * 128, n_blk = tbx * 128;
    int k_begin = tbz * ksplit;
    int k_end = (k_begin + ksplit < j.K) ? k_begin + ksplit : j.K;
    if (j.trim) {                                         // the rows that carry anything for this tile column, divided among the k-slices
        const int lo = (int)j.fa[tbx] * j.fB, hi = (int)j.fb[tbx] * j.fB;
        const int per = ((hi - lo + j.nsplit - 1) / j.nsplit + BKH - 1) / BKH * BKH;
        k_begin = lo + tbz * per;
        k_end = (k_begin + per < hi) ? k_begin + per : hi;      // a last k-tile may run past hi: structural zeros of B for this tile column (or past K: masked)
    }

    const int lk = tid >> 4, lc = (tid & 15) * 8;
    const unsigned la = 2u * (j.a0 + (unsigned)m_blk + (unsigned)lc), lb = 2u * (j.b0 + (unsigned)n_blk + (unsigned)lc), lz = 2u * (j.zero + (unsigned)lc);
    const unsigned sa1 = 2u * j.SA1, sa2 = 2u * j.SA2, sb1 = 2u * j.SB1, sb2 = 2u * j.SB2;
    st_u32x4 va[NP], vb[NP];
    auto ld = [&](const unsigned byte_off) { return *reinterpret_cast<const st_u32x4*>(reinterpret_cast<const char*>(j.base) + byte_off); };
    auto gload = [&](const int kt) {
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const int k = kt + lk + 16 * p;
            const unsigned kc = (unsigned)(k < j.K ? k : j.K - 1);
            const unsigned b = __umulhi(kc, j.magic);
            const unsigned t = (unsigned)j.t_lo + kc - __umul24(b, (unsigned)j.Tv);
            const unsigned oa = __umul24(b, sa1) + __umul24(t, sa2) + la;
            const unsigned ob = __umul24(b, sb1) + __umul24(t, sb2) + lb;
            const unsigned live = (unsigned)((k - j.K) >> 31);
            va[p] = ld((oa & live) | (lz & ~live));
            vb[p] = ld(ob);
        }
    };
    auto lstore = [&](const int buf) {
        h16_t* as = As + buf * TS + lk * LD + lc;
        h16_t* bs = Bs + buf * TS + lk * LD + lc;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            *reinterpret_cast<st_u32x4*>(as + p * 16 * LD) = va[p];
            *reinterpret_cast<st_u32x4*>(bs + p * 16 * LD) = vb[p];
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
        // transpose read: lane l = 16 G + i supplies the address of 4 contiguous elements of k row (i >> 2), columns 4 (i & 3) .. + 3 of the
        // [4 k][16 rows] block its 16-lane group reads, and receives the 4 k of column i.  MFMA lane l is row (l & 31) = 16 (G & 1) + i,
        // k group (l >> 5) = G >> 1: block origin k = 16 s + 8 (G >> 1) + 4 q, row = 32 blk + 16 (G & 1).
        const int G = lane >> 4, i16 = lane & 15;
        const int t_off = (8 * (G >> 1) + (i16 >> 2)) * LD + 16 * (G & 1) + 4 * (i16 & 3);
        const int a_off = t_off + wm * 64, b_off = t_off + wn * 64;
        for (int kt = k_begin; kt < k_end; kt += BKH) {
            const bool more = kt + BKH < k_end;
            gload(more ? kt + BKH : kt);
            __builtin_amdgcn_sched_barrier(0);
            const h16_t* as = As + cur * TS + a_off;
            const h16_t* bs = Bs + cur * TS + b_off;
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                union { st_s16x4 h[2]; frag_t f; } a[2], b[2];
#pragma unroll
                for (int x = 0; x < 2; ++x)
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        a[x].h[q] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((st_s16x4 __attribute__((address_space(3)))*)(as + (16 * s + 4 * q) * LD + 32 * x));
                        b[x].h[q] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((st_s16x4 __attribute__((address_space(3)))*)(bs + (16 * s + 4 * q) * LD + 32 * x));
                    }
                acc[0][0] = mfma16x<HT>(a[0].f, b[0].f, acc[0][0]);
                acc[0][1] = mfma16x<HT>(a[0].f, b[1].f, acc[0][1]);
                acc[1][0] = mfma16x<HT>(a[1].f, b[0].f, acc[1][0]);
                acc[1][1] = mfma16x<HT>(a[1].f, b[1].f, acc[1][1]);
            }
            __builtin_amdgcn_sched_barrier(0);
            lstore(cur ^ 1);
            __syncthreads();
            cur ^= 1;
        }
    }
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) epi(m_blk + wm * 64 + 32 * mi, n_blk + wn * 64, acc[mi]);
}

// ------------------------------------------------------------------------------ host side
template <int HT, class EPI>
static inline int launch16_nt(const Rows16& ra, const Rows16& rb, const EPI& epi, int M, int Nc, int K, int nsplit, hipStream_t s, bool allow64 = true, const Crop16* crop = nullptr)
{
    dim3 grid((Nc + 127) / 128, (M + 127) / 128, nsplit > 1 ? nsplit : 1);
    Crop16 cr{}; cr.mode = 0; if (crop) cr = *crop; cr.nsplit = nsplit > 1 ? nsplit : 1;
    const bool k64 = allow64 && K % 64 == 0 && (nsplit <= 1 || (K / 64) % nsplit == 0);
    if (k64) {
        int ksplit = K; if (nsplit > 1) ksplit = st_round_up((K + nsplit - 1) / nsplit, 64);
        constexpr size_t lds = (size_t)4 * 128 * (64 + 8) * sizeof(h16_t);
        const int rc = ::ensure_dyn_lds((const void*)gemm16_nt_kernel<HT, 64, EPI>, "gemm16_nt_kernel"); if (rc) return rc;
        hipLaunchKernelGGL((gemm16_nt_kernel<HT, 64, EPI>), grid, dim3(256), lds, s, ra, rb, epi, K, ksplit, cr);
    } else {
        int ksplit = K; if (nsplit > 1) ksplit = st_round_up((K + nsplit - 1) / nsplit, 32);
        constexpr size_t lds = (size_t)4 * 128 * (32 + 8) * sizeof(h16_t);
        hipLaunchKernelGGL((gemm16_nt_kernel<HT, 32, EPI>), grid, dim3(256), lds, s, ra, rb, epi, K, ksplit, cr);
    }
    return 0;
}
template <int HT, class EPI>
static inline bool nt256_fits(int K, int nsplit) { return K % 64 == 0 && K / 64 >= (nsplit > 1 ? nsplit : 1); }
template <int HT, class EPI>
static inline int launch16_nt256(const Rows16& ra, const Rows16& rb, const EPI& epi, int M, int Nc, int K, int nsplit, hipStream_t s, int dbg = 0)
{
    int ksplit = K; if (nsplit > 1) ksplit = st_round_up((K + nsplit - 1) / nsplit, 64);
    constexpr size_t lds = (size_t)3 * (256 + 128) * 128;
    const int rc = ::ensure_dyn_lds((const void*)gemm16_nt256_kernel<HT, EPI>, "gemm16_nt256_kernel"); if (rc) return rc;
    hipLaunchKernelGGL((gemm16_nt256_kernel<HT, EPI>), dim3((Nc + 127) / 128, (M + 255) / 256, nsplit > 1 ? nsplit : 1), dim3(768), lds, s, ra, rb, epi, K, ksplit, dbg);
    return 0;
}
template <int HT, int BKH>
static inline int launch16_tn(const TN16Job& j, const StoreC& epi, int M, int Nc, int nsplit, hipStream_t s)
{
    int ksplit = j.K; if (nsplit > 1) ksplit = st_round_up((j.K + nsplit - 1) / nsplit, BKH);
    TN16Job jj = j; jj.nsplit = nsplit > 1 ? nsplit : 1;
    constexpr size_t lds = (size_t)4 * BKH * (128 + 32) * sizeof(h16_t);
    if (lds > 65536) { const int rc = ::ensure_dyn_lds((const void*)gemm16_tn_kernel<HT, BKH>, "gemm16_tn_kernel"); if (rc) return rc; }
    hipLaunchKernelGGL((gemm16_tn_kernel<HT, BKH>), dim3((Nc + 127) / 128, (M + 127) / 128, nsplit > 1 ? nsplit : 1), dim3(256), lds, s, jj, epi, ksplit);
    return 0;
}

}  // namespace stg
