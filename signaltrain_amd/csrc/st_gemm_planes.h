// GEMMs on bfloat16 PLANES (gfx950): C[M,N] = sum_k A[M,k] B[N,k] with both operands K-contiguous ("NT x NT"), each operand either
//   * PRE-SPLIT: PL bfloat16 planes in HBM ([plane][same layout as the fp32 tensor]), written once per step by planes_kernel -- the
//     k-loop then only moves 16-byte pieces global -> register -> LDS, no conversion, no arithmetic; or
//   * fp32 in HBM, split as it is staged (an activation that has a single consumer): 2 conversions + 4 subtractions per plane and
//     float4, placed in the shadow of the MFMAs.
// PL = 3: x = x1 + x2 + x3 (bfloat16 each, both remainders exact), the product from the six partial products of order >= 2^-16,
//         fp32 accumulation: fp32-grade results from the bf16 matrix pipe (ST_PREC_F32X3, include/signaltrain_hip.h).
// PL = 1: operands rounded to bfloat16 (ST_PREC_BF16*): the "real bf16" form -- bases kept as bf16 copies, no conversion in the loop.
// Why planes: on gfx950 v_mfma_f32_*_f32 executes at the vector-ALU fp32 rate and overlaps no vector work (tools/ubench/*.hip:
// an fp32 MFMA wave and a VALU wave on one SIMD take the SUM of their times), while v_mfma_f32_32x32x16_bf16 runs on the matrix
// pipe beside the VALU: six of them cost 3/8 of the fp32 MFMA's cycles.
//
// Tile: (32 * WAVES_M) x 96, one 32 x 96 strip per wave (3 accumulators), k-tile 16 = one MFMA k-step.  LDS per buffer and plane:
// [rows][16 + 8] bfloat16 (48-byte pitch: the ds_read_b128 of 16 consecutive rows hits 16 distinct 16-byte slots); two buffers.
// Software pipeline per k-tile, ONE basic block:  fragment reads of tile t | the MFMAs of tile t, term-major (consecutive MFMAs write
// different accumulators), each followed by one staging step of tile t+1 | global loads of tile t+2 | barrier.
#pragma once
#include "st_gemm.h"

namespace stg {

// ---------------------------------------------------------------- pre-split operand (the bases: written once per step by wplanes_kernel)
// K-CHUNK-MAJOR planes:  element (row, k) of plane p lives at  base[((k / 16) * rows + row) * (16 * PL) + p * 16 + k % 16]  -- the
// 16 * PL values of one (k-chunk, row) are contiguous (96 bytes for PL = 3) and so are consecutive rows: a workgroup's operand tile of
// one k-step is ONE contiguous block, fetched as consecutive 16-byte pieces by consecutive lanes.  (Row-major planes would hand each
// lane pair a 32-byte piece of a different cache line per k-step: the texture path serves a line per cycle, and the planes of 128 rows
// do not stay in L1 between the four k-steps that share a line -- measured 2x slower than the fp32 kernel.)
// Rows past `rows` (tile overhang) are clamped: their outputs are masked by the epilogue.
struct ChunkP {
    static constexpr bool kPre = true;
    const unsigned short* base; int rows;
};
template <class L, class = void> struct is_pre { static constexpr bool value = false; };
template <class L> struct is_pre<L, decltype((void)L::kPre)> { static constexpr bool value = L::kPre; };

// ---------------------------------------------------------------- per-operand staging state
// fp32 operand, split in the kernel: items = float4 along k (4 per row and k-tile).
template <class L, int ROWS, int NT_, int PL, bool PRE = is_pre<L>::value>
struct Stager {
    static_assert(!L::kTN, "the planes kernel takes K-contiguous operands");
    static constexpr int N = ROWS * 4, IT = (N + NT_ - 1) / NT_, NS = IT * PL;
    int row[IT], k4[IT]; bool valid[IT]; RowState st[IT];
    float4 reg[2][IT]; bool ok[2][IT];                  // two prefetch sets: tile t+2 is requested while tile t+1 is being staged
    float sr[IT][4]; unsigned short* dst[IT]; int psz[IT];
    __device__ __forceinline__ void init(const L& l, const int blk0, const int tid) {
#pragma unroll
        for (int p = 0; p < IT; ++p) {
            const int idx = tid + NT_ * p; valid[p] = idx < N; const int id = valid[p] ? idx : 0;
            row[p] = id >> 2; k4[p] = (id & 3) * 4; st[p] = l.row_state(blk0 + row[p]);
        }
    }
    __device__ __forceinline__ void gload(const L& l, const int kt, const int set) {
#pragma unroll
        for (int p = 0; p < IT; ++p) {
            const int k = kt + k4[p];
            if constexpr (L::kOff) {
                const bool o = !L::kCheck || (k >= st[p].lo && k < st[p].hi);
                ok[set][p] = o; reg[set][p] = ldg128(l.dummy(), o ? st[p].o + (unsigned)k : 0u);
            } else {
                const Src s = l.src(st[p], k);
                if constexpr (L::kCheck) { ok[set][p] = s.ok; reg[set][p] = *reinterpret_cast<const float4*>(s.ok ? s.p : l.dummy()); }
                else { ok[set][p] = true; reg[set][p] = *reinterpret_cast<const float4*>(s.p); }
            }
        }
    }
    __device__ __forceinline__ void begin(const L& l, unsigned short* tile, const int plane_sz, unsigned short* dump, const int LD, const int set) {
#pragma unroll
        for (int p = 0; p < IT; ++p) {
            const float4 v = ok[set][p] ? l.post(reg[set][p]) : make_float4(0.f, 0.f, 0.f, 0.f);
            sr[p][0] = v.x; sr[p][1] = v.y; sr[p][2] = v.z; sr[p][3] = v.w;
            const bool skip = N % NT_ != 0 && !valid[p];
            dst[p] = skip ? dump : tile + row[p] * LD + k4[p]; psz[p] = skip ? 0 : plane_sz;
        }
    }
    __device__ __forceinline__ void stage(const int s, const int) {         // s = item * PL + plane
        const int u = s / PL, pp = s % PL;
        const uint2 pl = make_uint2(st_cvt_pk_bf16(sr[u][0], sr[u][1]), st_cvt_pk_bf16(sr[u][2], sr[u][3]));
        *reinterpret_cast<uint2*>(dst[u] + pp * psz[u]) = pl;
        if (pp + 1 < PL) { sr[u][0] -= st_bf16_lo(pl.x); sr[u][1] -= st_bf16_hi(pl.x); sr[u][2] -= st_bf16_lo(pl.y); sr[u][3] -= st_bf16_hi(pl.y); }
    }
};
// pre-split operand: items = 16-byte pieces; piece q of a row = (plane q / 2, k-half q % 2).
template <class L, int ROWS, int NT_, int PL>
struct Stager<L, ROWS, NT_, PL, true> {
    static constexpr int PPR = 2 * PL;                       // pieces per row and k-chunk
    static constexpr int N = ROWS * PPR, IT = (N + NT_ - 1) / NT_, NS = IT;
    bool valid[IT]; const unsigned short* src[IT]; unsigned cstride;      // elements between k-chunks
    uint4 reg[2][IT];
    unsigned short* dst[IT]; int loff[IT];
    __device__ __forceinline__ void init(const L& l, const int blk0, const int tid) {
        cstride = (unsigned)l.rows * (16 * PL);
#pragma unroll
        for (int p = 0; p < IT; ++p) {
            const int idx = tid + NT_ * p; valid[p] = idx < N; const int id = valid[p] ? idx : 0;
            const int r = id / PPR, q = id - r * PPR;
            const int gr = blk0 + r < l.rows ? blk0 + r : l.rows - 1;
            src[p] = l.base + ((size_t)gr * PPR + q) * 8;
            loff[p] = (q >> 1) * (ROWS * 24 + 16) + r * 24 + 8 * (q & 1);      // LDS offset in the buffer: plane q / 2 (ROWS x 24 + 16 elements each), row r, k-half q % 2
        }
    }
    __device__ __forceinline__ void gload(const L&, const int kt, const int set) {
#pragma unroll
        for (int p = 0; p < IT; ++p) reg[set][p] = *reinterpret_cast<const uint4*>(src[p] + (size_t)(kt >> 4) * cstride);
    }
    __device__ __forceinline__ void begin(const L&, unsigned short* tile, const int plane_sz, unsigned short* dump, const int LD, const int) {
#pragma unroll
        for (int p = 0; p < IT; ++p) {
            const bool skip = N % NT_ != 0 && !valid[p];
            dst[p] = skip ? dump : tile + loff[p];
        }
    }
    __device__ __forceinline__ void stage(const int s, const int set) { *reinterpret_cast<uint4*>(dst[s]) = reg[set][s]; }
};

// timing-only ablation (tools: build with -DST_PL_ABLATE=bits; results INVALID): 1 no global loads in the loop | 2 no barrier |
// 4 no staging steps | 8 fragments read once | 16 no MFMAs
#ifndef ST_PL_ABLATE
#define ST_PL_ABLATE 0
#endif
template <int WAVES_M, int PL, int MI, class AL, class BL, class EPI>
__global__ void __launch_bounds__(WAVES_M * 64)
gemm_planes_kernel(const AL al, const BL bl, const EPI epi, const int K, const int ksplit)
{
    static_assert(PL == 1 || PL == 3, "one bfloat16 plane (rounded operands) or the three-plane split");
    constexpr int BKP = 16, LD = BKP + 8;
    constexpr int BM = 32 * MI * WAVES_M, NT = 64 * WAVES_M;      // a wave owns MI x NJ accumulator blocks: (32 MI) x 96
    constexpr int A_SZ = BM * LD + 16, B_SZ = BN * LD + 16;       // one plane of one buffer (+32 bytes: the three planes of a pre-split operand are written by ONE
                                                                  // ds_write_b128 -- 8-lane groups hold pieces of all three -- and must not share banks)
    constexpr int NTERM = PL == 3 ? 6 : 1, NM = NTERM * NJ * MI;
    extern __shared__ __attribute__((aligned(16))) unsigned short plds[];        // [2][PL][A_SZ] | [2][PL][B_SZ] | 16-byte dump slot per thread
    unsigned short* const As = plds;
    unsigned short* const Bs = plds + 2 * PL * A_SZ;
    unsigned short* const dump = plds + 2 * PL * (A_SZ + B_SZ) + 8 * threadIdx.x;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int tbx, tby, tbz; xcd_tile(tbx, tby, tbz);
    const int m_blk = tby * BM, n_blk = tbx * BN;
    const int k_begin = tbz * ksplit;
    const int k_end = (k_begin + ksplit < K) ? k_begin + ksplit : K;

    Stager<AL, BM, NT, PL> sa; Stager<BL, BN, NT, PL> sb;
    sa.init(al, m_blk, tid); sb.init(bl, n_blk, tid);
    constexpr int NSA = Stager<AL, BM, NT, PL>::NS, NSB = Stager<BL, BN, NT, PL>::NS, NS = NSA + NSB;

    f32x16 acc[MI][NJ];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[mi][j][i] = 0.f;

    if (k_begin < k_end) {
        sa.gload(al, k_begin, 0); sb.gload(bl, k_begin, 0);
        sa.begin(al, As, A_SZ, dump, LD, 0); sb.begin(bl, Bs, B_SZ, dump, LD, 0);
#pragma unroll
        for (int s = 0; s < NSA; ++s) sa.stage(s, 0);
#pragma unroll
        for (int s = 0; s < NSB; ++s) sb.stage(s, 0);
        { const int k1 = k_begin + BKP < k_end ? k_begin + BKP : k_begin; sa.gload(al, k1, 1); sb.gload(bl, k1, 1); }     // tile 1 -> set 1
        __syncthreads();
        const int h = lane >> 5, l31 = lane & 31;
        const int a_off = (wave * 32 * MI + l31) * LD + 8 * h;
        const int b_off = l31 * LD + 8 * h;
        st_bf16x8 a[MI][PL], b[NJ][PL];
        // one k-tile: tile `kt` sits in LDS buffer CUR; the registers of set 1 - CUR hold tile kt + 16 (staged now), set CUR receives tile kt + 32
        auto ktile = [&](const int kt, const int CUR) {
            const unsigned short* as = As + CUR * PL * A_SZ + a_off;
            const unsigned short* bs = Bs + CUR * PL * B_SZ + b_off;
#if ST_PL_ABLATE & 8
            if (kt == k_begin)
#endif
            {
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int p = 0; p < PL; ++p) a[mi][p] = *reinterpret_cast<const st_bf16x8*>(as + p * A_SZ + 32 * mi * LD);
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int p = 0; p < PL; ++p) b[j][p] = *reinterpret_cast<const st_bf16x8*>(bs + p * B_SZ + 32 * j * LD);
            }
            __builtin_amdgcn_sched_barrier(0);
            constexpr int TA[6] = {0, 2, 1, 0, 1, 0}, TB[6] = {2, 0, 1, 1, 0, 0};      // smallest partial products first
#pragma unroll
            for (int m = 0; m < NM; ++m) {
                const int t = m / (NJ * MI), j = (m / MI) % NJ, mi = m % MI;
#if !(ST_PL_ABLATE & 16)
                acc[mi][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi][PL == 3 ? TA[t] : 0], b[j][PL == 3 ? TB[t] : 0], acc[mi][j], 0, 0, 0);
#else
                acc[mi][j][0] += __builtin_bit_cast(float, (int)a[mi][0][0]) + __builtin_bit_cast(float, (int)b[j][0][0]);
#endif
                if (m == 0) {
                    // tile kt + 16 leaves its registers (past the end: stale data into a buffer that is not read again) ...
                    sa.begin(al, As + (CUR ^ 1) * PL * A_SZ, A_SZ, dump, LD, CUR ^ 1); sb.begin(bl, Bs + (CUR ^ 1) * PL * B_SZ, B_SZ, dump, LD, CUR ^ 1);
#if !(ST_PL_ABLATE & 1)
                    // ... and tile kt + 32 is requested into the other set: a whole iteration to arrive
                    const int k2 = kt + 2 * BKP;
                    const int kl = k2 < k_end ? k2 : kt;
                    sa.gload(al, kl, CUR); sb.gload(bl, kl, CUR);
#endif
                }
                // one staging step behind each MFMA: step s behind MFMA s * NM / NS (several per MFMA when there are more steps)
#if !(ST_PL_ABLATE & 4)
#pragma unroll
                for (int s = 0; s < NS; ++s)
                    if ((s * NM) / NS == m) { if (s < NSA) sa.stage(s, CUR ^ 1); else sb.stage(s - NSA, CUR ^ 1); }
#endif
                __builtin_amdgcn_sched_barrier(0);
            }
#if !(ST_PL_ABLATE & 2)
            __syncthreads();
#endif
        };
        for (int kt = k_begin; kt < k_end; kt += 2 * BKP) {
            ktile(kt, 0);
            if (kt + BKP < k_end) ktile(kt + BKP, 1);
        }
    }
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) epi(m_blk + (wave * MI + mi) * 32, n_blk, acc[mi]);
}

template <int WAVES_M, int PL, int MI = 1, class AL, class BL, class EPI>
static inline int launch_planes(const AL& al, const BL& bl, const EPI& epi, int M, int Nc, int K, int nsplit, hipStream_t s)
{
    constexpr int BM = 32 * MI * WAVES_M;
    constexpr size_t lds = ((size_t)2 * PL * ((BM + BN) * 24 + 32) + 8 * 64 * WAVES_M) * sizeof(unsigned short);
    int ksplit = K;
    if (nsplit > 1) ksplit = st_round_up((K + nsplit - 1) / nsplit, 16);
    dim3 grid((Nc + BN - 1) / BN, (M + BM - 1) / BM, nsplit > 1 ? nsplit : 1);
    if (lds > 65536) { const int rc = ::ensure_dyn_lds((const void*)gemm_planes_kernel<WAVES_M, PL, MI, AL, BL, EPI>, "gemm_planes_kernel"); if (rc) return rc; }
    hipLaunchKernelGGL((gemm_planes_kernel<WAVES_M, PL, MI, AL, BL, EPI>), grid, dim3(WAVES_M * 64), lds, s, al, bl, epi, K, ksplit);
    return 0;
}

// fp32 bases -> k-chunk-major bfloat16 planes (one launch for all of them).  A job reads a row-major [rows][K] matrix; with in2 the
// output rows alternate between two sources (row j = in[j >> 1] for even j, in2[j >> 1] for odd j: the (re, im) column order of the
// analysis GEMM).  One thread per (row, four consecutive k).
struct WPlanesJob { const float* in; const float* in2; unsigned short* out; int rows, K, ld; };
struct WPlanesArgs { WPlanesJob job[4]; unsigned blk0[5]; int njobs; };
template <int PL>
__global__ void __launch_bounds__(256)
wplanes_kernel(const WPlanesArgs a)
{
    int j = 0;
#pragma unroll
    for (int q = 1; q < 4; ++q) if (q < a.njobs && blockIdx.x >= a.blk0[q]) j = q;
    const WPlanesJob jb = a.job[j];
    const int kq = jb.K >> 2;
    const size_t i = (size_t)(blockIdx.x - a.blk0[j]) * 256 + threadIdx.x;
    if (i >= (size_t)jb.rows * kq) return;
    const int row = (int)(i / kq), k = ((int)(i - (size_t)row * kq)) << 2;
    const float* src = jb.in2 ? ((row & 1) ? jb.in2 : jb.in) + (size_t)(row >> 1) * jb.ld : jb.in + (size_t)row * jb.ld;
    const float4 v = *reinterpret_cast<const float4*>(src + k);
    unsigned short* o = jb.out + ((size_t)(k >> 4) * jb.rows + row) * (16 * PL) + (k & 15);
    if constexpr (PL == 3) {
        uint2 pl[3]; st_split3(v.x, v.y, v.z, v.w, pl);
#pragma unroll
        for (int p = 0; p < 3; ++p) *reinterpret_cast<uint2*>(o + 16 * p) = pl[p];
    } else *reinterpret_cast<uint2*>(o) = make_uint2(st_cvt_pk_bf16(v.x, v.y), st_cvt_pk_bf16(v.z, v.w));
}

}  // namespace stg
