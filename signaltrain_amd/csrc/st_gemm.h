// st_gemm.h -- fp32 MFMA GEMM family for the learned-basis STFT front/back end (gfx950).
//
// One kernel template, C[M x Nc] = A[M x K] * B[K x Nc], specialised by operand *loaders*
// (which gather the overlapped audio frames straight from the waveform -- no im2col copy) and
// by an *epilogue* (polar transform, plain store, split-K slab store).
//
// Tiling (CDNA4, 64-wide waves): workgroup = WAVES_M waves stacked along M; each wave owns a
// 32 x 96 output strip = 3 accumulators of v_mfma_f32_32x32x2_f32 (exact fp32, 64 cyc/issue ==
// dependent latency, so 3 independent accumulators keep the pipe full).  BN = 96 because the
// spectral width is 2*513 = 1026 = 10.7 * 96 (11 tiles, 2.8 % pad) whereas 128-wide tiles
// would waste 12.5 % on the Nyquist bin.  BK = 32.  Operand tiles are staged k-major in LDS
// (As[k][m], Bs[k][n]) so that an MFMA operand fetch is one conflict-free ds_read_b32 per lane
// (lanes 0-31 / 32-63 read two consecutive k rows).  fp32 MFMA is so slow relative to LDS/L2
// (64 cycles per 32x32x2) that register-staged double buffering with one barrier per k-tile is
// sufficient: 64 MFMA-issue cycles x 48 per wave per barrier.
#pragma once
#include "st_common.h"

namespace stg {

constexpr int BK = 32;
constexpr int BN = 96;
constexpr int NJ = BN / 32;   // MFMA column tiles per wave

// ------------------------------------------------------------------------------ loaders
// "NT-type" operand: memory is contiguous along the reduction index k.
//   RowState row_state(row)        -- per output row/col, computed once per thread
//   float4   load(st, k)           -- 4 consecutive k (k % 4 == 0); zero outside [lo,hi)
// "TN-type" operand: memory is contiguous along the non-reduction index.
//   float4   load(k, col)          -- 4 consecutive cols at reduction index k

struct RowState {
    const float* p;   // element k of this row lives at p[k] (only dereferenced for lo <= k < hi)
    int lo, hi;
};

// Rows = overlapped frames of a batch of signals: row r = (b, t), element k = sig[b, H*t - pad + k].
// Replaces Conv1d's implicit im2col (cls_fe_dft.py:55-56) and the framed operand of every backward GEMM.
struct FramedNT {
    static constexpr bool kTN = false;
    const float* sig; int Ls, T, H, pad, R, Kw; float scale;
    __device__ RowState row_state(int r) const {
        RowState s; s.p = sig; s.lo = 0; s.hi = 0;
        if (r < R) {
            const int b = r / T, t = r - b * T;
            const int start = H * t - pad;                 // frame start in the unpadded signal (bit-exact contract)
            s.p = sig + (size_t)b * Ls + start;
            s.lo = start < 0 ? -start : 0;
            s.hi = (Ls - start) < Kw ? (Ls - start) : Kw;
            if (s.hi < s.lo) s.hi = s.lo;
        }
        return s;
    }
    __device__ float4 load(const RowState& s, int k) const {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (k >= s.lo && k < s.hi) {
            v = *reinterpret_cast<const float4*>(s.p + k);
            v.x *= scale; v.y *= scale; v.z *= scale; v.w *= scale;
        }
        return v;
    }
};

// Same frames, TN-type (reduction index = frame row r, contiguous along the tap n).
struct FramedTN {
    static constexpr bool kTN = true;
    const float* sig; int Ls, T, H, pad, R, Kw; float scale;
    __device__ float4 load(int r, int n) const {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r < R && n < Kw) {
            const int b = r / T, t = r - b * T;
            const int pos = H * t - pad + n;
            if (pos >= 0 && pos < Ls) {
                v = *reinterpret_cast<const float4*>(sig + (size_t)b * Ls + pos);
                v.x *= scale; v.y *= scale; v.z *= scale; v.w *= scale;
            }
        }
        return v;
    }
};

// Dense row-major [rows][ld] matrix, reduction along the row (NT-type).
struct PlainNT {
    static constexpr bool kTN = false;
    const float* base; int rows, ld, K;
    __device__ RowState row_state(int r) const {
        RowState s; s.p = base + (size_t)(r < rows ? r : 0) * ld; s.lo = 0; s.hi = (r < rows) ? K : 0;
        return s;
    }
    __device__ float4 load(const RowState& s, int k) const {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (k < s.hi) v = *reinterpret_cast<const float4*>(s.p + k);
        return v;
    }
};

// Dense row-major [K][ld] matrix, reduction along the rows (TN-type).
struct PlainTN {
    static constexpr bool kTN = true;
    const float* base; int K, ld, cols;
    __device__ float4 load(int k, int c) const {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (k < K && c < cols) v = *reinterpret_cast<const float4*>(base + (size_t)k * ld + c);
        return v;
    }
};

// Analysis bases as the B operand: GEMM column j -> (bin = j>>1, re/im = j&1); only the F used rows
// of the [N,N] parameters are ever touched (cls_fe_dft.py:55-56 computes all N then slices).
struct AnalysisW {
    static constexpr bool kTN = false;
    const float* Wr; const float* Wi; int F, N;
    __device__ RowState row_state(int j) const {
        const int bin = j >> 1;
        RowState s; s.lo = 0;
        const bool ok = bin < F;
        s.p = ((j & 1) ? Wi : Wr) + (size_t)(ok ? bin : 0) * N;
        s.hi = ok ? N : 0;
        return s;
    }
    __device__ float4 load(const RowState& s, int k) const {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (k < s.hi) v = *reinterpret_cast<const float4*>(s.p + k);
        return v;
    }
};

// ------------------------------------------------------------------------------ epilogues
// D layout of v_mfma_f32_32x32x2_f32: reg i of lane l holds
//   row = (i&3) + 8*(i>>2) + 4*(l>>5),  col = l&31.
__device__ __forceinline__ int d_row(int i, int lane) { return (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5); }

struct StoreC {       // out[(z*M + row)*ld + col]; z = blockIdx.z (split-K slab) when slabbed
    float* out; int M, Nc, ld; size_t slab;
    __device__ void operator()(int m0, int n0, const f32x16 (&acc)[NJ]) const {
        const int lane = threadIdx.x & 63;
        float* o = out + (size_t)blockIdx.z * slab;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int col = n0 + 32 * j + (lane & 31);
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int row = m0 + d_row(i, lane);
                if (row < M && col < Nc) o[(size_t)row * ld + col] = acc[j][i];
            }
        }
    }
};

// Analysis epilogue: nn_proc.py:309-310 fused.  Columns are interleaved (re,im) pairs of one bin in
// adjacent lanes; a lane-pair exchange gives both, even lanes store (re, mag), odd lanes (im, phs).
struct PolarStore {
    float* re; float* im; float* mag; float* phs; int R, F;
    __device__ void operator()(int m0, int n0, const f32x16 (&acc)[NJ]) const {
        const int lane = threadIdx.x & 63;
        const bool odd = lane & 1;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int bin = (n0 + 32 * j + (lane & 31)) >> 1;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float v = acc[j][i];
                const float o = __shfl_xor(v, 1);
                const float vr = odd ? o : v, vi = odd ? v : o;
                const int row = m0 + d_row(i, lane);
                if (row < R && bin < F) {
                    const size_t idx = (size_t)row * F + bin;
                    if (!odd) {
                        if (re) re[idx] = vr;
                        if (mag) mag[idx] = sqrtf(vr * vr + vi * vi);
                    } else {
                        if (im) im[idx] = vi;
                        if (phs) phs[idx] = atan2f(vi, vr + 1e-7f);
                    }
                }
            }
        }
    }
};

// ------------------------------------------------------------------------------ kernel
template <int WAVES_M, class AL, class BL, class EPI>
__global__ void __launch_bounds__(WAVES_M * 64)
gemm_kernel(const AL al, const BL bl, const EPI epi, const int K, const int ksplit)
{
    constexpr int BM = 32 * WAVES_M, NT = 64 * WAVES_M;
    constexpr int LDA = AL::kTN ? BM + 4 : BM + 2;     // TN: 16-B aligned rows for ds_write_b128; NT: 4*LD = 8 (mod 32) -> conflict-free transposing ds_write_b32
    constexpr int LDB = BL::kTN ? BN + 4 : BN + 2;
    constexpr int A_IT = BM * (BK / 4) / NT;           // float4 items per thread per k-tile (= 4)
    constexpr int B_IT = BN * (BK / 4) / NT;           // 6 / 4 / 3 for WAVES_M = 2 / 3 / 4
    static_assert(BM * (BK / 4) % NT == 0 && BN * (BK / 4) % NT == 0, "tile/threads mismatch");
    __shared__ __attribute__((aligned(16))) float As[2 * BK * LDA];
    __shared__ __attribute__((aligned(16))) float Bs[2 * BK * LDB];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m_blk = blockIdx.y * BM, n_blk = blockIdx.x * BN;
    const int k_begin = blockIdx.z * ksplit;
    const int k_end = (k_begin + ksplit < K) ? k_begin + ksplit : K;

    // per-thread item coordinates (fixed across k-tiles)
    int a_i[A_IT], a_k[A_IT], b_i[B_IT], b_k[B_IT];
    RowState a_st[AL::kTN ? 1 : A_IT], b_st[BL::kTN ? 1 : B_IT];
#pragma unroll
    for (int p = 0; p < A_IT; ++p) {
        const int idx = tid + NT * p;
        if constexpr (AL::kTN) { a_i[p] = (idx % (BM / 4)) * 4; a_k[p] = idx / (BM / 4); }
        else { const int t2 = idx >> 2; a_i[p] = t2 % BM; a_k[p] = 16 * (t2 / BM) + 4 * (idx & 3); a_st[p] = al.row_state(m_blk + a_i[p]); }
    }
#pragma unroll
    for (int p = 0; p < B_IT; ++p) {
        const int idx = tid + NT * p;
        if constexpr (BL::kTN) { b_i[p] = (idx % (BN / 4)) * 4; b_k[p] = idx / (BN / 4); }
        else { const int t2 = idx >> 2; b_i[p] = t2 % BN; b_k[p] = 16 * (t2 / BN) + 4 * (idx & 3); b_st[p] = bl.row_state(n_blk + b_i[p]); }
    }

    float4 ra[A_IT], rb[B_IT];
    auto gload = [&](int kt) {
#pragma unroll
        for (int p = 0; p < A_IT; ++p) {
            if constexpr (AL::kTN) ra[p] = al.load(kt + a_k[p], m_blk + a_i[p]);
            else ra[p] = al.load(a_st[p], kt + a_k[p]);
        }
#pragma unroll
        for (int p = 0; p < B_IT; ++p) {
            if constexpr (BL::kTN) rb[p] = bl.load(kt + b_k[p], n_blk + b_i[p]);
            else rb[p] = bl.load(b_st[p], kt + b_k[p]);
        }
    };
    auto lstore = [&](int buf) {
        float* as = As + buf * BK * LDA;
        float* bs = Bs + buf * BK * LDB;
#pragma unroll
        for (int p = 0; p < A_IT; ++p) {
            if constexpr (AL::kTN) *reinterpret_cast<float4*>(as + a_k[p] * LDA + a_i[p]) = ra[p];
            else {
                float* q = as + a_k[p] * LDA + a_i[p];
                q[0] = ra[p].x; q[LDA] = ra[p].y; q[2 * LDA] = ra[p].z; q[3 * LDA] = ra[p].w;
            }
        }
#pragma unroll
        for (int p = 0; p < B_IT; ++p) {
            if constexpr (BL::kTN) *reinterpret_cast<float4*>(bs + b_k[p] * LDB + b_i[p]) = rb[p];
            else {
                float* q = bs + b_k[p] * LDB + b_i[p];
                q[0] = rb[p].x; q[LDB] = rb[p].y; q[2 * LDB] = rb[p].z; q[3 * LDB] = rb[p].w;
            }
        }
    };

    f32x16 acc[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;

    if (k_begin < k_end) {
        gload(k_begin);
        lstore(0);
        __syncthreads();
        int cur = 0;
        const int a_off = (lane >> 5) * LDA + wave * 32 + (lane & 31);
        const int b_off = (lane >> 5) * LDB + (lane & 31);
        for (int kt = k_begin; kt < k_end; kt += BK) {
            const bool more = kt + BK < k_end;
            if (more) gload(kt + BK);
            const float* as = As + cur * BK * LDA + a_off;
            const float* bs = Bs + cur * BK * LDB + b_off;
#pragma unroll
            for (int kk = 0; kk < BK / 2; ++kk) {
                const float a = as[2 * kk * LDA];
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    const float b = bs[2 * kk * LDB + 32 * j];
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j], 0, 0, 0);
                }
            }
            if (more) lstore(cur ^ 1);
            __syncthreads();
            cur ^= 1;
        }
    }
    epi(m_blk + wave * 32, n_blk, acc);
}

template <int WAVES_M, class AL, class BL, class EPI>
static inline void launch(const AL& al, const BL& bl, const EPI& epi, int M, int Nc, int K, int nsplit,
                          hipStream_t s)
{
    constexpr int BM = 32 * WAVES_M;
    int ksplit = K;
    if (nsplit > 1) ksplit = st_round_up((K + nsplit - 1) / nsplit, BK);
    const int nz = (K + ksplit - 1) / ksplit;
    dim3 grid((Nc + BN - 1) / BN, (M + BM - 1) / BM, nz);
    hipLaunchKernelGGL((gemm_kernel<WAVES_M, AL, BL, EPI>), grid, dim3(WAVES_M * 64), 0, s, al, bl, epi, K, ksplit);
}

}  // namespace stg
