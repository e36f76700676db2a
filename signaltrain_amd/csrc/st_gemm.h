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
// would waste 12.5 % on the Nyquist bin.  BK = 32.
//
// LDS staging, two layouts chosen by how the operand lies in memory:
//  * "NT" operand (memory contiguous along k): row-major tile Xs[row][36]; a global float4 becomes
//    ONE ds_write_b128 and the wave's MFMA fragments for the whole k-tile are 4 ds_read_b128 per
//    32-row block (lane (m, h) takes k = 16h..16h+15: the reduction order inside a tile is free,
//    A and B only have to agree).  Pitch 36 floats keeps both conflict-free.
//  * "TN" operand (memory contiguous along m/n): k-major tile Xs[k][rows+4], ds_write_b128 in,
//    one ds_read_b32 per MFMA operand out.
// Global loads are branch-free (clamped pointer + select) so all 7-10 loads of the next k-tile are
// in flight while the current tile's 48 MFMAs per wave run; scaling (x/2, nn_proc.py:307) is applied
// at the LDS store so no load result is touched early.
//
// Zero-padding frames (frames that lie completely in the Conv1d padding / in the cropped margins of
// the transposed conv) are never computed: the framed loaders enumerate only the live frames of each
// window (RowMap) -- 23 of 25 analysis frames, 7 of 9 synthesis frames at the default geometry.
#pragma once
#include "st_common.h"

// > 64 KB of dynamic LDS needs a per-(device, kernel) attribute: st_api.hip's checked, cached helper
static int ensure_dyn_lds(const void* fn, const char* name);
namespace stg {

constexpr int BK = 32;
constexpr int BN = 96;
constexpr int NJ = BN / 32;   // MFMA column tiles per wave
constexpr int PK = BK + 4;    // row pitch (floats) of an NT tile in LDS

// Compact row index r' (live frames only) -> (window b, frame t); full row = b*T + t.
// r / Tv by multiply-high with magic = floor(2^32 / Tv) + 1 (exact for r < 2^32 / Tv; the host keeps row counts below
// 2^24): the loaders call this once per global load inside the k-loop, where a run-time integer division costs ~20 VALU
// instructions -- it was most of the 8 VALU-per-MFMA of the weight-gradient GEMMs.  magic = 0 means Tv = 1.
struct RowMap {
    int Tv, t_lo, T; unsigned magic;
    int fm;     // round 5: 0 = window-major compact rows (r = b * Tv + (t - t_lo)); fm = B > 0: FRAME-major (r = (t - t_lo) * B + b, magic = that of B) --
                // the rows of one frame are contiguous, so a 128-row tile has one frame index (or a short run of them) and with it ONE live tap range:
                // the partly padded / partly cropped frames' structural zeros (cls_fe_dft.py:28-31, :113) can be skipped per tile (st_gemm_tn.h)
    __host__ __device__ int rows(int B) const { return B * Tv; }
    __device__ void split(int r, int& b, int& t) const {
        if (fm) {
            const int q = magic ? (int)__umulhi((unsigned)r, magic) : r;
            b = r - (int)__umul24((unsigned)q, (unsigned)fm);
            t = t_lo + q;
            return;
        }
        b = magic ? (int)__umulhi((unsigned)r, magic) : r;
        t = t_lo + (r - (int)__umul24((unsigned)b, (unsigned)Tv));
    }
    __device__ int full(int r) const { int b, t; split(r, b, t); return (int)__umul24((unsigned)b, (unsigned)T) + t; }
};
static inline unsigned rowmap_magic(int Tv) { return Tv > 1 ? (unsigned)(((unsigned long long)1 << 32) / (unsigned)Tv) + 1u : 0u; }
// Frames t with a non-empty intersection [H t - pad, H t - pad + N) x [0, Ls)
static inline RowMap live_frames(int T, int H, int N, int pad, int Ls)
{
    int lo = 0, hi = T - 1;
    while (lo < T && H * lo - pad + N <= 0) ++lo;
    while (hi >= lo && H * hi - pad >= Ls) --hi;
    RowMap m; m.t_lo = lo; m.Tv = hi - lo + 1; m.T = T; m.fm = 0;
    if (m.Tv <= 0) { m.Tv = T; m.t_lo = 0; }
    m.magic = rowmap_magic(m.Tv);
    return m;
}
static inline RowMap all_frames(int T) { RowMap m; m.Tv = T; m.t_lo = 0; m.T = T; m.fm = 0; m.magic = rowmap_magic(T); return m; }
// the same live frames enumerated frame-major over B windows (B = 1: both orders coincide)
static inline RowMap frame_major(RowMap m, int B) { m.fm = B; m.magic = B > 1 ? rowmap_magic(B) : 0u; return m; }

// float4 at a 32-bit ELEMENT offset from a wave-uniform base: saddr + voffset addressing, no 64-bit arithmetic per load
// (the host keeps every operand below 2^30 elements).
__device__ __forceinline__ float4 ldg128(const float* __restrict__ base, const unsigned elem)
{
    return *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(base) + (size_t)(elem << 2));
}

struct Src { const float* p; bool ok; };

// A TN-type operand as data: element (k, col) lives at  base[b * S1 + t * S2 + col]  with (b, t) = RowMap::split(k < clampR ? k : 0),
// valid iff k < klim && col < collim.  Lets ONE code path address either operand of a TN x TN GEMM from per-lane parameters
// (gemm_kernel's unified micro-tile items) instead of evaluating both loaders' off() in every lane.
struct TNParams { const float* base; unsigned magic, S1, S2; int Tv, t_lo, clampR, klim, collim; };

struct RowState {
    const float* p;   // element k of this row lives at p[k] (only dereferenced for lo <= k < hi)
    unsigned o;       // the same as a 32-bit element offset from the loader's dummy() base (loaders with kOff; may wrap below zero for k < lo)
    int lo, hi;
};

// ------------------------------------------------------------------------------ loaders
// A loader either needs per-load validity (kCheck: the load is clamped to a valid address and the value replaced by
// zero) or not: operands whose out-of-range rows/columns only feed output elements that the epilogue masks are simply
// clamped (garbage in, masked out), which removes the compare/select VALU work from the k-loop.
//
// Rows = overlapped frames of a batch of signals: compact row r -> (b, t), element k = sig[b, H*t - pad + k].
// Replaces Conv1d's implicit im2col (cls_fe_dft.py:55-56) and the framed operand of every backward GEMM.
// PADDED = true: `sig` is the workspace copy [B][pad + Ls + pad] with zero margins and the input scale already
// applied (st_pad_scale / ola_loss_kernel write it once per step): every frame element exists, no validity logic.
template <bool PADDED>
struct FramedNT {
    static constexpr bool kTN = false;
    static constexpr bool kCheck = !PADDED;
    static constexpr bool kOff = true;
    const float* sig; int Ls, H, pad, R, Kw; float scale; RowMap map;
    __device__ const float* dummy() const { return sig; }
    __device__ RowState row_state(int r) const {
        RowState s; s.p = sig; s.o = 0; s.lo = 0; s.hi = 0;
        if (PADDED) {
            int b, t; map.split(r < R ? r : 0, b, t);         // rows >= R: clamped, their outputs are masked
            s.o = __umul24((unsigned)b, (unsigned)(Ls + 2 * pad)) + __umul24((unsigned)H, (unsigned)t);   // padded coordinate of frame start (bit-exact contract: H*t - pad + pad)
            s.p = sig + s.o;
            s.hi = Kw;
        } else if (r < R) {
            int b, t; map.split(r, b, t);
            const int start = H * t - pad;                    // frame start in the unpadded signal
            s.o = __umul24((unsigned)b, (unsigned)Ls) + (unsigned)start;
            s.p = sig + (size_t)b * Ls + start;
            s.lo = start < 0 ? -start : 0;
            s.hi = (Ls - start) < Kw ? (Ls - start) : Kw;
            if (s.hi < s.lo) s.hi = s.lo;
        }
        return s;
    }
    __device__ Src src(const RowState& s, int k) const { return Src{s.p + k, PADDED || (k >= s.lo && k < s.hi)}; }
    __device__ float4 post(float4 v) const {
        if (!PADDED) { v.x *= scale; v.y *= scale; v.z *= scale; v.w *= scale; }
        return v;
    }
};

// Same frames, TN-type (reduction index = compact frame row r, contiguous along the tap n).  With PADDED the rows past
// R are clamped: the other operand of a weight-gradient GEMM zeroes them.
template <bool PADDED>
struct FramedTN {
    static constexpr bool kTN = true;
    static constexpr bool kCheck = !PADDED;
    static constexpr bool kOff = true;
    const float* sig; int Ls, H, pad, R, Kw; float scale; RowMap map;
    __device__ const float* dummy() const { return sig; }
    __device__ Src src(int r, int n) const {
        if (PADDED) {
            int b, t; map.split(r < R ? r : 0, b, t);
            return Src{sig + (size_t)b * (Ls + 2 * pad) + H * t + n, true};
        }
        Src s{sig, false};
        if (r < R && n < Kw) {
            int b, t; map.split(r, b, t);
            const int pos = H * t - pad + n;
            s.ok = pos >= 0 && pos < Ls;
            s.p = sig + (size_t)b * Ls + pos;
        }
        return s;
    }
    // element offset from dummy() of (row r, tap n); ok = the load is valid (else the caller loads offset 0 and zeroes)
    __device__ unsigned off(int r, int n, bool& ok) const {
        int b, t;
        map.split(r < R ? r : 0, b, t);
        if (PADDED) {
            ok = true;
            return __umul24((unsigned)b, (unsigned)(Ls + 2 * pad)) + __umul24((unsigned)H, (unsigned)t) + (unsigned)n;
        }
        const int pos = H * t - pad + n;
        ok = r < R && n < Kw && pos >= 0 && pos < Ls;
        return ok ? __umul24((unsigned)b, (unsigned)Ls) + (unsigned)pos : 0u;
    }
    __device__ float4 post(float4 v) const {
        if (!PADDED) { v.x *= scale; v.y *= scale; v.z *= scale; v.w *= scale; }
        return v;
    }
    static constexpr bool kParams = PADDED;            // expressible as TNParams (the padded form only: no per-element validity)
    __device__ TNParams tn_params() const { return TNParams{sig, map.magic, (unsigned)(Ls + 2 * pad), (unsigned)H, map.Tv, map.t_lo, R, 0x7fffffff, 0x7fffffff}; }
};

// Dense row-major [rows][ld] matrix, reduction along the row (NT-type); compact rows via RowMap.  Requires K % BK == 0;
// rows past R are clamped (masked outputs).
struct PlainNT {
    static constexpr bool kTN = false;
    static constexpr bool kCheck = false;
    static constexpr bool kOff = true;
    const float* base; int R, ld, K; RowMap map;
    __device__ const float* dummy() const { return base; }
    __device__ RowState row_state(int r) const {
        RowState s; s.lo = 0; s.hi = K;
        s.o = __umul24((unsigned)map.full(r < R ? r : 0), (unsigned)ld);
        s.p = base + s.o;
        return s;
    }
    __device__ Src src(const RowState& s, int k) const { return Src{s.p + k, true}; }
    __device__ float4 post(float4 v) const { return v; }
};

// Dense row-major [K][ld] matrix, reduction along the (compact) rows (TN-type): rows past K must read as zero.
struct PlainTN {
    static constexpr bool kTN = true;
    static constexpr bool kCheck = true;
    static constexpr bool kOff = true;
    const float* base; int K, ld, cols; RowMap map;
    __device__ const float* dummy() const { return base; }
    __device__ Src src(int k, int c) const {
        const bool ok = k < K && c < cols;
        return Src{base + (size_t)(ok ? map.full(k) : 0) * ld + c, ok};
    }
    __device__ unsigned off(int k, int c, bool& ok) const {
        ok = k < K && c < cols;
        return ok ? __umul24((unsigned)map.full(k), (unsigned)ld) + (unsigned)c : 0u;
    }
    __device__ float4 post(float4 v) const { return v; }
    static constexpr bool kParams = true;
    __device__ TNParams tn_params() const { return TNParams{base, map.magic, (unsigned)(map.T * ld), (unsigned)ld, map.Tv, map.t_lo, K, K, cols}; }
};

// Analysis bases as the B operand: GEMM column j -> (bin = j>>1, re/im = j&1); only the F used rows
// of the [N,N] parameters are ever touched (cls_fe_dft.py:55-56 computes all N then slices).  Columns past 2F are
// clamped (the polar epilogue masks bin >= F).
struct AnalysisW {
    static constexpr bool kTN = false;
    static constexpr bool kCheck = false;
    static constexpr bool kOff = false;     // two independent base pointers: rows keep 64-bit addresses
    const float* Wr; const float* Wi; int F, N;
    __device__ const float* dummy() const { return Wr; }
    __device__ RowState row_state(int j) const {
        const int bin = j >> 1;
        RowState s; s.lo = 0; s.hi = N; s.o = 0;
        s.p = ((j & 1) ? Wi : Wr) + (size_t)(bin < F ? bin : 0) * N;
        return s;
    }
    __device__ Src src(const RowState& s, int k) const { return Src{s.p + k, true}; }
    __device__ float4 post(float4 v) const { return v; }
};

// XCD-aware workgroup -> tile mapping.  The hardware hands consecutive workgroups (x fastest, then y, then z) to the 8
// XCDs round-robin, and every XCD has its own L2: with the natural mapping the tiles that share an A row-block, a B
// column-block or a split-K slice are spread over all eight L2s and every operand is fetched up to 8 times.  Here
// workgroup L is given logical index  start(L % 8) + L / 8, so XCD j owns a CONTIGUOUS range of (k-slice, tile row,
// tile column) and its L2 sees one band of A / one slice of K.  Bijective for any grid size.
#ifndef ST_XCD_SWIZZLE
#define ST_XCD_SWIZZLE 1
#endif
// nz_: number of z-slices that take part (default: the whole grid; st_gemm_tn.h keeps its last slice out)
__device__ __forceinline__ void xcd_tile(int& bx, int& by, int& bz, const int nz_ = 0)
{
    bx = blockIdx.x; by = blockIdx.y; bz = blockIdx.z;
#if ST_XCD_SWIZZLE
    const int nx = gridDim.x, ny = gridDim.y, nz = nz_ > 0 ? nz_ : (int)gridDim.z;
    const int T = nx * ny * nz;
    if (T < 16) return;
    const int L = bx + nx * (by + ny * bz);
    const int j = L & 7, q = T >> 3, r = T & 7;
    const int Lp = j * q + (j < r ? j : r) + (L >> 3);
    bz = Lp / (nx * ny);
    const int rem = Lp - bz * (nx * ny);
    if (ny > nx) { bx = rem / ny; by = rem - bx * ny; }      // many tile rows: an XCD owns a band of tile COLUMNS (a slice of B stays in its L2)
    else { by = rem / nx; bx = rem - by * nx; }               // else a band of tile rows
#endif
}

// Two-dimensional variant for GEMMs whose BOTH operands are larger than what a band leaves in one L2 (the analysis forward: 46 x 11 tiles, A = 10.5 MB
// of waveform, B = 4.2 MB of bases; with bands of tile columns every XCD streamed all of A: 100 MB fetched for 12.6 MB of operands, flagged since round 1).
// The eight XCDs own a 4 x 2 grid of blocks of (tile rows, tile columns); an XCD's contiguous range of the block-major tile sequence lies in its block (+- a
// few tiles: block sizes are not all equal), whose A rows (1/4) and B panel (1/2) fit a 4 MB L2 together.  No split-K (bz = 0).
__device__ __forceinline__ void xcd_tile_2d(int& bx, int& by, int& bz)
{
    bx = blockIdx.x; by = blockIdx.y; bz = 0;
#if ST_XCD_SWIZZLE
    const int nx = gridDim.x, ny = gridDim.y;
    const int T = nx * ny;
    if (T < 64 || nx < 2 || ny < 4) { xcd_tile(bx, by, bz); return; }
    const int L = bx + nx * by;
    const int j = L & 7, q = T >> 3, r = T & 7;
    int p = j * q + (j < r ? j : r) + (L >> 3);               // position in the block-major sequence
#pragma unroll
    for (int blk = 0; blk < 8; ++blk) {
        const int rb = blk >> 1, cb = blk & 1;
        const int r0 = (rb * ny) >> 2, r1 = ((rb + 1) * ny) >> 2, c0 = (cb * nx) >> 1, c1 = ((cb + 1) * nx) >> 1;
        const int sz = (r1 - r0) * (c1 - c0);
        if (p < sz) { const int rr = p / (c1 - c0); by = r0 + rr; bx = c0 + (p - rr * (c1 - c0)); return; }
        p -= sz;
    }
#endif
}

// ------------------------------------------------------------------------------ epilogues
// D layout of v_mfma_f32_32x32x2_f32: reg i of lane l holds
//   row = (i&3) + 8*(i>>2) + 4*(l>>5),  col = l&31.
__device__ __forceinline__ int d_row(int i, int lane) { return (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5); }

struct StoreC {       // out[(z*slab) + full_row*ld + col]; z = blockIdx.z (split-K slab)
    float* out; int M, Nc, ld; size_t slab; RowMap map;
    __device__ StoreC slab_shifted(int dz) const { StoreC e = *this; e.out += (size_t)dz * slab; return e; }
    template <int NJX>                                            // NJX = 32-column blocks per wave (3: the 32 x 96 strips; 2: st_gemm16.h)
    __device__ void operator()(int m0, int n0, const f32x16 (&acc)[NJX]) const {
        const int lane = threadIdx.x & 63;
        int tbx, tby, tbz; xcd_tile(tbx, tby, tbz);               // the split-K slice this workgroup computed (see xcd_tile)
        float* o = out + (size_t)tbz * slab;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int row = m0 + d_row(i, lane);
            if (row < M) {
                float* orow = o + (size_t)map.full(row) * ld;
#pragma unroll
                for (int j = 0; j < NJX; ++j) {
                    const int col = n0 + 32 * j + (lane & 31);
                    if (col < Nc) orow[col] = acc[j][i];
                }
            }
        }
    }
};

struct BiasStore {    // out[row*ld + col] = acc + bias[col]   (Conv1d bias; bias may be null)
    float* out; const float* bias; int M, Nc, ld;
    __device__ void operator()(int m0, int n0, const f32x16 (&acc)[NJ]) const {
        const int lane = threadIdx.x & 63;
        float bv[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) { const int col = n0 + 32 * j + (lane & 31); bv[j] = (bias && col < Nc) ? bias[col] : 0.f; }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int row = m0 + d_row(i, lane);
            if (row < M) {
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    const int col = n0 + 32 * j + (lane & 31);
                    if (col < Nc) out[(size_t)row * ld + col] = acc[j][i] + bv[j];
                }
            }
        }
    }
};

// Analysis epilogue: nn_proc.py:309-310 fused.  Columns are interleaved (re,im) pairs of one bin in adjacent lanes.
// Registers are processed in pairs (rows r, r+1): one lane-pair exchange hands the even lane the full complex value
// of row r and the odd lane that of row r+1, so every lane does ONE sqrt and ONE atan2 per register pair.
struct PolarStore {
    float* re; float* im; float* mag; float* phs; int R, F; RowMap map;
    // round 4, wide geometries: mag / phs ALSO (or, with mag == phs == NULL, only) in the feature-major layout of the wide autoencoder path,
    // V[t][b * FP + bin] with RV = B * FP columns per frame row (st_ae_wide.h); NULL: off
    float* Vm = nullptr; float* Vp = nullptr; int FP = 0; unsigned RV = 0;
    template <int NJX>
    __device__ void operator()(int m0, int n0, const f32x16 (&acc)[NJX]) const {
        const int lane = threadIdx.x & 63;
        const bool odd = lane & 1;
#pragma unroll
        for (int i = 0; i < 16; i += 2) {
            const int row = m0 + d_row(i + (odd ? 1 : 0), lane);
            int wb = 0, wt = 0; map.split(row < R ? row : 0, wb, wt);
            const size_t rbase = (size_t)((int)__umul24((unsigned)wb, (unsigned)map.T) + wt) * F;
            const size_t vbase = (size_t)wt * RV + (size_t)wb * FP;
#pragma unroll
            for (int j = 0; j < NJX; ++j) {
                const int bin = (n0 + 32 * j + (lane & 31)) >> 1;
                const float v0 = acc[j][i], v1 = acc[j][i + 1];
                const float got = __shfl_xor(odd ? v0 : v1, 1);      // even lane receives im(row r), odd lane re(row r+1)
                const float vr = odd ? got : v0, vi = odd ? v1 : got;
                if (row < R && bin < F) {
                    const size_t idx = rbase + bin;
                    if (re) re[idx] = vr;
                    if (im) im[idx] = vi;
                    if (mag || phs || Vm) {
                        const float mg = __builtin_amdgcn_sqrtf(vr * vr + vi * vi), ph = st_atan2f(vi, vr + 1e-7f);      // v_sqrt_f32, 1 ulp
                        if (mag) mag[idx] = mg;
                        if (phs) phs[idx] = ph;
                        if (Vm) { Vm[vbase + bin] = mg; Vp[vbase + bin] = ph; }
                    }
                }
            }
        }
    }
};

template <class E> constexpr bool kPolarEpi = false;
template <> constexpr bool kPolarEpi<PolarStore> = true;
template <class L, class = void> struct has_tn_params { static constexpr bool value = false; };
template <class L> struct has_tn_params<L, decltype((void)L::kParams)> { static constexpr bool value = L::kParams; };

// ------------------------------------------------------------------------------ kernel
// The timing-only ablation switches are COMPILE-TIME (build with -DST_GEMM_ABLATE for tools/gemm_ablate*.py): as run-time
// tests they split the k-loop body into several basic blocks, and the compiler then shuttled all 48 accumulator registers
// between AGPRs and VGPRs on every iteration (96 extra instructions per 24 MFMAs).
#ifdef ST_GEMM_ABLATE
#define ST_DBG(bit_) (dbg & (bit_))
#else
#define ST_DBG(bit_) false
#endif
// BKT = k-tile depth (32: 64 KB LDS/WG, 2 WGs/CU; 16: 36 KB, 4 WGs/CU -- better for the small-M split-K GEMMs).
// dbg: timing-only ablation switches (bit0 skip loads/stores in the k-loop, bit1 skip barriers, bit2 skip MFMAs).
// MI = 32-row blocks per wave: MI = 1 is the 32 x 96 wave strip; WAVES_M = 1, MI = 3 lets ONE wave own a 96 x 96 tile
// (9 accumulators): same global traffic per MFMA as three waves sharing the tile, half the LDS fragment reads (B is read
// once instead of three times) and no cross-wave barrier stalls -- for the split-K weight-gradient GEMMs.
// XT: M/N-contiguous (TN-type) operands are transposed on the way into LDS -- a thread loads a 4(k) x 4(row) micro-tile as four
// float4 and writes four k-quads -- into a K-QUAD-MAJOR tile  X[k / 4][row][k % 4]  (16-byte slots, no padding), so that the MFMA
// fragments of such an operand are fetched exactly like those of a K-contiguous one: ds_read_b128 of 4 consecutive k, HK / 4 reads
// per 32 rows per k-tile instead of HK scalar ds_read_b32 (the k-major staging costs the weight-gradient GEMMs and the synthesis
// frames GEMM ~17 points of MFMA utilisation against the NT x NT GEMMs: 4 LDS instructions + a wait per 3 MFMAs).  Slot of
// (k-quad q, row r):  q * ROWS + swz(r),  swz = (r & ~3) | ((r + (r >> 3)) & 3): the rotation of the row-within-quad
// index by the 8-row block (and by the k-quad, for the service groups that straddle two k-quads) makes both the ds_write_b128
// of the four micro-tile rows and the ds_read_b128 of 16 consecutive rows hit 16 distinct 16-byte slots -- conflict-free both
// ways.  (Round 1 tried a row-major [row][BK+4] transposed tile: its four row writes piled onto two bank groups, 55 % slower.)
// MEASURED (round 2, B = 256): SQ_LDS_BANK_CONFLICT 0 and a third fewer LDS-array cycles than the k-major staging, yet the
// weight-gradient GEMM is SLOWER (173 vs 145 us) -- the kernel is not bound by its fragment reads.  Kept selectable
// (st_set_tuning(7001)); the default stays k-major.
__device__ __forceinline__ int xq_slot(const int row, const int q) { (void)q; return (row & ~3) | ((row + (row >> 3)) & 3); }
template <int WAVES_M, int BKT, int MI, bool XT, class AL, class BL, class EPI>
__global__ void __launch_bounds__(WAVES_M * 64)
gemm_kernel(const AL al, const BL bl, const EPI epi, const int K, const int ksplit, const int dbg)
{
    constexpr int BM = 32 * WAVES_M * MI, NT = 64 * WAVES_M;
    constexpr int PKT = BKT + 4;                         // row pitch: 36 (BK 32) / 20 (BK 16) floats, both conflict-free for b128
    constexpr bool A_T = AL::kTN && XT, A_K = AL::kTN && !XT;       // transposed staging / k-major staging
    constexpr bool B_T = BL::kTN && XT, B_K = BL::kTN && !XT;
    constexpr int LDA = A_K ? BM + 4 : PKT;             // k-major [BK][BM+4] ; row-major [BM][PKT] ; A_T: k-quad-major [BK/4][BM][4]
    constexpr int LDB = B_K ? BN + 4 : PKT;
    constexpr int A_SZ = A_T ? BKT * BM : (A_K ? BKT * LDA : BM * LDA);
    constexpr int B_SZ = B_T ? BKT * BN : (B_K ? BKT * LDB : BN * LDB);
    constexpr int KQ = BKT / 4;                          // float4 per row per k-tile
    constexpr int A_N = A_T ? (BM / 4) * KQ : BM * KQ;   // items per k-tile: float4s, or 4x4 micro-tiles (4 float4 loads each)
    constexpr int B_N = B_T ? (BN / 4) * KQ : BN * KQ;
    constexpr int A_LD = A_T ? 4 : 1, B_LD = B_T ? 4 : 1;
    constexpr int A_IT = (A_N + NT - 1) / NT, B_IT = (B_N + NT - 1) / NT;
    constexpr int HK = BKT / 2;                          // k per lane-half: lane (m, h) covers k = HK*h .. HK*h + HK-1
    __shared__ __attribute__((aligned(16))) float As[2 * A_SZ];
    __shared__ __attribute__((aligned(16))) float Bs[2 * B_SZ];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int tbx, tby, tbz;
    if constexpr (kPolarEpi<EPI>) xcd_tile_2d(tbx, tby, tbz); else xcd_tile(tbx, tby, tbz);      // the analysis forward (never split-K): 2-D blocks of tiles per XCD
    const int m_blk = tby * BM, n_blk = tbx * BN;
    const int k_begin = tbz * ksplit;
    const int k_end = (k_begin + ksplit < K) ? k_begin + ksplit : K;

    // per-thread item coordinates (fixed across k-tiles): (row-in-tile, k-in-tile) and LDS offset
    int a_i[A_IT], a_k[A_IT], a_l[A_IT], b_i[B_IT], b_k[B_IT], b_l[B_IT];
    bool a_v[A_IT], b_v[B_IT];
    RowState a_st[AL::kTN ? 1 : A_IT], b_st[BL::kTN ? 1 : B_IT];
#pragma unroll
    for (int p = 0; p < A_IT; ++p) {
        const int idx = tid + NT * p;
        a_v[p] = idx < A_N;
        const int id = a_v[p] ? idx : 0;
        if constexpr (A_T) { a_i[p] = (id % (BM / 4)) * 4; a_k[p] = (id / (BM / 4)) * 4; a_l[p] = 0; }   // micro-tile: rows a_i..a_i+3 (fastest across lanes: coalesced), k = a_k..a_k+3
        else if constexpr (A_K) { a_i[p] = (id % (BM / 4)) * 4; a_k[p] = id / (BM / 4); a_l[p] = a_k[p] * LDA + a_i[p]; }
        else { a_i[p] = id / KQ; a_k[p] = (id % KQ) * 4; a_l[p] = a_i[p] * LDA + a_k[p]; a_st[p] = al.row_state(m_blk + a_i[p]); }
    }
#pragma unroll
    for (int p = 0; p < B_IT; ++p) {
        const int idx = (B_T ? NT - 1 - tid : tid) + NT * p;      // micro-tile items: B starts from the last thread, so a tile with fewer items than threads spreads A and B over different waves
        b_v[p] = idx < B_N;
        const int id = b_v[p] ? idx : 0;
        if constexpr (B_T) { b_i[p] = (id % (BN / 4)) * 4; b_k[p] = (id / (BN / 4)) * 4; b_l[p] = 0; }
        else if constexpr (B_K) { b_i[p] = (id % (BN / 4)) * 4; b_k[p] = id / (BN / 4); b_l[p] = b_k[p] * LDB + b_i[p]; }
        else { b_i[p] = id / KQ; b_k[p] = (id % KQ) * 4; b_l[p] = b_i[p] * LDB + b_k[p]; b_st[p] = bl.row_state(n_blk + b_i[p]); }
    }

    // TN x TN with both operands transposed on the way in (the weight-gradient GEMMs): ONE micro-tile per thread -- threads
    // [0, A_N) take A's, threads [A_N, A_N + B_N) take B's -- instead of a (half-idle) A item and a B item each: 16 prefetch
    // registers instead of 32, which is what keeps four waves per SIMD resident.
    constexpr bool UNI = A_T && B_T && (A_N + B_N <= NT) && A_IT == 1 && B_IT == 1 && has_tn_params<AL>::value && has_tn_params<BL>::value;
    const bool u_isA = tid < A_N;
    int u_i = 0, u_k = 0;
    TNParams up{};
    if constexpr (UNI) {
        const int id = u_isA ? tid : (tid - A_N < B_N ? tid - A_N : 0); const int RQ = u_isA ? BM / 4 : BN / 4; u_i = (id % RQ) * 4; u_k = (id / RQ) * 4;
        const TNParams pa = al.tn_params(), pb = bl.tn_params();
        up.base = u_isA ? pa.base : pb.base; up.magic = u_isA ? pa.magic : pb.magic; up.S1 = u_isA ? pa.S1 : pb.S1; up.S2 = u_isA ? pa.S2 : pb.S2;
        up.Tv = u_isA ? pa.Tv : pb.Tv; up.t_lo = u_isA ? pa.t_lo : pb.t_lo; up.clampR = u_isA ? pa.clampR : pb.clampR;
        up.klim = u_isA ? pa.klim : pb.klim; up.collim = u_isA ? pa.collim : pb.collim;
        u_i += u_isA ? m_blk : n_blk;                    // global column of the micro-tile's first row
    }
    float4 ra[UNI ? 1 : A_IT][A_LD], rb[UNI ? 1 : B_IT][UNI ? 1 : B_LD];
    bool oa[UNI ? 1 : A_IT][A_LD], ob[UNI ? 1 : B_IT][UNI ? 1 : B_LD];
    auto gload = [&](int kt) {
        if constexpr (UNI) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int k = kt + u_k + q;
                const unsigned kc = (unsigned)(k < up.clampR ? k : 0);
                const unsigned b = up.magic ? __umulhi(kc, up.magic) : kc;
                const unsigned t = (unsigned)up.t_lo + (kc - __umul24(b, (unsigned)up.Tv));
                const bool ok = k < up.klim && u_i < up.collim;
                oa[0][q] = ok;
                ra[0][q] = ldg128(up.base, ok ? __umul24(b, up.S1) + __umul24(t, up.S2) + (unsigned)u_i : 0u);
            }
            return;
        }
#pragma unroll
        for (int p = 0; p < A_IT; ++p)
#pragma unroll
            for (int q = 0; q < A_LD; ++q) {
                if constexpr (AL::kTN) { bool ok; const unsigned o = al.off(kt + a_k[p] + q, m_blk + a_i[p], ok); oa[p][q] = ok; ra[p][q] = ldg128(al.dummy(), o); }
                else if constexpr (AL::kOff) {
                    const int k = kt + a_k[p];
                    const bool ok = !AL::kCheck || (k >= a_st[p].lo && k < a_st[p].hi);
                    oa[p][q] = ok; ra[p][q] = ldg128(al.dummy(), ok ? a_st[p].o + (unsigned)k : 0u);
                } else {
                    const Src s = al.src(a_st[p], kt + a_k[p]);
                    if constexpr (AL::kCheck) { oa[p][q] = s.ok; ra[p][q] = *reinterpret_cast<const float4*>(s.ok ? s.p : al.dummy()); }
                    else { oa[p][q] = true; ra[p][q] = *reinterpret_cast<const float4*>(s.p); }
                }
            }
#pragma unroll
        for (int p = 0; p < B_IT; ++p)
#pragma unroll
            for (int q = 0; q < B_LD; ++q) {
                if constexpr (BL::kTN) { bool ok; const unsigned o = bl.off(kt + b_k[p] + q, n_blk + b_i[p], ok); ob[p][q] = ok; rb[p][q] = ldg128(bl.dummy(), o); }
                else if constexpr (BL::kOff) {
                    const int k = kt + b_k[p];
                    const bool ok = !BL::kCheck || (k >= b_st[p].lo && k < b_st[p].hi);
                    ob[p][q] = ok; rb[p][q] = ldg128(bl.dummy(), ok ? b_st[p].o + (unsigned)k : 0u);
                } else {
                    const Src s = bl.src(b_st[p], kt + b_k[p]);
                    if constexpr (BL::kCheck) { ob[p][q] = s.ok; rb[p][q] = *reinterpret_cast<const float4*>(s.ok ? s.p : bl.dummy()); }
                    else { ob[p][q] = true; rb[p][q] = *reinterpret_cast<const float4*>(s.p); }
                }
            }
    };
    auto lstore = [&](int buf) {
        float* as = As + buf * A_SZ;
        float* bs = Bs + buf * B_SZ;
        const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
        if constexpr (UNI) {
            if (tid < A_N + B_N) {
                float4 v[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = oa[0][q] ? (u_isA ? al.post(ra[0][q]) : bl.post(ra[0][q])) : zero;
                const int q_ = u_k >> 2, li = u_i - (u_isA ? m_blk : n_blk);
                float* base_ = (u_isA ? as : bs) + q_ * ((u_isA ? BM : BN) * 4);
                *reinterpret_cast<float4*>(base_ + 4 * xq_slot(li + 0, q_)) = make_float4(v[0].x, v[1].x, v[2].x, v[3].x);
                *reinterpret_cast<float4*>(base_ + 4 * xq_slot(li + 1, q_)) = make_float4(v[0].y, v[1].y, v[2].y, v[3].y);
                *reinterpret_cast<float4*>(base_ + 4 * xq_slot(li + 2, q_)) = make_float4(v[0].z, v[1].z, v[2].z, v[3].z);
                *reinterpret_cast<float4*>(base_ + 4 * xq_slot(li + 3, q_)) = make_float4(v[0].w, v[1].w, v[2].w, v[3].w);
            }
            return;
        }
#pragma unroll
        for (int p = 0; p < A_IT; ++p) {
            if (A_N % NT != 0 && !a_v[p]) continue;
            if constexpr (A_T) {                       // 4(k) x 4(m) micro-tile -> four k-quads
                float4 v[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = oa[p][q] ? al.post(ra[p][q]) : zero;
                const int q_ = a_k[p] >> 2;
                float* base_ = as + q_ * (BM * 4);
                *reinterpret_cast<float4*>(base_ + 4 * xq_slot(a_i[p] + 0, q_)) = make_float4(v[0].x, v[1].x, v[2].x, v[3].x);
                *reinterpret_cast<float4*>(base_ + 4 * xq_slot(a_i[p] + 1, q_)) = make_float4(v[0].y, v[1].y, v[2].y, v[3].y);
                *reinterpret_cast<float4*>(base_ + 4 * xq_slot(a_i[p] + 2, q_)) = make_float4(v[0].z, v[1].z, v[2].z, v[3].z);
                *reinterpret_cast<float4*>(base_ + 4 * xq_slot(a_i[p] + 3, q_)) = make_float4(v[0].w, v[1].w, v[2].w, v[3].w);
            } else {
                *reinterpret_cast<float4*>(as + a_l[p]) = oa[p][0] ? al.post(ra[p][0]) : zero;
            }
        }
#pragma unroll
        for (int p = 0; p < B_IT; ++p) {
            if (B_N % NT != 0 && !b_v[p]) continue;
            if constexpr (B_T) {
                float4 v[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = ob[p][q] ? bl.post(rb[p][q]) : zero;
                const int q_ = b_k[p] >> 2;
                float* base_ = bs + q_ * (BN * 4);
                *reinterpret_cast<float4*>(base_ + 4 * xq_slot(b_i[p] + 0, q_)) = make_float4(v[0].x, v[1].x, v[2].x, v[3].x);
                *reinterpret_cast<float4*>(base_ + 4 * xq_slot(b_i[p] + 1, q_)) = make_float4(v[0].y, v[1].y, v[2].y, v[3].y);
                *reinterpret_cast<float4*>(base_ + 4 * xq_slot(b_i[p] + 2, q_)) = make_float4(v[0].z, v[1].z, v[2].z, v[3].z);
                *reinterpret_cast<float4*>(base_ + 4 * xq_slot(b_i[p] + 3, q_)) = make_float4(v[0].w, v[1].w, v[2].w, v[3].w);
            } else {
                *reinterpret_cast<float4*>(bs + b_l[p]) = ob[p][0] ? bl.post(rb[p][0]) : zero;
            }
        }
    };

    f32x16 acc[MI][NJ];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[mi][j][i] = 0.f;

    if (k_begin < k_end) {
        gload(k_begin);
        lstore(0);
        __syncthreads();
        int cur = 0;
        const int h = lane >> 5, l31 = lane & 31;
        // NT: lane reads HK consecutive floats of its row; TN: lane reads column l31 of rows HK*h .. HK*h+HK-1
        const int a_off = A_T ? 0 : (A_K ? (HK * h) * LDA + wave * (32 * MI) + l31 : (wave * (32 * MI) + l31) * LDA + HK * h);
        constexpr int A_MI = A_K ? 32 : 32 * LDA;              // LDS offset between the wave's 32-row blocks
        const int b_off = B_T ? 0 : (B_K ? (HK * h) * LDB + l31 : l31 * LDB + HK * h);
        // k-quad-major operands: float offset of (row, k-quad HK/4 * h + q) -- per lane, fixed across k-tiles
        int a_xq[A_T ? MI : 1][A_T ? HK / 4 : 1], b_xq[B_T ? NJ : 1][B_T ? HK / 4 : 1];
        if constexpr (A_T) {
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int q = 0; q < HK / 4; ++q) { const int kq = (HK / 4) * h + q; a_xq[mi][q] = (kq * BM + xq_slot((wave * MI + mi) * 32 + l31, kq)) * 4; }
        }
        if constexpr (B_T) {
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int q = 0; q < HK / 4; ++q) { const int kq = (HK / 4) * h + q; b_xq[j][q] = (kq * BN + xq_slot(32 * j + l31, kq)) * 4; }
        }
        for (int kt = k_begin; kt < k_end; kt += BKT) {
            // branch-free body: the last iteration re-loads its own tile (harmless) instead of skipping the prefetch,
            // so the whole loop is ONE basic block and the accumulators stay put
            const bool more = kt + BKT < k_end;
            if (!ST_DBG(1)) gload(more ? kt + BKT : kt);
            __builtin_amdgcn_sched_barrier(0);           // phases stay in order: prefetch issue | fragments + MFMAs | LDS stores (else the
                                                         // stores and their vmcnt waits get hoisted above the MFMAs)
            const float* as = As + cur * A_SZ + a_off;
            const float* bs = Bs + cur * B_SZ + b_off;
            if (!ST_DBG(4)) {
                float af[MI][HK], bf[NJ][HK];
                if constexpr (!A_K) {
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                        for (int q = 0; q < HK / 4; ++q) {
                            const float4 v = *reinterpret_cast<const float4*>(A_T ? as + a_xq[A_T ? mi : 0][A_T ? q : 0] : as + mi * A_MI + 4 * q);
                            af[mi][4 * q] = v.x; af[mi][4 * q + 1] = v.y; af[mi][4 * q + 2] = v.z; af[mi][4 * q + 3] = v.w;
                        }
                }
                if constexpr (!B_K) {
#pragma unroll
                    for (int j = 0; j < NJ; ++j)
#pragma unroll
                        for (int q = 0; q < HK / 4; ++q) {
                            const float4 v = *reinterpret_cast<const float4*>(B_T ? bs + b_xq[B_T ? j : 0][B_T ? q : 0] : bs + 32 * j * LDB + 4 * q);
                            bf[j][4 * q] = v.x; bf[j][4 * q + 1] = v.y; bf[j][4 * q + 2] = v.z; bf[j][4 * q + 3] = v.w;
                        }
                }
#pragma unroll
                for (int kk = 0; kk < HK; ++kk) {
                    float a[MI], b[NJ];
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi) { if constexpr (A_K) a[mi] = as[kk * LDA + mi * A_MI]; else a[mi] = af[mi][kk]; }
#pragma unroll
                    for (int j = 0; j < NJ; ++j) { if constexpr (B_K) b[j] = bs[kk * LDB + 32 * j]; else b[j] = bf[j][kk]; }
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                        for (int j = 0; j < NJ; ++j)
                            acc[mi][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi], b[j], acc[mi][j], 0, 0, 0);
                }
                if constexpr (A_K && B_K && MI == 1) {
                    // k-major operands: one scalar A and NJ scalar B fragments per k-step.  Recipe: the LDS reads of step k+1 are
                    // issued before the MFMAs of step k (left alone the compiler emits ds_read / s_waitcnt / MFMA triplets).
                    __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
                    for (int kk = 0; kk < HK; ++kk) {
                        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                        __builtin_amdgcn_sched_group_barrier(0x008, MI * NJ, 0);
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            if (!ST_DBG(1)) lstore(cur ^ 1);
            if (!ST_DBG(2)) __syncthreads();
            cur ^= 1;
        }
    }
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) epi(m_blk + (wave * MI + mi) * 32, n_blk, acc[mi]);
}

// ------------------------------------------------------------------------------ mixed-precision variant
// bf16 operands, fp32 accumulation (v_mfma_f32_32x32x16_bf16: 16x the fp32 MFMA rate) for the bf16 configurations of
// BASELINE.json.  Global data stays fp32 (master weights, activations, gradients): the operands are rounded to bf16
// (round-to-nearest-even, v_cvt_pk_bf16_f32) while they are staged into LDS, so loaders and epilogues are shared with
// the fp32 kernel and the result equals "round both operands to bf16, multiply, accumulate in fp32".
// LDS: both operands K-contiguous, [row][32 + 8] bf16 (80 B pitch: conflict-free 128-bit fragment reads).  M/N-contiguous
// (TN-type) operands are transposed on the way in: a thread loads a 4(k) x 4(m) micro-tile as four float4 and writes
// four 8-byte k-quads.  One k-tile = 32 = two MFMA k-steps; a lane's fragment is 8 consecutive k (k = 16 s + 8 (lane/32) + i)
// for both operands, so the reduction order inside a step is whatever the hardware uses -- identically for A and B.
typedef __bf16 st_bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 st_f16x8 __attribute__((ext_vector_type(8)));
// v_cvt_pk_bf16_f32 (round to nearest even) as volatile asm: left to the compiler, the conversions were (1) vectorised as
// (x,z)/(y,w) pairs and re-shuffled with four extra VALU instructions per float4, and (2) hoisted above the MFMA phase of
// the k-loop -- the sched_barrier only orders machine instructions -- so the wave waited for its prefetch right after
// issuing it.  A volatile asm stays where lstore() puts it, behind the MFMAs.
__device__ __forceinline__ unsigned st_pack_bf16(float a, float b)
{
    unsigned r; asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r;
}
// fp16 (HT = 2, BASELINE configs[4] "fp16 mixed precision"): IEEE round to nearest even; a value beyond the fp16 range becomes
// inf, exactly as under the reference's Apex amp -- the non-finite gradient norm then makes the optimizer kernel skip the step
// (clip_adam_kernel) and the host lowers the loss scale.  The one operand that is saturated at its source instead is the
// polar backward's output (polar_bwd_kernel: 1e7-sized sub-gradients on silent frames).
__device__ __forceinline__ unsigned st_pack_f16(float a, float b)
{
    unsigned r; asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r;
}
template <int HT> __device__ __forceinline__ unsigned st_pack_h(float a, float b) { if constexpr (HT == 2) return st_pack_f16(a, b); else return st_pack_bf16(a, b); }

// fp32 operands as THREE bfloat16 planes (PL = 3):  x = x1 + x2 + x3,  x1 = bf16(x), x2 = bf16(x - x1), x3 = bf16(x - x1 - x2)
// (both differences are exact in fp32), 24 significand bits in all.  The product a*b is then formed from the six partial products of
// order >= 2^-16 -- a1b3, a3b1, a2b2, a1b2, a2b1, a1b1, each exact in the fp32 accumulator's input -- and the three dropped ones
// are below 2^-24 relative: fp32-grade products (measured against float64: not worse than the fp32 MFMA path, tests/test_gpu_split.py)
// on the bf16 MATRIX pipe.  Why: on gfx950 v_mfma_f32_*_f32 runs at the vector-ALU fp32 rate (157 TF) and does not overlap vector
// work at all (tools/ubench), while 6 x v_mfma_f32_32x32x16_bf16 cost 0.375 of the fp32 MFMA's cycles and run beside the VALU.
__device__ __forceinline__ float st_bf16_lo(unsigned p) { return __uint_as_float(p << 16); }
__device__ __forceinline__ float st_bf16_hi(unsigned p) { return __uint_as_float(p & 0xffff0000u); }
// (conversions through __builtin_convertvector, not the volatile asm of st_pack_bf16: the scheduler has to see them as VALU work it
// may place between the MFMAs)
__device__ __forceinline__ unsigned st_cvt_pk_bf16(const float a, const float b)
{
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    union { bf16x2_t h; unsigned u; } r; r.h = __builtin_convertvector((f32x2_t){a, b}, bf16x2_t); return r.u;
}
__device__ __forceinline__ void st_split3(const float x, const float y, const float z, const float w, uint2 (&pl)[3])
{
    float r[4] = {x, y, z, w};
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        pl[p] = make_uint2(st_cvt_pk_bf16(r[0], r[1]), st_cvt_pk_bf16(r[2], r[3]));
        if (p < 2) { r[0] -= st_bf16_lo(pl[p].x); r[1] -= st_bf16_hi(pl[p].x); r[2] -= st_bf16_lo(pl[p].y); r[3] -= st_bf16_hi(pl[p].y); }
    }
}
// HT: 1 = bfloat16 operands (v_mfma_f32_32x32x16_bf16), 2 = float16 operands (v_mfma_f32_32x32x16_f16); same rate, same layout.
// PL: 1 = operands rounded to HT; 3 = the three-plane bfloat16 split of fp32 operands (HT = 1).  BKH: k-tile depth (32 or 16).
#ifndef ST_SPLIT_VALU_PER_MFMA
#define ST_SPLIT_VALU_PER_MFMA 5
#endif
// The kernel proper is gemm_half_body (tile coordinates passed in): gemm_half_kernel runs it for one operand set, gemm_half_pair_kernel for TWO
// independent GEMMs of the same shape in one launch (round 3: the layer-1 / layer-9 GEMMs of the two autoencoders on the wide path are 64 x 33.8 k-output
// launches that fill a third of the chip and cost 10-28 us each mostly in launch, ramp and drain: ten launches -> five).
template <int WAVES_M, int HT, int PL, int BKH, class AL, class BL, class EPI>
__device__ __forceinline__ void
gemm_half_body(const AL& al, const BL& bl, const EPI& epi, const int K, const int ksplit, const int tbx, const int tby, const int tbz)
{
    static_assert(PL == 1 || (PL == 3 && HT == 1), "the split needs the fp32 exponent range: bfloat16 planes only");
    constexpr int BM = 32 * WAVES_M, NT = 64 * WAVES_M;
    constexpr int LD = BKH + 8;                                   // bf16 elements per LDS row
    constexpr int A_SZ = BM * LD, B_SZ = BN * LD;                 // one plane of one buffer
    constexpr int KQ = BKH / 4;
    constexpr int A_N = AL::kTN ? (BM / 4) * KQ : BM * KQ;       // items: float4 along k (NT) or 4x4 micro-tiles (TN)
    constexpr int B_N = BL::kTN ? (BN / 4) * KQ : BN * KQ;
    constexpr int A_IT = (A_N + NT - 1) / NT, B_IT = (B_N + NT - 1) / NT;
    constexpr int A_LDS = AL::kTN ? 4 : 1, B_LDS = BL::kTN ? 4 : 1;    // global float4 loads per item
    extern __shared__ __attribute__((aligned(16))) unsigned short half_lds[];      // [2 buffers][PL planes][A_SZ] | [2][PL][B_SZ] | PL = 3: a dump slot per thread
    unsigned short* const As = half_lds;
    unsigned short* const Bs = half_lds + 2 * PL * A_SZ;
    unsigned short* const dump = half_lds + 2 * PL * (A_SZ + B_SZ) + 4 * threadIdx.x;   // threads without an item store there: no branch in the loop body

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m_blk = tby * BM, n_blk = tbx * BN;
    const int k_begin = tbz * ksplit;
    const int k_end = (k_begin + ksplit < K) ? k_begin + ksplit : K;

    int a_i[A_IT], a_k[A_IT], b_i[B_IT], b_k[B_IT];
    bool a_v[A_IT], b_v[B_IT];
    RowState a_st[AL::kTN ? 1 : A_IT], b_st[BL::kTN ? 1 : B_IT];
#pragma unroll
    for (int p = 0; p < A_IT; ++p) {
        const int idx = tid + NT * p; a_v[p] = idx < A_N; const int id = a_v[p] ? idx : 0;
        if constexpr (AL::kTN) { a_i[p] = (id % (BM / 4)) * 4; a_k[p] = (id / (BM / 4)) * 4; }   // m fastest across lanes: coalesced global loads (a k-fastest order has conflict-free LDS writes but measured 10 % slower)
        else { a_i[p] = id / KQ; a_k[p] = (id % KQ) * 4; a_st[p] = al.row_state(m_blk + a_i[p]); }
    }
#pragma unroll
    for (int p = 0; p < B_IT; ++p) {
        const int idx = tid + NT * p; b_v[p] = idx < B_N; const int id = b_v[p] ? idx : 0;
        if constexpr (BL::kTN) { b_i[p] = (id % (BN / 4)) * 4; b_k[p] = (id / (BN / 4)) * 4; }
        else { b_i[p] = id / KQ; b_k[p] = (id % KQ) * 4; b_st[p] = bl.row_state(n_blk + b_i[p]); }
    }

    float4 ra[A_IT][A_LDS], rb[B_IT][B_LDS];
    bool oa[A_IT][A_LDS], ob[B_IT][B_LDS];
    auto gload = [&](int kt) {
#pragma unroll
        for (int p = 0; p < A_IT; ++p)
#pragma unroll
            for (int q = 0; q < A_LDS; ++q) {
                if constexpr (AL::kTN) { bool ok; const unsigned o = al.off(kt + a_k[p] + q, m_blk + a_i[p], ok); oa[p][q] = ok; ra[p][q] = ldg128(al.dummy(), o); }
                else if constexpr (AL::kOff) {
                    const int k = kt + a_k[p];
                    const bool ok = !AL::kCheck || (k >= a_st[p].lo && k < a_st[p].hi);
                    oa[p][q] = ok; ra[p][q] = ldg128(al.dummy(), ok ? a_st[p].o + (unsigned)k : 0u);
                } else {
                    const Src s = al.src(a_st[p], kt + a_k[p]);
                    if constexpr (AL::kCheck) { oa[p][q] = s.ok; ra[p][q] = *reinterpret_cast<const float4*>(s.ok ? s.p : al.dummy()); }
                    else { oa[p][q] = true; ra[p][q] = *reinterpret_cast<const float4*>(s.p); }
                }
            }
#pragma unroll
        for (int p = 0; p < B_IT; ++p)
#pragma unroll
            for (int q = 0; q < B_LDS; ++q) {
                if constexpr (BL::kTN) { bool ok; const unsigned o = bl.off(kt + b_k[p] + q, n_blk + b_i[p], ok); ob[p][q] = ok; rb[p][q] = ldg128(bl.dummy(), o); }
                else if constexpr (BL::kOff) {
                    const int k = kt + b_k[p];
                    const bool ok = !BL::kCheck || (k >= b_st[p].lo && k < b_st[p].hi);
                    ob[p][q] = ok; rb[p][q] = ldg128(bl.dummy(), ok ? b_st[p].o + (unsigned)k : 0u);
                } else {
                    const Src s = bl.src(b_st[p], kt + b_k[p]);
                    if constexpr (BL::kCheck) { ob[p][q] = s.ok; rb[p][q] = *reinterpret_cast<const float4*>(s.ok ? s.p : bl.dummy()); }
                    else { ob[p][q] = true; rb[p][q] = *reinterpret_cast<const float4*>(s.p); }
                }
            }
    };
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    // four consecutive k of one row -> LDS: one 8-byte store per plane (plane p of a buffer sits p * SZ elements behind plane 0)
    auto put4 = [&](unsigned short* dst, const int plane_sz, const float x, const float y, const float z, const float w) {
        if constexpr (PL == 3) {
            uint2 pl[3]; st_split3(x, y, z, w, pl);
#pragma unroll
            for (int p = 0; p < 3; ++p) *reinterpret_cast<uint2*>(dst + p * plane_sz) = pl[p];
        } else { (void)plane_sz; *reinterpret_cast<uint2*>(dst) = make_uint2(st_pack_h<HT>(x, y), st_pack_h<HT>(z, w)); }
    };
    auto lstore = [&](int buf) {
        unsigned short* as = As + buf * PL * A_SZ;
        unsigned short* bs = Bs + buf * PL * B_SZ;
#pragma unroll
        for (int p = 0; p < A_IT; ++p) {
            const bool skip = A_N % NT != 0 && !a_v[p];
            if (PL != 3 && skip) continue;
            if constexpr (AL::kTN) {
                float4 v[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = oa[p][q] ? al.post(ra[p][q]) : zero4;
                unsigned short* const t = as + a_i[p] * LD + a_k[p];
                put4(skip ? dump : t + 0 * LD, skip ? 0 : A_SZ, v[0].x, v[1].x, v[2].x, v[3].x);
                put4(skip ? dump : t + 1 * LD, skip ? 0 : A_SZ, v[0].y, v[1].y, v[2].y, v[3].y);
                put4(skip ? dump : t + 2 * LD, skip ? 0 : A_SZ, v[0].z, v[1].z, v[2].z, v[3].z);
                put4(skip ? dump : t + 3 * LD, skip ? 0 : A_SZ, v[0].w, v[1].w, v[2].w, v[3].w);
            } else {
                const float4 v = oa[p][0] ? al.post(ra[p][0]) : zero4;
                put4(skip ? dump : as + a_i[p] * LD + a_k[p], skip ? 0 : A_SZ, v.x, v.y, v.z, v.w);
            }
        }
#pragma unroll
        for (int p = 0; p < B_IT; ++p) {
            const bool skip = B_N % NT != 0 && !b_v[p];
            if (PL != 3 && skip) continue;
            if constexpr (BL::kTN) {
                float4 v[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = ob[p][q] ? bl.post(rb[p][q]) : zero4;
                unsigned short* const t = bs + b_i[p] * LD + b_k[p];
                put4(skip ? dump : t + 0 * LD, skip ? 0 : B_SZ, v[0].x, v[1].x, v[2].x, v[3].x);
                put4(skip ? dump : t + 1 * LD, skip ? 0 : B_SZ, v[0].y, v[1].y, v[2].y, v[3].y);
                put4(skip ? dump : t + 2 * LD, skip ? 0 : B_SZ, v[0].z, v[1].z, v[2].z, v[3].z);
                put4(skip ? dump : t + 3 * LD, skip ? 0 : B_SZ, v[0].w, v[1].w, v[2].w, v[3].w);
            } else {
                const float4 v = ob[p][0] ? bl.post(rb[p][0]) : zero4;
                put4(skip ? dump : bs + b_i[p] * LD + b_k[p], skip ? 0 : B_SZ, v.x, v.y, v.z, v.w);
            }
        }
    };

    f32x16 acc[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;

    if constexpr (PL == 3) {
        // Software pipeline, one basic block per k-tile:  fragments of tile t  |  the 6 * NJ MFMAs of tile t, term-major (consecutive
        // MFMAs write different accumulators), each followed by ONE plane-stage of the split of tile t+1 (2 conversions, the 8-byte
        // LDS store of that plane, the 8 instructions that form the next remainder) so that the vector work issues in the MFMA's
        // shadow -- a dense MFMA run would stall the wave on the busy matrix pipe and serialise everything behind it (tools/ubench)
        // |  global loads of tile t+2 (their registers were consumed by the stages)  |  barrier.  sched_barrier pins the order.
        static_assert(BKH == 16, "PL == 3 runs one MFMA k-step per k-tile");
        constexpr int A_U = A_IT * (AL::kTN ? 4 : 1), B_U = B_IT * (BL::kTN ? 4 : 1), NU = A_U + B_U, NS = 3 * NU, NM = 6 * NJ;
        float sr[NU][4]; unsigned short* sdst[NU]; int ssz[NU];
        auto stage_begin = [&](const int buf) {
            unsigned short* as = As + buf * PL * A_SZ;
            unsigned short* bs = Bs + buf * PL * B_SZ;
#pragma unroll
            for (int p = 0; p < A_IT; ++p) {
                const bool skip = A_N % NT != 0 && !a_v[p];
                if constexpr (AL::kTN) {
                    float4 v[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q] = oa[p][q] ? al.post(ra[p][q]) : zero4;
                    const float m[4][4] = {{v[0].x, v[1].x, v[2].x, v[3].x}, {v[0].y, v[1].y, v[2].y, v[3].y}, {v[0].z, v[1].z, v[2].z, v[3].z}, {v[0].w, v[1].w, v[2].w, v[3].w}};
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int u = 4 * p + i;
#pragma unroll
                        for (int e = 0; e < 4; ++e) sr[u][e] = m[i][e];
                        sdst[u] = skip ? dump : as + (a_i[p] + i) * LD + a_k[p]; ssz[u] = skip ? 0 : A_SZ;
                    }
                } else {
                    const float4 v = oa[p][0] ? al.post(ra[p][0]) : zero4;
                    sr[p][0] = v.x; sr[p][1] = v.y; sr[p][2] = v.z; sr[p][3] = v.w;
                    sdst[p] = skip ? dump : as + a_i[p] * LD + a_k[p]; ssz[p] = skip ? 0 : A_SZ;
                }
            }
#pragma unroll
            for (int p = 0; p < B_IT; ++p) {
                const bool skip = B_N % NT != 0 && !b_v[p];
                if constexpr (BL::kTN) {
                    float4 v[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q] = ob[p][q] ? bl.post(rb[p][q]) : zero4;
                    const float m[4][4] = {{v[0].x, v[1].x, v[2].x, v[3].x}, {v[0].y, v[1].y, v[2].y, v[3].y}, {v[0].z, v[1].z, v[2].z, v[3].z}, {v[0].w, v[1].w, v[2].w, v[3].w}};
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int u = A_U + 4 * p + i;
#pragma unroll
                        for (int e = 0; e < 4; ++e) sr[u][e] = m[i][e];
                        sdst[u] = skip ? dump : bs + (b_i[p] + i) * LD + b_k[p]; ssz[u] = skip ? 0 : B_SZ;
                    }
                } else {
                    const int u = A_U + p;
                    const float4 v = ob[p][0] ? bl.post(rb[p][0]) : zero4;
                    sr[u][0] = v.x; sr[u][1] = v.y; sr[u][2] = v.z; sr[u][3] = v.w;
                    sdst[u] = skip ? dump : bs + b_i[p] * LD + b_k[p]; ssz[u] = skip ? 0 : B_SZ;
                }
            }
        };
        auto stage = [&](const int u, const int pp) {
            const uint2 pl = make_uint2(st_cvt_pk_bf16(sr[u][0], sr[u][1]), st_cvt_pk_bf16(sr[u][2], sr[u][3]));
            *reinterpret_cast<uint2*>(sdst[u] + pp * ssz[u]) = pl;
            if (pp < 2) { sr[u][0] -= st_bf16_lo(pl.x); sr[u][1] -= st_bf16_hi(pl.x); sr[u][2] -= st_bf16_lo(pl.y); sr[u][3] -= st_bf16_hi(pl.y); }
        };
        if (k_begin < k_end) {
            gload(k_begin);
            lstore(0);
            gload(k_begin + BKH < k_end ? k_begin + BKH : k_begin);
            __syncthreads();
            int cur = 0;
            const int h = lane >> 5, l31 = lane & 31;
            const int a_off = (wave * 32 + l31) * LD + 8 * h;
            const int b_off = l31 * LD + 8 * h;
            for (int kt = k_begin; kt < k_end; kt += BKH) {
                const unsigned short* as = As + cur * PL * A_SZ + a_off;
                const unsigned short* bs = Bs + cur * PL * B_SZ + b_off;
                st_bf16x8 a[3], b[NJ][3];
#pragma unroll
                for (int p = 0; p < 3; ++p) a[p] = *reinterpret_cast<const st_bf16x8*>(as + p * A_SZ);
#pragma unroll
                for (int j = 0; j < NJ; ++j)
#pragma unroll
                    for (int p = 0; p < 3; ++p) b[j][p] = *reinterpret_cast<const st_bf16x8*>(bs + p * B_SZ + 32 * j * LD);
                __builtin_amdgcn_sched_barrier(0);
                constexpr int TA[6] = {0, 2, 1, 0, 1, 0}, TB[6] = {2, 0, 1, 1, 0, 0};      // smallest partial products first
#pragma unroll
                for (int m = 0; m < NM; ++m) {
                    const int t = m / NJ, j = m % NJ;
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[TA[t]], b[j][TB[t]], acc[j], 0, 0, 0);
                    if (m == 0) {
                        stage_begin(cur ^ 1);                    // tile t+1 (stale registers past the end: that buffer is not read again)
                        const int k2 = kt + 2 * BKH;             // its registers are free now: tile t+2 has the whole iteration to arrive
                        gload(k2 < k_end ? k2 : kt);
                    }
                    // stage s runs behind MFMA 1 + s * (NM - 1) / NS  (all stages spread over MFMAs 1 .. NM-1)
#pragma unroll
                    for (int sg = 0; sg < NS; ++sg)
                        if (1 + (sg * (NM - 1)) / NS == m) stage(sg / 3, sg % 3);
                    __builtin_amdgcn_sched_barrier(0);
                }
                __syncthreads();
                cur ^= 1;
            }
        }
    } else
    if (k_begin < k_end) {
        gload(k_begin);
        lstore(0);
        __syncthreads();
        int cur = 0;
        const int h = lane >> 5, l31 = lane & 31;
        const int a_off = (wave * 32 + l31) * LD + 8 * h;
        const int b_off = l31 * LD + 8 * h;
        for (int kt = k_begin; kt < k_end; kt += BKH) {
            const bool more = kt + BKH < k_end;
            gload(more ? kt + BKH : kt);                     // branch-free body (see gemm_kernel)
            __builtin_amdgcn_sched_barrier(0);
            const unsigned short* as = As + cur * PL * A_SZ + a_off;
            const unsigned short* bs = Bs + cur * PL * B_SZ + b_off;
#pragma unroll
            for (int ks = 0; ks < BKH / 16; ++ks) {
                if constexpr (PL == 3) {
                    st_bf16x8 a[3];
#pragma unroll
                    for (int p = 0; p < 3; ++p) a[p] = *reinterpret_cast<const st_bf16x8*>(as + p * A_SZ + 16 * ks);
#pragma unroll
                    for (int j = 0; j < NJ; ++j) {
                        st_bf16x8 b[3];
#pragma unroll
                        for (int p = 0; p < 3; ++p) b[p] = *reinterpret_cast<const st_bf16x8*>(bs + p * B_SZ + 32 * j * LD + 16 * ks);
                        // smallest partial products first
                        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], acc[j], 0, 0, 0);
                        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], acc[j], 0, 0, 0);
                        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], acc[j], 0, 0, 0);
                        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], acc[j], 0, 0, 0);
                        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], acc[j], 0, 0, 0);
                        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], acc[j], 0, 0, 0);
                    }
                } else if constexpr (HT == 2) {
                    const st_f16x8 a = *reinterpret_cast<const st_f16x8*>(as + 16 * ks);
#pragma unroll
                    for (int j = 0; j < NJ; ++j) {
                        const st_f16x8 b = *reinterpret_cast<const st_f16x8*>(bs + 32 * j * LD + 16 * ks);
                        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[j], 0, 0, 0);
                    }
                } else {
                    const st_bf16x8 a = *reinterpret_cast<const st_bf16x8*>(as + 16 * ks);
#pragma unroll
                    for (int j = 0; j < NJ; ++j) {
                        const st_bf16x8 b = *reinterpret_cast<const st_bf16x8*>(bs + 32 * j * LD + 16 * ks);
                        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[j], 0, 0, 0);
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            lstore(cur ^ 1);
            __syncthreads();
            cur ^= 1;
        }
    }
    epi(m_blk + wave * 32, n_blk, acc);
}
template <int WAVES_M, int HT, int PL, int BKH, class AL, class BL, class EPI>
__global__ void __launch_bounds__(WAVES_M * 64)
gemm_half_kernel(const AL al, const BL bl, const EPI epi, const int K, const int ksplit)
{
    int tbx, tby, tbz; xcd_tile(tbx, tby, tbz);
    gemm_half_body<WAVES_M, HT, PL, BKH>(al, bl, epi, K, ksplit, tbx, tby, tbz);
}
// z-slices [0, nzh) belong to the first GEMM, [nzh, 2 nzh) to the second.  An epilogue that derives its split-K slab from the launch's z index (StoreC,
// through xcd_tile) sees z = nzh + slice for the second GEMM: the launcher hands that epilogue an output pointer moved back by nzh slabs.
template <int WAVES_M, int HT, int PL, int BKH, class AL, class BL, class EPI>
__global__ void __launch_bounds__(WAVES_M * 64)
gemm_half_pair_kernel(const AL al0, const BL bl0, const EPI epi0, const AL al1, const BL bl1, const EPI epi1, const int K, const int ksplit, const int nzh)
{
    int tbx, tby, tbz; xcd_tile(tbx, tby, tbz);
    if (tbz < nzh) gemm_half_body<WAVES_M, HT, PL, BKH>(al0, bl0, epi0, K, ksplit, tbx, tby, tbz);
    else gemm_half_body<WAVES_M, HT, PL, BKH>(al1, bl1, epi1, K, ksplit, tbx, tby, tbz - nzh);
}

template <int WAVES_M, int HT = 1, int PL = 1, int BKH = (PL == 3 ? 16 : 32), class AL, class BL, class EPI>
static inline int launch_half_pair(const AL& al0, const BL& bl0, const EPI& epi0, const AL& al1, const BL& bl1, const EPI& epi1,
                                   int M, int Nc, int K, int nsplit, hipStream_t s)
{
    constexpr int BM = 32 * WAVES_M;
    constexpr size_t lds = ((size_t)2 * PL * (BM + BN) * (BKH + 8) + (PL == 3 ? 4 * 64 * WAVES_M : 0)) * sizeof(unsigned short);
    int ksplit = K;
    if (nsplit > 1) ksplit = st_round_up((K + nsplit - 1) / nsplit, 32);
    const int nzh = nsplit > 1 ? nsplit : 1;
    dim3 grid((Nc + BN - 1) / BN, (M + BM - 1) / BM, 2 * nzh);
    if (lds > 65536) { const int rc = ::ensure_dyn_lds((const void*)gemm_half_pair_kernel<WAVES_M, HT, PL, BKH, AL, BL, EPI>, "gemm_half_pair_kernel"); if (rc) return rc; }
    hipLaunchKernelGGL((gemm_half_pair_kernel<WAVES_M, HT, PL, BKH, AL, BL, EPI>), grid, dim3(WAVES_M * 64), lds, s, al0, bl0, epi0, al1, bl1, epi1, K, ksplit, nzh);
    return 0;
}

template <int WAVES_M, int HT = 1, int PL = 1, int BKH = (PL == 3 ? 16 : 32), class AL, class BL, class EPI>
static inline int launch_half(const AL& al, const BL& bl, const EPI& epi, int M, int Nc, int K, int nsplit, hipStream_t s)
{
    constexpr int BM = 32 * WAVES_M;
    constexpr size_t lds = ((size_t)2 * PL * (BM + BN) * (BKH + 8) + (PL == 3 ? 4 * 64 * WAVES_M : 0)) * sizeof(unsigned short);
    int ksplit = K;
    if (nsplit > 1) ksplit = st_round_up((K + nsplit - 1) / nsplit, 32);
    dim3 grid((Nc + BN - 1) / BN, (M + BM - 1) / BM, nsplit > 1 ? nsplit : 1);
    if (lds > 65536) { const int rc = ::ensure_dyn_lds((const void*)gemm_half_kernel<WAVES_M, HT, PL, BKH, AL, BL, EPI>, "gemm_half_kernel"); if (rc) return rc; }
    hipLaunchKernelGGL((gemm_half_kernel<WAVES_M, HT, PL, BKH, AL, BL, EPI>), grid, dim3(WAVES_M * 64), lds, s, al, bl, epi, K, ksplit);
    return 0;
}

template <int WAVES_M, int BKT, int MI = 1, bool XT = false, class AL, class BL, class EPI>
static inline void launch(const AL& al, const BL& bl, const EPI& epi, int M, int Nc, int K, int nsplit,
                          hipStream_t s, int dbg = 0)
{
    constexpr int BM = 32 * WAVES_M * MI;
    int ksplit = K;
    if (nsplit > 1) ksplit = st_round_up((K + nsplit - 1) / nsplit, BK);
    // exactly nsplit z-slices: a slice that starts past K stores zeros, so consumers sum a fixed slab count
    dim3 grid((Nc + BN - 1) / BN, (M + BM - 1) / BM, nsplit > 1 ? nsplit : 1);
    hipLaunchKernelGGL((gemm_kernel<WAVES_M, BKT, MI, XT, AL, BL, EPI>), grid, dim3(WAVES_M * 64), 0, s, al, bl, epi, K, ksplit, dbg);
}

}  // namespace stg
