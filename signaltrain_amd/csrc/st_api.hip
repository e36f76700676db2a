// st_api.hip -- extern "C" entry points of libsignaltrain_hip.so (see include/signaltrain_hip.h).
// All device work is launched on the caller's stream; no allocation, no synchronisation.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <math.h>
#include <mutex>
#include <set>
#include <utility>
#include "st_common.h"
#include "st_gemm.h"
#include "st_gemm_planes.h"
#include "st_gemm_tn.h"
#include "st_gemm16.h"
#include "st_misc.h"
#include "st_ae.h"
#include "st_ae_wide.h"
#include "st_ae_split.h"
#include "st_ae32.h"
#include "st_dp.h"
#include "st_feed.h"

// ------------------------------------------------------------------------------ errors
static thread_local char g_err[512] = "";
int st_fail(int code, const char* fmt, ...)
{
    va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof(g_err), fmt, ap); va_end(ap);
    return code;
}
int st_check_launch(const char* what)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return st_fail(ST_ERR_LAUNCH, "%s: %s", what, hipGetErrorString(e));
    return ST_OK;
}
static void prof_mark(const char* name, void* stream);
#define ST_LAUNCHED(what) do { int rc_ = st_check_launch(what); if (rc_ != ST_OK) return rc_; prof_mark(what, stream); } while (0)
#define ST_TRY(x) do { int rc_ = (x); if (rc_ != ST_OK) return rc_; } while (0)
#define ST_REQ(cond, ...) do { if (!(cond)) return st_fail(ST_ERR_ARG, __VA_ARGS__); } while (0)

extern "C" const char* st_last_error(void) { return g_err; }

// ------------------------------------------------------------------------------ optional event profiling
// Off by default (zero overhead).  When enabled, every kernel launch of the library is followed by a
// hipEventRecord on the launch stream; st_profile_report() turns consecutive events into per-kernel
// durations.  Used by bench.py's roofline leg only -- never inside the timed region.
#define ST_PROF_MAX 4096
static bool g_prof = false;
static int g_prof_n = 0;
static hipEvent_t g_prof_ev[ST_PROF_MAX];
static const char* g_prof_name[ST_PROF_MAX];
static bool g_prof_init = false;
static void prof_mark(const char* name, void* stream)
{
    if (!g_prof || g_prof_n >= ST_PROF_MAX) return;
    (void)hipEventRecord(g_prof_ev[g_prof_n], st_stream(stream));
    g_prof_name[g_prof_n++] = name;
}
extern "C" int st_profile_enable(int on)
{
    if (on && !g_prof_init) {
        for (int i = 0; i < ST_PROF_MAX; ++i)
            if (hipEventCreate(&g_prof_ev[i]) != hipSuccess) return st_fail(ST_ERR_LAUNCH, "hipEventCreate");
        g_prof_init = true;
    }
    g_prof = on != 0; g_prof_n = 0;
    return ST_OK;
}
// Writes "name total_ms count\n" lines (aggregated by kernel name) into buf; caller must have synchronised the stream.
extern "C" int st_profile_report(char* buf, int buflen)
{
    if (!buf || buflen <= 0) return st_fail(ST_ERR_ARG, "st_profile_report: bad buffer");
    const char* names[64]; double tot[64]; int cnt[64]; int nn = 0;
    for (int i = 1; i < g_prof_n; ++i) {
        if (!strcmp(g_prof_name[i], "begin")) continue;
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, g_prof_ev[i - 1], g_prof_ev[i]) != hipSuccess) continue;
        int k = 0; for (; k < nn; ++k) if (!strcmp(names[k], g_prof_name[i])) break;
        if (k == nn) { if (nn == 64) continue; names[nn] = g_prof_name[i]; tot[nn] = 0; cnt[nn] = 0; ++nn; }
        tot[k] += ms; cnt[k]++;
    }
    int off = 0; buf[0] = 0;
    for (int k = 0; k < nn; ++k) off += snprintf(buf + off, off < buflen ? buflen - off : 0, "%s %.6f %d\n", names[k], tot[k], cnt[k]);
    g_prof_n = 0;
    return ST_OK;
}
extern "C" int st_version(void) { return 100; }
extern "C" int st_kp(int F) { return st_kp_of(F); }

// ------------------------------------------------------------------------------ geometry / layout
extern "C" int st_geometry(double scale_factor, double shrink_factor, int legacy, int K, int B, st_dims* o)
{
    ST_REQ(o && scale_factor > 0 && shrink_factor > 0, "st_geometry: bad arguments");
    const int chunk = (int)(8192 * scale_factor);              // nn_proc.py:357
    const int out_chunk = (int)(chunk / shrink_factor);        // nn_proc.py:358
    int ft = 1024, hop = 384;                                  // nn_proc.py:370-371
    if (legacy) { ft = (int)(ft * scale_factor); hop = (int)(hop * scale_factor); }   // nn_proc.py:374-376
    o->B = B; o->L = chunk; o->N = ft; o->H = hop; o->K = K;
    o->T = (int)(ceil(chunk / (double)hop) + ceil(ft / (double)hop));                 // nn_proc.py:378
    o->OT = (int)(ceil(out_chunk / (double)hop) + ceil(ft / (double)hop));            // nn_proc.py:379
    o->y = (o->OT - 1) * hop - ft;                                                    // nn_proc.py:380
    o->F = ft / 2 + 1;
    return ST_OK;
}

static int check_dims(const st_dims* d)
{
    ST_REQ(d, "null dims");
    ST_REQ(d->B > 0 && d->L > 0 && d->N > 0 && d->H > 0 && d->T > 0 && d->OT > 0 && d->K >= 0, "non-positive dimension");
    ST_REQ(d->F == d->N / 2 + 1, "F must be N/2+1");
    ST_REQ(d->N % 32 == 0 && d->H % 4 == 0 && d->L % 4 == 0 && d->y % 4 == 0, "N%%32, H%%4, L%%4, y%%4 required");
    ST_REQ(d->y == (d->OT - 1) * d->H - d->N && d->y > 0 && d->y <= d->L, "y must equal (OT-1)*H-N");
    ST_REQ(d->OT <= d->T, "OT must be <= T");
    ST_REQ(d->K <= 16, "at most 16 knobs");
    ST_REQ(d->prec >= ST_PREC_F32 && d->prec <= ST_PREC_F32X3, "st_dims.prec = %d is not an ST_PREC_* level", d->prec);
    ST_REQ(d->loss_scale >= 0.f && d->loss_scale <= 3.0e38f, "st_dims.loss_scale must be 0 (none) or a positive finite scale");
    // the kernels address every operand with 32-bit element offsets from a wave-uniform base and index rows with 24-bit
    // multiplies: rows (B*T) < 2^24, the largest per-batch operands (B*T x KP spectra, B x (L + 2N) padded signals) < 2^30 elements
    ST_REQ((size_t)d->B * d->T < ((size_t)1 << 24) && (size_t)d->B * d->T * st_kp_of(d->F) < ((size_t)1 << 30) &&
           (size_t)d->B * ((size_t)d->L + 2 * (size_t)d->N) < ((size_t)1 << 30), "batch too large for 32-bit element offsets (B=%d)", d->B);
    return ST_OK;
}

static void ae_shapes(const st_dims* d, int* out, int* in)
{
    const int o[9] = {64, 32, 16, 16, 16, 16, 32, 64, d->OT};
    const int i[9] = {d->T, 64, 32, 16, 16 + d->K, 16, 16, 32, 64};
    memcpy(out, o, sizeof(o)); memcpy(in, i, sizeof(i));
}

extern "C" int64_t st_param_offsets(const st_dims* d, int64_t* offs)
{
    if (check_dims(d) != ST_OK) return -1;
    int64_t off = 0; int n = 0;
    auto put = [&](int64_t sz) { if (offs) offs[n] = off; ++n; off += (sz + 3) / 4 * 4; };
    for (int s = 0; s < 4; ++s) put((int64_t)d->N * d->N);
    int out[9], in[9]; ae_shapes(d, out, in);
    for (int a = 0; a < 2; ++a)
        for (int l = 0; l < 9; ++l) { put((int64_t)out[l] * in[l]); put(out[l]); }
    return off;
}

struct Layout {            // everything derived from dims that the host side needs
    int64_t offs[40]; int64_t total; int64_t n_stft; int PG; sta::AEOffsets go; int KP;
};
static int make_layout(const st_dims* d, Layout* L)
{
    ST_TRY(check_dims(d));
    L->total = st_param_offsets(d, L->offs);
    L->n_stft = L->offs[4];
    L->PG = (int)(L->offs[22] - L->offs[4]);
    for (int l = 0; l < 9; ++l) {
        L->go.w[l] = (int)(L->offs[4 + 2 * l] - L->offs[4]);
        L->go.b[l] = (int)(L->offs[5 + 2 * l] - L->offs[4]);
    }
    L->KP = st_kp_of(d->F);
    return ST_OK;
}

// ------------------------------------------------------------------------------ launch parameters
static int num_cus()
{
    static int n = 0;
    if (!n) { int dev = 0; (void)hipGetDevice(&dev); hipDeviceProp_t p; if (hipGetDeviceProperties(&p, dev) == hipSuccess) n = p.multiProcessorCount; if (n <= 0) n = 256; }
    return n;
}
static int g_dbg = 0;   // timing-only ablation switches (bench/diagnostics); results are invalid when non-zero
extern "C" int st_set_debug(int v) { g_dbg = v; return ST_OK; }
namespace sta { extern __device__ unsigned long long g_ae_stage_cycles[32]; }
// Diagnostics: read (and clear) the per-stage s_memtime accumulators of ae_bwd_kernel (st_set_debug(256)).
extern "C" int st_debug_read_stage_cycles(unsigned long long* out32);
static int g_dp_inline = 1;  // st_dp_train_step: 1 = the last (exposed) exchange is issued in line on the compute stream (round 6), 0 = on the communicator stream between two hand-offs (st_set_tuning(8300 / 8301))
static int g_ae_save = 1;    // fused geometries, fp32 autoencoder layers: 1 = the forward kernel keeps the activations and the backward reads them (round 6), 0 = the backward recomputes them (st_set_tuning(8200 / 8201))
static int g_ae_split = -1;  // autoencoder backward of the fused geometries: 0 = the single kernel (st_set_tuning(8000)), 1 = the two kernels of st_ae_split.h (8001),
                             // -1 = by precision (8002, default): fp32 -> single (179.5 us against 87.0 + 92.4 us at B = 256 -- equal: the fp32 MFMA holds the vector ALUs, a partner
                             // wave has nothing to overlap with -- and the split moves 66 MB more per step: h4 / d a4 / tails hand-over); 16-bit Linear layers -> split (the
                             // matrix pipe is then a separate unit and two waves per SIMD overlap it with the ELU / conversion / transpose work: 102 -> 48 + 44 us at
                             // B = 256, 354 -> 152 + 138 us at B = 1024, bf16_all)
static int g_ae32 = 1;       // 16-bit autoencoder forward of the fused geometries on 32-row groups / v_mfma_f32_32x32x16 (st_ae32.h); 0 = the 16-row kernel of st_ae.h (st_set_tuning(8100 + n))
static int g_pl_bf16 = 0;    // ST_PREC_BF16*: analysis / frames GEMMs on the plane kernel with ONE plane (bf16 copies of the bases, k-chunk-major).  MEASURED SLOWER at B = 256
                             // (analysis 53.8 vs 49.8 us + 12 us for the copies): three MFMAs per 16-deep k-tile and barrier; needs a 64-deep tile   (st_set_tuning(9400 + n))
static int g_wg_split = 0;   // ST_PREC_F32X3: weight-gradient GEMMs on the in-kernel three-plane split instead of the fp32 MFMA kernel (see ST_GEMM_WG)   (st_set_tuning(9300 + n))
static int g_pl_dgrad = 0;   // ST_PREC_F32X3: synthesis data gradient on the plane kernel (measured slower than the fp32 MFMA kernel at B = 256: 57 vs 45 us)   (st_set_tuning(9200 + n))
static int g_pl_shape = 3;   // analysis plane GEMM tile (ST_PREC_F32X3): 0 = 4 waves x (32 x 96) [91.9 us], 1 = 2 waves x (64 x 96) [117], 2 = 4 waves x (64 x 96) [112],
                             // 3 = 8 waves x (32 x 96) = 256 x 96, one workgroup per CU: a quarter less L2 traffic at the same two waves per SIMD [88.1]   (st_set_tuning(9100 + n))
static int g_g16 = 1;        // 16-bit configurations: the fused step's GEMMs on pre-rounded 16-bit operands (st_gemm16.h); 0 = gemm_half_kernel on fp32 operands (st_set_tuning(9600), diagnostics)
static int g_g16_bk = 64;    // k-tile depth of its TN kernel (st_set_tuning(9632 / 9664))
static int g_g16_dma = 0;    // bit 0 / bit 1: its analysis forward / synthesis data-gradient GEMM on the LDS-DMA kernel gemm16_nt256_kernel (st_set_tuning(9690 + bits)).  MEASURED EQUAL to
                             // gemm16_nt_kernel (analysis forward, bf16, B = 256 / 1024: 34.2 / 138.6 us against 35.5 / 137.2): both deliver ~20 B/clk/CU from L2 to LDS, see st_gemm16.h
static int g_g16_abl = 0;    // TIMING ONLY (results invalid): analysis forward epilogue ablation, bit0 no mag/phs, bit1 no re/im, bit2 frame rows at 16-byte aligned (wrong) offsets (st_set_tuning(9680 + bits))
static int g_g16_split = 0;  // k-slices of its weight-gradient GEMMs (0: by residency; st_set_tuning(9700 + n))
static int g_nt128 = 1;      // fp32 synthesis frames / data-gradient GEMMs on the 128 x 128-tile NT kernel (st_gemm_tn.h); 0 = gemm_kernel<2, ...> (st_set_tuning(9950), diagnostics)
static int g_tn128 = 1;      // weight-gradient GEMMs on the 128 x 128-tile kernel (st_gemm_tn.h) where it applies; 0 = gemm_kernel<3, ...> (st_set_tuning(9500), diagnostics)
static int g_tn_bk = 32;     // its k-tile depth (st_set_tuning(9516 / 9532))
static int g_tn_fm = 3;      // round 5: frame-major reduction order + per-tile-column row ranges (structural zeros skipped) in the 128 x 128 weight-gradient GEMMs: bit 0 synthesis
                             // (25 % of its reduction rows are cropped taps: 41.7 -> 37.0 us), bit 1 analysis (7.6 %: 113.8 -> 109.5 us)   (st_set_tuning(9540 + bits), diagnostics)
static int g_g16_crop = 15;  // round 5, 16-bit operand pipeline: structural zeros skipped (frame-major rows): bit 0 frames GEMM (dead tile columns return), bit 1 data gradient (live taps per tile row),
                             // bit 2 / bit 3 synthesis / analysis weight gradient (reduction rows per tile column)   (st_set_tuning(9560 + bits), diagnostics)
static int g_frs_nt = 1;     // synthesis frames GEMM against the transposed fold (both operands K-contiguous); 0 = the k-major form (st_set_tuning(9000), diagnostics)
static int g_xt = 0;       // 1: M/N-contiguous operands staged k-quad-major (st_gemm.h XT; st_set_tuning(7001), diagnostics).  MEASURED SLOWER at B=256 although
                           // conflict-free with a third fewer LDS cycles: analysis wgrad 173 vs 145 us, synthesis frames 63 vs 60 us (16 more prefetch
                           // registers -> 4 instead of 4.5 waves per SIMD, and 16 v_mov per micro-tile): the k-major staging stays the default
static const int NORM_E_PARTIALS = 32;     // |g| partials of the autoencoder gradient range (st_dims.clip_all) when they come from l1_partial_kernel
static const int NORM_E_MAX = 4096;        // room for the per-block partials of the reducing kernels themselves (post_ae_kernel / wide_grad_finish_kernel)
static int g_wide_pair = 1;  // wide geometries, 16-bit: the layer-1 / layer-9 GEMMs of the two autoencoders as ONE launch each (gemm_half_pair_kernel); 0 = two launches (st_set_tuning(9960 + n), diagnostics)
static int g_wide_direct = 1; // wide geometries: the analysis epilogue writes the wide autoencoder path's feature-major input itself (no wide_in_kernel in the fused step); 0 = copy kernel (st_set_tuning(9970 + n), diagnostics)
static int g_wide_dvp = 1;   // wide geometries: layer-1 data gradient (+ polar backward) as one fused kernel; 0 = two GEMMs + polar_bwd (st_set_tuning(9900), diagnostics)
static int g_nt_mi = 0;     // fp32 NT x NT GEMMs with 64 x 96 wave tiles (MI = 2): 0 off; bit 0 analysis forward <4,16,2>, bit 1 <2,32,2>, bit 2 frames / dgrad <2,16,2>  (st_set_tuning(9800 + n), experiments)
static int g_an_bk = 32;   // k-tile depth of the analysis forward GEMM (see ST_GEMM_AN)
static int g_bk = 16;   // k-tile depth of the GEMM family (16: 36 KB LDS/WG -> 4 WGs/CU; 32: 64 KB -> 2 WGs/CU)
static int g_wg_mode_set(int v);
static int g_wsplit_max = 16, g_wsplit_div = 200, g_an_waves = 4, g_syn_split = 3, g_frs_split = 3, g_wide_fused = 1, g_wsplit_half = 0;      // frames: 3, 4 measured equal, 6 slower (only 66 k-tiles to split)
extern "C" int st_set_tuning(int bk)
{
#ifdef ST_DIAG      // timing-only ablations (results INVALID by construction): compiled only into diagnostic builds (make EXTRA=-DST_DIAG), never into the product library
    if (bk >= 96800 && bk < 96928) { g_g16_abl = bk - 96800; return ST_OK; }      // 16-bit analysis GEMM (bits 3..5: its k-loop)
#else
    if ((bk >= 96800 && bk < 96928) || (bk >= 9680 && bk < 9690)) return st_fail(ST_ERR_ARG, "st_set_tuning(%d): timing-only ablation, needs a -DST_DIAG build", bk);
#endif
    if (bk >= 9970 && bk < 9980) { g_wide_direct = bk - 9970; return ST_OK; }
    if (bk >= 9960) { g_wide_pair = bk - 9960; return ST_OK; }
    if (bk >= 9950 && bk < 9960) { g_nt128 = bk - 9950; return ST_OK; }
    if (bk >= 9900) { g_wide_dvp = bk - 9900; return ST_OK; }
    if (bk >= 9800) { g_nt_mi = bk - 9800; return ST_OK; }
    if (bk >= 9700) { g_g16_split = bk - 9700; return ST_OK; }
    if (bk >= 9690 && bk < 9700) { g_g16_dma = bk - 9690; return ST_OK; }
#ifdef ST_DIAG
    if (bk >= 9680 && bk < 9690) { g_g16_abl = bk - 9680; return ST_OK; }
#endif
    if (bk >= 9600) { const int v = bk - 9600; if (v == 32 || v == 64) g_g16_bk = v; else g_g16 = v; return ST_OK; }
    if (bk >= 9560 && bk < 9576) { g_g16_crop = bk - 9560; return ST_OK; }
    if (bk >= 9540 && bk < 9544) { g_tn_fm = bk - 9540; return ST_OK; }
    if (bk >= 9500) { const int v = bk - 9500; if (v == 16 || v == 32) g_tn_bk = v; else g_tn128 = v; return ST_OK; }
    if (bk >= 9400) { g_pl_bf16 = bk - 9400; return ST_OK; }
    if (bk >= 9300) { g_wg_split = bk - 9300; return ST_OK; }
    if (bk >= 9200) { g_pl_dgrad = bk - 9200; return ST_OK; }
    if (bk >= 9100) { g_pl_shape = bk - 9100; return ST_OK; }
    if (bk >= 9000) { g_frs_nt = bk - 9000; return ST_OK; }
    if (bk >= 8100 && bk < 8110) { g_ae32 = bk - 8100; return ST_OK; }           // 8100 / 8101: 16-bit autoencoder forward on 16-row / 32-row groups (st_ae32.h)
    if (bk == 8300 || bk == 8301) { g_dp_inline = bk - 8300; return ST_OK; }      // last exchange of the data-parallel step: communicator stream / in line
    if (bk == 8200 || bk == 8201) { g_ae_save = bk - 8200; return ST_OK; }      // autoencoder backward: recompute / read the kept activations
    if (bk >= 8000) { g_ae_split = bk == 8002 ? -1 : bk - 8000; return ST_OK; }     // 8000 / 8001 / 8002: single-kernel / split autoencoder backward / by precision
    if (bk >= 7000) { g_xt = bk - 7000; return ST_OK; }
    if (bk >= 6000) { g_wsplit_half = bk - 6000; return ST_OK; }  // 6000 + n: split-K of the half (one-basis) analysis weight-gradient GEMMs of st_loss_backward_stage (0: as the full GEMM)
    if (bk >= 5000) { g_wide_fused = bk - 5000; return ST_OK; }   // 5000 / 5001: wide AE path all-GEMM / fused inner layers
    if (bk >= 4000) { g_frs_split = bk - 4000; return (g_frs_split >= 1 && g_frs_split <= 6) ? ST_OK : st_fail(ST_ERR_ARG, "frames split must be 1..6"); }
    if (bk >= 3000) { g_syn_split = bk - 3000; return ST_OK; }    // 3000 + n: synthesis split-K (<= 3: consumers sum at most 3 slabs)
    if (bk >= 2000) { g_an_waves = bk - 2000; return ST_OK; }     // 2000 + n: waves per workgroup of the analysis forward GEMM
    if (bk >= 1000) { g_wsplit_div = bk - 1000; return ST_OK; }  // 1000 + n: rows per weight-gradient k-slice (diagnostics)
    if (bk >= 200) { g_wsplit_max = bk - 200; return ST_OK; }      // 200 + n: cap of the weight-gradient split-K (diagnostics)
    if (bk >= 100) return g_wg_mode_set(bk - 100);           // 100 / 101: weight-gradient tile mode (diagnostics)
    if (bk != 16 && bk != 32) return st_fail(ST_ERR_ARG, "bk must be 16 or 32"); g_bk = bk; g_an_bk = bk; return ST_OK;
}
// The diagnostic switches above as ONE readable state: st_get_tuning() reports them in a fixed order, st_reset_tuning() restores the shipped
// defaults.  The product path never sets them; tests/conftest.py asserts after every test that the state is back at ST_TUNING_DEFAULTS
// (a wrong default can then not ship unnoticed, and a test cannot leak a switch into the next one).
#define ST_TUNING_LIST(X) X(g_dbg, 0) X(g_ae_split, -1) X(g_ae_save, 1) X(g_dp_inline, 1) X(g_pl_bf16, 0) X(g_wg_split, 0) X(g_pl_dgrad, 0) X(g_pl_shape, 3) X(g_g16, 1) X(g_g16_bk, 64) X(g_g16_dma, 0) \
    X(g_g16_abl, 0) X(g_g16_split, 0) X(g_nt128, 1) X(g_tn128, 1) X(g_tn_bk, 32) X(g_frs_nt, 1) X(g_xt, 0) X(g_wide_pair, 1) X(g_wide_dvp, 1) X(g_nt_mi, 0) X(g_an_bk, 32) \
    X(g_bk, 16) X(g_wsplit_max, 16) X(g_wsplit_div, 200) X(g_an_waves, 4) X(g_syn_split, 3) X(g_frs_split, 3) X(g_wide_fused, 1) X(g_wsplit_half, 0) X(g_wg_mode, 0) X(g_ae32, 1) X(g_wide_direct, 1) X(g_tn_fm, 3) X(g_g16_crop, 15)
static int g_wg_mode = 0;
extern "C" int st_get_tuning(int* out, int n)
{
    int i = 0;
#define X(v_, d_) if (out && i < n) out[i] = v_; ++i;
    ST_TUNING_LIST(X)
#undef X
    return i;                      // number of switches (call with out = NULL to size the array)
}
extern "C" int st_tuning_defaults(int* out, int n)
{
    int i = 0;
#define X(v_, d_) if (out && i < n) out[i] = d_; ++i;
    ST_TUNING_LIST(X)
#undef X
    return i;
}
extern "C" int st_reset_tuning(void)
{
#define X(v_, d_) v_ = d_;
    ST_TUNING_LIST(X)
#undef X
    return ST_OK;
}
// Arithmetic of a call = st_dims::prec (ST_PREC_*; round 1 had a process-wide switch here, which raced between engines):
// half type (0 none / 1 bfloat16 / 2 float16) of the STFT GEMM operands and of the autoencoder layers.
// gemm_ht: 3 = fp32 operands as three bfloat16 planes (ST_PREC_F32X3; st_gemm.h gemm_half_kernel PL = 3)
static inline int gemm_ht(int prec) { return (prec == ST_PREC_BF16 || prec == ST_PREC_BF16_ALL) ? 1 : ((prec == ST_PREC_F16 || prec == ST_PREC_F16_ALL) ? 2 : (prec == ST_PREC_F32X3 ? 3 : 0)); }
static inline int ae_ht(int prec) { return prec == ST_PREC_BF16_ALL ? 1 : (prec == ST_PREC_F16_ALL ? 2 : 0); }
static inline float loss_scale_of(const st_dims* d) { return d->loss_scale > 0.f ? d->loss_scale : 1.0f; }
// every ST_GEMM* user has `d` (const st_dims*) in scope
#define ST_GEMM_BK(BK_, W_, ...) do { const int ht_ = gemm_ht(d->prec); \
                              if (ht_ == 1) stg::launch_half<W_, 1>(__VA_ARGS__); else if (ht_ == 2) stg::launch_half<W_, 2>(__VA_ARGS__); \
                              else if (ht_ == 3) { ST_TRY((stg::launch_half<W_, 1, 3>(__VA_ARGS__))); } \
                              else if (g_xt) { if ((BK_) == 16) stg::launch<W_, 16, 1, true>(__VA_ARGS__, g_dbg); else stg::launch<W_, 32, 1, true>(__VA_ARGS__, g_dbg); } \
                              else if ((BK_) == 16) stg::launch<W_, 16>(__VA_ARGS__, g_dbg); else stg::launch<W_, 32>(__VA_ARGS__, g_dbg); } while (0)
#define ST_GEMM(W_, ...) ST_GEMM_BK(g_bk, W_, __VA_ARGS__)
// the analysis forward GEMM (K = N = 1024, two 4-wave workgroups per CU either way) runs 5 % faster with 32-deep k-tiles
// (half the barriers); every other GEMM of the step is faster with 16 (more workgroups per CU)
#define ST_GEMM_AN(W_, ...) ST_GEMM_BK(g_an_bk, W_, __VA_ARGS__)
// weight-gradient GEMMs: g_wg_mode 0 = three waves share a 96x96 tile (32x96 strips), 1 = one wave per 96x96 tile
static int g_wg_mode_set(int v) { g_wg_mode = v; return ST_OK; }
// ST_PREC_F32X3: the weight-gradient GEMMs reduce along the ROWS of both operands (k-major staging, 4x4 register transposes); their
// in-kernel three-plane split measured slower than the fp32 MFMA kernel (176 vs 141 us, 62 vs 55 us at B = 256), so that precision
// level keeps them on the fp32 kernel (st_set_tuning(9301) selects the split form).
#define ST_GEMM_WG(...) do { if (g_wg_mode == 1 && gemm_ht(d->prec) == 0) stg::launch<1, 16, 3>(__VA_ARGS__, g_dbg); \
                             else if (g_wg_mode == 2 && gemm_ht(d->prec) == 0) stg::launch<3, 16, 1, true>(__VA_ARGS__, g_dbg); \
                             else if (gemm_ht(d->prec) == 3 && !g_wg_split) stg::launch<3, 16>(__VA_ARGS__, g_dbg); else ST_GEMM(3, __VA_ARGS__); } while (0)
// Kernels with > 64 KB of dynamic LDS need the attribute once per (device, kernel); the result is checked (round 1 discarded
// it behind non-atomic flags: a failure surfaced later as an opaque launch error).
static int ensure_dyn_lds(const void* fn, const char* name)
{
    static std::mutex mu; static std::set<std::pair<int, const void*>> done;
    int dev = 0; (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lk(mu);
    if (done.count({dev, fn})) return ST_OK;
    const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return st_fail(ST_ERR_LAUNCH, "hipFuncSetAttribute(%s, 160 KB dynamic LDS): %s", name, hipGetErrorString(e));
    done.insert({dev, fn});
    return ST_OK;
}
#define ST_DYN_LDS(kernel_) ST_TRY(ensure_dyn_lds(reinterpret_cast<const void*>(&kernel_), #kernel_))
static const int AE_FWD_NW = 8, AE_BWD_NW = 4;
static int synth_live_rows(const st_dims* d);
static bool ae_is_wide(const st_dims* d);
// Waves per workgroup of the fused forward kernel: 8, or 11 where that removes the tail round.  A wave walks whole 16-row groups,
// one workgroup per CU: at B = 256 there are 8448 = 256 * 33 groups -- 4.125 per wave with 8 waves (five rounds, the fifth 12 %
// full), exactly 3 per wave with 11 (2.75 waves per SIMD, the fp32 kernel's 152 registers allow 3).  Time goes with
// rounds x waves sharing a SIMD: 5 x 8 against 3 x 11.  The 16-bit instantiations (176 registers) stay at 8.
static int ae_fwd_groups(const st_dims* d) { return d->B * (st_kp_of(d->F) / 32); }
static int ae_fwd_nw(const st_dims* d)
{
    if (ae_is_wide(d) || ae_ht(d->prec) != 0) return AE_FWD_NW;
    const int groups = ae_fwd_groups(d), c = num_cus();
    auto cost = [&](int nw) { int grid = (groups + nw - 1) / nw; if (grid > c) grid = c; const int rounds = (groups + grid * nw - 1) / (grid * nw); return rounds * nw; };
    return cost(11) < cost(AE_FWD_NW) ? 11 : AE_FWD_NW;
}
// Wide path, layers 2..8 (ae_inner_fwd_kernel: 116-126 registers, up to four waves per SIMD): waves per workgroup out of {8, 9, 12} by
// rounds x waves sharing the busiest SIMD.  B = 64 at the 65536-sample window is 2112 = 256 * 8.25 groups: with 8 waves 64 of the 2048 waves walked a
// second group (two rounds for 3 % more work); 9 waves on 235 workgroups take one.
static int ae_inner_nw(const st_dims* d)
{
    const int groups = ae_fwd_groups(d), c = num_cus();
    int best = AE_FWD_NW, best_cost = 1 << 30;
    for (int nw : {8, 9, 12}) {
        int grid = (groups + nw - 1) / nw; if (grid > c) grid = c;
        const int rounds = (groups + grid * nw - 1) / (grid * nw), cost = rounds * ((nw + 3) / 4);
        if (cost < best_cost) { best_cost = cost; best = nw; }
    }
    return best;
}
static int ae_inner_grid(const st_dims* d) { const int nw = ae_inner_nw(d); int g = (ae_fwd_groups(d) + nw - 1) / nw; int c = num_cus(); return g < c ? g : c; }
static int ae_fwd_grid(const st_dims* d) { const int nw = ae_fwd_nw(d); int g = (ae_fwd_groups(d) + nw - 1) / nw; int c = num_cus(); return g < c ? g : c; }
static int ae_bwd_grid(const st_dims* d) { int groups = d->B * (st_kp_of(d->F) / 32); int g = (groups + AE_BWD_NW - 1) / AE_BWD_NW; int c = num_cus() / 2; if (c < 1) c = 1; return g < c ? g : c; }
// split-K factors.  fp32 MFMA tiles are long serial chains (48 MFMAs x 64 cycles per k-tile per wave), so a GEMM
// needs >= ~2 waves per SIMD (2048 waves) to overlap its load/LDS phases; the small-M synthesis GEMMs and the
// 121-tile weight-gradient GEMMs get there by splitting K and summing the slabs in the consumer kernel.
static int wgrad_split(int R) { int s = R / g_wsplit_div; if (s < 1) s = 1; if (s > g_wsplit_max) s = g_wsplit_max; return s; }
// ... and never more k-slices than keep all workgroups co-resident: the 3-wave weight-gradient workgroup (25 KB of LDS, 95 registers)
// fits six to a CU, and a second, partly filled round costs more than the shorter k-chains save (measured sawtooth at B = 256,
// 121 tiles: 10 / 12 / 16 slices -> 151 / 153 / 160 us incl. the slab reduce; 11, 13 -> 160, 169)
static int wgrad_split_tiles(int R, int M, int Nc)
{
    int s = wgrad_split(R);
    const int tiles = ((M + 95) / 96) * ((Nc + 95) / 96), slots = 6 * num_cus();
    int fit = slots / (tiles > 0 ? tiles : 1); if (fit < 1) fit = 1;
    if (fit >= 2) fit &= ~1;                       // even slice counts: odd ones leave a ragged last slice (k-slices are multiples of 32 rows)
    return s < fit ? s : fit;
}
static int synth_split(int R) { return R >= 4096 ? 1 : g_syn_split; }   // consumers (ola_loss_kernel, ae_bwd_kernel) sum at most 3 slabs

extern "C" int st_ae_fwd_partials(const st_dims* d) { return ae_fwd_grid(d) * ae_fwd_nw(d); }
extern "C" int st_ola_loss_partials(const st_dims* d) { return d->B * ((d->y + 255) / 256); }
extern "C" int st_norm_partials(const st_dims* d) { return stm::norm_partial_count(d->F, d->N); }
// k-slices of the 128 x 128-tile weight-gradient GEMM (st_gemm_tn.h): as many as fill the CUs with one workgroup each, never slices
// shorter than 64 reduction rows
static int tn_split(int R, int N)
{
    const int tiles = (N / 128) * (N / 128);
    int s = num_cus() / (tiles > 0 ? tiles : 1); if (s > 16) s = 16;
    const int cap = R / 64; if (s > cap) s = cap;
    return s < 1 ? 1 : s;
}
extern "C" size_t st_wgrad_ws_floats(const st_dims* d)
{
    int s = wgrad_split_tiles(d->B * d->T, st_kp_of(d->F), d->N);
    const int s2 = tn_split(d->B * d->T, d->N); if (s2 > s) s = s2;
    return (size_t)s * st_kp_of(d->F) * d->N + (size_t)64 * 2 * d->N;       // + the Nyquist partials of the 128 x 128-tile form
}
static size_t synth_wgrad_ws_floats(const st_dims* d) { return st_wgrad_ws_floats(d); }
// Round 6: a second slab area for the SYNTHESIS weight gradient alone.  In the data-parallel step its slabs are summed on the communicator stream beside the autoencoder
// backward; with a buffer of their own the analysis weight-gradient GEMM (which reuses the first area) needs no communicator -> compute wait before it starts -- one
// barrier packet (~6 us of bubble on this stack) less on the compute stream.
static size_t synth_wgrad_ws_floats(const st_dims* d);      // = st_wgrad_ws_floats(d): every slab-count rule of the fp32 and 16-bit weight-gradient launches is capped by that area's size
extern "C" int st_synth_slabs(const st_dims* d) { return synth_split(synth_live_rows(d)); }
// split-K slabs of the synthesis FRAMES GEMM (summed by ola_loss_kernel, which takes up to 6; the dgrad slabs are summed
// inside ae_bwd_kernel where every extra slab costs 8 loads per row group, hence the separate, smaller count above)
static int frames_split(int R) { return R >= 4096 ? 1 : g_frs_split; }
extern "C" int st_synth_frame_slabs(const st_dims* d) { return frames_split(synth_live_rows(d)); }
// Wide geometries (T > 32 or OT > 16) run the autoencoders as feature-major GEMMs (st_ae_wide.h) and need workspace
// for the activations [features][B*FP]; the fused kernels of st_ae.h need none in forward.
static bool ae_is_wide(const st_dims* d) { return d->T > 32 || d->OT > 16; }
struct WideWS {
    float *W1p[2], *W5p[2], *V[2], *H[2][8], *E9[2];         // forward: H[a][j] = output of layer j+1 (H[a][3] has 16 + K rows: [h4 ; knobs])
    float *DA[2][9], *TL[2], *slabs, *inner_ws;               // backward: dA_l, skip/residual tails, split-K slabs (one set per net), workgroup partials of the fused inner kernel
    // every layer-input buffer (V, H[.][j]) has ONE extra row (index = that layer's IN) of ones: bias gradient via the wgrad GEMM
    size_t fwd_floats, floats; int Tp, nsplit; size_t R, SL; int so[10];
};
static void wide_carve(const st_dims* d, float* base, WideWS* w)
{
    const int FP = st_kp_of(d->F) / 2;
    const size_t R = (size_t)d->B * FP;
    const int hrows[8] = {64, 32, 16, 16 + d->K, 16, 16, 32, 64};
    const int drows[9] = {64, 32, 16, 16, 16, 16, 32, 64, d->OT};
    int out[9], in[9]; ae_shapes(d, out, in);
    w->R = R; w->Tp = st_round_up(d->T, 32);      // 32: k-tile of the bf16 GEMM kernel (level-2 precision)
    { long ns = (long)(R / 128); w->nsplit = (int)(ns < 1 ? 1 : (ns > 256 ? 256 : ns)); }   // wgrad K = R: short k-chains on many workgroups
    w->so[0] = 0;
    for (int l = 0; l < 9; ++l) w->so[l + 1] = w->so[l] + st_round_up(out[l] * (in[l] + 1), 64);
    w->SL = (size_t)w->so[9];
    size_t off = 0;
    auto take = [&](size_t n) { float* p = base ? base + off : nullptr; off += (n + 63) / 64 * 64; return p; };
    for (int a = 0; a < 2; ++a) {
        w->W1p[a] = take((size_t)64 * w->Tp); w->W5p[a] = take(16 * 32);
        w->V[a] = take((size_t)(d->T + 1) * R);
        for (int j = 0; j < 8; ++j) w->H[a][j] = take((size_t)(hrows[j] + 1) * R);
        w->E9[a] = take((size_t)d->OT * R);
    }
    w->fwd_floats = off;
    for (int a = 0; a < 2; ++a) {
        for (int l = 0; l < 9; ++l) w->DA[a][l] = take((size_t)drows[l] * R);
        w->TL[a] = take((size_t)d->OT * R);
    }
    w->slabs = take((size_t)2 * w->nsplit * w->SL);
    { Layout L; if (make_layout(d, &L) == ST_OK) w->inner_ws = take((size_t)ae_bwd_grid(d) * 2 * L.PG); else w->inner_ws = nullptr; }
    w->floats = off;
}
// Fused geometries: the autoencoder workspace is [h4 exchange | d a4 exchange | workgroup gradient partials]; the first two are the
// 16-wide code of both nets and its gradient, [net][group][lane] float4 each, that connect the forward kernel and the two halves
// of the split backward (st_ae_split.h).  g_ae_split = 0 runs the single-kernel backward (st_set_tuning(8000), diagnostics).
static const int AE_SPLIT_NW = 8;
static size_t ae_h4_floats(const st_dims* d) { return (size_t)2 * ae_fwd_groups(d) * 256; }
static int ae_split_grid(const st_dims* d) { int g = (ae_fwd_groups(d) + AE_SPLIT_NW - 1) / AE_SPLIT_NW; int c = num_cus() / 2; if (c < 1) c = 1; return g < c ? g : c; }
// The split form pays where the MFMAs are long (fp32: measured 181 vs 188 us at B = 256); with 16-bit operands the matrix time is a
// few microseconds, the kernel is all instruction issue, and two kernels only add a second prologue (bf16_all: 128 vs 124 us): those
// precisions keep the single kernel.
// 16-bit operands in the split form were tried (decoder + encoder halves 49 + 45 us against 101 us for the single kernel, the step did not
// move) and are NOT instantiated: the compiler emitted a cross-block MFMA-result hazard in the 16-bit encoder half (tools/check_mfma_hazards.py).
static bool ae_use_split(const st_dims* d) { return (g_ae_split < 0 ? ae_ht(d->prec) != 0 : g_ae_split != 0) && !ae_is_wide(d) && !(g_dbg & 256); }
// Round 6: kept activations of the fused fp32 autoencoders ([net][group][17 tiles][64 lanes] float4, st_ae.h): the workspace always has room for them where the
// autoencoder layers run in fp32 (the tuning switch picks the kernels, not the size); they sit BEHIND the workgroup partials.
static size_t ae_sv_floats(const st_dims* d) { return (!ae_is_wide(d) && ae_ht(d->prec) == 0) ? (size_t)2 * ae_fwd_groups(d) * sta::AE_SV_TILES * 256 : 0; }
static int ae_parts_max(const st_dims* d) { return ae_split_grid(d) > ae_bwd_grid(d) ? ae_split_grid(d) : ae_bwd_grid(d); }
static bool ae_use_saved(const st_dims* d) { return g_ae_save && !ae_is_wide(d) && ae_ht(d->prec) == 0 && !ae_use_split(d); }
static float* ae_sv_ptr(const st_dims* d, const Layout& L, float* aews) { return aews + 2 * ae_h4_floats(d) + (size_t)ae_parts_max(d) * 2 * L.PG; }
extern "C" size_t st_ae_fwd_ws_floats(const st_dims* d)
{
    if (check_dims(d) != ST_OK) return 0;
    if (!ae_is_wide(d)) return ae_h4_floats(d);             // optional for the forward alone (ws may be NULL: h4 is then not kept)
    WideWS w; wide_carve(d, nullptr, &w); return w.fwd_floats;
}
extern "C" size_t st_ae_kept_activation_bytes(const st_dims* d)
{
    if (check_dims(d) != ST_OK || !ae_use_saved(d)) return 0;
    return ae_sv_floats(d) * sizeof(float);
}
extern "C" size_t st_ae_bwd_ws_floats(const st_dims* d)
{
    Layout L; if (make_layout(d, &L) != ST_OK) return 0;
    if (ae_is_wide(d)) { WideWS w; wide_carve(d, nullptr, &w); return w.floats; }
    return 2 * ae_h4_floats(d) + (size_t)ae_parts_max(d) * 2 * L.PG + ae_sv_floats(d);
}
static int ae_wide_fwd(const st_dims* d, const Layout& L, const float* mag, const float* phs, const float* knobs,
                       const float* ae_m, const float* ae_p, float* mag_hat, float* phs_hat, float* AA, float* reg_partial,
                       WideWS& w, void* stream, unsigned short* AA16 = nullptr, bool in_done = false);

// ------------------------------------------------------------------------------ per-op entry points
// Feature-major destination of the polar epilogue on wide geometries (st_gemm.h PolarStore::Vm / Vp): the wide autoencoder path's input layout
struct PolarWide { float* Vm; float* Vp; int FP; unsigned RV; };
static inline void polar_wide_set(stg::PolarStore& ep, const PolarWide* pw) { if (pw && pw->Vm) { ep.Vm = pw->Vm; ep.Vp = pw->Vp; ep.FP = pw->FP; ep.RV = pw->RV; } }
// `padded`: sig is the workspace copy [B][N + L + N] (zero margins, input scale applied) written by pad_scale_kernel.
static int analysis_fwd_impl(const st_dims* d, const float* sig, bool padded, const float* Wr, const float* Wi, float in_scale,
                             float* re, float* im, float* mag, float* phs, void* stream, bool dead_frames_done = false, const PolarWide* pw = nullptr)
{
    const stg::RowMap map = stg::live_frames(d->T, d->H, d->N, d->N, d->L);   // frames entirely inside the Conv1d padding are skipped
    const int R = map.rows(d->B);
    stg::AnalysisW bl{Wr, Wi, d->F, d->N};
    stg::PolarStore ep{re, im, mag, phs, R, d->F, map};
    polar_wide_set(ep, pw);
    if (padded) {
        stg::FramedNT<true> al{sig, d->L, d->H, d->N, R, d->N, 1.0f, map};
        if (gemm_ht(d->prec) == 0 && (g_nt_mi & 1)) stg::launch<4, 16, 2>(al, bl, ep, R, 2 * d->F, d->N, 1, st_stream(stream), g_dbg);
        else if (gemm_ht(d->prec) == 0 && (g_nt_mi & 2)) stg::launch<2, 32, 2>(al, bl, ep, R, 2 * d->F, d->N, 1, st_stream(stream), g_dbg);
        else if (g_an_waves == 2) ST_GEMM_AN(2, al, bl, ep, R, 2 * d->F, d->N, 1, st_stream(stream));
        else if (g_an_waves == 3) ST_GEMM_AN(3, al, bl, ep, R, 2 * d->F, d->N, 1, st_stream(stream));
        else ST_GEMM_AN(4, al, bl, ep, R, 2 * d->F, d->N, 1, st_stream(stream));
    }
    else { stg::FramedNT<false> al{sig, d->L, d->H, d->N, R, d->N, in_scale, map}; ST_GEMM(4, al, bl, ep, R, 2 * d->F, d->N, 1, st_stream(stream)); }
    ST_LAUNCHED("analysis_fwd");
    if (map.Tv < d->T && !dead_frames_done) {   // ... and are exact zeros (re=im=mag=0, phs=atan2(0,1e-7)=0)
        hipLaunchKernelGGL(stm::zero_dead_frames_kernel, dim3(d->B * (d->T - map.Tv)), dim3(256), 0, st_stream(stream),
                           re, im, mag, phs, d->T, d->F, map.t_lo, map.Tv);
        ST_LAUNCHED("zero_dead_frames");
    }
    return ST_OK;
}
extern "C" int st_analysis_fwd(const st_dims* d, const float* x, const float* Wr, const float* Wi, float in_scale,
                               float* re, float* im, float* mag, float* phs, void* stream)
{
    ST_TRY(check_dims(d)); ST_REQ(x && Wr && Wi, "st_analysis_fwd: null input");
    return analysis_fwd_impl(d, x, false, Wr, Wi, in_scale, re, im, mag, phs, stream);
}
static int pad_scale(const float* in, float* out, int B, int Ls, int pad, float s, void* stream)
{
    hipLaunchKernelGGL(stm::pad_scale_kernel, dim3(((Ls + 2 * pad) / 4 + 255) / 256, B), dim3(256), 0, st_stream(stream), in, out, Ls, pad, s);
    ST_LAUNCHED("pad_scale");
    return ST_OK;
}

// AA16 != NULL (fused step of the 16-bit GEMM configurations): the spectra are written rounded to the operand type, not as fp32
static int ae_fwd_impl(const st_dims* d, const float* mag, const float* phs, const float* knobs,
                       const float* ae_m, const float* ae_p, float* mag_hat, float* phs_hat, float* AA,
                       float* reg_partial, float* ws, void* stream, unsigned short* AA16 = nullptr, bool wide_in_done = false, float* sv = nullptr);
extern "C" int st_ae_fwd(const st_dims* d, const float* mag, const float* phs, const float* knobs,
                         const float* ae_m, const float* ae_p, float* mag_hat, float* phs_hat, float* AA,
                         float* reg_partial, float* ws, void* stream)
{
    return ae_fwd_impl(d, mag, phs, knobs, ae_m, ae_p, mag_hat, phs_hat, AA, reg_partial, ws, stream);
}
static int ae_fwd_impl(const st_dims* d, const float* mag, const float* phs, const float* knobs,
                       const float* ae_m, const float* ae_p, float* mag_hat, float* phs_hat, float* AA,
                       float* reg_partial, float* ws, void* stream, unsigned short* AA16, bool wide_in_done, float* sv)
{      // sv != NULL (fused fp32 geometries, training step): the activations are kept for the backward (ae_sv_floats)
    Layout L; ST_TRY(make_layout(d, &L));
    const int aa_ht = AA16 ? gemm_ht(d->prec) : 0;
    ST_REQ(mag && phs && (knobs || d->K == 0) && ae_m && ae_p && ((mag_hat && phs_hat && AA) || (!mag_hat && !phs_hat && !AA && ws)), "st_ae_fwd: null pointer");
    if (!knobs) knobs = ae_m;      // K == 0: the kernels issue one clamped, masked load of knobs[0] -- any resident float will do
    if (ae_is_wide(d)) {
        ST_REQ(mag_hat, "st_ae_fwd: the code-only pass exists for the fused geometries only");
        ST_REQ(ws, "st_ae_fwd: this geometry (T=%d, OT=%d) needs st_ae_fwd_ws_floats() floats of workspace", d->T, d->OT);
        WideWS w; wide_carve(d, ws, &w);
        return ae_wide_fwd(d, L, mag, phs, knobs, ae_m, ae_p, mag_hat, phs_hat, AA, reg_partial, w, stream, AA16, wide_in_done);
    }
    ST_REQ((size_t)d->B * d->T * d->F < ((size_t)1 << 30) && (size_t)d->B * d->OT * L.KP < ((size_t)1 << 30),
           "st_ae_fwd: batch too large for the kernel's 32-bit element offsets (B=%d)", d->B);
    const size_t lds = ((size_t)2 * sta::CL::FWD_TOTAL + (size_t)(L.KP / 2)) * sizeof(float);      // two forward images + the per-bin frequency weights
    const float expfac = (float)(7.0 / d->F);
#define ST_AE_FWD_LAUNCH(HT_) do { ST_DYN_LDS((sta::ae_fwd_kernel<AE_FWD_NW, HT_>)); \
        hipLaunchKernelGGL((sta::ae_fwd_kernel<AE_FWD_NW, HT_>), dim3(ae_fwd_grid(d)), dim3(AE_FWD_NW * 64), lds, st_stream(stream), \
                           mag, phs, knobs, ae_m, ae_p, L.go, mag_hat, phs_hat, AA, reg_partial, \
                           d->B, d->T, d->OT, d->F, d->K, L.KP, expfac, ws, AA16, aa_ht); } while (0)
    const bool use32 = g_ae32 && ae_ht(d->prec) != 0 && L.KP / 2 <= 17 * 32 && AE_FWD_NW == 8;
#define ST_AE_FWD32_LAUNCH(HT_) do { ST_DYN_LDS((sta::ae_fwd32_kernel<AE_FWD_NW, HT_>)); \
        hipLaunchKernelGGL((sta::ae_fwd32_kernel<AE_FWD_NW, HT_>), dim3(ae_fwd_grid(d)), dim3(AE_FWD_NW * 64), (size_t)sta::ae32_lds_floats(L.KP / 2) * sizeof(float), st_stream(stream), \
                           mag, phs, knobs, ae_m, ae_p, L.go, mag_hat, phs_hat, AA, reg_partial, \
                           d->B, d->T, d->OT, d->F, d->K, L.KP, expfac, ws, AA16, aa_ht); } while (0)
    switch (ae_ht(d->prec)) {
    case 1: if (use32) ST_AE_FWD32_LAUNCH(1); else ST_AE_FWD_LAUNCH(1); break;
    case 2: if (use32) ST_AE_FWD32_LAUNCH(2); else ST_AE_FWD_LAUNCH(2); break;
    default:
        if (sv) {
            ST_REQ(mag_hat, "st_ae_fwd: internal: kept activations on a code-only pass");
#define ST_AE_FWD_SV(NW_) do { ST_DYN_LDS((sta::ae_fwd_kernel<NW_, 0, true>)); \
            hipLaunchKernelGGL((sta::ae_fwd_kernel<NW_, 0, true>), dim3(ae_fwd_grid(d)), dim3(NW_ * 64), lds, st_stream(stream), \
                               mag, phs, knobs, ae_m, ae_p, L.go, mag_hat, phs_hat, AA, reg_partial, d->B, d->T, d->OT, d->F, d->K, L.KP, expfac, ws, AA16, aa_ht, sv); } while (0)
            if (ae_fwd_nw(d) == 11) ST_AE_FWD_SV(11); else ST_AE_FWD_SV(AE_FWD_NW);
#undef ST_AE_FWD_SV
        } else if (ae_fwd_nw(d) == 11) {
            ST_DYN_LDS((sta::ae_fwd_kernel<11, 0>));
            hipLaunchKernelGGL((sta::ae_fwd_kernel<11, 0>), dim3(ae_fwd_grid(d)), dim3(11 * 64), lds, st_stream(stream),
                               mag, phs, knobs, ae_m, ae_p, L.go, mag_hat, phs_hat, AA, reg_partial, d->B, d->T, d->OT, d->F, d->K, L.KP, expfac, ws, AA16, aa_ht);
        } else ST_AE_FWD_LAUNCH(0);
    }
#undef ST_AE_FWD_LAUNCH
#undef ST_AE_FWD32_LAUNCH
    ST_LAUNCHED("ae_fwd"); return ST_OK;
}

extern "C" int st_synth_fold(const st_dims* d, const float* Sr, const float* Si, float* Sfold, void* stream)
{
    ST_TRY(check_dims(d)); ST_REQ(Sr && Si && Sfold, "st_synth_fold: null pointer");
    const int KP = st_kp_of(d->F);
    hipLaunchKernelGGL(stm::fold_kernel, dim3(KP), dim3(256), 0, st_stream(stream), Sr, Si, Sfold, d->N, d->F, KP);
    ST_LAUNCHED("synth_fold"); return ST_OK;
}

static stg::RowMap synth_live(const st_dims* d) { return stg::live_frames(d->OT, d->H, d->N, d->N, d->y); }
static int synth_live_rows(const st_dims* d) { return synth_live(d).rows(d->B); }

// SfoldT != NULL: the transposed fold [N][KP] the fused forward builds (prep_kernel) -- both operands K-contiguous
static int synthesis_frames_impl(const st_dims* d, const float* AA, const float* Sfold, const float* SfoldT, float* frs, void* stream);
extern "C" int st_synthesis_frames(const st_dims* d, const float* AA, const float* Sfold, float* frs, void* stream)
{
    ST_TRY(check_dims(d)); ST_REQ(AA && Sfold && frs, "st_synthesis_frames: null pointer");
    return synthesis_frames_impl(d, AA, Sfold, nullptr, frs, stream);
}
// Fraction of the synthesis frames' taps that the crop of cls_fe_dft.py:113 throws away (the structural zeros the work list skips): 25 % at the 8192-sample window
// (7 live frames), 4 % at the 65536-sample window (44 live frames)
static double synth_crop_fraction(const st_dims* d)
{
    const stg::RowMap ms = synth_live(d);
    long live = 0;
    for (int t = ms.t_lo; t < ms.t_lo + ms.Tv; ++t) { int lo, hi; stg::ntw_live_taps(t, t, d->H, d->N, d->N, d->y, lo, hi); live += hi - lo; }
    return 1.0 - (double)live / ((double)ms.Tv * (double)d->N);
}
// the frame-major row order splits a compact row r < R = frames * B into (frame, window) with a multiply-high by the reciprocal of B
// (RowMap::magic, rowmap_magic()), which is exact only while r * B < 2^32.  Every switch that turns the frame-major order on asks this first; past the bound
// (B = 32768 at 7 live frames) the launch keeps the window-major order, i.e. the untrimmed / uncropped kernels of round 4.
static inline bool fm_div_exact(int R, int B) { return R >= 0 && B >= 1 && (unsigned long long)(unsigned)R * (unsigned long long)(unsigned)B < (1ull << 32); }
extern "C" int st_fm_div_exact(int R, int B) { return fm_div_exact(R, B) ? 1 : 0; }
// the 128 x 128-tile work-list kernel (st_gemm_tn.h): fp32 products, up to 255 tile rows.  Used where one workgroup per CU covers the whole GEMM in one round (the rule
// of rounds 3-4: small batches at any geometry) or where the crop is worth the frame-major row order (>= 10 % of the taps: every geometry of the 8192-sample window) --
// several rounds are then fine (B = 512: 61 / 69 us against 107 / 107 on gemm_kernel<2, ...>).  MEASURED the other way at the 65536-sample window (B = 64, 4 % cropped, 510 workgroups
// in two rounds): 80 / 77 us against 64 / 65 -- a tile's 128 rows there are 64 windows 194 KB apart; that geometry stays on the small-tile kernel.  Whether a list fits (and with
// how many k-slices) is the builder's decision (ntw_frames / ntw_dgrad return false -> gemm_kernel<4, ...> / <2, ...>).
static bool use_nt128(const st_dims* d, int M, int Nc, int ncus = 0)      // a pure function of (d, ncus) once ncus > 0 (st_nt128_worklist's override: ADVICE round 5)
{
    if (!(g_nt128 && gemm_ht(d->prec) == 0 && d->N % 32 == 0 && (M + 127) / 128 <= 255 && fm_div_exact(M, d->B))) return false;
    if (ncus <= 0) ncus = num_cus();
    return ((M + 127) / 128) * ((Nc + 127) / 128) * 2 <= ncus || synth_crop_fraction(d) >= 0.10;
}
static int synthesis_frames_impl(const st_dims* d, const float* AA, const float* Sfold, const float* SfoldT, float* frs, void* stream)
{
    const int KP = st_kp_of(d->F);
    const stg::RowMap ms = synth_live(d);      // output frames that land wholly in the cropped margins are never needed
    const int R = ms.rows(d->B);
    stg::PlainNT al{AA, R, KP, KP, ms};
    stg::PlainTN bl{Sfold, KP, d->N, d->N, stg::all_frames(1)};
    // frs holds st_synth_frame_slabs() split-K slabs [B*OT, N]; st_ola_loss sums them
    stg::StoreC ep{frs, R, d->N, d->N, (size_t)d->B * d->OT * d->N, ms};
    if (SfoldT && g_frs_nt) {
        stg::PlainNT bt{SfoldT, d->N, KP, KP, stg::all_frames(1)};
        stg::NTWork wk;
        if (use_nt128(d, R, d->N) && stg::ntw_frames(wk, ms, d->B, d->H, d->N, d->N, d->y, KP, frames_split(R), num_cus())) {
            // frame-major rows; only the tile columns whose taps survive the crop (cls_fe_dft.py:113): the others are never read by ola_loss_kernel
            const stg::NTRows ra = stg::ntrows_frame_major(AA, (unsigned)(d->OT * KP), (unsigned)KP, ms, d->B, R), rb{SfoldT, (unsigned)KP, 0u, 0u, 1, 0, d->N};
            const stg::StoreSlab es{frs, R, d->N, d->N, (size_t)d->B * d->OT * d->N, stg::frame_major(ms, d->B)};
            ST_TRY(stg::launch_nt128(ra, rb, es, wk, st_stream(stream)));
        }
        else if (R >= 4096) ST_GEMM(4, al, bt, ep, R, d->N, KP, 1, st_stream(stream));
        else if (gemm_ht(d->prec) == 0 && (g_nt_mi & 4)) stg::launch<2, 16, 2>(al, bt, ep, R, d->N, KP, frames_split(R), st_stream(stream), g_dbg);
        else ST_GEMM(2, al, bt, ep, R, d->N, KP, frames_split(R), st_stream(stream));
    }
    else if (R >= 4096) ST_GEMM(4, al, bl, ep, R, d->N, KP, 1, st_stream(stream));
    else ST_GEMM(2, al, bl, ep, R, d->N, KP, frames_split(R), st_stream(stream));
    ST_LAUNCHED("synthesis_frames"); return ST_OK;
}

// Host-side view of the work list of the 128 x 128-tile synthesis GEMMs (st_gemm_tn.h, round 5): which = 0 the frames GEMM, 1 the data gradient.
// out[i] = the packed entry i (tile row (8 bits) | tile column (6) << 8 | slab (2) << 14 | first zero-filled slab (2; 0 = none) << 16 | kind (1) << 18 | first k unit (6) << 19 |
// k units (7) << 25); head6 = {slabs, col_h, col_stride, frame-major windows per frame, k-tiles (of 32) per k unit, k-tiles of the whole reduction}.  Returns the number of
// entries (= workgroups), 0 when this geometry does not run on the work-list kernel, < 0 on bad arguments.  No device work: the CPU test-suite checks the list against the
// cropping rule of cls_fe_dft.py:113.
extern "C" int st_nt128_worklist(const st_dims* d, int which, int ncus, unsigned* out, int cap, int* head6)
{
    if (check_dims(d) != ST_OK || !out || !head6 || cap < 0 || (which != 0 && which != 1)) return -1;
    const int KP = st_kp_of(d->F);
    const stg::RowMap ms = synth_live(d);
    const int R = ms.rows(d->B);
    if (ncus <= 0) ncus = num_cus();
    stg::NTWork wk; wk.n = 0;
    if (!use_nt128(d, R, which ? KP : d->N, ncus)) return 0;
    const bool ok = which ? stg::ntw_dgrad(wk, ms, d->B, d->H, d->N, d->N, d->y, d->F, KP, R >= 4096 ? 1 : synth_split(R), ncus)
                          : stg::ntw_frames(wk, ms, d->B, d->H, d->N, d->N, d->y, KP, frames_split(R), ncus);
    if (!ok) return 0;
    head6[0] = wk.nslabs; head6[1] = wk.col_h; head6[2] = wk.col_stride; head6[3] = d->B; head6[4] = wk.kunit; head6[5] = wk.kt_total;
    for (int i = 0; i < wk.n && i < cap; ++i) out[i] = wk.e[i];
    return wk.n;
}

static int ola_loss_impl(const st_dims* d, const float* frs, const float* x, const float* y_true,
                         float* y_hat, float* dsyn, int dsyn_pad, float* loss_partial, void* stream, unsigned short* dsyn16 = nullptr)
{
    const float inv = loss_scale_of(d) / ((float)d->B * (float)d->y);     // d loss / d y_hat, times the loss scale (train.py:134-135)
    {   // four samples per thread (round 5) where the geometry and the pointers allow 16-byte accesses: always inside the fused step
        const int ns = st_synth_frame_slabs(d), nslot = (d->y + 255) / 256;
        auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
        const bool ok4 = d->H % 4 == 0 && d->N % 4 == 0 && d->y % 4 == 0 && d->L % 4 == 0 && dsyn_pad % 4 == 0 && ns >= 1 && ns <= 3 && (d->N + d->H - 1) / d->H <= 3 &&
                         al16(frs) && al16(x) && al16(y_true) && al16(y_hat) && al16(dsyn) && al16(dsyn16) && ((size_t)d->B * d->OT * d->N) % 4 == 0;
        if (ok4) {
            const dim3 grid((d->y / 4 + 255) / 256, d->B);
            const size_t slab = (size_t)d->B * d->OT * d->N;
            unsigned short* d16 = dsyn ? dsyn16 : nullptr; const int ht = dsyn16 ? gemm_ht(d->prec) : 0;
#define ST_OLA4(NS_) hipLaunchKernelGGL((stm::ola_loss4_kernel<NS_>), grid, dim3(256), 0, st_stream(stream), frs, x, y_true, y_hat, dsyn, loss_partial, d->L, d->N, d->H, d->OT, d->y, inv, slab, dsyn_pad, nslot, d16, ht)
            if (ns == 3) ST_OLA4(3); else if (ns == 2) ST_OLA4(2); else ST_OLA4(1);
#undef ST_OLA4
            ST_LAUNCHED("ola_loss");
            return ST_OK;
        }
    }
    hipLaunchKernelGGL(stm::ola_loss_kernel, dim3((d->y + 255) / 256, d->B), dim3(256), 0, st_stream(stream),
                       frs, x, y_true, y_hat, dsyn, loss_partial, d->L, d->N, d->H, d->OT, d->y, inv,
                       st_synth_frame_slabs(d), (size_t)d->B * d->OT * d->N, dsyn_pad, dsyn ? dsyn16 : nullptr, dsyn16 ? gemm_ht(d->prec) : 0);
    ST_LAUNCHED("ola_loss");
    return ST_OK;
}
extern "C" int st_ola_loss(const st_dims* d, const float* frs, const float* x, const float* y_true,
                           float* y_hat, float* dsyn, float* loss_partial, void* stream)
{
    ST_TRY(check_dims(d)); ST_REQ(frs, "st_ola_loss: null pointer");
    return ola_loss_impl(d, frs, x, y_true, y_hat, dsyn, 0, loss_partial, stream);
}

static int synthesis_dgrad_impl(const st_dims* d, const float* dsyn, bool padded, const float* Sfold, float* dAA, void* stream)
{
    const int KP = st_kp_of(d->F);
    const stg::RowMap ms = synth_live(d);
    const int R = ms.rows(d->B);
    // dfrs[b,t,n] = dfull[b, H t + n] with dfull = zero-pad(dsyn, N each side)  == frames of dsyn with pad N.
    // Rows of dAA for dead frames are NOT written (st_ae_bwd treats them as zero).
    // dAA holds st_synth_slabs() split-K slabs [B*OT, KP]; st_ae_bwd sums them
    stg::PlainNT bl{Sfold, KP, d->N, d->N, stg::all_frames(1)};
    stg::StoreC ep{dAA, R, KP, KP, (size_t)d->B * d->OT * KP, ms};
    const int ns = R >= 4096 ? 1 : synth_split(R);
    if (padded) {
        stg::FramedNT<true> al{dsyn, d->y, d->H, d->N, R, d->N, 1.0f, ms};
        stg::NTWork wk;
        if (use_nt128(d, R, KP) && stg::ntw_dgrad(wk, ms, d->B, d->H, d->N, d->N, d->y, d->F, KP, ns, num_cus())) {
            // frame-major rows, per tile row only the taps that lie inside d syn (the margins of the padded copy are zeros), Nyquist columns apart
            const stg::NTRows ra = stg::ntrows_frame_major(dsyn, (unsigned)(d->y + 2 * d->N), (unsigned)d->H, ms, d->B, R);
            const stg::NTRows rb = wk.col_h ? stg::NTRows{Sfold, (unsigned)((KP / 2) * d->N), (unsigned)d->N, stg::rowmap_magic(d->F - 1), d->F - 1, 0, 2 * (d->F - 1)}
                                            : stg::NTRows{Sfold, (unsigned)d->N, 0u, 0u, 1, 0, KP};
            const stg::StoreSlab es{dAA, R, KP, KP, (size_t)d->B * d->OT * KP, stg::frame_major(ms, d->B)};
            ST_TRY(stg::launch_nt128(ra, rb, es, wk, st_stream(stream)));
        }
        else if (R >= 4096) ST_GEMM(4, al, bl, ep, R, KP, d->N, ns, st_stream(stream));
        else if (gemm_ht(d->prec) == 0 && (g_nt_mi & 4)) stg::launch<2, 16, 2>(al, bl, ep, R, KP, d->N, ns, st_stream(stream), g_dbg);
        else ST_GEMM(2, al, bl, ep, R, KP, d->N, ns, st_stream(stream));
    } else {
        stg::FramedNT<false> al{dsyn, d->y, d->H, d->N, R, d->N, 1.0f, ms};
        if (R >= 4096) ST_GEMM(4, al, bl, ep, R, KP, d->N, ns, st_stream(stream)); else ST_GEMM(2, al, bl, ep, R, KP, d->N, ns, st_stream(stream));
    }
    ST_LAUNCHED("synthesis_dgrad");
    return ST_OK;
}
extern "C" int st_synthesis_dgrad(const st_dims* d, const float* dsyn, const float* Sfold, float* dAA, void* stream)
{
    ST_TRY(check_dims(d)); ST_REQ(dsyn && Sfold && dAA, "st_synthesis_dgrad: null pointer");
    return synthesis_dgrad_impl(d, dsyn, false, Sfold, dAA, stream);
}

// The 128 x 128-tile form (st_gemm_tn.h): fp32 products (ST_PREC_F32, and ST_PREC_F32X3 whose weight gradients stay on the fp32 MFMA), operands
// in the padded workspace layout, N a multiple of 256 (so that the F - 1 non-Nyquist bins of each basis are whole 128-row tiles)
static bool use_tn128(const st_dims* d, bool padded)
{
    const int ht = gemm_ht(d->prec);
    return g_tn128 && padded && (ht == 0 || (ht == 3 && !g_wg_split)) && d->N % 256 == 0 && g_wg_mode == 0;
}
// Frame-major reduction order (round 5): k -> frame t_lo + k / B, window k % B.  The operands keep their layout; only the strides swap roles and the
// origin moves to frame t_lo.  B >= 2 (the kernel divides by the inner count with a multiply-high); B = 1 keeps the window-major order (identical rows).
struct TNFrameMajor { stg::TNOperand a, b; stg::RowMap map; stg::FrameTrim trim; bool on; };
static TNFrameMajor tn_frame_major(const stg::TNOperand& ta, const stg::TNOperand& tb, const stg::RowMap& live, int B, int H, int Ntaps, int pad, int Ls, bool enable)
{
    TNFrameMajor f; f.on = enable && B >= 2 && fm_div_exact(live.Tv * B, B); f.a = ta; f.b = tb; f.map = live; f.trim = stg::FrameTrim{}; f.trim.on = 0;
    if (!f.on) return f;
    f.a = stg::TNOperand{ta.base + (size_t)live.t_lo * ta.S2, ta.S2, ta.S1};
    f.b = stg::TNOperand{tb.base + (size_t)live.t_lo * tb.S2, tb.S2, tb.S1};
    f.map.Tv = B; f.map.t_lo = 0; f.map.magic = stg::rowmap_magic(B); f.map.fm = 0;      // the kernel's (outer, inner) = (frame, window)
    f.trim = stg::frame_trim(live, B, H, Ntaps, pad, Ls);
    return f;
}
static int wgrad_tn128(const st_dims* d, const stg::TNOperand& ta, const stg::TNOperand& tb, const float* zeros, const stg::RowMap& map, int R,
                       float* ws, int ns, stm::NyqJob* nyq, void* stream, const stg::FrameTrim* trim)
{
    const int KP = st_kp_of(d->F);
    const int mh = (d->N / 2) / 128;
    // Nyquist partials [P][2][N] sit behind the ns slabs (st_wgrad_ws_floats reserves the room)
    float* part = ws + (size_t)ns * KP * d->N;
    int P = 0;
    const unsigned c0 = (unsigned)(d->F - 1), c1 = (unsigned)(KP / 2 + d->F - 1);
    if (g_tn_bk == 16) ST_TRY((stg::launch_tn128<16>(ta, tb, zeros, map, R, d->N, mh, (unsigned)(KP / 2), d->N, ws, d->N, (size_t)KP * d->N, ns, st_stream(stream), part, c0, c1, &P, trim)));
    else ST_TRY((stg::launch_tn128<32>(ta, tb, zeros, map, R, d->N, mh, (unsigned)(KP / 2), d->N, ws, d->N, (size_t)KP * d->N, ns, st_stream(stream), part, c0, c1, &P, trim)));
    nyq->part = part; nyq->P = P; nyq->on = 1;
    return ST_OK;
}
static int synthesis_wgrad_impl(const st_dims* d, const float* AA, const float* dsyn, bool padded, float* ws,
                                float* gSr, float* gSi, float* norm_partial, void* stream, int* defer_slabs = nullptr, stm::NyqJob* defer_nyq = nullptr)
{   // defer_slabs: the caller sums the slabs itself (post_ae_kernel); receives the slab count (and the Nyquist job)
    const int KP = st_kp_of(d->F);
    const stg::RowMap ms = synth_live(d);
    const int R = ms.rows(d->B);
    int ns = wgrad_split_tiles(R, KP, d->N);
    stm::NyqJob nyq{}; nyq.on = 0;
    const stg::TNOperand ta{AA, (unsigned)(d->OT * KP), (unsigned)KP}, tb{dsyn, (unsigned)(d->y + 2 * d->N), (unsigned)d->H};
    if (use_tn128(d, padded) && stg::tn128_fits(ta, tb, dsyn, ms, d->N, d->N, (size_t)d->B * d->OT * KP, (size_t)d->B * (d->y + 2 * d->N))) {
        ns = tn_split(R, d->N);
        const TNFrameMajor fm = tn_frame_major(ta, tb, ms, d->B, d->H, d->N, d->N, d->y, (g_tn_fm & 1) != 0);      // taps of d syn's frames outside the crop are zeros: skipped per tile column
        ST_TRY(wgrad_tn128(d, fm.a, fm.b, dsyn, fm.map, R, ws, ns, &nyq, stream, &fm.trim));
    } else {
        stg::PlainTN al{AA, R, KP, KP, ms};
        stg::StoreC ep{ws, KP, d->N, d->N, (size_t)KP * d->N, stg::all_frames(1)};
        if (padded) { stg::FramedTN<true> bl{dsyn, d->y, d->H, d->N, R, d->N, 1.0f, ms}; ST_GEMM_WG(al, bl, ep, KP, d->N, R, ns, st_stream(stream)); }
        else { stg::FramedTN<false> bl{dsyn, d->y, d->H, d->N, R, d->N, 1.0f, ms}; ST_GEMM_WG(al, bl, ep, KP, d->N, R, ns, st_stream(stream)); }
    }
    ST_LAUNCHED("synthesis_wgrad");
    if (defer_slabs) { *defer_slabs = ns; if (defer_nyq) *defer_nyq = nyq; return ST_OK; }
    hipLaunchKernelGGL(stm::wgrad_reduce_kernel, dim3(st_norm_partials(d)), dim3(256), 0, st_stream(stream),
                       ws, ns, gSr, gSi, norm_partial, d->N, d->F, KP, 1, 0, 2 * d->F, (float*)nullptr, nyq);
    ST_LAUNCHED("synthesis_wgrad_reduce");
    return ST_OK;
}
extern "C" int st_synthesis_wgrad(const st_dims* d, const float* AA, const float* dsyn, float* ws,
                                  float* gSr, float* gSi, float* norm_partial, void* stream)
{
    ST_TRY(check_dims(d)); ST_REQ(AA && dsyn && ws && gSr && gSi && norm_partial, "st_synthesis_wgrad: null pointer");
    return synthesis_wgrad_impl(d, AA, dsyn, false, ws, gSr, gSi, norm_partial, stream);
}


// ------------------------------------------------------------------------------ wide-geometry autoencoders (st_ae_wide.h)
// BM = 64, k-tile 16 (every K below is a multiple of 16 or checked).  Level-2 precision: the 16-bit kernel (k-tile 32: W1 is padded to a multiple of 32
// columns).  The weight-gradient GEMMs reduce over the R = B * 528 columns: a multiple of 32 for even batches; for ODD batches (R = 16 mod 32) they run
// the same kernel on 16-deep k-tiles (round 5) -- until round 4 an odd batch dropped the whole wide path to fp32 layers (and said so: st_effective_prec),
// the last place where the arithmetic of a call depended on its batch size.  wide_ht, a local of every user of ST_WGEMM: 0 fp32 / 1 bf16 / 2 fp16.
static inline int wide_half_type(const st_dims* d, int R) { (void)R; return ae_ht(d->prec); }
// The arithmetic a call REALLY runs.  Since round 5 that is the request for every geometry and batch; the entry stays so that callers (and the tests) can
// keep asserting it instead of assuming it.
extern "C" int st_effective_prec(const st_dims* d)
{
    if (!d) return -1;
    if (ae_is_wide(d) && ae_ht(d->prec) && wide_half_type(d, d->B * (st_kp_of(d->F) / 2)) == 0)
        return d->prec == ST_PREC_BF16_ALL ? ST_PREC_BF16 : ST_PREC_F16;
    return d->prec;
}
#define ST_WGEMM(...) do { if (wide_ht == 1) stg::launch_half<2, 1>(__VA_ARGS__); else if (wide_ht == 2) stg::launch_half<2, 2>(__VA_ARGS__); \
                           else stg::launch<2, 16>(__VA_ARGS__, g_dbg); } while (0)
// the same GEMM for both autoencoders: ONE launch in the 16-bit configurations (gemm_half_pair_kernel), two on the fp32 kernel
#define ST_WGEMM_PAIR(A0_, B0_, E0_, A1_, B1_, E1_, M_, N_, K_, NS_, S_) do { \
        if (wide_ht == 1 && g_wide_pair) stg::launch_half_pair<2, 1>(A0_, B0_, E0_, A1_, B1_, E1_, M_, N_, K_, NS_, S_); \
        else if (wide_ht == 2 && g_wide_pair) stg::launch_half_pair<2, 2>(A0_, B0_, E0_, A1_, B1_, E1_, M_, N_, K_, NS_, S_); \
        else { ST_WGEMM(A0_, B0_, E0_, M_, N_, K_, NS_, S_); ST_WGEMM(A1_, B1_, E1_, M_, N_, K_, NS_, S_); } } while (0)
static int ae_wide_fwd(const st_dims* d, const Layout& L, const float* mag, const float* phs, const float* knobs,
                       const float* ae_m, const float* ae_p, float* mag_hat, float* phs_hat, float* AA, float* reg_partial,
                       WideWS& w, void* stream, unsigned short* AA16, bool in_done)
{
    hipStream_t s = st_stream(stream);
    const int FP = L.KP / 2, F = d->F, T = d->T, OT = d->OT, R = (int)w.R, Tp = w.Tp;
    const int wide_ht = wide_half_type(d, R);
    ST_REQ(w.R * (size_t)(T > 64 ? T : 64) < ((size_t)1 << 30), "wide autoencoder path: batch too large (B=%d)", d->B);
    const stg::RowMap id = stg::all_frames(1);
    int out[9], in[9]; ae_shapes(d, out, in);
    if (!in_done) {      // the fused step's analysis GEMM wrote V itself and prep_kernel did the side jobs (round 4); the per-op entry copies here
        stw::PadJobs pj; int blk = 0;
        for (int a = 0; a < 2; ++a) {
            const float* ae = a ? ae_p : ae_m;
            pj.src[2 * a] = ae + L.go.w[0]; pj.dst[2 * a] = w.W1p[a]; pj.rows[2 * a] = 64; pj.cols[2 * a] = T; pj.pitch[2 * a] = Tp;
            pj.blk0[2 * a] = blk; blk += (64 * Tp + 255) / 256;
            pj.src[2 * a + 1] = ae + L.go.w[4]; pj.dst[2 * a + 1] = w.W5p[a]; pj.rows[2 * a + 1] = 16; pj.cols[2 * a + 1] = 16 + d->K; pj.pitch[2 * a + 1] = 32;
            pj.blk0[2 * a + 1] = blk; blk += 2;
        }
        pj.blk0[4] = blk;
        ST_REQ(FP % 4 == 0 && (size_t)(T + d->K) * d->B * (FP / 4) < ((size_t)1 << 31), "wide autoencoder path: batch too large (B=%d)", d->B);
        const int n_copy = (int)(((size_t)(T + d->K) * d->B * (FP / 4) + 255) / 256);
        hipLaunchKernelGGL(stw::wide_in_kernel, dim3(n_copy + blk), dim3(256), 0, s, mag, phs, knobs, w.V[0], w.V[1], w.H[0][3], w.H[1][3],
                           d->B, T, F, FP, d->K, n_copy, pj);
        ST_LAUNCHED("ae_wide_in");
    }
    // layers 1..8: GEMM for layer 1 (K = T); layers 2..8 either one fused kernel for both nets (default) or seven more GEMMs
    if (g_wide_fused) {
        stg::PlainNT al0{w.W1p[0], out[0], Tp, Tp, id}, al1{w.W1p[1], out[0], Tp, Tp, id};
        stg::PlainTN bl0{w.V[0], in[0], R, R, id}, bl1{w.V[1], in[0], R, R, id};
        stw::ActStore ep0{w.H[0][0], ae_m + L.go.b[0], out[0], R, FP, F}, ep1{w.H[1][0], ae_p + L.go.b[0], out[0], R, FP, F};
        ST_WGEMM_PAIR(al0, bl0, ep0, al1, bl1, ep1, out[0], R, Tp, 1, s);
    }
    else for (int a = 0; a < 2; ++a) {
        const float* ae = a ? ae_p : ae_m;
        for (int l = 0; l < 8; ++l) {
            const float* Wl = l == 0 ? w.W1p[a] : (l == 4 ? w.W5p[a] : ae + L.go.w[l]);
            const int kp = l == 0 ? Tp : (l == 4 ? 32 : in[l]);                       // padded reduction length = row pitch of Wl
            const float* Hin = l == 0 ? w.V[a] : w.H[a][l - 1];
            stg::PlainNT al{Wl, out[l], kp, kp, id};
            stg::PlainTN bl{Hin, in[l], R, R, id};
            stw::ActStore ep{w.H[a][l], ae + L.go.b[l], out[l], R, FP, F};
            ST_WGEMM(al, bl, ep, out[l], R, kp, 1, s);
        }
    }
    if (g_wide_fused) {
        const size_t lds = (size_t)2 * sta::CL::FWD_TOTAL * sizeof(float);
#define ST_AE_INNER_FWD_(NW_, HT_) do { ST_DYN_LDS((sta::ae_inner_fwd_kernel<NW_, HT_>)); \
            hipLaunchKernelGGL((sta::ae_inner_fwd_kernel<NW_, HT_>), dim3(ae_inner_grid(d)), dim3(NW_ * 64), lds, s, \
                               w.H[0][0], w.H[1][0], knobs, ae_m, ae_p, L.go, w.H[0][7], w.H[1][7], d->B, F, d->K, L.KP); } while (0)
#define ST_AE_INNER_FWD(HT_) do { const int nw_ = ae_inner_nw(d); if (nw_ == 9) ST_AE_INNER_FWD_(9, HT_); else if (nw_ == 12) ST_AE_INNER_FWD_(12, HT_); else ST_AE_INNER_FWD_(AE_FWD_NW, HT_); } while (0)
        switch (wide_ht) { case 1: ST_AE_INNER_FWD(1); break; case 2: ST_AE_INNER_FWD(2); break; default: ST_AE_INNER_FWD(0); }
#undef ST_AE_INNER_FWD
#undef ST_AE_INNER_FWD_
    }
    {
        stg::PlainNT al0{ae_m + L.go.w[8], OT, 64, 64, id}, al1{ae_p + L.go.w[8], OT, 64, 64, id};
        stg::PlainTN bl0{w.H[0][7], 64, R, R, id}, bl1{w.H[1][7], 64, R, R, id};
        stw::OutStore ep0{w.E9[0], mag_hat, w.V[0] + (size_t)(T - OT) * R, ae_m + L.go.b[8], OT, R, FP, F, 0};
        stw::OutStore ep1{w.E9[1], phs_hat, w.V[1] + (size_t)(T - OT) * R, ae_p + L.go.b[8], OT, R, FP, F, 1};
        ST_WGEMM_PAIR(al0, bl0, ep0, al1, bl1, ep1, OT, R, 64, 1, s);
    }
    ST_LAUNCHED("ae_wide_fwd");
    if (AA) {
        const float expfac = (float)(7.0 / d->F);
        hipLaunchKernelGGL(stw::wide_polar_out_kernel, dim3(st_ae_fwd_partials(d)), dim3(256), 0, s, mag_hat, phs_hat, AA, reg_partial,
                           d->B, OT, F, FP, L.KP, expfac, AA16, AA16 ? gemm_ht(d->prec) : 0);
        ST_LAUNCHED("ae_wide_polar_out");
    }
    return ST_OK;
}

// one weight-gradient GEMM of the wide path: dW_l (+ bias via the ones row) as split-K slabs of net a
static void wide_wgrad(const st_dims* d, WideWS& w, int a, int l, const int* out, const int* in, hipStream_t s, const int wide_ht)
{
    const int R = (int)w.R;
    const stg::RowMap id = stg::all_frames(1);
    const float* Hin = l == 0 ? w.V[a] : w.H[a][l - 1];
    stg::PlainNT al{w.DA[a][l], out[l], R, R, id};
    stg::PlainNT bl{Hin, in[l] + 1, R, R, id};
    stg::StoreC ep{w.slabs + (size_t)a * w.nsplit * w.SL + w.so[l], out[l], in[l] + 1, in[l] + 1, w.SL, id};
    if (wide_ht && R % 32) {      // odd batch: the reduction length is 16 mod 32 -> 16-deep k-tiles of the same kernel
        if (wide_ht == 1) stg::launch_half<2, 1, 1, 16>(al, bl, ep, out[l], in[l] + 1, R, w.nsplit, s); else stg::launch_half<2, 2, 1, 16>(al, bl, ep, out[l], in[l] + 1, R, w.nsplit, s);
        return;
    }
    ST_WGEMM(al, bl, ep, out[l], in[l] + 1, R, w.nsplit, s);
}

// ... of both nets in one launch: net 1's slabs follow net 0's (slab z of the pair launch = net * nsplit + slice), so both epilogues share one origin
static void wide_wgrad_pair(const st_dims* d, WideWS& w, int l, const int* out, const int* in, hipStream_t s, const int wide_ht)
{
    const int R = (int)w.R;
    const stg::RowMap id = stg::all_frames(1);
    stg::PlainNT al0{w.DA[0][l], out[l], R, R, id}, al1{w.DA[1][l], out[l], R, R, id};
    stg::PlainNT bl0{l == 0 ? w.V[0] : w.H[0][l - 1], in[l] + 1, R, R, id}, bl1{l == 0 ? w.V[1] : w.H[1][l - 1], in[l] + 1, R, R, id};
    stg::StoreC ep0{w.slabs + w.so[l], out[l], in[l] + 1, in[l] + 1, w.SL, id};
    stg::StoreC ep1 = ep0;
    if (!(g_wide_pair && wide_ht)) ep1.out = w.slabs + (size_t)w.nsplit * w.SL + w.so[l];      // two launches: each with its own z = 0 .. nsplit - 1
    if (wide_ht && R % 32) {      // odd batch: 16-deep k-tiles (see wide_wgrad)
        if (g_wide_pair) {
            if (wide_ht == 1) stg::launch_half_pair<2, 1, 1, 16>(al0, bl0, ep0, al1, bl1, ep1, out[l], in[l] + 1, R, w.nsplit, s);
            else stg::launch_half_pair<2, 2, 1, 16>(al0, bl0, ep0, al1, bl1, ep1, out[l], in[l] + 1, R, w.nsplit, s);
        } else if (wide_ht == 1) { stg::launch_half<2, 1, 1, 16>(al0, bl0, ep0, out[l], in[l] + 1, R, w.nsplit, s); stg::launch_half<2, 1, 1, 16>(al1, bl1, ep1, out[l], in[l] + 1, R, w.nsplit, s); }
        else { stg::launch_half<2, 2, 1, 16>(al0, bl0, ep0, out[l], in[l] + 1, R, w.nsplit, s); stg::launch_half<2, 2, 1, 16>(al1, bl1, ep1, out[l], in[l] + 1, R, w.nsplit, s); }
        return;
    }
    ST_WGEMM_PAIR(al0, bl0, ep0, al1, bl1, ep1, out[l], in[l] + 1, R, w.nsplit, s);
}

// Where the gradient w.r.t. the autoencoder inputs goes on the fused path: straight through the polar backward into d G (wide_dv_polar_kernel)
struct PolarSink { const float* re; const float* im; const float* g_mag; float* dG; unsigned short* dG16; };
static int ae_wide_bwd(const st_dims* d, const Layout& L, const float* mag, const float* phs, const float* knobs,
                       const float* ae_m, const float* ae_p, const float* mag_hat, const float* phs_hat, const float* dAA,
                       const float* g_mag_hat, float reg_coef, float* dmag, float* dphs, WideWS& w, float* g_m, float* g_p,
                       bool have_fwd, void* stream, const PolarSink* sink = nullptr, bool* sink_used = nullptr,
                       const stw::SynReduce* syn = nullptr, bool* syn_done = nullptr, float* norm_e = nullptr, int* n_norm_e = nullptr)
{
    hipStream_t s = st_stream(stream);
    const int FP = L.KP / 2, F = d->F, T = d->T, OT = d->OT, R = (int)w.R, Tp = w.Tp;
    const int wide_ht = wide_half_type(d, R);
    const stg::RowMap id = stg::all_frames(1);
    int out[9], in[9]; ae_shapes(d, out, in);
    // forward state (activations + ELU outputs of layer 9): recomputed into the workspace unless the fused step's own
    // forward just left it there (same workspace, same layout); no user-visible outputs
    if (!have_fwd) ST_TRY(ae_wide_fwd(d, L, mag, phs, knobs, ae_m, ae_p, nullptr, nullptr, nullptr, nullptr, w, stream));
    stw::OnesRows rows;
    for (int a = 0; a < 2; ++a) {
        rows.p[9 * a] = w.V[a] + (size_t)in[0] * R;
        for (int j = 0; j < 8; ++j) rows.p[9 * a + 1 + j] = w.H[a][j] + (size_t)in[j + 1] * R;
    }
    const float expfac = (float)(7.0 / d->F);
    const stg::RowMap ms = synth_live(d);
    {
        const size_t n = (size_t)d->B * OT * FP;
        ST_REQ(n < ((size_t)1 << 30), "wide autoencoder path: batch too large (B=%d)", d->B);
        int grid = (int)((n + 255) / 256); if (grid > 8192) grid = 8192;
        hipLaunchKernelGGL(stw::wide_dout_kernel, dim3(grid + 18 * d->B), dim3(256), 0, s, dAA, st_synth_slabs(d), (size_t)d->B * OT * L.KP,
                           mag_hat, phs_hat, w.E9[0], w.E9[1], w.V[0] + (size_t)(T - OT) * R, g_mag_hat, reg_coef, expfac,
                           w.DA[0][8], w.DA[1][8], w.TL[0], w.TL[1], d->B, OT, F, FP, L.KP, ms.t_lo, ms.t_lo + ms.Tv - 1, grid, rows);
        ST_LAUNCHED("ae_wide_dout");
    }
    // data gradient through W_l into dA_{l-1} (layer 5: only the 16 code columns; the knobs take no gradient)
    auto dgrad = [&](int a, int l, bool plain) {
        const float* ae = a ? ae_p : ae_m;
        const float* Wl = l == 0 ? w.W1p[a] : (l == 4 ? w.W5p[a] : ae + L.go.w[l]);
        const int pitch = l == 0 ? Tp : (l == 4 ? 32 : in[l]);
        const int m = l == 0 ? T : (l == 4 ? 16 : in[l]);
        stg::PlainTN al{Wl, out[l], pitch, m, id};
        stg::PlainTN bl{w.DA[a][l], out[l], R, R, id};
        if (l == 0) { stw::DvStore ep{a ? dphs : dmag, w.TL[a], T, OT, R, FP, F}; ST_WGEMM(al, bl, ep, T, R, out[l], 1, s); }
        else if (plain) { stg::StoreC ep{w.DA[a][l - 1], m, R, R, 0, id}; ST_WGEMM(al, bl, ep, m, R, out[l], 1, s); }     // dH only: the fused kernel applies ELU'
        else { stw::DgradStore ep{w.DA[a][l - 1], w.H[a][l - 1], m, R}; ST_WGEMM(al, bl, ep, m, R, out[l], 1, s); }
    };
    stw::GradTab tab;
    for (int l = 0; l < 9; ++l) { tab.so[l] = w.so[l]; tab.out[l] = out[l]; tab.in[l] = in[l]; tab.gw[l] = L.go.w[l]; tab.gb[l] = L.go.b[l]; }
    tab.so[9] = w.so[9];
    int inner_parts = 0;
    if (g_wide_fused) {
        // layer 9 as GEMMs, layers 8..2 in one fused kernel (both nets), layer 1 as GEMMs
        wide_wgrad_pair(d, w, 8, out, in, s, wide_ht);
        {   // d H8 = W9^T d A9 of both nets (the fused kernel applies ELU')
            stg::PlainTN al0{ae_m + L.go.w[8], out[8], in[8], in[8], id}, al1{ae_p + L.go.w[8], out[8], in[8], in[8], id};
            stg::PlainTN bl0{w.DA[0][8], out[8], R, R, id}, bl1{w.DA[1][8], out[8], R, R, id};
            stg::StoreC ep0{w.DA[0][7], in[8], R, R, 0, id}, ep1{w.DA[1][7], in[8], R, R, 0, id};
            ST_WGEMM_PAIR(al0, bl0, ep0, al1, bl1, ep1, in[8], R, out[8], 1, s);
        }
        {
            const size_t lds = (size_t)sta::ae_bwd_lds_floats(AE_BWD_NW) * sizeof(float);
            const int grid = ae_bwd_grid(d);
#define ST_AE_INNER_BWD(HT_) do { ST_DYN_LDS((sta::ae_bwd_kernel<AE_BWD_NW, false, true, HT_, 0>)); \
                hipLaunchKernelGGL((sta::ae_bwd_kernel<AE_BWD_NW, false, true, HT_, 0>), dim3(grid, 2), dim3(AE_BWD_NW * 64), lds, s, \
                                   (const float*)w.H[0][0], (const float*)w.H[1][0], knobs, ae_m, ae_p, L.go, L.PG, \
                                   (const float*)w.DA[0][7], (const float*)w.DA[1][7], (const float*)nullptr, (const float*)nullptr, 0.f, 0.f, \
                                   w.DA[0][0], w.DA[1][0], w.inner_ws, d->B, T, OT, F, d->K, L.KP, 0, 0, 1, (size_t)0, 0); } while (0)
            switch (wide_ht) { case 1: ST_AE_INNER_BWD(1); break; case 2: ST_AE_INNER_BWD(2); break; default: ST_AE_INNER_BWD(0); }
#undef ST_AE_INNER_BWD
            inner_parts = grid;                          // summed by the second role of wide_grad_finish_kernel below (was a launch of its own)
        }
        for (int l = 1; l < 8; ++l) tab.out[l] = 0;                 // the finish kernel only scatters layers 1 and 9
        wide_wgrad_pair(d, w, 0, out, in, s, wide_ht);
        if (!g_wide_dvp) for (int a = 0; a < 2; ++a) dgrad(a, 0, false);
        if (norm_e && 2 * ((w.so[9] + 63) / 64 + (L.PG + 63) / 64) > NORM_E_MAX) norm_e = nullptr;
        hipLaunchKernelGGL(stw::wide_grad_finish_kernel, dim3((w.so[9] + 63) / 64 + (L.PG + 63) / 64 + (syn ? st_norm_partials(d) : 0), 2), dim3(256), 0, s, w.slabs, w.nsplit, w.SL, tab, g_m, g_p,
                           (w.so[9] + 63) / 64, (const float*)w.inner_ws, inner_parts, L.PG, syn ? *syn : stw::SynReduce{}, norm_e);
        if (syn && syn_done) *syn_done = true;
        if (norm_e && n_norm_e) *n_norm_e = 2 * ((w.so[9] + 63) / 64 + (L.PG + 63) / 64);
        if (g_wide_dvp) {
            stw::DvPolarArgs q;
            q.DA1m = w.DA[0][0]; q.DA1p = w.DA[1][0]; q.TLm = w.TL[0]; q.TLp = w.TL[1]; q.W1m = ae_m + L.go.w[0]; q.W1p = ae_p + L.go.w[0];
            q.re = sink ? sink->re : nullptr; q.im = sink ? sink->im : nullptr; q.g_mag = sink ? sink->g_mag : nullptr;
            q.dG = sink ? sink->dG : nullptr; q.dG16 = sink ? sink->dG16 : nullptr;
            q.dmag = sink ? nullptr : dmag; q.dphs = sink ? nullptr : dphs;          // fused step: d mag / d phs never leave the kernel
            q.ht = gemm_ht(d->prec) <= 2 ? gemm_ht(d->prec) : 0; q.sat = gemm_ht(d->prec) == 2 ? 65504.0f : 0.0f;
            q.B = d->B; q.T = T; q.OT = OT; q.F = F; q.FP = FP; q.KP = L.KP;
            const int TP16 = (T + 15) / 16 * 16, groups = d->B * (FP / 16);
            ST_REQ(TP16 <= 192, "wide autoencoder path: T = %d frames exceeds the layer-1 data-gradient kernel's 192", T);
            const size_t lds = (size_t)2 * (wide_ht ? 32 : 64) * TP16 * sizeof(float);          // 16-bit images take half the room
            int grid = ((groups + 7) / 8) * (TP16 / 16); if (grid > num_cus()) grid = num_cus();      // units of (8 adjacent groups, 16-frame tile), shared equally; one workgroup per CU
                                                                                                      // (two or three per CU measured slower: 146.6 / 155.5 us against 141.1 for the whole
                                                                                                      // wide backward -- every workgroup stages the layer-1 weights, ~10 us of dependent loads)
#define ST_DVP(HT_) do { ST_DYN_LDS((stw::wide_dv_polar_kernel<HT_>)); hipLaunchKernelGGL((stw::wide_dv_polar_kernel<HT_>), dim3(grid), dim3(512), lds, s, q); } while (0)
            switch (wide_ht) { case 1: ST_DVP(1); break; case 2: ST_DVP(2); break; default: ST_DVP(0); }
#undef ST_DVP
            if (sink_used) *sink_used = sink != nullptr;
        }
    } else {
        for (int a = 0; a < 2; ++a) {
            for (int l = 8; l >= 0; --l) { wide_wgrad(d, w, a, l, out, in, s, wide_ht); dgrad(a, l, false); }
            hipLaunchKernelGGL(stw::wide_grad_finish_kernel, dim3((w.so[9] + 63) / 64, 1), dim3(256), 0, s,
                               w.slabs + (size_t)a * w.nsplit * w.SL, w.nsplit, w.SL, tab, a ? g_p : g_m, a ? g_p : g_m);
        }
    }
    ST_LAUNCHED("ae_wide_bwd");
    return ST_OK;
}
#undef ST_WGEMM

static int ae_bwd_impl(const st_dims* d, const float* mag, const float* phs, const float* knobs,
                       const float* ae_m, const float* ae_p, const float* mag_hat, const float* phs_hat,
                       const float* dAA, const float* g_mag_hat, float reg_coef, float* dmag, float* dphs, float* ws,
                       float* g_m, float* g_p, bool have_fwd, void* stream, bool* defer_reduce = nullptr,
                       const PolarSink* sink = nullptr, bool* sink_used = nullptr, const stw::SynReduce* syn = nullptr, bool* syn_done = nullptr,
                       float* norm_e = nullptr, int* n_norm_e = nullptr)
{
    // defer_reduce: in -> the caller will sum the workgroup partials itself (post_ae_kernel, together with the polar backward);
    // out -> false if this geometry's path already reduced them (wide geometries)
    Layout L; ST_TRY(make_layout(d, &L));
    ST_REQ(mag && phs && (knobs || d->K == 0) && ae_m && ae_p && mag_hat && phs_hat && dAA && dmag && dphs && ws && g_m && g_p, "st_ae_bwd: null pointer");
    if (!knobs) knobs = ae_m;
    if (ae_is_wide(d)) {
        if (defer_reduce) *defer_reduce = false;
        WideWS w; wide_carve(d, ws, &w);
        return ae_wide_bwd(d, L, mag, phs, knobs, ae_m, ae_p, mag_hat, phs_hat, dAA, g_mag_hat, reg_coef, dmag, dphs, w, g_m, g_p, have_fwd, stream, sink, sink_used, syn, syn_done, norm_e, n_norm_e);
    }
    const size_t lds = (size_t)sta::ae_bwd_lds_floats(AE_BWD_NW) * sizeof(float);
    static_assert((size_t)sta::ae_bwd_lds_floats(AE_BWD_NW) * sizeof(float) <= 160 * 1024, "ae_bwd LDS budget");
    ST_REQ((size_t)st_synth_slabs(d) * d->B * d->OT * L.KP < ((size_t)1 << 30) && (size_t)d->B * d->T * L.KP < ((size_t)1 << 30),
           "st_ae_bwd: batch too large for the kernel's 32-bit element offsets (B=%d)", d->B);
    const float expfac = (float)(7.0 / d->F);
    const stg::RowMap live = synth_live(d);
    float* h4x = ws;                                      // workspace: [h4 | d a4 | workgroup partials] (st_ae_bwd_ws_floats)
    float* da4x = ws + ae_h4_floats(d);
    float* parts = ws + 2 * ae_h4_floats(d);
    int grid = ae_bwd_grid(d);
    if (ae_use_split(d)) {
        // two kernels at two waves per SIMD (st_ae_split.h): decoder half (layers 5..9, from the code h4 the forward kernel kept),
        // then encoder half (layers 1..4, from d a4 and the tails the first one left)
        if (!have_fwd) {                                   // per-op entry without a preceding forward in this workspace: code-only forward pass
            st_dims dd = *d; dd.prec = d->prec;
            ST_TRY(st_ae_fwd(&dd, mag, phs, knobs, ae_m, ae_p, nullptr, nullptr, nullptr, nullptr, h4x, stream));
        }
        grid = ae_split_grid(d);
        static_assert((size_t)sta::ae_split_lds_floats<1>(AE_SPLIT_NW) * sizeof(float) <= 160 * 1024 && (size_t)sta::ae_split_lds_floats<2>(AE_SPLIT_NW) * sizeof(float) <= 160 * 1024, "split ae_bwd LDS budget");
#define ST_AE_PART(PART_, HT_, GM_) do { ST_DYN_LDS((sta::ae_bwd_part_kernel<AE_SPLIT_NW, PART_, HT_, GM_>)); \
        hipLaunchKernelGGL((sta::ae_bwd_part_kernel<AE_SPLIT_NW, PART_, HT_, GM_>), dim3(grid, 2), dim3(AE_SPLIT_NW * 64), \
                           (size_t)sta::ae_split_lds_floats<PART_>(AE_SPLIT_NW) * sizeof(float), st_stream(stream), \
                           mag, phs, knobs, ae_m, ae_p, L.go, L.PG, mag_hat, phs_hat, dAA, g_mag_hat, reg_coef, expfac, dmag, dphs, parts, \
                           (const float*)h4x, da4x, d->B, d->T, d->OT, d->F, d->K, L.KP, live.t_lo, live.t_lo + live.Tv - 1, st_synth_slabs(d), \
                           (size_t)d->B * d->OT * L.KP); } while (0)
#define ST_AE_PARTS(HT_) do { if (g_mag_hat) ST_AE_PART(1, HT_, true); else ST_AE_PART(1, HT_, false); ST_LAUNCHED("ae_bwd_dec"); ST_AE_PART(2, HT_, false); ST_LAUNCHED("ae_bwd_enc"); } while (0)
        switch (ae_ht(d->prec)) { case 1: ST_AE_PARTS(1); break; case 2: ST_AE_PARTS(2); break; default: ST_AE_PARTS(0); }
#undef ST_AE_PARTS
#undef ST_AE_PART
        if (defer_reduce && *defer_reduce) return ST_OK;
        hipLaunchKernelGGL(stm::ae_grad_reduce_kernel, dim3((L.PG + 63) / 64, 2), dim3(256), 0, st_stream(stream), parts, grid, L.PG, g_m, g_p);
        ST_LAUNCHED("ae_grad_reduce"); return ST_OK;
    }
#define ST_AE_BWD_LAUNCH(TIMED_, HT_, VAR_) do { ST_DYN_LDS((sta::ae_bwd_kernel<AE_BWD_NW, TIMED_, false, HT_, VAR_>)); \
    hipLaunchKernelGGL((sta::ae_bwd_kernel<AE_BWD_NW, TIMED_, false, HT_, VAR_>), dim3(grid, 2), dim3(AE_BWD_NW * 64), lds, st_stream(stream), \
                       mag, phs, knobs, ae_m, ae_p, L.go, L.PG, mag_hat, phs_hat, dAA, g_mag_hat, reg_coef, expfac, \
                       dmag, dphs, parts, d->B, d->T, d->OT, d->F, d->K, L.KP, live.t_lo, live.t_lo + live.Tv - 1, st_synth_slabs(d), (size_t)d->B * d->OT * L.KP, g_dbg); } while (0)
    // kernel variant: 1 = an upstream d/d mag_hat arrives (autograd entry), 2 = T - OT == 16 (tails already in registers), 0 = neither
    const int var = g_mag_hat ? 1 : (d->T - d->OT == 16 ? 2 : 0);
    if (have_fwd && ae_use_saved(d)) {               // round 6: the forward of this workspace kept the activations -- no recompute
        const float* sv = ae_sv_ptr(d, L, ws);
#define ST_AE_BWD_SV(VAR_, TIMED_) do { ST_DYN_LDS((sta::ae_bwd_kernel<AE_BWD_NW, TIMED_, false, 0, VAR_, true>)); \
        hipLaunchKernelGGL((sta::ae_bwd_kernel<AE_BWD_NW, TIMED_, false, 0, VAR_, true>), dim3(grid, 2), dim3(AE_BWD_NW * 64), lds, st_stream(stream), \
                       mag, phs, knobs, ae_m, ae_p, L.go, L.PG, mag_hat, phs_hat, dAA, g_mag_hat, reg_coef, expfac, \
                       dmag, dphs, parts, d->B, d->T, d->OT, d->F, d->K, L.KP, live.t_lo, live.t_lo + live.Tv - 1, st_synth_slabs(d), (size_t)d->B * d->OT * L.KP, g_dbg, sv); } while (0)
        if (g_dbg & 256) { ST_REQ(var == 2, "the stage-timer build covers the fp32 training step at T - OT == 16 only"); ST_AE_BWD_SV(2, true); }      // tools/ae_stage_times.py
        else if (var == 1) ST_AE_BWD_SV(1, false); else if (var == 2) ST_AE_BWD_SV(2, false); else ST_AE_BWD_SV(0, false);
#undef ST_AE_BWD_SV
        ST_LAUNCHED("ae_bwd");
        if (defer_reduce && *defer_reduce) return ST_OK;
        hipLaunchKernelGGL(stm::ae_grad_reduce_kernel, dim3((L.PG + 63) / 64, 2), dim3(256), 0, st_stream(stream), parts, grid, L.PG, g_m, g_p);
        ST_LAUNCHED("ae_grad_reduce"); return ST_OK;
    }
#define ST_AE_BWD_VARS(HT_) do { if (var == 1) ST_AE_BWD_LAUNCH(false, HT_, 1); else if (var == 2) ST_AE_BWD_LAUNCH(false, HT_, 2); else ST_AE_BWD_LAUNCH(false, HT_, 0); } while (0)
    if (g_dbg & 256) {                               // stage-timer build (tools/ae_stage_times.py): the default-geometry training variant
        ST_REQ(var == 2 && ae_ht(d->prec) == 0, "the stage-timer build covers the fp32 training step at T - OT == 16 only");
        ST_AE_BWD_LAUNCH(true, 0, 2);
    }
    else switch (ae_ht(d->prec)) { case 1: ST_AE_BWD_VARS(1); break; case 2: ST_AE_BWD_VARS(2); break; default: ST_AE_BWD_VARS(0); }
#undef ST_AE_BWD_VARS
#undef ST_AE_BWD_LAUNCH
    ST_LAUNCHED("ae_bwd");
    if (defer_reduce && *defer_reduce) return ST_OK;
    hipLaunchKernelGGL(stm::ae_grad_reduce_kernel, dim3((L.PG + 63) / 64, 2), dim3(256), 0, st_stream(stream),
                       parts, grid, L.PG, g_m, g_p);
    ST_LAUNCHED("ae_grad_reduce"); return ST_OK;
}
extern "C" int st_ae_bwd(const st_dims* d, const float* mag, const float* phs, const float* knobs,
                         const float* ae_m, const float* ae_p, const float* mag_hat, const float* phs_hat,
                         const float* dAA, const float* g_mag_hat, float reg_coef, float* dmag, float* dphs, float* ws,
                         float* g_m, float* g_p, void* stream)
{
    return ae_bwd_impl(d, mag, phs, knobs, ae_m, ae_p, mag_hat, phs_hat, dAA, g_mag_hat, reg_coef, dmag, dphs, ws, g_m, g_p, false, stream);
}

extern "C" int st_polar_bwd(const st_dims* d, const float* re, const float* im, const float* dmag, const float* dphs,
                            const float* g_mag, float* dG, void* stream)
{
    ST_TRY(check_dims(d)); ST_REQ(re && im && dmag && dphs && dG, "st_polar_bwd: null pointer");
    const int R = d->B * d->T, KP = st_kp_of(d->F);
    hipLaunchKernelGGL(stm::polar_bwd_kernel, dim3((KP / 2 + 255) / 256, R), dim3(256), 0, st_stream(stream),
                       re, im, dmag, dphs, g_mag, dG, R, d->F, KP, gemm_ht(d->prec) == 2 ? 65504.0f : 0.0f);
    ST_LAUNCHED("polar_bwd"); return ST_OK;
}

// half = -1: both bases in one GEMM (M = KP rows of dG^T); half = 0 / 1: only the real / imaginary basis (M = KP/2), so
// that in data parallel the first half's all-reduce runs under the second half's GEMM (st_loss_backward_stage).
static int analysis_wgrad_impl(const st_dims* d, const float* dG, const float* sig, bool padded, float in_scale, float* ws,
                               float* gWr, float* gWi, float* norm_partial, void* stream, int half = -1, float* stage = nullptr, int stage_bf16 = 0)
{
    const int KP = st_kp_of(d->F);
    const stg::RowMap ma = stg::live_frames(d->T, d->H, d->N, d->N, d->L);   // all-zero frames contribute nothing
    const int R = ma.rows(d->B);
    int ns = wgrad_split_tiles(R, KP, d->N);
    if (half >= 0) {
        // One-basis GEMM: 6 x 11 tiles of 96 x 96, one 3-wave workgroup per CU at a time, so the run time goes with
        // ceil(tiles * ns / CUs) / ns (measured sawtooth, B=256: ns = 7, 11, 15 are the minima, 8 / 12 / 16 cost +40..60 us).
        // Pick the split in [ns/2, ns] that fills its last round best; never above the slab count the workspace holds.
        const int tiles = ((KP / 2 + 95) / 96) * ((d->N + 95) / 96), cus = num_cus();
        int best = ns; double bf = -1.0;
        for (int c = ns; c >= (ns + 1) / 2 && c >= 1; --c) {
            const int w = tiles * c, rounds = (w + cus - 1) / cus;
            const double f = (double)w / ((double)rounds * cus);
            if (f > bf + 1e-9) { bf = f; best = c; }
        }
        ns = (g_wsplit_half > 0 && g_wsplit_half < ns) ? g_wsplit_half : best;
    }
    const int M = half < 0 ? KP : KP / 2, m0 = half > 0 ? KP / 2 : 0;
    stm::NyqJob nyq{}; nyq.on = 0;
    const stg::TNOperand ta{dG, (unsigned)(d->T * KP), (unsigned)KP}, tb{sig, (unsigned)(d->L + 2 * d->N), (unsigned)d->H};
    if (half < 0 && use_tn128(d, padded) && stg::tn128_fits(ta, tb, sig, ma, d->N, d->N, (size_t)d->B * d->T * KP, (size_t)d->B * (d->L + 2 * d->N))) {
        ns = tn_split(R, d->N);
        const TNFrameMajor fm = tn_frame_major(ta, tb, ma, d->B, d->H, d->N, d->N, d->L, (g_tn_fm & 2) != 0);      // taps of the partly padded frames (Conv1d padding, cls_fe_dft.py:28-31) are zeros: skipped per tile column
        ST_TRY(wgrad_tn128(d, fm.a, fm.b, sig, fm.map, R, ws, ns, &nyq, stream, &fm.trim));
    } else {
        stg::PlainTN al{dG + m0, R, KP, M, ma};
        stg::StoreC ep{ws + (size_t)m0 * d->N, M, d->N, d->N, (size_t)KP * d->N, stg::all_frames(1)};
        if (padded) { stg::FramedTN<true> bl{sig, d->L, d->H, d->N, R, d->N, 1.0f, ma}; ST_GEMM_WG(al, bl, ep, M, d->N, R, ns, st_stream(stream)); }
        else { stg::FramedTN<false> bl{sig, d->L, d->H, d->N, R, d->N, in_scale, ma}; ST_GEMM_WG(al, bl, ep, M, d->N, R, ns, st_stream(stream)); }
    }
    ST_LAUNCHED("analysis_wgrad");
    // grid: the gradient rows of this call, then (whole tensor / second half) the Nyquist blocks resp. the unused partial slots
    const int nrows = half < 0 ? 2 * d->F : d->F, extra = half == 0 ? 0 : stm::nyq_blocks(d->N);
    hipLaunchKernelGGL(stm::wgrad_reduce_kernel, dim3(nrows + extra), dim3(256), 0, st_stream(stream),
                       ws, ns, gWr, gWi, norm_partial, d->N, d->F, KP, 0, half > 0 ? d->F : 0, nrows, stage, nyq, 0, stage_bf16);
    ST_LAUNCHED("analysis_wgrad_reduce");
    return ST_OK;
}
extern "C" int st_analysis_wgrad(const st_dims* d, const float* dG, const float* x, float in_scale, float* ws,
                                 float* gWr, float* gWi, float* norm_partial, void* stream)
{
    ST_TRY(check_dims(d)); ST_REQ(dG && x && ws && gWr && gWi && norm_partial, "st_analysis_wgrad: null pointer");
    return analysis_wgrad_impl(d, dG, x, false, in_scale, ws, gWr, gWi, norm_partial, stream);
}

extern "C" int st_finalize_scalars(const st_dims* d, const float* loss_partial, const float* reg_partial,
                                   const float* norm_a, const float* norm_s, float inv_world, float* scalars, void* stream)
{
    ST_TRY(check_dims(d)); ST_REQ(scalars, "st_finalize_scalars: null pointer");
    const float inv_y = 1.0f / ((float)d->B * (float)d->y);
    const float reg_scale = (float)(2e-5 / 10.0) / ((float)d->B * (float)d->OT * (float)d->F);   // loss_functions.py:36
    const stm::FinArgs f{loss_partial, st_ola_loss_partials(d), reg_partial, st_ae_fwd_partials(d),
                         norm_a, st_norm_partials(d), norm_s, st_norm_partials(d), inv_y, reg_scale, inv_world, nullptr, 0};
    hipLaunchKernelGGL(stm::finalize_kernel, dim3(1), dim3(256), 0, st_stream(stream), f, scalars);
    ST_LAUNCHED("finalize_scalars"); return ST_OK;
}
static stm::FinArgs fin_args(const st_dims* d, const float* loss_partial, const float* reg_partial, const float* norm_a, const float* norm_s, float norm_scale)
{
    const float inv_y = 1.0f / ((float)d->B * (float)d->y);
    const float reg_scale = (float)(2e-5 / 10.0) / ((float)d->B * (float)d->OT * (float)d->F);   // loss_functions.py:36
    return stm::FinArgs{loss_partial, st_ola_loss_partials(d), reg_partial, st_ae_fwd_partials(d),
                        norm_a, st_norm_partials(d), norm_s, norm_s ? st_norm_partials(d) : 0, inv_y, reg_scale, norm_scale, nullptr, 0};
}

// fin != nullptr: the clip coefficient (and, if fin->loss_partial, the loss scalars) are derived inside the kernel from
// the partial sums -- no separate finalize launch; `scalars` is then an output.
static int clip_adam_impl(float* params, float* grads, float* m, float* v, int64_t n_total, int64_t n_stft /* clipped range */,
                          float* scalars, float grad_scale, float lr, float beta1, float beta2, float eps, int step,
                          const stm::FinArgs* fin, void* stream, bool dev_hyper = false)
{
    ST_REQ(params && grads && m && v && scalars, "st_clip_adam: null pointer");
    ST_REQ(n_total % 4 == 0 && n_stft % 4 == 0 && n_stft <= n_total && step >= 1, "st_clip_adam: bad sizes/step");
    const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
    const float neg_step = (float)(-(double)lr / bc1);
    const float bc2s = (float)sqrt(bc2);
    const float w1 = (float)(1.0 - (double)beta1), w2 = (float)(1.0 - (double)beta2);
    int grid = (int)((n_total / 4 + 255) / 256); if (grid > 2048) grid = 2048;
    if (dev_hyper) {          // step number and learning rate from scalars[6], scalars[7] (captured step, st_graph_*)
        ST_REQ(fin, "device-side hyper-parameters need the fused finalize");
        hipLaunchKernelGGL((stm::clip_adam_kernel<true, true>), dim3(grid), dim3(256), 0, st_stream(stream),
                           params, grads, m, v, n_total / 4, n_stft / 4, scalars, grad_scale, 0.f, w1, beta2, w2, 1.f, eps, *fin, beta1);
    } else if (fin)
        hipLaunchKernelGGL(stm::clip_adam_kernel<true>, dim3(grid), dim3(256), 0, st_stream(stream),
                           params, grads, m, v, n_total / 4, n_stft / 4, scalars, grad_scale, neg_step, w1, beta2, w2, bc2s, eps, *fin);
    else
        hipLaunchKernelGGL(stm::clip_adam_kernel<false>, dim3(grid), dim3(256), 0, st_stream(stream),
                           params, grads, m, v, n_total / 4, n_stft / 4, scalars, grad_scale, neg_step, w1, beta2, w2, bc2s, eps, stm::FinArgs{});
    ST_LAUNCHED("clip_adam"); return ST_OK;
}
extern "C" int st_clip_adam(float* params, float* grads, float* m, float* v, int64_t n_total, int64_t n_stft,
                            const float* scalars, float grad_scale, float lr, float beta1, float beta2, float eps, int step,
                            void* stream)
{
    return clip_adam_impl(params, grads, m, v, n_total, n_stft, const_cast<float*>(scalars), grad_scale, lr, beta1, beta2, eps, step, nullptr, stream);
}

extern "C" int st_debug_read_stage_cycles(unsigned long long* out32)
{
    unsigned long long z[32] = {0};
    if (hipMemcpyFromSymbol(out32, HIP_SYMBOL(sta::g_ae_stage_cycles), sizeof(z)) != hipSuccess) return st_fail(ST_ERR_LAUNCH, "memcpyFromSymbol");
    if (hipMemcpyToSymbol(HIP_SYMBOL(sta::g_ae_stage_cycles), z, sizeof(z)) != hipSuccess) return st_fail(ST_ERR_LAUNCH, "memcpyToSymbol");
    return ST_OK;
}

// ------------------------------------------------------------------------------ workspace
struct WS {
    // dAA, frs: NOT every element is written per step (the pitch-padding columns [F, FP) of dAA's halves, the all-cropped tiles of frs): no consumer may read those -- the
    // contract is stated at st_workspace_bytes in the header; the Python engine zero-fills the buffer once when it allocates it
    float *re, *im, *mag, *phs, *mag_hat, *phs_hat, *AA, *dAA, *Sfold, *SfoldT, *frs, *y_hat, *dsyn, *dmag, *dphs, *dG, *xp;
    float *wg, *wg2, *aews, *loss_p, *reg_p, *norm_a, *norm_s, *norm_e;      // wg2: the synthesis weight gradient's own slabs in the data-parallel step (synth_wgrad_ws_floats)
    // bfloat16 planes of the operands that are written once per step (st_gemm_planes.h): [3][same layout as the fp32 tensor]
    unsigned short *pl_W, *pl_Sfold, *pl_SfoldT;      // k-chunk-major: [K / 16][rows][3][16]
    // 16-bit GEMM operands (st_gemm16.h), each written by its producer in the layout of its fp32 counterpart: padded x/2, the analysis
    // bases as rows (bin, re | im) [2F][N], the folded synthesis bases [KP][N] and [N][KP], the spectra, the padded d syn, d G
    unsigned short *xp16, *W16, *Sfold16, *SfoldT16, *AA16, *dsyn16, *dG16;
    int n_norm_e;      // > 0: the autoencoder backward of this call left that many |g| partials of its gradients in norm_e (clip_all needs no l1_partial launch)
    bool g16;          // this call runs the 16-bit operand pipeline (set by the entry point after carve(): use_g16(); the autograd entries and the
                       // four-stage schedule keep fp32 operands + gemm_half_kernel)
    size_t bytes;
};
static void carve(const st_dims* d, void* base, WS* w)
{
    const size_t RT = (size_t)d->B * d->T, RO = (size_t)d->B * d->OT, F = d->F, KP = st_kp_of(d->F), N = d->N;
    size_t off = 0;
    auto take = [&](size_t n) { float* p = base ? reinterpret_cast<float*>(base) + off : nullptr; off += (n + 63) / 64 * 64; return p; };
    w->re = take(RT * F); w->im = take(RT * F); w->mag = take(RT * F); w->phs = take(RT * F);
    w->mag_hat = take(RO * F); w->phs_hat = take(RO * F);
    const size_t nsl = st_synth_slabs(d), nfs = st_synth_frame_slabs(d);
    w->AA = take(RO * KP); w->dAA = take(nsl * RO * KP);
    w->Sfold = take(KP * N); w->SfoldT = take(KP * N); w->frs = take(nfs * RO * N);
    w->y_hat = take((size_t)d->B * d->y); w->dsyn = take((size_t)d->B * (d->y + 2 * d->N));   // dsyn padded [B][N + y + N]
    w->xp = take((size_t)d->B * (d->L + 2 * d->N));                                               // x/2 padded  [B][N + L + N]
    w->dmag = take(RT * F); w->dphs = take(RT * F); w->dG = take(RT * KP);
    w->wg = take(st_wgrad_ws_floats(d)); w->aews = take(st_ae_bwd_ws_floats(d));
    w->loss_p = take(st_ola_loss_partials(d)); w->reg_p = take(st_ae_fwd_partials(d));
    w->norm_a = take(st_norm_partials(d)); w->norm_s = take(st_norm_partials(d)); w->norm_e = take(NORM_E_MAX); w->n_norm_e = 0;
    auto take16 = [&](size_t n) { return reinterpret_cast<unsigned short*>(take((n + 1) / 2)); };      // always sized for three planes: 19 MB
    w->pl_W = take16((size_t)3 * 2 * F * N); w->pl_Sfold = take16((size_t)3 * KP * N); w->pl_SfoldT = take16((size_t)3 * KP * N);
    // (+ 256: the 128-wide tiles of the TN kernel read up to 96 elements past the last row of an M/N-contiguous operand; masked outputs)
    w->xp16 = take16((size_t)d->B * (d->L + 2 * d->N) + 256); w->W16 = take16((size_t)2 * F * N); w->Sfold16 = take16(KP * N); w->SfoldT16 = take16(KP * N);
    w->AA16 = take16(RO * KP + 256); w->dsyn16 = take16((size_t)d->B * (d->y + 2 * d->N) + 256); w->dG16 = take16(RT * KP + 256);
    w->wg2 = take(synth_wgrad_ws_floats(d));      // last: every earlier offset stays what it was
    w->g16 = false;
    w->bytes = off * sizeof(float);
}
// float offsets of the forward state inside a workspace carved for `d` (after st_model_fwd with save_for_backward):
// re, im, mag, phs [B][T][F]; mag_hat, phs_hat [B][OT][F]; AA [B*OT][KP] (an_real at columns [0, F), an_imag at [KP/2, KP/2 + F)); y_hat [B][y].
// For diagnostics (st_model.forward(return_acts=True), nn_proc.py:311-338): the caller views its own buffer, nothing is copied.
extern "C" int st_workspace_offsets(const st_dims* d, int64_t* offs8)
{
    ST_TRY(check_dims(d)); ST_REQ(offs8, "st_workspace_offsets: null pointer");
    WS w; carve(d, reinterpret_cast<void*>(sizeof(float)), &w);       // a non-null dummy base: pointer differences are the offsets
    float* base = reinterpret_cast<float*>(sizeof(float));
    float* p[8] = {w.re, w.im, w.mag, w.phs, w.mag_hat, w.phs_hat, w.AA, w.y_hat};
    for (int i = 0; i < 8; ++i) offs8[i] = (int64_t)(p[i] - base);
    return ST_OK;
}
// The ten return_acts tensors of ONE autoencoder (nn_proc.py:77-126), [B][F][width] each, concatenated in `acts`
// (widths 64, 32, 16, 16, 16 + K, 16, 16, 32, 64, OT; st_ae_acts_floats() floats).  v: [B][T][F] (mag or phs); ae: that autoencoder's
// packed parameters (st_param_offsets order); sf != 0: the skip-filter output (magnitude net), else the bare ELU output (phase net).
extern "C" size_t st_ae_acts_floats(const st_dims* d) { return (size_t)d->B * d->F * (64 + 32 + 16 + 16 + 16 + d->K + 16 + 16 + 32 + 64 + d->OT); }
extern "C" int st_ae_acts(const st_dims* d, const float* v, const float* knobs, const float* ae, int sf, float* acts, void* stream)
{
    Layout L; ST_TRY(make_layout(d, &L));
    ST_REQ(v && (knobs || d->K == 0) && ae && acts, "st_ae_acts: null pointer");
    if (!knobs) knobs = ae;
    stm::AeActsArgs a;
    a.v = v; a.knobs = knobs; a.ae = ae; a.B = d->B; a.T = d->T; a.OT = d->OT; a.F = d->F; a.K = d->K; a.sf = sf;
    for (int l = 0; l < 9; ++l) { a.w_off[l] = L.go.w[l]; a.b_off[l] = L.go.b[l]; }
    const int widths[10] = {64, 32, 16, 16, 16 + d->K, 16, 16, 32, 64, d->OT};
    size_t off = 0;
    for (int i = 0; i < 10; ++i) { a.out[i] = acts + off; off += (size_t)d->B * d->F * widths[i]; }
    hipLaunchKernelGGL(stm::ae_acts_kernel, dim3((d->B * d->F + 63) / 64), dim3(64), 0, st_stream(stream), a);
    ST_LAUNCHED("ae_acts"); return ST_OK;
}
extern "C" size_t st_workspace_bytes(const st_dims* d)
{
    if (check_dims(d) != ST_OK) return 0;
    WS w; carve(d, nullptr, &w); return w.bytes;
}
// ONE workspace for every batch 1 .. d->B at every arithmetic level and clip scope: the exact size above is monotonic in none of them (header).
extern "C" size_t st_workspace_bytes_max(const st_dims* d)
{
    if (check_dims(d) != ST_OK) return 0;
    size_t best = 0;
    st_dims q = *d;
    for (int b = 1; b <= d->B; ++b)
        for (int prec = ST_PREC_F32; prec <= ST_PREC_F32X3; ++prec)
            for (int ca = 0; ca < 2; ++ca) {
                q.B = b; q.prec = prec; q.clip_all = ca;
                WS w; carve(&q, nullptr, &w);
                if (w.bytes > best) best = w.bytes;
            }
    return best;
}

// ------------------------------------------------------------------------------ the plane GEMMs of the fused step (ST_PREC_F32X3)
// Operands written once per step -- the padded waveform, the F used rows of the analysis bases, the folded synthesis bases in both
// orientations -- become three bfloat16 planes in one elementwise launch; the spectra AA and d syn (one consumer each) are split as
// their GEMM stages them.  The weight-gradient GEMMs (reduction along the rows of both operands) keep the fp32 MFMA kernel.
static bool use_planes(const st_dims* d) { return (gemm_ht(d->prec) == 3 || (gemm_ht(d->prec) == 1 && g_pl_bf16)) && d->N % 16 == 0 && st_kp_of(d->F) % 16 == 0; }
static int planes_of(const st_dims* d) { return gemm_ht(d->prec) == 3 ? 3 : 1; }
static int planes_prepare(const st_dims* d, const float* Wr, const float* Wi, WS& w, void* stream)
{
    const int KP = st_kp_of(d->F);
    stg::WPlanesArgs a; a.njobs = 3;
    a.job[0] = stg::WPlanesJob{Wr, Wi, w.pl_W, 2 * d->F, d->N, d->N};                 // rows (bin, re | im) interleaved
    a.job[1] = stg::WPlanesJob{w.Sfold, nullptr, w.pl_Sfold, KP, d->N, d->N};
    a.job[2] = stg::WPlanesJob{w.SfoldT, nullptr, w.pl_SfoldT, d->N, KP, KP};
    a.job[3] = a.job[2];
    unsigned blk = 0;
    for (int j = 0; j < 3; ++j) { a.blk0[j] = blk; blk += (unsigned)(((size_t)a.job[j].rows * (a.job[j].K / 4) + 255) / 256); }
    a.blk0[3] = blk; a.blk0[4] = blk;
    if (planes_of(d) == 3) hipLaunchKernelGGL(stg::wplanes_kernel<3>, dim3(blk), dim3(256), 0, st_stream(stream), a);
    else hipLaunchKernelGGL(stg::wplanes_kernel<1>, dim3(blk), dim3(256), 0, st_stream(stream), a);
    ST_LAUNCHED("planes");
    return ST_OK;
}
static int analysis_fwd_planes(const st_dims* d, WS& w, float* re, float* im, float* mag, float* phs, void* stream, const PolarWide* pw = nullptr)
{
    const stg::RowMap map = stg::live_frames(d->T, d->H, d->N, d->N, d->L);
    const int R = map.rows(d->B);
    stg::FramedNT<true> al{w.xp, d->L, d->H, d->N, R, d->N, 1.0f, map};
    stg::ChunkP bl{w.pl_W, 2 * d->F};
    stg::PolarStore ep{re, im, mag, phs, R, d->F, map};
    polar_wide_set(ep, pw);
    if (planes_of(d) == 1) ST_TRY((stg::launch_planes<4, 1>(al, bl, ep, R, 2 * d->F, d->N, 1, st_stream(stream))));
    else if (g_pl_shape == 1) ST_TRY((stg::launch_planes<2, 3, 2>(al, bl, ep, R, 2 * d->F, d->N, 1, st_stream(stream))));
    else if (g_pl_shape == 2) ST_TRY((stg::launch_planes<4, 3, 2>(al, bl, ep, R, 2 * d->F, d->N, 1, st_stream(stream))));
    else if (g_pl_shape == 3) ST_TRY((stg::launch_planes<8, 3, 1>(al, bl, ep, R, 2 * d->F, d->N, 1, st_stream(stream))));
    else ST_TRY((stg::launch_planes<4, 3>(al, bl, ep, R, 2 * d->F, d->N, 1, st_stream(stream))));
    ST_LAUNCHED("analysis_fwd");
    return ST_OK;
}
static int synthesis_frames_planes(const st_dims* d, WS& w, void* stream)
{
    const int KP = st_kp_of(d->F);
    const stg::RowMap ms = synth_live(d);
    const int R = ms.rows(d->B);
    stg::PlainNT al{w.AA, R, KP, KP, ms};
    stg::ChunkP bt{w.pl_SfoldT, d->N};
    stg::StoreC ep{w.frs, R, d->N, d->N, (size_t)d->B * d->OT * d->N, ms};
    if (planes_of(d) == 1) {
        if (R >= 4096) ST_TRY((stg::launch_planes<4, 1>(al, bt, ep, R, d->N, KP, 1, st_stream(stream))));
        else ST_TRY((stg::launch_planes<2, 1>(al, bt, ep, R, d->N, KP, frames_split(R), st_stream(stream))));
    }
    else if (R >= 4096) ST_TRY((stg::launch_planes<4, 3>(al, bt, ep, R, d->N, KP, 1, st_stream(stream))));
    else ST_TRY((stg::launch_planes<2, 3>(al, bt, ep, R, d->N, KP, frames_split(R), st_stream(stream))));
    ST_LAUNCHED("synthesis_frames"); return ST_OK;
}
static int synthesis_dgrad_planes(const st_dims* d, WS& w, void* stream)
{
    const int KP = st_kp_of(d->F);
    const stg::RowMap ms = synth_live(d);
    const int R = ms.rows(d->B);
    stg::FramedNT<true> al{w.dsyn, d->y, d->H, d->N, R, d->N, 1.0f, ms};
    stg::ChunkP bl{w.pl_Sfold, KP};
    stg::StoreC ep{w.dAA, R, KP, KP, (size_t)d->B * d->OT * KP, ms};
    const int ns = R >= 4096 ? 1 : synth_split(R);
    if (planes_of(d) == 1) {
        if (R >= 4096) ST_TRY((stg::launch_planes<4, 1>(al, bl, ep, R, KP, d->N, ns, st_stream(stream))));
        else ST_TRY((stg::launch_planes<2, 1>(al, bl, ep, R, KP, d->N, ns, st_stream(stream))));
    }
    else if (R >= 4096) ST_TRY((stg::launch_planes<4, 3>(al, bl, ep, R, KP, d->N, ns, st_stream(stream))));
    else ST_TRY((stg::launch_planes<2, 3>(al, bl, ep, R, KP, d->N, ns, st_stream(stream))));
    ST_LAUNCHED("synthesis_dgrad"); return ST_OK;
}

// ------------------------------------------------------------------------------ the 16-bit operand pipeline (st_gemm16.h)
static bool use_g16(const st_dims* d)
{
    const int ht = gemm_ht(d->prec);
    return g_g16 && (ht == 1 || ht == 2) && !g_pl_bf16 && d->N % 128 == 0 && num_cus() > 0;
}
#define ST_G16(CALL_) do { if (gemm_ht(d->prec) == 2) ST_TRY((stg::CALL_<2>)); else ST_TRY((stg::CALL_<1>)); } while (0)
static int analysis_fwd16(const st_dims* d, WS& w, float* re, float* im, float* mag, float* phs, void* stream, const PolarWide* pw = nullptr)
{
    const stg::RowMap map = stg::live_frames(d->T, d->H, d->N, d->N, d->L);
    const int R = map.rows(d->B);
    stg::Rows16 ra = stg::rows16(w.xp16, (unsigned)(d->L + 2 * d->N), (unsigned)d->H, map, R);
    if (g_g16_abl & 4) ra.S2 &= ~7u;              // timing only: 16-byte aligned frame rows
    const stg::Rows16 rb = stg::rows16_plain(w.W16, (unsigned)d->N, 2 * d->F);
    stg::PolarStore ep{re, im, mag, phs, R, d->F, map};
    polar_wide_set(ep, pw);
    if (g_g16_abl & 1) ep.mag = ep.phs = nullptr;
    if (g_g16_abl & 2) ep.re = ep.im = nullptr;
    if ((g_g16_dma & 1) && d->N % 64 == 0) {
        if (gemm_ht(d->prec) == 2) ST_TRY((stg::launch16_nt256<2>(ra, rb, ep, R, 2 * d->F, d->N, 1, st_stream(stream), g_g16_abl >> 3)));
        else ST_TRY((stg::launch16_nt256<1>(ra, rb, ep, R, 2 * d->F, d->N, 1, st_stream(stream), g_g16_abl >> 3)));
    }
    else if (gemm_ht(d->prec) == 2) ST_TRY((stg::launch16_nt<2>(ra, rb, ep, R, 2 * d->F, d->N, 1, st_stream(stream), g_g16_bk != 32)));
    else ST_TRY((stg::launch16_nt<1>(ra, rb, ep, R, 2 * d->F, d->N, 1, st_stream(stream), g_g16_bk != 32)));
    ST_LAUNCHED("analysis_fwd"); return ST_OK;
}
static int synthesis_frames16(const st_dims* d, WS& w, void* stream)
{
    const int KP = st_kp_of(d->F);
    const stg::RowMap ms = synth_live(d);
    const int R = ms.rows(d->B);
    const bool crop = (g_g16_crop & 1) && d->N % 128 == 0 && fm_div_exact(R, d->B);
    const stg::Rows16 ra = crop ? stg::rows16_frame_major(w.AA16, (unsigned)(d->OT * KP), (unsigned)KP, ms, d->B, R) : stg::rows16(w.AA16, (unsigned)(d->OT * KP), (unsigned)KP, ms, R);
    const stg::Rows16 rb = stg::rows16_plain(w.SfoldT16, (unsigned)KP, d->N);
    stg::StoreC ep{w.frs, R, d->N, d->N, (size_t)d->B * d->OT * d->N, crop ? stg::frame_major(ms, d->B) : ms};
    const stg::Crop16 cr{crop ? 1 : 0, d->B, d->H, d->N, d->N, d->y, ms.t_lo, R, 1};      // tile columns without a tap inside the crop (cls_fe_dft.py:113) are not computed: ola_loss_kernel never reads them
    const int ns = frames_split(R);
    if (gemm_ht(d->prec) == 2) ST_TRY((stg::launch16_nt<2>(ra, rb, ep, R, d->N, KP, ns, st_stream(stream), true, &cr)));
    else ST_TRY((stg::launch16_nt<1>(ra, rb, ep, R, d->N, KP, ns, st_stream(stream), true, &cr)));
    ST_LAUNCHED("synthesis_frames"); return ST_OK;
}
static int synthesis_dgrad16(const st_dims* d, WS& w, void* stream)
{
    const int KP = st_kp_of(d->F);
    const stg::RowMap ms = synth_live(d);
    const int R = ms.rows(d->B);
    const bool crop = (g_g16_crop & 2) && !(g_g16_dma & 2) && fm_div_exact(R, d->B);
    const stg::Rows16 ra = crop ? stg::rows16_frame_major(w.dsyn16, (unsigned)(d->y + 2 * d->N), (unsigned)d->H, ms, d->B, R) : stg::rows16(w.dsyn16, (unsigned)(d->y + 2 * d->N), (unsigned)d->H, ms, R);
    const stg::Rows16 rb = stg::rows16_plain(w.Sfold16, (unsigned)d->N, KP);
    stg::StoreC ep{w.dAA, R, KP, KP, (size_t)d->B * d->OT * KP, crop ? stg::frame_major(ms, d->B) : ms};
    const stg::Crop16 cr{crop ? 2 : 0, d->B, d->H, d->N, d->N, d->y, ms.t_lo, R, 1};      // per tile row only the taps that lie inside d syn: the rest of the padded copy is zeros
    const int ns = R >= 4096 ? 1 : synth_split(R);
    if ((g_g16_dma & 2) && d->N % 64 == 0 && d->N / 64 >= ns) {
        if (gemm_ht(d->prec) == 2) ST_TRY((stg::launch16_nt256<2>(ra, rb, ep, R, KP, d->N, ns, st_stream(stream))));
        else ST_TRY((stg::launch16_nt256<1>(ra, rb, ep, R, KP, d->N, ns, st_stream(stream))));
    }
    else if (gemm_ht(d->prec) == 2) ST_TRY((stg::launch16_nt<2>(ra, rb, ep, R, KP, d->N, ns, st_stream(stream), true, &cr)));
    else ST_TRY((stg::launch16_nt<1>(ra, rb, ep, R, KP, d->N, ns, st_stream(stream), true, &cr)));
    ST_LAUNCHED("synthesis_dgrad"); return ST_OK;
}
// k-slices of a 16-bit weight-gradient GEMM: about two workgroups per CU (their time is staging, not matrix work), never slices under
// 128 reduction rows, never more slabs than the workspace holds
static int g16_wsplit(const st_dims* d, int R)
{
    const int KP = st_kp_of(d->F), tiles = ((KP + 127) / 128) * (d->N / 128);
    int s = g_g16_split > 0 ? g_g16_split : (2 * num_cus()) / (tiles > 0 ? tiles : 1);
    const int cap = R / 128; if (s > cap) s = cap;
    const int room = (int)(st_wgrad_ws_floats(d) / ((size_t)KP * d->N)); if (s > room) s = room;
    return s < 1 ? 1 : s;
}
// A: [rows (b, t)][KP] 16-bit (d G or the spectra), B: frames of a padded 16-bit signal whose first N elements are zero (the block of zeros)
static int wgrad16(const st_dims* d, const unsigned short* A, unsigned SA1, const unsigned short* Bsig, unsigned SB1, const stg::RowMap& map, int R,
                   float* slabs, int ns, void* stream, int m0 = 0, int M = -1,      // [m0, m0 + M): the output rows (= columns of A) of this launch; default all KP
                   int trim_Ls = 0)                                                  // > 0: frame-major reduction order, rows per tile column trimmed to the frames with a tap inside [pad, pad + Ls) (round 5)
{
    const int KP = st_kp_of(d->F);
    if (M < 0) M = KP;
    const unsigned short* lo = A < Bsig ? A : Bsig;
    ST_REQ((size_t)(A - lo) + (size_t)R / map.Tv * SA1 + 256 < ((size_t)1 << 30) && (size_t)(Bsig - lo) + (size_t)R / map.Tv * SB1 + 256 < ((size_t)1 << 30) && map.Tv >= 2 &&
           SA1 < (1u << 23) && SB1 < (1u << 23), "16-bit weight-gradient GEMM: operands out of the 32-bit / 24-bit addressing range (B=%d)", d->B);
    stg::TN16Job j;
    j.base = lo; j.a0 = (unsigned)(A - lo) + (unsigned)m0; j.b0 = (unsigned)(Bsig - lo); j.zero = j.b0;
    j.SA1 = SA1; j.SA2 = (unsigned)KP; j.SB1 = SB1; j.SB2 = (unsigned)d->H;
    j.magic = map.magic; j.Tv = map.Tv; j.t_lo = map.t_lo; j.K = R;
    j.trim = 0; j.fB = 0; j.nsplit = 1;
    for (int i = 0; i < 64; ++i) { j.fa[i] = 0; j.fb[i] = 0; }
    if (trim_Ls > 0 && d->B >= 2 && fm_div_exact(R, d->B)) {
        const stg::FrameTrim ft = stg::frame_trim(map, d->B, d->H, d->N, d->N, trim_Ls);
        // (outer, inner) = (frame, window): the strides swap roles, the origins move to frame t_lo (the zero block stays where it is)
        j.a0 += (unsigned)map.t_lo * (unsigned)KP; j.b0 += (unsigned)map.t_lo * (unsigned)d->H;
        j.SA1 = (unsigned)KP; j.SA2 = SA1; j.SB1 = (unsigned)d->H; j.SB2 = SB1;
        j.magic = stg::rowmap_magic(d->B); j.Tv = d->B; j.t_lo = 0;
        j.trim = ft.on; j.fB = d->B;
        for (int i = 0; i < 64; ++i) { j.fa[i] = ft.fa[i]; j.fb[i] = ft.fb[i]; }
    }
    stg::StoreC ep{slabs + (size_t)m0 * d->N, M, d->N, d->N, (size_t)KP * d->N, stg::all_frames(1)};
    const int ht = gemm_ht(d->prec);
    if (g_g16_bk == 32) { if (ht == 2) ST_TRY((stg::launch16_tn<2, 32>(j, ep, M, d->N, ns, st_stream(stream)))); else ST_TRY((stg::launch16_tn<1, 32>(j, ep, M, d->N, ns, st_stream(stream)))); }
    else { if (ht == 2) ST_TRY((stg::launch16_tn<2, 64>(j, ep, M, d->N, ns, st_stream(stream)))); else ST_TRY((stg::launch16_tn<1, 64>(j, ep, M, d->N, ns, st_stream(stream)))); }
    return ST_OK;
}

// ------------------------------------------------------------------------------ fused entry points
static int forward_impl(const st_dims* d, const Layout& L, const float* params, const float* x, const float* knobs,
                        const float* y_true, float* y_hat, float* mag, float* mag_hat, WS& w, bool save, void* stream)
{
    const float* Wr = params + L.offs[0]; const float* Wi = params + L.offs[1];
    const float* Sr = params + L.offs[2]; const float* Si = params + L.offs[3];
    const float* ae_m = params + L.offs[4]; const float* ae_p = params + L.offs[22];
    // wide geometries (round 4): the analysis epilogue writes mag / phs straight into the wide autoencoder path's feature-major input
    const bool wide_direct = ae_is_wide(d) && g_wide_direct && (size_t)d->B * (L.KP / 2) * (size_t)(d->T > 64 ? d->T : 64) < ((size_t)1 << 30);
    WideWS ww; if (wide_direct) wide_carve(d, w.aews, &ww);
    PolarWide pwide{nullptr, nullptr, 0, 0u};
    if (wide_direct) pwide = PolarWide{ww.V[0], ww.V[1], L.KP / 2, (unsigned)ww.R};
    // saved-for-backward state always lives in the workspace; user-visible outputs are copies
    {   // one launch: x/2 (nn_proc.py:307) with the Conv1d padding materialised, the Hermitian fold of the synthesis bases
        // (cls_fe_dft.py:109-110 on the weights) and the exact zeros of the frames that lie wholly in the padding
        const stg::RowMap map = stg::live_frames(d->T, d->H, d->N, d->N, d->L);
        stm::PrepArgs a;
        a.x = x; a.xp = w.xp; a.Ls = d->L; a.pad = d->N; a.scale = 0.5f;
        a.nbx = ((d->L + 2 * d->N) / 4 + 255) / 256; a.n_pad = a.nbx * d->B;
        a.Sr = Sr; a.Si = Si; a.Sfold = w.Sfold; a.SfoldT = w.SfoldT; a.N = d->N; a.F = d->F; a.KP = L.KP; a.n_fold = (L.KP / 32) * (d->N / 32);
        a.re = save ? w.re : nullptr; a.im = save ? w.im : nullptr; a.mag = w.mag; a.phs = w.phs; a.T = d->T; a.t_lo = map.t_lo; a.Tv = map.Tv;
        const int n_dead = d->B * (d->T - map.Tv);
        a.n_dead = n_dead; a.ht = w.g16 ? gemm_ht(d->prec) : 0; a.n_w16 = 0;
        a.xp16 = w.xp16; a.Sfold16 = w.Sfold16; a.SfoldT16 = w.SfoldT16; a.W16 = w.W16; a.Wr = Wr; a.Wi = Wi;
        if (a.ht) a.n_w16 = (int)(((size_t)2 * d->F * (d->N / 4) + 255) / 256);
        a.wd = stm::PrepWide{}; a.wd.n_vpad = a.wd.n_kn = a.wd.n_pj = 0;
        if (wide_direct) {       // wide geometries: the side jobs of the feature-major input (see stm::PrepWide)
            const int FP = L.KP / 2, Tp = ww.Tp;
            a.wd.Vm = ww.V[0]; a.wd.Vp = ww.V[1]; a.wd.FP = FP; a.wd.B = d->B; a.wd.R = (unsigned)ww.R;
            a.wd.H4Km = ww.H[0][3]; a.wd.H4Kp = ww.H[1][3]; a.wd.knobs = knobs; a.wd.K = d->K;
            int blk = 0;
            for (int n = 0; n < 2; ++n) {
                const float* ae = n ? ae_p : ae_m;
                a.wd.pj.src[2 * n] = ae + L.go.w[0]; a.wd.pj.dst[2 * n] = ww.W1p[n]; a.wd.pj.rows[2 * n] = 64; a.wd.pj.cols[2 * n] = d->T; a.wd.pj.pitch[2 * n] = Tp;
                a.wd.pj.blk0[2 * n] = blk; blk += (64 * Tp + 255) / 256;
                a.wd.pj.src[2 * n + 1] = ae + L.go.w[4]; a.wd.pj.dst[2 * n + 1] = ww.W5p[n]; a.wd.pj.rows[2 * n + 1] = 16; a.wd.pj.cols[2 * n + 1] = 16 + d->K; a.wd.pj.pitch[2 * n + 1] = 32;
                a.wd.pj.blk0[2 * n + 1] = blk; blk += 2;
            }
            a.wd.pj.blk0[4] = blk; a.wd.n_pj = blk;
            a.wd.n_vpad = (int)(((size_t)d->B * d->T * (FP - d->F) + 255) / 256); if (a.wd.n_vpad < 1) a.wd.n_vpad = 1;
            a.wd.n_kn = (int)(((size_t)d->K * d->B * (FP / 4) + 255) / 256);
        }
        hipLaunchKernelGGL(stm::prep_kernel, dim3(a.n_pad + a.n_fold + n_dead + a.wd.n_vpad + a.wd.n_kn + a.wd.n_pj + a.n_w16), dim3(256), 0, st_stream(stream), a);
        ST_LAUNCHED("prep");
    }
    const bool planes = use_planes(d) && !w.g16;
    // a training call (target given, no user-visible |STFT| requested) on the direct wide path needs mag / phs ONLY in the feature-major layout
    const bool polar_rows = !(wide_direct && y_true && !mag);
    float* const pmag = polar_rows ? w.mag : nullptr; float* const pphs = polar_rows ? w.phs : nullptr;
    if (w.g16) ST_TRY(analysis_fwd16(d, w, save ? w.re : nullptr, save ? w.im : nullptr, pmag, pphs, stream, &pwide));
    else if (planes) {
        ST_TRY(planes_prepare(d, Wr, Wi, w, stream));
        ST_TRY(analysis_fwd_planes(d, w, save ? w.re : nullptr, save ? w.im : nullptr, pmag, pphs, stream, &pwide));
    } else
    ST_TRY(analysis_fwd_impl(d, w.xp, true, Wr, Wi, 1.0f, save ? w.re : nullptr, save ? w.im : nullptr, pmag, pphs, stream, true, &pwide));
    float* sv = nullptr;
    if (save && ae_use_saved(d)) { Layout Ls; ST_TRY(make_layout(d, &Ls)); sv = ae_sv_ptr(d, Ls, w.aews); }      // round 6: the activations stay for the backward
    ST_TRY(ae_fwd_impl(d, w.mag, w.phs, knobs, ae_m, ae_p, w.mag_hat, w.phs_hat, w.AA, w.reg_p,
                       (ae_is_wide(d) || (save && ae_use_split(d))) ? w.aews : nullptr, stream, w.g16 ? w.AA16 : nullptr, wide_direct, sv));     // fused geometries: the code h4 is kept for the split backward
    if (w.g16) ST_TRY(synthesis_frames16(d, w, stream)); else
    if (planes) ST_TRY(synthesis_frames_planes(d, w, stream)); else
    ST_TRY(synthesis_frames_impl(d, w.AA, w.Sfold, w.SfoldT, w.frs, stream));
    ST_TRY(ola_loss_impl(d, w.frs, x, y_true, y_hat ? y_hat : w.y_hat, (save && y_true) ? w.dsyn : nullptr, d->N,
                         y_true ? w.loss_p : nullptr, stream, w.g16 ? w.dsyn16 : nullptr));
    const size_t nm = (size_t)d->B * d->T * d->F * sizeof(float), nh = (size_t)d->B * d->OT * d->F * sizeof(float);
    if (mag && hipMemcpyAsync(mag, w.mag, nm, hipMemcpyDeviceToDevice, st_stream(stream)) != hipSuccess) return st_fail(ST_ERR_LAUNCH, "copy mag");
    if (mag_hat && hipMemcpyAsync(mag_hat, w.mag_hat, nh, hipMemcpyDeviceToDevice, st_stream(stream)) != hipSuccess) return st_fail(ST_ERR_LAUNCH, "copy mag_hat");
    return ST_OK;
}

// backward of everything behind d syn (workspace holds the forward state): autograd of train.py:138.
// phase 1 = synthesis dgrad/wgrad + autoencoders + polar backward (fills grads[n_stft/2 ..));
// phase 2 = analysis weight gradient (fills rows [0,F) of the first two tensors).
static int backward_syn(const st_dims* d, const Layout& L, float* grads, WS& w, void* stream, int* defer_slabs = nullptr, stm::NyqJob* defer_nyq = nullptr)
{
    if (w.g16) {
        ST_TRY(synthesis_dgrad16(d, w, stream));
        const stg::RowMap ms = synth_live(d);
        const int R = ms.rows(d->B), KP = st_kp_of(d->F), ns = g16_wsplit(d, R);
        ST_TRY(wgrad16(d, w.AA16, (unsigned)(d->OT * KP), w.dsyn16, (unsigned)(d->y + 2 * d->N), ms, R, w.wg, ns, stream, 0, -1, (g_g16_crop & 4) ? d->y : 0));
        ST_LAUNCHED("synthesis_wgrad");
        if (defer_slabs) { *defer_slabs = ns; if (defer_nyq) { *defer_nyq = stm::NyqJob{}; defer_nyq->on = 0; } return ST_OK; }
        stm::NyqJob nq{}; nq.on = 0;
        hipLaunchKernelGGL(stm::wgrad_reduce_kernel, dim3(st_norm_partials(d)), dim3(256), 0, st_stream(stream),
                           w.wg, ns, grads + L.offs[2], grads + L.offs[3], w.norm_s, d->N, d->F, KP, 1, 0, 2 * d->F, (float*)nullptr, nq);
        ST_LAUNCHED("synthesis_wgrad_reduce");
        return ST_OK;
    }
    if (use_planes(d) && g_pl_dgrad) ST_TRY(synthesis_dgrad_planes(d, w, stream)); else
    ST_TRY(synthesis_dgrad_impl(d, w.dsyn, true, w.Sfold, w.dAA, stream));
    return synthesis_wgrad_impl(d, w.AA, w.dsyn, true, w.wg, grads + L.offs[2], grads + L.offs[3], w.norm_s, stream, defer_slabs, defer_nyq);
}
static int backward_ae(const st_dims* d, const Layout& L, const float* params, float* grads,
                       const float* knobs, const float* g_mag_hat, const float* g_mag, float reg_coef, WS& w, void* stream, int syn_slabs = 0,
                       const stm::NyqJob* syn_nyq = nullptr)
{   // syn_slabs > 0: the synthesis weight-gradient slabs in w.wg are still to be summed (done by post_ae_kernel)
    const float* ae_m = params + L.offs[4]; const float* ae_p = params + L.offs[22];
    bool deferred = true, sink_used = false, syn_done = false;
    const PolarSink sink{w.re, w.im, g_mag, w.g16 ? nullptr : w.dG, w.g16 ? w.dG16 : nullptr};
    stw::SynReduce syn{};
    if (syn_slabs > 0 && ae_is_wide(d)) {              // wide geometries: the synthesis slab sum rides in the gradient-finish launch of the autoencoder backward
        syn.wg = w.wg; syn.nz = syn_slabs; syn.gSr = grads + L.offs[2]; syn.gSi = grads + L.offs[3]; syn.norm_s = w.norm_s; syn.N = d->N; syn.F = d->F; syn.KP = L.KP;
        syn.nyq = stm::NyqJob{}; syn.nyq.on = 0; if (syn_nyq) syn.nyq = *syn_nyq;
    }
    ST_TRY(ae_bwd_impl(d, w.mag, w.phs, knobs, ae_m, ae_p, w.mag_hat, w.phs_hat, w.dAA, g_mag_hat, reg_coef, w.dmag, w.dphs,
                       w.aews, grads + L.offs[4], grads + L.offs[22], true, stream, &deferred, &sink, &sink_used, syn.wg ? &syn : nullptr, &syn_done,
                       (d->clip_all && ae_is_wide(d)) ? w.norm_e : nullptr, &w.n_norm_e));      // the forward left its AE state in w.aews
    if (syn.wg && !syn_done) {                         // the path taken had no launch to ride in (all-GEMM variant): the reduce as a launch of its own
        hipLaunchKernelGGL(stm::wgrad_reduce_kernel, dim3(st_norm_partials(d)), dim3(256), 0, st_stream(stream),
                           syn.wg, syn.nz, syn.gSr, syn.gSi, syn.norm_s, d->N, d->F, L.KP, 1, 0, 2 * d->F, (float*)nullptr, syn.nyq);
        ST_LAUNCHED("synthesis_wgrad_reduce");
    }
    if (syn.wg) syn_slabs = 0;
    if (sink_used) return ST_OK;                       // wide geometries: the polar backward ran inside wide_dv_polar_kernel
    if (!deferred) {
        ST_REQ(syn_slabs == 0, "internal: deferred synthesis slabs on a path without post_ae_kernel");
        if (w.g16) {                       // wide geometries: the polar backward is its own launch; d G goes out in the GEMM operand type
            const int R = d->B * d->T, KP = st_kp_of(d->F);
            hipLaunchKernelGGL(stm::polar_bwd_kernel, dim3((KP / 2 + 255) / 256, R), dim3(256), 0, st_stream(stream),
                               w.re, w.im, w.dmag, w.dphs, g_mag, (float*)nullptr, R, d->F, KP, gemm_ht(d->prec) == 2 ? 65504.0f : 0.0f, w.dG16, gemm_ht(d->prec));
            ST_LAUNCHED("polar_bwd"); return ST_OK;
        }
        return st_polar_bwd(d, w.re, w.im, w.dmag, w.dphs, g_mag, w.dG, stream);
    }
    stm::PostAeArgs a;
    a.ws = w.aews + 2 * ae_h4_floats(d); a.nparts = ae_use_split(d) ? ae_split_grid(d) : ae_bwd_grid(d); a.PG = L.PG; a.g_m = grads + L.offs[4]; a.g_p = grads + L.offs[22];
    a.n_red_x = (L.PG + 63) / 64; a.n_red = 2 * a.n_red_x;
    a.re = w.re; a.im = w.im; a.dmag = w.dmag; a.dphs = w.dphs; a.g_mag = g_mag; a.dG = w.dG; a.F = d->F; a.KP = L.KP;
    a.gx = (L.KP / 2 + 255) / 256; a.sat = gemm_ht(d->prec) == 2 ? 65504.0f : 0.0f;
    a.n_polar = a.gx * d->B * d->T;
    a.polar_total = 0; a.polar_magic = 0;
    if (stm::polar_flat_ok((long long)d->B * d->T, L.KP / 2)) {
        a.polar_total = (unsigned)((long long)d->B * d->T * (L.KP / 2)); a.polar_magic = stm::polar_magic(L.KP / 2);
        a.n_polar = (int)((a.polar_total + 256 * stm::POLAR_EPT - 1) / (256 * stm::POLAR_EPT));
    }
    a.wg = w.wg; a.wg_nz = syn_slabs; a.gSr = grads + L.offs[2]; a.gSi = grads + L.offs[3]; a.norm_s = w.norm_s; a.N = d->N;
    a.nyq = stm::NyqJob{}; a.nyq.on = 0; if (syn_nyq) a.nyq = *syn_nyq;
    a.dG16 = nullptr; a.ht = 0;
    if (w.g16) { a.dG16 = w.dG16; a.ht = gemm_ht(d->prec); a.dG = nullptr; }      // the analysis weight-gradient GEMM is the only consumer on this path
    a.norm_e = nullptr;
    if (d->clip_all && a.n_red <= NORM_E_MAX) { a.norm_e = w.norm_e; w.n_norm_e = a.n_red; }
    hipLaunchKernelGGL(stm::post_ae_kernel, dim3(a.n_red + a.n_polar + (syn_slabs > 0 ? st_norm_partials(d) : 0)), dim3(256), 0, st_stream(stream), a);
    ST_LAUNCHED("post_ae");
    return ST_OK;
}
static int backward_p1(const st_dims* d, const Layout& L, const float* params, float* grads,
                       const float* knobs, const float* g_mag_hat, const float* g_mag, float reg_coef, WS& w, void* stream)
{
    int syn_slabs = 0; stm::NyqJob syn_nyq{}; syn_nyq.on = 0;
    ST_TRY(backward_syn(d, L, grads, w, stream, &syn_slabs, &syn_nyq));      // the slab sum rides in a later launch: post_ae_kernel (fused geometries) / wide_grad_finish_kernel (wide ones)
    return backward_ae(d, L, params, grads, knobs, g_mag_hat, g_mag, reg_coef, w, stream, syn_slabs, &syn_nyq);
}
static int backward_p2(const st_dims* d, const Layout& L, float* grads, const float* x, WS& w, void* stream, float* stage = nullptr, int stage_bf16 = 0)
{
    (void)x;
    if (w.g16) {
        const stg::RowMap ma = stg::live_frames(d->T, d->H, d->N, d->N, d->L);
        const int R = ma.rows(d->B), KP = st_kp_of(d->F), ns = g16_wsplit(d, R);
        ST_TRY(wgrad16(d, w.dG16, (unsigned)(d->T * KP), w.xp16, (unsigned)(d->L + 2 * d->N), ma, R, w.wg, ns, stream, 0, -1, (g_g16_crop & 8) ? d->L : 0));
        ST_LAUNCHED("analysis_wgrad");
        stm::NyqJob nq{}; nq.on = 0;
        hipLaunchKernelGGL(stm::wgrad_reduce_kernel, dim3(st_norm_partials(d)), dim3(256), 0, st_stream(stream),
                           w.wg, ns, grads + L.offs[0], grads + L.offs[1], w.norm_a, d->N, d->F, KP, 0, 0, 2 * d->F, stage, nq, 0, stage_bf16);
        ST_LAUNCHED("analysis_wgrad_reduce");
        return ST_OK;
    }
    return analysis_wgrad_impl(d, w.dG, w.xp, true, 1.0f, w.wg, grads + L.offs[0], grads + L.offs[1], w.norm_a, stream, -1, stage, stage_bf16);
}
// ONE basis (half 0 = real, 1 = imaginary) of the analysis weight gradient -- its GEMM over that basis' rows and its slab reduce (gradient rows, packed
// stage rows [half * F, half * F + F)) -- so that the data-parallel exchange of the first basis runs under the GEMM of the second (st_dp_train_step,
// exchange flag 2): only F * N values (2.1 MB; 1.05 MB bf16-packed) stay exposed behind the last GEMM of the step.  Each half fills the chip with twice
// the k-slices of the whole-tensor launch (half the tiles), so the two launches together do the work of the one.  nyq_io carries the Nyquist partials'
// description (128 x 128-tile form: formed for BOTH bases by the first launch) from half 0 to half 1.
static int analysis_wgrad_half(const st_dims* d, const Layout& L, float* grads, WS& w, int half, float* stage, int stage_bf16, stm::NyqJob* nyq_io, void* stream)
{
    const int KP = st_kp_of(d->F), F = d->F, N = d->N, m0 = half ? KP / 2 : 0;
    const stg::RowMap ma = stg::live_frames(d->T, d->H, d->N, d->N, d->L);
    const int R = ma.rows(d->B);
    const int room = (int)((st_wgrad_ws_floats(d) - (size_t)64 * 2 * N) / ((size_t)KP * N));      // slabs the workspace holds (the Nyquist partials sit behind them)
    const int per = stm::nyq_blocks(N) / 2;
    int ns;
    const stg::TNOperand ta{w.dG + m0, (unsigned)(d->T * KP), (unsigned)KP}, tb{w.xp, (unsigned)(d->L + 2 * d->N), (unsigned)d->H};
    if (w.g16) {
        const int tiles = ((KP / 2 + 127) / 128) * (N / 128);
        ns = g_g16_split > 0 ? g_g16_split : (2 * num_cus()) / (tiles > 0 ? tiles : 1);
        if (ns > R / 128) ns = R / 128; if (ns > room) ns = room; if (ns < 1) ns = 1;
        ST_TRY(wgrad16(d, w.dG16, (unsigned)(d->T * KP), w.xp16, (unsigned)(d->L + 2 * d->N), ma, R, w.wg, ns, stream, m0, KP / 2, (g_g16_crop & 8) ? d->L : 0));
        if (!half) { *nyq_io = stm::NyqJob{}; nyq_io->on = 0; }
    } else if (use_tn128(d, true) && stg::tn128_fits(ta, tb, w.xp, ma, N / 2, N, (size_t)d->B * d->T * KP, (size_t)d->B * (d->L + 2 * d->N))) {
        const int mh = (N / 2) / 128, tiles = mh * (N / 128);
        ns = num_cus() / (tiles > 0 ? tiles : 1); if (ns > 16) ns = 16; if (ns > R / 64) ns = R / 64; if (ns > room) ns = room; if (ns < 1) ns = 1;
        float* part = w.wg + (size_t)room * KP * N;
        int P = 0;
        const TNFrameMajor fm = tn_frame_major(ta, tb, ma, d->B, d->H, N, N, d->L, (g_tn_fm & 2) != 0);
        // half 0 also forms the Nyquist partials of BOTH bases (its A origin is column 0: c0 = F - 1, c1 = KP / 2 + F - 1); half 1 has no Nyquist slice
        if (g_tn_bk == 16) ST_TRY((stg::launch_tn128<16>(fm.a, fm.b, w.xp, fm.map, R, N / 2, mh, (unsigned)(KP / 2), N, w.wg + (size_t)m0 * N, N, (size_t)KP * N, ns, st_stream(stream),
                                                          half ? nullptr : part, (unsigned)(F - 1), (unsigned)(KP / 2 + F - 1), &P, &fm.trim)));
        else ST_TRY((stg::launch_tn128<32>(fm.a, fm.b, w.xp, fm.map, R, N / 2, mh, (unsigned)(KP / 2), N, w.wg + (size_t)m0 * N, N, (size_t)KP * N, ns, st_stream(stream),
                                            half ? nullptr : part, (unsigned)(F - 1), (unsigned)(KP / 2 + F - 1), &P, &fm.trim)));
        if (!half) { nyq_io->part = part; nyq_io->P = P; nyq_io->on = 1; }
    } else {
        ST_LAUNCHED("analysis_wgrad");
        if (!half) { *nyq_io = stm::NyqJob{}; nyq_io->on = 0; }
        return analysis_wgrad_impl(d, w.dG, w.xp, true, 1.0f, w.wg, grads + L.offs[0], grads + L.offs[1], w.norm_a, stream, half, stage, stage_bf16);
    }
    ST_LAUNCHED("analysis_wgrad");
    hipLaunchKernelGGL(stm::wgrad_reduce_kernel, dim3(F + per), dim3(256), 0, st_stream(stream),
                       w.wg, ns, grads + L.offs[0], grads + L.offs[1], w.norm_a, N, F, KP, 0, half * F, F, stage, *nyq_io, half * per, stage_bf16);
    ST_LAUNCHED("analysis_wgrad_reduce");
    return ST_OK;
}
static int backward_impl(const st_dims* d, const Layout& L, const float* params, float* grads, const float* x,
                         const float* knobs, const float* g_mag_hat, const float* g_mag, float reg_coef, WS& w, void* stream)
{
    ST_TRY(backward_p1(d, L, params, grads, knobs, g_mag_hat, g_mag, reg_coef, w, stream));
    return backward_p2(d, L, grads, x, w, stream);
}

extern "C" int st_model_fwd(const st_dims* d, const float* params, const float* x, const float* knobs,
                            float* y_hat, float* mag, float* mag_hat, void* ws, int save_for_backward, void* stream)
{
    Layout L; ST_TRY(make_layout(d, &L));
    ST_REQ(params && x && (knobs || d->K == 0) && ws, "st_model_fwd: null pointer");
    if (!knobs) knobs = params;
    WS w; carve(d, ws, &w);
    return forward_impl(d, L, params, x, knobs, nullptr, y_hat, mag, mag_hat, w, save_for_backward != 0, stream);
}

extern "C" int st_model_bwd(const st_dims* d, const float* params, float* grads, const float* x, const float* knobs,
                            const float* g_y_hat, const float* g_mag_hat, const float* g_mag, void* ws, void* stream)
{
    Layout L; ST_TRY(make_layout(d, &L));
    ST_REQ(params && grads && x && (knobs || d->K == 0) && g_y_hat && ws, "st_model_bwd: null pointer");
    if (!knobs) knobs = params;
    WS w; carve(d, ws, &w);
    ST_TRY(pad_scale(g_y_hat, w.dsyn, d->B, d->y, d->N, 2.0f, stream));     // dsyn = 2 * g_y_hat, padded for the framed loaders
    return backward_impl(d, L, params, grads, x, knobs, g_mag_hat, g_mag, 0.0f, w, stream);
}

// d (anything downstream) / d knobs for arbitrary upstream gradients -- what autograd hands to a knobs tensor that requires grad (nn_proc.py:92-93: the
// knob settings are repeated over the rows of a window and concatenated in front of fnn_addknobs of BOTH autoencoders, nn_proc.py:332-333).  With
// d a5 the gradient at that layer's pre-activation, d knobs[b][k] = sum over nets, rows of window b, outputs o of  W5[o][16 + k] * d a5[row][o]: the row sum is
// exactly the layer's BIAS gradient of a batch that holds window b alone.  So: one forward + backward per window (B = 1), then 16 x K multiply-adds
// on the two bias gradients.  Deliberately the slow, exact route (B launch-bound passes, ~0.3 ms each): the reference's training never asks for this
// gradient (knobs are data), so the hot kernels carry no per-window reduction for it.  grads_scratch: L.total floats, overwritten; the saved-for-backward
// state of `ws` afterwards belongs to the LAST window -- run the forward again before st_model_bwd.  At geometries where a single window cannot take
// the requested 16-bit arithmetic (st_effective_prec: the wide autoencoder path, odd batch) these passes run the autoencoder layers in fp32.
__global__ void knob_grad_kernel(const float* __restrict__ Wm, const float* __restrict__ gbm, const float* __restrict__ Wp, const float* __restrict__ gbp,
                                 const int K, float* __restrict__ out)
{
    const int k = threadIdx.x;
    if (k >= K) return;
    float s = 0.f, t = 0.f;
    for (int o = 0; o < 16; ++o) { s += Wm[o * (16 + K) + 16 + k] * gbm[o]; t += Wp[o * (16 + K) + 16 + k] * gbp[o]; }
    out[k] = s + t;
}
extern "C" int st_model_knob_grad(const st_dims* d, const float* params, float* grads_scratch, const float* x, const float* knobs,
                                  const float* g_y_hat, const float* g_mag_hat, const float* g_mag, void* ws, float* g_knobs, void* stream)
{
    Layout L; ST_TRY(make_layout(d, &L));
    ST_REQ(params && grads_scratch && x && knobs && g_y_hat && ws && g_knobs, "st_model_knob_grad: null pointer");
    ST_REQ(d->K >= 1 && d->K <= 64, "st_model_knob_grad: K = %d", d->K);
    st_dims d1 = *d; d1.B = 1;
    Layout L1; ST_TRY(make_layout(&d1, &L1));
    const size_t w5 = (size_t)L1.offs[4] + (size_t)L1.go.w[4], b5 = (size_t)L1.offs[4] + (size_t)L1.go.b[4];
    for (int b = 0; b < d->B; ++b) {
        const float* xb = x + (size_t)b * d->L; const float* kb = knobs + (size_t)b * d->K;
        ST_TRY(st_model_fwd(&d1, params, xb, kb, nullptr, nullptr, nullptr, ws, 1, stream));
        ST_TRY(st_model_bwd(&d1, params, grads_scratch, xb, kb, g_y_hat + (size_t)b * d->y, g_mag_hat ? g_mag_hat + (size_t)b * d->OT * d->F : nullptr,
                            g_mag ? g_mag + (size_t)b * d->T * d->F : nullptr, ws, stream));
        hipLaunchKernelGGL(knob_grad_kernel, dim3(1), dim3(64), 0, st_stream(stream), params + w5, grads_scratch + b5, params + w5 + L1.PG, grads_scratch + b5 + L1.PG,
                           d->K, g_knobs + (size_t)b * d->K);
    }
    ST_LAUNCHED("knob_grad");
    return ST_OK;
}

extern "C" int st_loss_backward(const st_dims* d, const float* params, float* grads, const float* x, const float* knobs,
                                const float* y_true, float* y_hat, float* mag, float* mag_hat, void* ws,
                                float* scalars, void* stream)
{
    Layout L; ST_TRY(make_layout(d, &L));
    ST_REQ(params && grads && x && (knobs || d->K == 0) && y_true && ws && scalars, "st_loss_backward: null pointer");
    if (!knobs) knobs = params;
    WS w; carve(d, ws, &w);
    w.g16 = use_g16(d);
    prof_mark("begin", stream);
    ST_TRY(forward_impl(d, L, params, x, knobs, y_true, y_hat, mag, mag_hat, w, true, stream));
    const float reg_coef = loss_scale_of(d) * (float)(2e-5 / 10.0) / ((float)d->B * (float)d->OT * (float)d->F);   // loss_functions.py:36
    ST_TRY(backward_impl(d, L, params, grads, x, knobs, nullptr, nullptr, reg_coef, w, stream));
    // grads carry the loss scale (if any); the published norm is that of the unscaled gradient
    ST_TRY(st_finalize_scalars(d, w.loss_p, w.reg_p, w.norm_a, w.norm_s, 1.0f / loss_scale_of(d), scalars, stream));
    return ST_OK;
}

// Data-parallel split of st_loss_backward: after phase 1 the synthesis + autoencoder gradients
// (grads[offs[2] ..)) are final and can be all-reduced while phase 2 (analysis wgrad) runs.
extern "C" int st_loss_backward_p1(const st_dims* d, const float* params, float* grads, const float* x, const float* knobs,
                                   const float* y_true, void* ws, void* stream)
{
    Layout L; ST_TRY(make_layout(d, &L));
    ST_REQ(params && grads && x && (knobs || d->K == 0) && y_true && ws, "st_loss_backward_p1: null pointer");
    if (!knobs) knobs = params;
    WS w; carve(d, ws, &w);
    w.g16 = use_g16(d);
    prof_mark("begin", stream);
    ST_TRY(forward_impl(d, L, params, x, knobs, y_true, nullptr, nullptr, nullptr, w, true, stream));
    const float reg_coef = loss_scale_of(d) * (float)(2e-5 / 10.0) / ((float)d->B * (float)d->OT * (float)d->F);
    return backward_p1(d, L, params, grads, knobs, nullptr, nullptr, reg_coef, w, stream);
}
extern "C" int st_loss_backward_p2(const st_dims* d, float* grads, const float* x, void* ws, float* scalars, void* stream)
{
    Layout L; ST_TRY(make_layout(d, &L));
    ST_REQ(grads && x && ws && scalars, "st_loss_backward_p2: null pointer");
    WS w; carve(d, ws, &w);
    w.g16 = use_g16(d);
    ST_TRY(backward_p2(d, L, grads, x, w, stream));
    return st_finalize_scalars(d, w.loss_p, w.reg_p, w.norm_a, w.norm_s, 1.0f / loss_scale_of(d), scalars, stream);
}

// As st_loss_backward_p2, and the 2F live rows of the two analysis gradients are ALSO written packed into `stage` [2F][N]:
// the data-parallel all-reduce then moves 4.2 MB instead of the 6.3 MB contiguous range that spans the dead rows of the
// first tensor (st_unstage_analysis copies the reduced rows back).
extern "C" int st_loss_backward_p2_staged(const st_dims* d, float* grads, float* stage, const float* x, void* ws, float* scalars, void* stream)
{
    Layout L; ST_TRY(make_layout(d, &L));
    ST_REQ(grads && stage && x && ws && scalars, "st_loss_backward_p2_staged: null pointer");
    WS w; carve(d, ws, &w);
    w.g16 = use_g16(d);
    (void)scalars;      // the loss scalars are published by st_dp_clip_adam (no single-block finalize between this GEMM and the all-reduce)
    return backward_p2(d, L, grads, x, w, stream, stage);
}
extern "C" int st_unstage_analysis(const st_dims* d, float* grads, const float* stage, void* stream)
{
    Layout L; ST_TRY(make_layout(d, &L));
    ST_REQ(grads && stage, "st_unstage_analysis: null pointer");
    const size_t live = (size_t)d->F * d->N * sizeof(float);
    if (hipMemcpyAsync(grads + L.offs[0], stage, live, hipMemcpyDeviceToDevice, st_stream(stream)) != hipSuccess ||
        hipMemcpyAsync(grads + L.offs[1], stage + (size_t)d->F * d->N, live, hipMemcpyDeviceToDevice, st_stream(stream)) != hipSuccess)
        return st_fail(ST_ERR_LAUNCH, "st_unstage_analysis: copy failed");
    return ST_OK;
}

// Finer data-parallel split (dp.DataParallel): stage s leaves one gradient range final, in the order
//   0: forward + loss + synthesis dgrad/wgrad   -> the two synthesis bases           grads[offs[2], offs[4])
//   1: autoencoders + polar backward             -> both autoencoders                 grads[offs[4], total)
//   2: analysis weight gradient, real basis      -> rows [0,F) of the first tensor    grads[offs[0], +F*N)
//   3: analysis weight gradient, imaginary basis -> rows [0,F) of the second tensor   grads[offs[1], +F*N)  + scalars
// so each all-reduce runs under the next stage and only the last, smallest range (2.1 MB) is exposed.
extern "C" int st_loss_backward_stage(const st_dims* d, const float* params, float* grads, const float* x, const float* knobs,
                                      const float* y_true, void* ws, float* scalars, int stage, void* stream)
{
    Layout L; ST_TRY(make_layout(d, &L));
    ST_REQ(params && grads && x && (knobs || d->K == 0) && y_true && ws && scalars, "st_loss_backward_stage: null pointer");
    if (!knobs) knobs = params;
    ST_REQ(stage >= 0 && stage < 4, "st_loss_backward_stage: stage %d not in 0..3", stage);
    WS w; carve(d, ws, &w);
    const float reg_coef = loss_scale_of(d) * (float)(2e-5 / 10.0) / ((float)d->B * (float)d->OT * (float)d->F);
    switch (stage) {
    case 0:
        prof_mark("begin", stream);
        ST_TRY(forward_impl(d, L, params, x, knobs, y_true, nullptr, nullptr, nullptr, w, true, stream));
        return backward_syn(d, L, grads, w, stream);
    case 1:
        return backward_ae(d, L, params, grads, knobs, nullptr, nullptr, reg_coef, w, stream);
    case 2:
        return analysis_wgrad_impl(d, w.dG, w.xp, true, 1.0f, w.wg, grads + L.offs[0], grads + L.offs[1], w.norm_a, stream, 0);
    default:
        ST_TRY(analysis_wgrad_impl(d, w.dG, w.xp, true, 1.0f, w.wg, grads + L.offs[0], grads + L.offs[1], w.norm_a, stream, 1));
        return st_finalize_scalars(d, w.loss_p, w.reg_p, w.norm_a, w.norm_s, 1.0f / loss_scale_of(d), scalars, stream);
    }
}

static int train_step_impl(const st_dims* d, float* params, float* grads, float* m, float* v, const float* x,
                           const float* knobs, const float* y_true, void* ws, float* scalars,
                           float lr, float beta1, float beta2, float eps, int step, void* stream, bool dev_hyper);
extern "C" int st_train_step(const st_dims* d, float* params, float* grads, float* m, float* v, const float* x,
                             const float* knobs, const float* y_true, void* ws, float* scalars,
                             float lr, float beta1, float beta2, float eps, int step, void* stream)
{
    return train_step_impl(d, params, grads, m, v, x, knobs, y_true, ws, scalars, lr, beta1, beta2, eps, step, stream, false);
}
static int train_step_impl(const st_dims* d, float* params, float* grads, float* m, float* v, const float* x,
                           const float* knobs, const float* y_true, void* ws, float* scalars,
                           float lr, float beta1, float beta2, float eps, int step, void* stream, bool dev_hyper)
{
    Layout L; ST_TRY(make_layout(d, &L));
    ST_REQ(params && grads && x && (knobs || d->K == 0) && y_true && ws && scalars, "st_train_step: null pointer");
    if (!knobs) knobs = params;
    WS w; carve(d, ws, &w);
    w.g16 = use_g16(d);
    prof_mark("begin", stream);
    ST_TRY(forward_impl(d, L, params, x, knobs, y_true, nullptr, nullptr, nullptr, w, true, stream));
    const float reg_coef = loss_scale_of(d) * (float)(2e-5 / 10.0) / ((float)d->B * (float)d->OT * (float)d->F);   // loss_functions.py:36
    ST_TRY(backward_impl(d, L, params, grads, x, knobs, nullptr, nullptr, reg_coef, w, stream));
    // loss scalars + clip coefficient inside the optimizer kernel (st_loss_backward + st_clip_adam minus one launch)
    const float inv_s = 1.0f / loss_scale_of(d);            // the gradients carry the loss scale: unscale inside the optimizer
    stm::FinArgs f = fin_args(d, w.loss_p, w.reg_p, w.norm_a, w.norm_s, inv_s);
    if (d->clip_all) {                                       // train.py:136: the norm runs over every parameter
        if (w.n_norm_e > 0) { f.norm_e = w.norm_e; f.n_ne = w.n_norm_e; }      // the kernels that summed the autoencoder gradients left their |g| partials
        else {
            hipLaunchKernelGGL(stm::l1_partial_kernel, dim3(NORM_E_PARTIALS), dim3(256), 0, st_stream(stream),
                               grads + L.n_stft, L.total - L.n_stft, 1.0f, w.norm_e);
            ST_LAUNCHED("l1_partial_ae");
            f.norm_e = w.norm_e; f.n_ne = NORM_E_PARTIALS;
        }
    }
    return clip_adam_impl(params, grads, m, v, L.total, d->clip_all ? L.total : L.n_stft, scalars, inv_s, lr, beta1, beta2, eps, step, &f, stream, dev_hyper);
}

extern "C" int st_dp_clip_adam(const st_dims* d, float* params, float* grads, float* m, float* v, void* ws,
                               float* scalars, float grad_scale, float lr, float beta1, float beta2, float eps,
                               int step, void* stream)
{
    Layout L; ST_TRY(make_layout(d, &L));
    ST_REQ(params && grads && m && v && ws && scalars, "st_dp_clip_adam: null pointer");
    WS w; carve(d, ws, &w);
    w.g16 = use_g16(d);
    // L1 norm of the all-reduced, 1/world-scaled STFT gradient: identical on every rank, no second collective
    const int np = st_norm_partials(d);
    const int64_t n_clip = d->clip_all ? L.total : L.n_stft;
    const float gs = grad_scale / loss_scale_of(d);          // 1/world and the loss scale leave the gradient together
    hipLaunchKernelGGL(stm::l1_partial_kernel, dim3(np), dim3(256), 0, st_stream(stream),
                       grads, n_clip, gs, w.norm_a);
    ST_LAUNCHED("l1_partial");
    stm::FinArgs f = fin_args(d, w.loss_p, w.reg_p, w.norm_a, nullptr, 1.0f);     // this rank's loss terms + the norm of the REDUCED gradient (l1 partials above)
    f.n_na = np;
    return clip_adam_impl(params, grads, m, v, L.total, n_clip, scalars, gs, lr, beta1, beta2, eps, step, &f, stream);
}

// ------------------------------------------------------------------------------ device-side data feed
extern "C" int st_compressor_4c(const float* x, const float* knobs_wc, float sr, int B, int L, int ysz, float* y, void* stream)
{
    ST_REQ(x && knobs_wc && y, "st_compressor_4c: null pointer");
    ST_REQ(B > 0 && L > 0 && ysz > 0 && ysz <= L && sr > 0.f, "st_compressor_4c: bad sizes (B=%d L=%d ysz=%d)", B, L, ysz);
    hipLaunchKernelGGL(stm::compressor_4c_kernel, dim3(B), dim3(256), 0, st_stream(stream), x, knobs_wc, sr, L, ysz, y);
    ST_LAUNCHED("compressor_4c"); return ST_OK;
}

// scratch of st_synth_comp4c for the full-featured path: [gain curve B * L | world knobs 4 B] and, for power-of-two windows beyond the in-LDS FFT
// (8192 < L <= 65536), [1/f noise B * L | per-window peaks | four-step FFT buffer 2 * min(B, 1024) * L]
static const int FEED_FFT_SUB = 1024;      // windows per pass-1 / pass-2 launch pair of the long-window noise (a launch pair costs ~50 us whatever its size: few, large ones)
static bool feed_long_fft(int L) { return L > stf::FFT_MAX && (L & (L - 1)) == 0 && L / stf::PL_N1 <= 256; }
extern "C" size_t st_synth_comp4c_scratch_floats(int B, int L)
{
    if (B <= 0 || L <= 0) return 0;
    size_t n = (size_t)B * (L + 4);
    if (feed_long_fft(L)) n += (size_t)B * L + (size_t)st_round_up(B, 64) + (size_t)2 * (B < FEED_FFT_SUB ? B : FEED_FFT_SUB) * L;
    return n;
}
extern "C" int st_synth_comp4c(unsigned seed, unsigned long long first_window, int B, int L, int ysz, int K, float sr,
                               const float* knob_lo, const float* knob_hi, int augment, int chooser, const float* pink_in,
                               float* x, float* y, float* knobs, float* scratch, void* stream)
{
    ST_REQ(x && y && knobs && knob_lo && knob_hi, "st_synth_comp4c: null pointer");
    ST_REQ(B > 0 && L > 0 && ysz > 0 && ysz <= L && sr > 0.f && K == 4, "st_synth_comp4c: bad sizes (B=%d L=%d ysz=%d K=%d)", B, L, ysz, K);
    ST_REQ(chooser == -1 || chooser == 0 || chooser == 1 || chooser == 2 || chooser == 4 || chooser == 6 || chooser == 7 || chooser == 100, "st_synth_comp4c: signal family %d is not built (the compressor's set is 0,1,2,4,6,7)", chooser);
    const bool fft_ok = L <= stf::FFT_MAX && (L & (L - 1)) == 0;
    const bool long_fft = !fft_ok && !pink_in && scratch && feed_long_fft(L);      // scratch is then st_synth_comp4c_scratch_floats(B, L) floats (contract)
    ST_REQ(fft_ok || pink_in || long_fft, "st_synth_comp4c: a %d-sample window needs either the 1/f noise from the caller (pink_in) or -- powers of two up to 65536 -- "
           "st_synth_comp4c_scratch_floats() floats of scratch for the library's own transform", L);
    stf::FeedArgs a;
    a.x = x; a.y = y; a.knobs = knobs; a.pink_in = pink_in; a.pink_peak = nullptr; a.seed = seed; a.first = first_window;
    a.L = L; a.ysz = ysz; a.K = K; a.sr = sr; a.augment = augment; a.chooser = chooser;
    for (int k = 0; k < 4; ++k) { a.lo[k] = knob_lo[k]; a.hi[k] = knob_hi[k]; }
    const bool split = scratch && L % 64 == 0;
    a.gc = split ? scratch : nullptr; a.kw = split ? scratch + (size_t)B * L : nullptr;
    if (long_fft) {
        // the window's 1/f noise by the library's own four-step inverse FFT (st_feed.h pink_long_pass1 / 2), 1024 windows at a time through the FFT buffer
        float* pink = scratch + (size_t)B * (L + 4);
        float* peak = pink + (size_t)B * L;
        float2* fbuf = reinterpret_cast<float2*>(peak + st_round_up(B, 64));
        const int N2 = L / stf::PL_N1;
        for (int b0 = 0; b0 < B; b0 += FEED_FFT_SUB) {
            const int nb = B - b0 < FEED_FFT_SUB ? B - b0 : FEED_FFT_SUB;
            hipLaunchKernelGGL(stf::pink_long_pass1_kernel, dim3(stf::PL_N1 / stf::PL_G, nb), dim3(256), 0, st_stream(stream), seed, first_window + (unsigned long long)b0, L, chooser, fbuf, peak + b0);
            hipLaunchKernelGGL(stf::pink_long_pass2_kernel, dim3(N2 / stf::PL_G, nb), dim3(256), 0, st_stream(stream), seed, first_window + (unsigned long long)b0, L, chooser,
                               (const float2*)fbuf, pink + (size_t)b0 * L, peak + b0);
        }
        ST_LAUNCHED("pink_long");
        a.pink_in = pink; a.pink_peak = peak;
    }
    const size_t lds = a.pink_in ? (split ? 0 : (size_t)stm::COMP_CH * sizeof(float)) : (size_t)stf::FFT_MAX * sizeof(float2);
    if (lds >= 65536) {      // exactly 64 KB of dynamic LDS beside a few static bytes: ask for it explicitly (the 160 KB request of ensure_dyn_lds is refused for a kernel with static LDS)
        static std::mutex mu; static std::set<int> done;
        int dev = 0; (void)hipGetDevice(&dev);
        std::lock_guard<std::mutex> lk(mu);
        if (!done.count(dev)) {
            const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&stf::synth_comp4c_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return st_fail(ST_ERR_LAUNCH, "hipFuncSetAttribute(synth_comp4c_kernel, %zu B dynamic LDS): %s", lds, hipGetErrorString(e));
            done.insert(dev);
        }
    }
    hipLaunchKernelGGL(stf::synth_comp4c_kernel, dim3(B), dim3(256), lds, st_stream(stream), a);
    ST_LAUNCHED("synth_comp4c");
    if (split) {
        hipLaunchKernelGGL(stm::comp_smooth_kernel, dim3((B + 63) / 64), dim3(64), 0, st_stream(stream), a.gc, a.kw, sr, B, L);
        ST_LAUNCHED("comp_smooth");
        hipLaunchKernelGGL(stm::comp_apply_kernel, dim3((ysz + 255) / 256, B), dim3(256), 0, st_stream(stream), x, a.gc, L, ysz, y);
        ST_LAUNCHED("comp_apply");
    }
    return ST_OK;
}

// ------------------------------------------------------------------------------ generic learned-basis front end (a15)
// cls_fe_dct_bases.Analysis / Synthesis: Conv1d(1 -> C, k = KW, stride = hop, padding = pad, bias) and
// ConvTranspose1d(C -> 1, k = KW, stride = hop) + crop, forward and backward, on the same framed GEMM family.
static int fe_check(int B, int L, int C, int KW, int hop, int pad)
{
    ST_REQ(B > 0 && L > 0 && C > 0 && KW > 0 && hop > 0 && pad >= 0, "front end: non-positive dimension");
    ST_REQ(KW % 16 == 0 && C % 16 == 0 && hop % 4 == 0 && pad % 4 == 0 && L % 4 == 0, "front end: KW%%16, C%%16, hop%%4, pad%%4, L%%4 required");
    return ST_OK;
}
extern "C" int st_fe_frames(int L, int KW, int hop, int pad) { return (L + 2 * pad - KW) / hop + 1; }
extern "C" size_t st_fe_ws_floats(int B, int L, int C, int KW, int hop, int pad)
{
    const int T = st_fe_frames(L, KW, hop, pad);
    const size_t R = (size_t)B * T;
    // frames [R][KW] + padded gradient signal [B][L + 2 KW] + split-K slabs of the weight gradient
    return R * KW + (size_t)B * ((size_t)L + 2 * (size_t)KW + 2 * (size_t)pad) + (size_t)wgrad_split((int)R) * C * KW + 1024;
}
extern "C" int st_fe_analysis_fwd(const float* x, int B, int L, const float* W, const float* bias, int C, int KW, int hop, int pad,
                                  float* out, void* stream)
{
    ST_TRY(fe_check(B, L, C, KW, hop, pad)); ST_REQ(x && W && out, "st_fe_analysis_fwd: null pointer");
    const int T = st_fe_frames(L, KW, hop, pad), R = B * T;
    ST_REQ(T > 0, "st_fe_analysis_fwd: window longer than the padded signal");
    stg::FramedNT<false> al{x, L, hop, pad, R, KW, 1.0f, stg::all_frames(T)};
    stg::PlainNT bl{W, C, KW, KW, stg::all_frames(1)};
    stg::BiasStore ep{out, bias, R, C, C};
    stg::launch<2, 16>(al, bl, ep, R, C, KW, 1, st_stream(stream), g_dbg);
    ST_LAUNCHED("fe_analysis_fwd"); return ST_OK;
}
static int fe_frames_ola(const float* xft, int B, int T, const float* W, int C, int KW, int hop, int crop, int len,
                         float* frs, float* out, void* stream)
{
    const int R = B * T;
    stg::PlainNT al{xft, R, C, C, stg::all_frames(1)};
    stg::PlainTN bl{W, C, KW, KW, stg::all_frames(1)};
    stg::StoreC ep{frs, R, KW, KW, 0, stg::all_frames(1)};
    stg::launch<2, 16>(al, bl, ep, R, KW, C, 1, st_stream(stream), g_dbg);
    hipLaunchKernelGGL(stm::ola_crop_kernel, dim3((len + 255) / 256, B), dim3(256), 0, st_stream(stream), frs, out, T, KW, hop, crop, len);
    return ST_OK;
}
extern "C" int st_fe_synthesis_fwd(const float* xft, int B, int T, const float* W, int C, int KW, int hop, int crop,
                                   float* ws, float* out, void* stream)
{
    ST_REQ(xft && W && ws && out && B > 0 && T > 0, "st_fe_synthesis_fwd: bad arguments");
    ST_REQ(KW % 16 == 0 && C % 16 == 0 && hop % 4 == 0 && crop % 4 == 0, "st_fe_synthesis_fwd: KW%%16, C%%16, hop%%4, crop%%4 required");
    const int len = (T - 1) * hop + KW - 2 * crop;
    ST_REQ(len > 0, "st_fe_synthesis_fwd: nothing left after the crop");
    ST_TRY(fe_frames_ola(xft, B, T, W, C, KW, hop, crop, len, ws, out, stream));
    ST_LAUNCHED("fe_synthesis_fwd"); return ST_OK;
}
// d loss / d (x/2) of the whole model, for callers with something trainable UPSTREAM of st_model (the reference's autograd gives it for
// free; its own training never asks: x is data).  Must follow st_model_bwd on the same workspace: the analysis output gradient dG
// [B*T][KP] is still there.  gxh[b][n] = sum_t sum_k dG[b,t,k] W[k][n - (H t - N)]  (conv-transpose of both Conv1d's, cropped by the
// padding); the caller scales by the 1/2 of nn_proc.py:307 and adds the skip term of :340.  scratch: st_model_input_grad_ws_floats().
__global__ void __launch_bounds__(256)
wcat_kernel(const float* __restrict__ Wr, const float* __restrict__ Wi, float* __restrict__ out, const int F, const int N, const int KP)
{
    const int row = blockIdx.x, half = KP / 2;
    const bool is_im = row >= half; const int k = is_im ? row - half : row;
    for (int n = threadIdx.x; n < N; n += 256) out[(size_t)row * N + n] = k < F ? (is_im ? Wi : Wr)[(size_t)k * N + n] : 0.f;
}
extern "C" size_t st_model_input_grad_ws_floats(const st_dims* d)
{
    if (check_dims(d) != ST_OK) return 0;
    return (size_t)st_kp_of(d->F) * d->N + (size_t)d->B * d->T * d->N + 64;
}
extern "C" int st_model_input_grad(const st_dims* d, const float* params, void* ws, float* scratch, float* gxh, void* stream)
{
    Layout L; ST_TRY(make_layout(d, &L));
    ST_REQ(params && ws && scratch && gxh, "st_model_input_grad: null pointer");
    ST_REQ(d->L % 4 == 0 && d->H % 4 == 0, "st_model_input_grad: L %% 4 and hop %% 4 required");
    WS w; carve(d, ws, &w);
    float* Wc = scratch; float* frs = scratch + (size_t)L.KP * d->N;
    hipLaunchKernelGGL(wcat_kernel, dim3(L.KP), dim3(256), 0, st_stream(stream), params + L.offs[0], params + L.offs[1], Wc, d->F, d->N, L.KP);
    ST_TRY(fe_frames_ola(w.dG, d->B, d->T, Wc, L.KP, d->N, d->H, d->N, d->L, frs, gxh, stream));
    ST_LAUNCHED("model_input_grad");
    return ST_OK;
}
// autograd of st_fe_analysis_fwd: gW [C][KW], gbias [C] (may be null), gx [B][L] (may be null)
extern "C" int st_fe_analysis_bwd(const float* x, int B, int L, const float* W, int C, int KW, int hop, int pad, const float* g_out,
                                  float* ws, float* gW, float* gbias, float* gx, void* stream)
{
    ST_TRY(fe_check(B, L, C, KW, hop, pad)); ST_REQ(x && W && g_out && ws && gW, "st_fe_analysis_bwd: null pointer");
    const int T = st_fe_frames(L, KW, hop, pad), R = B * T;
    const int ns = wgrad_split(R);
    float* slabs = ws;                                       // [ns][C][KW]
    float* frs = ws + (size_t)ns * C * KW;                   // [R][KW]
    {
        stg::PlainTN al{g_out, R, C, C, stg::all_frames(1)};
        stg::FramedTN<false> bl{x, L, hop, pad, R, KW, 1.0f, stg::all_frames(T)};
        stg::StoreC ep{slabs, C, KW, KW, (size_t)C * KW, stg::all_frames(1)};
        stg::launch<3, 16>(al, bl, ep, C, KW, R, ns, st_stream(stream), g_dbg);
        const size_t n = (size_t)C * KW;
        hipLaunchKernelGGL(stm::sum_slabs_flat_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st_stream(stream), slabs, ns, n, gW);
    }
    if (gbias) hipLaunchKernelGGL(stm::col_sum_kernel, dim3((C + 255) / 256), dim3(256), 0, st_stream(stream), g_out, R, C, gbias);
    if (gx) ST_TRY(fe_frames_ola(g_out, B, T, W, C, KW, hop, pad, L, frs, gx, stream));   // conv-transpose of the output gradient, cropped by the padding
    ST_LAUNCHED("fe_analysis_bwd"); return ST_OK;
}
// autograd of st_fe_synthesis_fwd: gW [C][KW], g_xft [B][T][C] (may be null)
extern "C" int st_fe_synthesis_bwd(const float* xft, int B, int T, const float* W, int C, int KW, int hop, int crop, const float* g_wave,
                                   float* ws, float* gW, float* g_xft, void* stream)
{
    ST_REQ(xft && W && g_wave && ws && gW && B > 0 && T > 0, "st_fe_synthesis_bwd: bad arguments");
    ST_REQ(KW % 16 == 0 && C % 16 == 0 && hop % 4 == 0 && crop % 4 == 0, "st_fe_synthesis_bwd: KW%%16, C%%16, hop%%4, crop%%4 required");
    const int len = (T - 1) * hop + KW - 2 * crop, R = B * T;
    ST_REQ(len > 0 && len % 4 == 0, "st_fe_synthesis_bwd: bad output length %d", len);
    const int ns = wgrad_split(R);
    float* slabs = ws;                                       // [ns][C][KW]
    float* gp = ws + (size_t)ns * C * KW;                    // padded gradient signal [B][crop + len + crop]: its frames are d(frames)
    ST_TRY(pad_scale(g_wave, gp, B, len, crop, 1.0f, stream));
    if (g_xft) {
        stg::FramedNT<true> al{gp, len, hop, crop, R, KW, 1.0f, stg::all_frames(T)};
        stg::PlainNT bl{W, C, KW, KW, stg::all_frames(1)};
        stg::StoreC ep{g_xft, R, C, C, 0, stg::all_frames(1)};
        stg::launch<2, 16>(al, bl, ep, R, C, KW, 1, st_stream(stream), g_dbg);
    }
    {
        stg::PlainTN al{xft, R, C, C, stg::all_frames(1)};
        stg::FramedTN<true> bl{gp, len, hop, crop, R, KW, 1.0f, stg::all_frames(T)};
        stg::StoreC ep{slabs, C, KW, KW, (size_t)C * KW, stg::all_frames(1)};
        stg::launch<3, 16>(al, bl, ep, C, KW, R, ns, st_stream(stream), g_dbg);
        const size_t n = (size_t)C * KW;
        hipLaunchKernelGGL(stm::sum_slabs_flat_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st_stream(stream), slabs, ns, n, gW);
    }
    ST_LAUNCHED("fe_synthesis_bwd"); return ST_OK;
}

// ------------------------------------------------------------------------------ data parallel (st_dp.h): RCCL inside the library
extern "C" int st_dp_unique_id(void* id128)
{
    ST_REQ(id128, "st_dp_unique_id: null buffer");
    st_dp tmp; memset(&tmp, 0, sizeof(tmp));
    ST_TRY(stdp::bind(&tmp));
    ncclUniqueId id;
    ST_NCCL(&tmp, tmp.GetUniqueId(&id), "ncclGetUniqueId");
    static_assert(sizeof(id) == 128, "RCCL unique id is 128 bytes");
    memcpy(id128, &id, sizeof(id));
    return ST_OK;
}

extern "C" int st_dp_init(const void* id128, int rank, int world, st_dp** out)
{
    ST_REQ(id128 && out && world >= 1 && rank >= 0 && rank < world, "st_dp_init: bad arguments (rank %d of %d)", rank, world);
    st_dp* p = new st_dp; memset(p, 0, sizeof(*p));
    int rc = stdp::bind(p);
    if (rc != ST_OK) { delete p; return rc; }
    ncclUniqueId id; memcpy(&id, id128, sizeof(id));
    p->rank = rank; p->world = world;
    const ncclResult_t r = p->CommInitRank(&p->comm, world, id, rank);       // collective over all ranks: every rank must call this
    if (r != ncclSuccess) { rc = st_fail(ST_ERR_LAUNCH, "st_dp ncclCommInitRank(rank %d of %d): %s", rank, world, p->GetErrorString(r)); delete p; return rc; }
    hipError_t e = hipStreamCreateWithFlags(&p->cs, hipStreamNonBlocking);
    for (int i = 0; i < 4 && e == hipSuccess; ++i) e = hipEventCreateWithFlags(&p->ready[i], hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&p->done, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&p->wgfree, hipEventDisableTiming);
    if (e != hipSuccess) { rc = st_fail(ST_ERR_LAUNCH, "st_dp_init: stream/event creation: %s", hipGetErrorString(e)); (void)p->CommDestroy(p->comm); delete p; return rc; }
    *out = p;
    return ST_OK;
}

extern "C" int st_dp_destroy(st_dp* p)
{
    if (!p) return ST_OK;
    (void)hipStreamSynchronize(p->cs);
    if (p->comm) (void)p->CommDestroy(p->comm);
    for (int i = 0; i < 4; ++i) if (p->ready[i]) (void)hipEventDestroy(p->ready[i]);
    if (p->done) (void)hipEventDestroy(p->done);
    if (p->wgfree) (void)hipEventDestroy(p->wgfree);
    if (p->cs) (void)hipStreamDestroy(p->cs);
    delete p;
    return ST_OK;
}
extern "C" int st_dp_rank(const st_dp* p) { return p ? p->rank : -1; }
extern "C" int st_dp_world(const st_dp* p) { return p ? p->world : -1; }
// ncclGetVersion of the library the communicator is bound to (e.g. 22606), 0 if it does not export one: bench.py prints it beside the world size every
// rank reports, so that a multi-GPU line carries its own evidence of what the ranks met on
extern "C" int st_dp_rccl_version(const st_dp* p) { int v = 0; if (p && p->GetVersion && p->GetVersion(&v) == ncclSuccess) return v; return 0; }

// buf (n floats, in place, SUM) is all-reduced on the communicator stream once everything issued so far on `stream` has
// finished; returns at once.  st_dp_sync makes `stream` wait for all collectives issued since the last sync.
extern "C" int st_dp_allreduce(st_dp* p, float* buf, int64_t n, void* stream)
{
    ST_REQ(p && buf && n > 0, "st_dp_allreduce: bad arguments");
    hipEvent_t ev = p->ready[p->n_issued & 3]; p->n_issued++;
    ST_HIP(hipEventRecord(ev, st_stream(stream)), "event record");
    ST_HIP(hipStreamWaitEvent(p->cs, ev, 0), "stream wait");
    ST_NCCL(p, p->AllReduce(buf, buf, (size_t)n, ncclFloat32, ncclSum, p->comm, p->cs), "ncclAllReduce");
    return ST_OK;
}
// ... of `n` elements of `dt` on the communicator stream WITHOUT a new compute -> comm dependency (the caller has ordered the stream already)
static int dp_allreduce_on_cs(st_dp* p, void* buf, int64_t n, ncclDataType_t dt)
{
    ST_NCCL(p, p->AllReduce(buf, buf, (size_t)n, dt, ncclSum, p->comm, p->cs), "ncclAllReduce");
    return ST_OK;
}
// compute -> comm: everything issued so far on `stream` precedes what is issued on the communicator stream from here on
static int dp_fork(st_dp* p, void* stream)
{
    hipEvent_t ev = p->ready[p->n_issued & 3]; p->n_issued++;
    ST_HIP(hipEventRecord(ev, st_stream(stream)), "event record");
    ST_HIP(hipStreamWaitEvent(p->cs, ev, 0), "stream wait");
    return ST_OK;
}
extern "C" int st_dp_broadcast(st_dp* p, float* buf, int64_t n, int root, void* stream)
{
    ST_REQ(p && buf && n > 0 && root >= 0 && root < p->world, "st_dp_broadcast: bad arguments");
    hipEvent_t ev = p->ready[p->n_issued & 3]; p->n_issued++;
    ST_HIP(hipEventRecord(ev, st_stream(stream)), "event record");
    ST_HIP(hipStreamWaitEvent(p->cs, ev, 0), "stream wait");
    ST_NCCL(p, p->Broadcast(buf, buf, (size_t)n, ncclFloat32, root, p->comm, p->cs), "ncclBroadcast");
    return ST_OK;
}
extern "C" int st_dp_sync(st_dp* p, void* stream)
{
    ST_REQ(p, "st_dp_sync: null communicator");
    ST_HIP(hipEventRecord(p->done, p->cs), "event record");
    ST_HIP(hipStreamWaitEvent(st_stream(stream), p->done, 0), "stream wait");
    return ST_OK;
}

// One data-parallel optimisation step (train.py:112-151 on this rank's shard + the exchange of SURVEY.md 8e), all from C:
//   phase 1 (forward ... autoencoder / polar backward)            -> all-reduce grads[offs[2], total)  (8.45 MB)  || phase 2
//   phase 2 (analysis weight gradient, live rows packed in stage) -> all-reduce stage [2F][N]          (4.2 MB, exposed)
//   copy the reduced rows back, L1 norm of the reduced gradient, clip, Adam with grad_scale = 1/world.
// `stage`: caller-owned 2*F*N floats.  p == NULL or world == 1 falls through to st_train_step (no exchange).
extern "C" int st_dp_train_step(st_dp* p, const st_dims* d, float* params, float* grads, float* m, float* v, float* stage,
                                const float* x, const float* knobs, const float* y_true, void* ws, float* scalars,
                                float lr, float beta1, float beta2, float eps, int step, int force_exchange, void* stream)
{
    if (!p || (p->world == 1 && !(force_exchange & 1)))
        return st_train_step(d, params, grads, m, v, x, knobs, y_true, ws, scalars, lr, beta1, beta2, eps, step, stream);
    Layout L; ST_TRY(make_layout(d, &L));
    ST_REQ(stage, "st_dp_train_step: null staging buffer");
    ST_REQ(params && grads && x && (knobs || d->K == 0) && y_true && ws, "st_dp_train_step: null pointer");
    if (!knobs) knobs = params;
    const float gs = (1.0f / (float)p->world) / loss_scale_of(d);          // 1/world and the loss scale leave the gradient together
    // force_exchange is a bit set: 1 = run the exchange even with one rank; 2 = split the LAST exchange by basis (real rows under the GEMM of the imaginary
    // ones: 2.1 MB exposed instead of 4.2); 4 = that exchange on bfloat16 values (only where the autoencoder layers already run in 16 bits: *_ALL)
    const bool split_last = (force_exchange & 2) != 0;
    const int pack16 = ((force_exchange & 4) != 0 && ae_ht(d->prec) != 0) ? 1 : 0;
    // Exchanges, each issued the moment its gradients are final (the communicator stream orders them):
    //   synthesis bases (8.4 MB)  -- after their weight-gradient GEMM, BEFORE the autoencoder backward: hidden behind the longest
    //                                kernels of the step (autoencoder backward + polar backward + analysis weight gradient); the sum of
    //                                the GEMM's split-K slabs itself runs ON the communicator stream, beside the autoencoder backward
    //                                (round 3 ran it in line: a 9 us launch on the critical path that the single-GPU step folds into post_ae_kernel);
    //   autoencoders (67 KB)      -- after the autoencoder backward;
    //   analysis bases            -- the 2F live rows, packed (4.2 MB), after the last GEMM of the step: the exposed one (split / packed: see the flags).
    WS w; carve(d, ws, &w);
    w.g16 = use_g16(d);
    {
        prof_mark("begin", stream);
        ST_TRY(forward_impl(d, L, params, x, knobs, y_true, nullptr, nullptr, nullptr, w, true, stream));
        const float reg_coef = loss_scale_of(d) * (float)(2e-5 / 10.0) / ((float)d->B * (float)d->OT * (float)d->F);
        int syn_slabs = 0; stm::NyqJob syn_nyq{}; syn_nyq.on = 0;
        float* const wg_main = w.wg;
        if (g_dp_inline) w.wg = w.wg2;                                                  // round 6: the synthesis slabs in their own area (no wait before the analysis GEMM reuses the first)
        ST_TRY(backward_syn(d, L, grads, w, stream, &syn_slabs, &syn_nyq));             // GEMMs only; the slabs are summed on the communicator stream
        ST_TRY(dp_fork(p, stream));
        hipLaunchKernelGGL(stm::wgrad_reduce_kernel, dim3(st_norm_partials(d)), dim3(256), 0, p->cs,
                           w.wg, syn_slabs, grads + L.offs[2], grads + L.offs[3], w.norm_s, d->N, d->F, L.KP, 1, 0, 2 * d->F, (float*)nullptr, syn_nyq);
        w.wg = wg_main;
        if (!g_dp_inline) ST_HIP(hipEventRecord(p->wgfree, p->cs), "event record");     // the analysis weight-gradient GEMM reuses the slab buffer
        ST_TRY(dp_allreduce_on_cs(p, grads + L.offs[2], L.offs[4] - L.offs[2], ncclFloat32));
        // The clip norm is that of the REDUCED, 1/world-scaled gradient.  The two ranges whose exchange is hidden get their |g| partials on the communicator
        // stream right behind their collective (hidden as well); only the analysis rows' share is formed after the exposed collective, in the pass that
        // copies them back (unstage_l1_kernel).  Same kernels, same order, same data on every rank: the norm is bit-identical across ranks.
        hipLaunchKernelGGL(stm::l1_partial_kernel, dim3(st_norm_partials(d)), dim3(256), 0, p->cs, grads + L.offs[2], L.offs[4] - L.offs[2], gs, w.norm_s);
        ST_TRY(backward_ae(d, L, params, grads, knobs, nullptr, nullptr, reg_coef, w, stream, 0, nullptr));
        ST_TRY(st_dp_allreduce(p, grads + L.offs[4], L.total - L.offs[4], stream));
        if (d->clip_all) hipLaunchKernelGGL(stm::l1_partial_kernel, dim3(NORM_E_PARTIALS), dim3(256), 0, p->cs, grads + L.n_stft, L.total - L.n_stft, gs, w.norm_e);
        if (!g_dp_inline) ST_HIP(hipStreamWaitEvent(st_stream(stream), p->wgfree, 0), "stream wait");
    }
    const int64_t half_n = (int64_t)d->F * d->N;
    bool last_in_line = false;
    if (!split_last && g_dp_inline) {
        // Round 6: the LAST exchange is exposed whatever stream it runs on (nothing is left to run beside it), so it is issued IN LINE on the compute stream:
        // the two cross-stream hand-offs around it (compute -> communicator before, communicator -> compute after: ~10 us each on this stack, measured with one
        // rank as a 21 us hole in front of the optimizer kernel) become ONE join that is recorded HERE, behind the last hidden collective, and has fired long
        // before the GEMM below ends.  RCCL serialises the collectives of one communicator in issue order across streams; every rank issues the same order.
        ST_HIP(hipEventRecord(p->done, p->cs), "event record");
        ST_TRY(backward_p2(d, L, grads, x, w, stream, stage, pack16));
        ST_HIP(hipStreamWaitEvent(st_stream(stream), p->done, 0), "stream wait");
        ST_NCCL(p, p->AllReduce(stage, stage, (size_t)(2 * half_n), pack16 ? ncclBfloat16 : ncclFloat32, ncclSum, p->comm, st_stream(stream)), "ncclAllReduce");
        last_in_line = true;
    } else if (!split_last) {
        ST_TRY(backward_p2(d, L, grads, x, w, stream, stage, pack16));
        ST_TRY(dp_fork(p, stream));
        ST_TRY(dp_allreduce_on_cs(p, stage, 2 * half_n, pack16 ? ncclBfloat16 : ncclFloat32));
    } else {
        stm::NyqJob nyq{}; nyq.on = 0;
        for (int h = 0; h < 2; ++h) {
            ST_TRY(analysis_wgrad_half(d, L, grads, w, h, stage, pack16, &nyq, stream));
            void* part = pack16 ? (void*)(reinterpret_cast<unsigned short*>(stage) + (size_t)h * half_n) : (void*)(stage + (size_t)h * half_n);
            if (h == 1 && g_dp_inline) {        // the second half's exchange is the exposed one: in line, behind the join recorded when the first half's was issued
                ST_HIP(hipStreamWaitEvent(st_stream(stream), p->done, 0), "stream wait");
                ST_NCCL(p, p->AllReduce(part, part, (size_t)half_n, pack16 ? ncclBfloat16 : ncclFloat32, ncclSum, p->comm, st_stream(stream)), "ncclAllReduce");
                last_in_line = true;
                break;
            }
            ST_TRY(dp_fork(p, stream));
            ST_TRY(dp_allreduce_on_cs(p, part, half_n, pack16 ? ncclBfloat16 : ncclFloat32));
            if (g_dp_inline) ST_HIP(hipEventRecord(p->done, p->cs), "event record");
        }
    }
    if (!last_in_line) ST_TRY(st_dp_sync(p, stream));
    {
        const int np = st_norm_partials(d);
        hipLaunchKernelGGL(stm::unstage_l1_kernel, dim3(2 * d->F), dim3(256), 0, st_stream(stream), stage, grads + L.offs[0], grads + L.offs[1], d->F, d->N, gs, w.norm_a, np, pack16);
        ST_LAUNCHED("unstage_l1");
        stm::FinArgs f = fin_args(d, w.loss_p, w.reg_p, w.norm_a, w.norm_s, 1.0f);
        if (d->clip_all) { f.norm_e = w.norm_e; f.n_ne = NORM_E_PARTIALS; }
        return clip_adam_impl(params, grads, m, v, L.total, d->clip_all ? L.total : L.n_stft, scalars, gs, lr, beta1, beta2, eps, step, &f, stream);
    }
}


// ------------------------------------------------------------------------------ the whole step as one HIP graph
// st_train_step captured once and replayed: 11 kernel nodes + the step-tick head.  What changes from iteration to iteration --
// the step number (Adam's bias corrections) and the learning rate (1-cycle table, train.py:108,150) -- lives on the device
// (scalars[6], scalars[7]; the table is a caller-owned device array), so the graph needs no per-step update; the minibatch is read
// from the fixed device buffers x / knobs / y_true the graph was captured with (the caller refills them, e.g. by an index gather
// from a device-resident dataset).  On this path the launch work of a step is one hipGraphLaunch; the GPU-side cost of the
// kernel boundaries themselves is the same as for eager launches (MI355X_MICROARCH.md "boundary": eager == hipGraph).
struct st_graph { hipGraph_t graph; hipGraphExec_t exec; };

static int attr_prepare(const st_dims* d)
{
    // hipFuncSetAttribute is not a capturable call: make sure every >64 KB-LDS kernel this geometry / precision uses has its
    // attribute before the capture starts (ensure_dyn_lds is then a table hit inside the captured calls)
    const int ht = ae_ht(d->prec);
#define ST_PREP3(K0_, K1_, K2_) do { if (ht == 1) ST_DYN_LDS(K1_); else if (ht == 2) ST_DYN_LDS(K2_); else ST_DYN_LDS(K0_); } while (0)
    if (ae_is_wide(d)) {
        ST_PREP3((sta::ae_inner_fwd_kernel<AE_FWD_NW, 0>), (sta::ae_inner_fwd_kernel<AE_FWD_NW, 1>), (sta::ae_inner_fwd_kernel<AE_FWD_NW, 2>));
        ST_PREP3((sta::ae_inner_fwd_kernel<9, 0>), (sta::ae_inner_fwd_kernel<9, 1>), (sta::ae_inner_fwd_kernel<9, 2>));
        ST_PREP3((sta::ae_inner_fwd_kernel<12, 0>), (sta::ae_inner_fwd_kernel<12, 1>), (sta::ae_inner_fwd_kernel<12, 2>));
        ST_PREP3((sta::ae_bwd_kernel<AE_BWD_NW, false, true, 0, 0>), (sta::ae_bwd_kernel<AE_BWD_NW, false, true, 1, 0>), (sta::ae_bwd_kernel<AE_BWD_NW, false, true, 2, 0>));
        ST_PREP3((stw::wide_dv_polar_kernel<0>), (stw::wide_dv_polar_kernel<1>), (stw::wide_dv_polar_kernel<2>));
    } else {
        ST_PREP3((sta::ae_fwd_kernel<AE_FWD_NW, 0>), (sta::ae_fwd_kernel<AE_FWD_NW, 1>), (sta::ae_fwd_kernel<AE_FWD_NW, 2>));
        if (ht == 1) ST_DYN_LDS((sta::ae_fwd32_kernel<AE_FWD_NW, 1>)); else if (ht == 2) ST_DYN_LDS((sta::ae_fwd32_kernel<AE_FWD_NW, 2>));
        if (ht == 0) { ST_DYN_LDS((sta::ae_fwd_kernel<11, 0>)); ST_DYN_LDS((sta::ae_fwd_kernel<11, 0, true>)); ST_DYN_LDS((sta::ae_fwd_kernel<AE_FWD_NW, 0, true>));
                       ST_DYN_LDS((sta::ae_bwd_kernel<AE_BWD_NW, false, false, 0, 0, true>)); ST_DYN_LDS((sta::ae_bwd_kernel<AE_BWD_NW, false, false, 0, 1, true>)); ST_DYN_LDS((sta::ae_bwd_kernel<AE_BWD_NW, false, false, 0, 2, true>)); }
        ST_PREP3((sta::ae_bwd_part_kernel<AE_SPLIT_NW, 1, 0, false>), (sta::ae_bwd_part_kernel<AE_SPLIT_NW, 1, 1, false>), (sta::ae_bwd_part_kernel<AE_SPLIT_NW, 1, 2, false>));
        ST_PREP3((sta::ae_bwd_part_kernel<AE_SPLIT_NW, 2, 0, false>), (sta::ae_bwd_part_kernel<AE_SPLIT_NW, 2, 1, false>), (sta::ae_bwd_part_kernel<AE_SPLIT_NW, 2, 2, false>));
        if (d->T - d->OT == 16) ST_PREP3((sta::ae_bwd_kernel<AE_BWD_NW, false, false, 0, 2>), (sta::ae_bwd_kernel<AE_BWD_NW, false, false, 1, 2>), (sta::ae_bwd_kernel<AE_BWD_NW, false, false, 2, 2>));
        else ST_PREP3((sta::ae_bwd_kernel<AE_BWD_NW, false, false, 0, 0>), (sta::ae_bwd_kernel<AE_BWD_NW, false, false, 1, 0>), (sta::ae_bwd_kernel<AE_BWD_NW, false, false, 2, 0>));
    }
#undef ST_PREP3
    if (g_nt128 && gemm_ht(d->prec) == 0) ST_DYN_LDS((stg::gemm_nt128_kernel));
    if (use_g16(d)) {           // st_gemm16.h: 72 / 80 KB of LDS with 64-deep k-tiles
        if (gemm_ht(d->prec) == 2) {
            ST_DYN_LDS((stg::gemm16_nt256_kernel<2, stg::PolarStore>)); ST_DYN_LDS((stg::gemm16_nt256_kernel<2, stg::StoreC>));
            ST_DYN_LDS((stg::gemm16_nt_kernel<2, 64, stg::PolarStore>)); ST_DYN_LDS((stg::gemm16_nt_kernel<2, 64, stg::StoreC>)); ST_DYN_LDS((stg::gemm16_tn_kernel<2, 64>));
        } else {
            ST_DYN_LDS((stg::gemm16_nt256_kernel<1, stg::PolarStore>)); ST_DYN_LDS((stg::gemm16_nt256_kernel<1, stg::StoreC>));
            ST_DYN_LDS((stg::gemm16_nt_kernel<1, 64, stg::PolarStore>)); ST_DYN_LDS((stg::gemm16_nt_kernel<1, 64, stg::StoreC>)); ST_DYN_LDS((stg::gemm16_tn_kernel<1, 64>));
        }
    }
    if (use_planes(d)) {        // the 4-wave plane GEMM carries 67 KB of LDS (st_gemm_planes.h)
        ST_DYN_LDS((stg::gemm_planes_kernel<4, 3, 1, stg::FramedNT<true>, stg::ChunkP, stg::PolarStore>));
        ST_DYN_LDS((stg::gemm_planes_kernel<8, 3, 1, stg::FramedNT<true>, stg::ChunkP, stg::PolarStore>));
        ST_DYN_LDS((stg::gemm_planes_kernel<4, 3, 1, stg::PlainNT, stg::ChunkP, stg::StoreC>));
        ST_DYN_LDS((stg::gemm_planes_kernel<4, 3, 1, stg::FramedNT<true>, stg::ChunkP, stg::StoreC>));
    }
    (void)num_cus();
    return ST_OK;
}

extern "C" int st_graph_create(const st_dims* d, float* params, float* grads, float* m, float* v, const float* x,
                               const float* knobs, const float* y_true, void* ws, float* scalars,
                               const float* lr_table, int n_lr, float beta1, float beta2, float eps, void* stream, st_graph** out)
{
    Layout L; ST_TRY(make_layout(d, &L));
    ST_REQ(params && grads && m && v && x && (knobs || d->K == 0) && y_true && ws && scalars && lr_table && n_lr > 0 && out, "st_graph_create: bad arguments");
    if (!knobs) knobs = params;
    ST_REQ(!g_prof, "st_graph_create: switch the event profiling off first (event records would be captured)");
    ST_TRY(attr_prepare(d));
    hipStream_t s = st_stream(stream);
    if (hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal) != hipSuccess) return st_fail(ST_ERR_LAUNCH, "st_graph_create: hipStreamBeginCapture failed (is `stream` the legacy default stream?)");
    hipLaunchKernelGGL(stm::step_tick_kernel, dim3(1), dim3(64), 0, s, scalars, lr_table, n_lr);
    int rc = train_step_impl(d, params, grads, m, v, x, knobs, y_true, ws, scalars, 0.f, beta1, beta2, eps, 1, stream, true);
    hipGraph_t g = nullptr;
    const hipError_t e = hipStreamEndCapture(s, &g);
    if (rc != ST_OK) { if (g) (void)hipGraphDestroy(g); return rc; }
    if (e != hipSuccess || !g) return st_fail(ST_ERR_LAUNCH, "st_graph_create: capture failed: %s", hipGetErrorString(e));
    hipGraphExec_t ex = nullptr;
    if (hipGraphInstantiate(&ex, g, nullptr, nullptr, 0) != hipSuccess) { (void)hipGraphDestroy(g); return st_fail(ST_ERR_LAUNCH, "st_graph_create: hipGraphInstantiate failed"); }
    st_graph* p = new st_graph{g, ex};
    *out = p;
    return ST_OK;
}
extern "C" int st_graph_launch(st_graph* g, void* stream)
{
    ST_REQ(g && g->exec, "st_graph_launch: null graph");
    const hipError_t e = hipGraphLaunch(g->exec, st_stream(stream));
    if (e != hipSuccess) return st_fail(ST_ERR_LAUNCH, "hipGraphLaunch: %s", hipGetErrorString(e));
    return ST_OK;
}
extern "C" int st_graph_destroy(st_graph* g)
{
    if (!g) return ST_OK;
    if (g->exec) (void)hipGraphExecDestroy(g->exec);
    if (g->graph) (void)hipGraphDestroy(g->graph);
    delete g;
    return ST_OK;
}
