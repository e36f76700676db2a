/* signaltrain_hip.h -- C ABI of libsignaltrain_hip.so (MI355X / gfx950).
 *
 * The reference (drscotthawley/signaltrain) has no FFI / operator registry: its hot path is
 * plain PyTorch modules.  This header is therefore the boundary a maintainer
 * would bind (ctypes stub in INTEGRATION.md); every entry point names the reference code it
 * replaces.  Conventions:
 *   - all pointers are DEVICE pointers to contiguous fp32 unless stated; the caller (PyTorch)
 *     owns every buffer, including workspaces and saved activations;
 *   - every compute entry is asynchronous on `stream` (a hipStream_t passed as void*), does
 *     no allocation and no host synchronisation;
 *   - return value: 0 = ok, <0 = error; message via st_last_error();
 *   - shapes use the reference's names: B windows, L samples/window, N=ft_size, H=hop,
 *     T input frames, OT output frames, F=N/2+1 bins, K knobs, y output samples.
 */
#ifndef SIGNALTRAIN_HIP_H
#define SIGNALTRAIN_HIP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define ST_OK 0
#define ST_ERR_ARG (-1)      /* bad dimension / null pointer / misalignment */
#define ST_ERR_LAUNCH (-2)   /* HIP launch error */
#define ST_ERR_UNSUPPORTED (-3)

/* Arithmetic of a call (st_dims.prec).  Carried per call: two engines of different precision can share a process.
 *   ST_PREC_F32       fp32 MFMA everywhere -- the parity path (<= 1e-4 of the fp32 reference).
 *   ST_PREC_BF16      bf16 operands, fp32 accumulation in the analysis / synthesis GEMMs and their data / weight gradients
 *                     (BASELINE.json configs[2], [3]); operands are rounded (RNE) as they are staged, parameters,
 *                     activations, gradients, autoencoders, loss and Adam stay fp32.
 *   ST_PREC_BF16_ALL  additionally the nine Linear layers of both autoencoders (nn_proc.py:84-117 and their autograd):
 *                     weights and layer inputs / incoming gradients rounded to bf16, fp32 accumulation; bias, ELU, ELU',
 *                     skip / residual epilogue fp32.
 *   ST_PREC_F16, ST_PREC_F16_ALL   the same two levels with IEEE float16 operands (BASELINE.json configs[4], the
 *                     reference's Apex amp path, train.py:254-255), meant to run with st_dims.loss_scale > 1
 *                     (train.py:134-135 amp.scale_loss).  Conversion is IEEE: an out-of-range operand becomes inf, the
 *                     gradient norm turns non-finite and st_train_step / st_dp_clip_adam SKIP the update and count it in
 *                     scalars[5] (Apex's overflow handling; the host lowers the scale).  The polar backward (fp32, 1e7-sized
 *                     sub-gradients on silent frames) saturates its output to the fp16 range; loss and Adam fp32.
 *   ST_PREC_F32X3     fp32 results from the bf16 MATRIX pipe: every operand of the analysis / synthesis GEMMs (forward, data and
 *                     weight gradients) is split as it is staged into three bfloat16 planes x = x1 + x2 + x3 (24 significand bits,
 *                     both remainders exact) and each product is the sum of the six partial products of order >= 2^-16, accumulated
 *                     in fp32.  Accuracy is that of an fp32 GEMM (measured against float64: not worse than ST_PREC_F32); it is a
 *                     different ROUNDING of the same arithmetic, so it is held to the fp32 reference with the fp32 tolerance.  On
 *                     gfx950 the fp32 MFMA runs at the vector-ALU rate and does not overlap vector work; six bf16 MFMAs cost 3/8 of
 *                     its cycles and run beside the VALU.  Autoencoders, polar maps, loss and Adam are as in ST_PREC_F32.
 * Each 16-bit level equals the reference computed with those operands rounded the same way (the oracle has the same switches),
 * not the fp32 reference. */
#define ST_PREC_F32 0
#define ST_PREC_BF16 1
#define ST_PREC_BF16_ALL 2
#define ST_PREC_F16 3
#define ST_PREC_F16_ALL 4
#define ST_PREC_F32X3 5

/* Geometry (nn_proc.py:357-385) and arithmetic of one call. */
typedef struct st_dims {
    int B;   /* windows in this (per-GPU) minibatch                         */
    int L;   /* samples per input window  (8192*scale)                      */
    int N;   /* ft_size (1024)                                              */
    int H;   /* hop (384)                                                   */
    int T;   /* input STFT frames  = ceil(L/H)+ceil(N/H)                    */
    int OT;  /* output STFT frames = ceil(out/H)+ceil(N/H)                  */
    int F;   /* N/2+1                                                       */
    int K;   /* knobs                                                       */
    int y;   /* output samples = (OT-1)*H - N                               */
    int prec;          /* ST_PREC_* (0 = fp32)                                                                   */
    float loss_scale;  /* static loss scale S of the mixed-precision step (0 or 1: none).  The fused entry points
                          multiply d loss by S before the backward and divide the gradients by S in the optimizer
                          (train.py:134-135); the per-op backward entries are linear and simply pass S through.   */
    int clip_all;      /* 0: L1 clip over the 4 STFT tensors only (nn_proc.py:299-302, the default path);
                          1: over ALL parameters (train.py:136, what the reference does when Apex amp is on)      */
} st_dims;

/* Internal padded spectral pitch: re bins at [0,F), im bins at [KP/2, KP/2+F). */
int st_kp(int F);

const char* st_last_error(void);
int st_version(void);

/* Diagnostic switches (process-wide; NEVER set by the product path): st_set_tuning(code) selects kernel variants kept for comparison
 * (codes: st_api.hip).  They are one readable state: st_get_tuning / st_tuning_defaults fill out[0..n) in a fixed order and return the
 * number of switches (out may be NULL), st_reset_tuning restores the shipped defaults; the test-suite asserts the state is at the
 * defaults after every test.  Timing-only ablations (results invalid) exist only in -DST_DIAG builds. */
int st_set_tuning(int bk);
int st_get_tuning(int* out, int n);
int st_tuning_defaults(int* out, int n);
int st_reset_tuning(void);
/* The arithmetic a call with these dims ACTUALLY runs (an ST_PREC_* code).  Since round 5 that is d->prec for every geometry and batch (until round 4 the
 * wide autoencoder path -- T > 32 or OT > 16, e.g. the 65536-sample window -- ran fp32 Linear layers for odd batches; its weight-gradient GEMMs now take
 * 16-deep k-tiles when B * 528 is 16 mod 32).  Kept so that callers assert the arithmetic instead of assuming it (StepEngine.effective_dtype). */
int st_effective_prec(const st_dims* d);
/* Timing-only ablation switches for diagnostics (bit0: skip k-loop loads/stores, bit1: skip barriers, bit2: skip MFMAs);
 * results are INVALID when non-zero.  Never set by the product path. */
int st_set_debug(int v);
int st_debug_read_stage_cycles(unsigned long long* out32);   /* diagnostics: s_memtime per stage of ae_bwd (st_set_debug(256)) */

/* Optional per-kernel HIP-event profiling of the fused entry points (off by default; used by
 * bench.py's roofline leg outside the timed region).  st_profile_report fills buf with
 * "kernel_name total_ms launches\n" lines; synchronise the stream first. */
int st_profile_enable(int on);
int st_profile_report(char* buf, int buflen);

/* nn_proc.py:357-385 (st_model.__init__ geometry); lean scheme when legacy==0. Host only. */
int st_geometry(double scale_factor, double shrink_factor, int legacy, int K, int B, st_dims* out);

/* Flat parameter buffer layout: the 40 state_dict tensors in registration order,
 * each start aligned to 4 floats.  st_param_offsets fills offs[40] (in floats) and returns the
 * total length in floats (or <0). */
int64_t st_param_offsets(const st_dims* d, int64_t* offs /*[40]*/);

/* Workspace sizes in bytes for st_model_fwd / st_train_step (pure functions of dims).
 * Contract on its contents: the library never reads a workspace element that the same step has not written -- but it does not write every element.
 * Regions the trimmed kernels skip stay as the caller left them: the padding columns [F, FP) of both halves of every dAA slab (the data gradient
 * covers the F spectral columns only; the autoencoder backward masks f >= F) and the 128-tap tile columns of frs that hold only cropped taps
 * (st_ola_loss reads live taps only).  Zero-fill the buffer once after allocating it (signaltrain_amd.engine does: torch.zeros) if anything
 * other than the library -- a debug dump, a new consumer -- is going to look at those regions; stale bytes there may read as NaN. */
size_t st_workspace_bytes(const st_dims* d);
/* st_workspace_bytes answers for EXACTLY the dims it is given, and it is monotonic neither in the batch (the split-K slab counts of the weight-gradient /
 * synthesis GEMMs are picked per batch: 585 windows of 8192 samples need 85.6 MB more than 586) nor in the arithmetic level (fp32 autoencoder layers keep
 * their activations for the backward).  A host that allocates ONE workspace and then runs other batches or levels in it -- a last partial batch, an inference
 * remainder, a change of st_dims.prec on a live model -- sizes it with this: the maximum over every batch 1 .. d->B, every ST_PREC_* level and both clip
 * scopes (host arithmetic only; ~20 ms for 1024 windows).  0 for bad dims. */
size_t st_workspace_bytes_max(const st_dims* d);
/* The knobs pointer of every entry below: [B][K] fp32; with K == 0 (a model without knobs: nn_proc.py:92-93 concatenates an empty tensor) it may be NULL. */

/* ---------------------------------------------------------------- per-op entry points --- */
/* cls_fe_dft.py:50-58 Analysis.forward fused with nn_proc.py:309-310 (mag, phs).
 * x[B,L]; Wr,Wi[N,N] (rows >= F unused); in_scale multiplies x (0.5, nn_proc.py:307).
 * Outputs [B,T,F] each; any of re/im/mag/phs may be NULL. */
int st_analysis_fwd(const st_dims* d, const float* x, const float* Wr, const float* Wi, float in_scale,
                    float* re, float* im, float* mag, float* phs, void* stream);

/* nn_proc.py:77-126 AsymAutoEncoder.forward for both nets + nn_proc.py:322-326 polar->rect.
 * ae_m / ae_p: the 18 tensors of one autoencoder, packed contiguously in state_dict order with
 * the 4-float alignment of st_param_offsets (i.e. pointer into the flat parameter buffer).
 * Outputs: mag_hat, phs_hat [B,OT,F]; AA [B*OT, KP] (an_real | an_imag, padded);
 * reg_partial: per-wave partial sums of |mag_hat * exp(7 f/F)| (st_ae_fwd_partials() floats). */
int st_ae_fwd(const st_dims* d, const float* mag, const float* phs, const float* knobs,
              const float* ae_m, const float* ae_p, float* mag_hat, float* phs_hat, float* AA,
              float* reg_partial, float* ws, void* stream);
int st_ae_fwd_partials(const st_dims* d);
/* Floats of workspace st_ae_fwd uses in `ws`.  Fused kernels (T <= 32 and OT <= 16): OPTIONAL - when ws is given the
 * forward keeps each net's 16-wide code h4 there ([net][group][lane] float4) so that a following st_ae_bwd on the SAME
 * workspace starts its decoder half from it; ws may be NULL (h4 is then not kept and the backward recomputes it).  Wide
 * geometries (e.g. the 65536-sample window: T = 174, OT = 46) run the layers as feature-major GEMMs, keep their
 * activations there and REQUIRE it.  st_ae_bwd_ws_floats() covers the backward of either path (and contains the forward's
 * part at offset 0). */
size_t st_ae_fwd_ws_floats(const st_dims* d);

/* Hermitian fold of the synthesis bases (cls_fe_dft.py:109-110 expressed on the weights):
 * Sfold[KP,N]: rows [0,F) = Sr[k]+Sr[N-k], rows [KP/2,KP/2+F) = Si[k]-Si[N-k]; other rows 0. */
int st_synth_fold(const st_dims* d, const float* Sr, const float* Si, float* Sfold, void* stream);

/* Number of split-K slabs the two small-M synthesis GEMMs write: st_synth_frame_slabs() for st_synthesis_frames'
 * output frs (slabs x [B*OT,N], summed by st_ola_loss) and st_synth_slabs() for st_synthesis_dgrad's output dAA
 * (slabs x [B*OT,KP], summed by st_ae_bwd). */
int st_synth_slabs(const st_dims* d);
int st_synth_frame_slabs(const st_dims* d);

/* cls_fe_dft.py:112 ConvTranspose1d as a GEMM: frs[B*OT,N] = AA[B*OT,KP] * Sfold[KP,N] -- only what survives the crop of cls_fe_dft.py:113: frames that
 * lie wholly in the cropped margins are not computed, and (round 5, 128 x 128-tile path) of the partly cropped frames only the 128-tap tile columns that hold
 * a tap n with N <= H t + n < N + y.  st_ola_loss reads exactly those taps; everything else in frs is left as it was. */
int st_synthesis_frames(const st_dims* d, const float* AA, const float* Sfold, float* frs, void* stream);

/* Diagnostics, host only (no device work): the work list of the 128 x 128-tile synthesis GEMMs -- which = 0 st_synthesis_frames, 1 st_synthesis_dgrad's
 * padded form inside the fused step -- after the structural zeros of the cropped transposed convolution (cls_fe_dft.py:112-113) are dropped.
 * out[i] = tile row (8 bits) | tile column (6) << 8 | slab (2) << 14 | first zero-filled slab (2; 0 = none) << 16 | kind (1; 1: the two Nyquist columns of a
 * tile row) << 18 | first k unit (6) << 19 | k units (7) << 25; rows are the live frames enumerated frame-major (row = (t' - t'_lo) * B + b);
 * head6 = {slabs, col_h, col_stride, B, k-tiles (of 32) per k unit, k-tiles of the whole reduction} (tile column c covers output columns
 * (128 c % col_h) + (128 c / col_h) * col_stride ... + 128; col_h = 0: 128 c).  ncus <= 0: the current device's CU count; with ncus > 0 the result
 * is a pure function of (d, which, ncus) -- no device query.
 * Returns the entry count (one workgroup each; more than ncus = several rounds), 0 if the geometry does not use this kernel. */
int st_nt128_worklist(const st_dims* d, int which, int ncus, unsigned* out, int cap, int* head6);

/* Diagnostics, host only: 1 if the frame-major row order may be used for a GEMM with R = (live frames) * B compact rows, i.e. if the kernels' division of a
 * row index r < R by B through a multiply-high with floor(2^32 / B) + 1 is exact (r * B < 2^32 for every r); 0: those launches keep the window-major order
 * and the untrimmed / uncropped tile sets.  (cls_fe_dft.py:28-31, :112-113: the trimming is an optimisation of the padded / cropped frames, never a
 * change of results.) */
int st_fm_div_exact(int R, int B);

/* cls_fe_dft.py:112-113 overlap-add + crop, nn_proc.py:332,340 residual and x2,
 * loss_functions.py:9-10 log-cosh partial sums and d loss/d syn.
 * y_true may be NULL (inference): then only y_hat is produced.
 * loss_partial[b * st_ola_loss_partials() + s] = the sum over samples [256 s, 256 s + 256) of window b, always by the same summation tree
 * (quads in sample order, then a balanced binary tree over the quad sums): the bits do not depend on the pointers' alignment, which only
 * selects between the 16-byte and the 4-byte access form of the kernel. */
int st_ola_loss(const st_dims* d, const float* frs, const float* x, const float* y_true,
                float* y_hat, float* dsyn, float* loss_partial, void* stream);
int st_ola_loss_partials(const st_dims* d);

/* Backward of the synthesis GEMM wrt its input: dAA[B*OT,KP] = frames(dsyn) * Sfold^T.  Columns [F, KP/2) and [KP/2 + F, KP) (the padding of the
 * spectral pitch) are written as zeros by the small-tile kernels and LEFT AS THEY WERE by the work-list kernel (see st_workspace_bytes). */
int st_synthesis_dgrad(const st_dims* d, const float* dsyn, const float* Sfold, float* dAA, void* stream);

/* Weight gradient of the synthesis bases (folded then unfolded to Sr/Si [N,N]); also emits
 * partial sums of |g| for the L1 clip (nn_proc.py:299-302).  ws: split-K scratch. */
int st_synthesis_wgrad(const st_dims* d, const float* AA, const float* dsyn, float* ws,
                       float* gSr, float* gSi, float* norm_partial, void* stream);

/* Backward of both autoencoders (recomputes activations) incl. nn_proc.py:322-326 and the
 * L1 term of loss_functions.py:36.  Outputs dmag, dphs [B,T,F]; per-wave partial weight grads
 * in ws (st_ae_bwd_ws_floats()), reduced into g_m / g_p (same packing as ae_m / ae_p). */
int st_ae_bwd(const st_dims* d, const float* mag, const float* phs, const float* knobs,
              const float* ae_m, const float* ae_p, const float* mag_hat, const float* phs_hat,
              const float* dAA, const float* g_mag_hat /* optional extra d/d mag_hat */, float reg_coef,
              float* dmag, float* dphs, float* ws, float* g_m, float* g_p, void* stream);
size_t st_ae_bwd_ws_floats(const st_dims* d);
/* Round 6.  In the fused step (st_model_fwd with save_for_backward / st_loss_backward / st_train_step) of the fused geometries with fp32 autoencoder
 * layers, the forward kernel KEEPS the post-ELU activations of both nets -- what the reference's autograd keeps (nn_proc.py:77-126) -- in the
 * autoencoder workspace behind the gradient partials, and the backward reads them instead of recomputing the forward chain.  Returns the bytes the
 * forward writes (and the backward reads back) per call: 2 nets x B * ceil(F / 16) row groups x 17 KB (294 MB at B = 256); 0 where the path is not
 * taken (16-bit autoencoder layers, wide geometries, st_set_tuning(8200)).  st_ae_bwd on its own (no forward in the same workspace) recomputes. */
size_t st_ae_kept_activation_bytes(const st_dims* d);

/* Backward of nn_proc.py:309-310: dG[B*T,KP] (d re | d im) from (re,im,dmag,dphs). */
int st_polar_bwd(const st_dims* d, const float* re, const float* im, const float* dmag, const float* dphs,
                 const float* g_mag /* optional extra d/d mag */, float* dG, void* stream);

/* Weight gradient of the analysis bases: gWr/gWi[N,N] rows [0,F) written, rows >= F untouched
 * (they are structurally zero).  Emits |g| partial sums. */
int st_analysis_wgrad(const st_dims* d, const float* dG, const float* x, float in_scale, float* ws,
                      float* gWr, float* gWi, float* norm_partial, void* stream);
size_t st_wgrad_ws_floats(const st_dims* d);
int st_norm_partials(const st_dims* d);

/* nn_proc.py:299-302 L1 clip (STFT tensors = first 4 tensors of the flat buffer) fused with
 * torch.optim.Adam.step (train.py:147).  scalars: device float[8] written by st_finalize_scalars. */
int st_finalize_scalars(const st_dims* d, const float* loss_partial, const float* reg_partial,
                        const float* norm_partial_a, const float* norm_partial_s, float inv_world,
                        float* scalars /* [0]=loss [1]=logcosh [2]=reg [3]=l1norm [4]=clip_coef [5]=overflow */, void* stream);
int st_clip_adam(float* params, float* grads, float* m, float* v, int64_t n_total, int64_t n_stft,
                 const float* scalars, float grad_scale, float lr, float beta1, float beta2, float eps, int step,
                 void* stream);

/* ---------------------------------------------------------------- fused entry points ----- */
/* st_model.forward (nn_proc.py:392 -> :305-340), inference/validation: y_hat[B,y], mag[B,T,F],
 * mag_hat[B,OT,F].  params = flat parameter buffer; ws = st_workspace_bytes(). */
int st_model_fwd(const st_dims* d, const float* params, const float* x, const float* knobs,
                 float* y_hat, float* mag, float* mag_hat, void* ws, int save_for_backward, void* stream);

/* Autograd backward of st_model_fwd(save_for_backward=1) for arbitrary upstream gradients
 * g_y_hat[B,y] (required), g_mag_hat[B,OT,F], g_mag[B,T,F] (optional): fills the flat `grads`
 * (rows >= F of the analysis tensors are never written: keep them zero). */
int st_model_bwd(const st_dims* d, const float* params, float* grads, const float* x, const float* knobs,
                 const float* g_y_hat, const float* g_mag_hat, const float* g_mag, void* ws, void* stream);

/* One optimisation step of train.py:112-151 WITHOUT the optimiser: forward, loss, backward.
 * grads (flat, same layout as params) are overwritten; scalars[0..3] receive loss terms and the
 * STFT L1 norm (before any all-reduce).  y_hat/mag/mag_hat may be NULL. */
int st_loss_backward(const st_dims* d, const float* params, float* grads, const float* x, const float* knobs,
                     const float* y_true, float* y_hat, float* mag, float* mag_hat, void* ws,
                     float* scalars, void* stream);

/* Data-parallel split of st_loss_backward.  After p1 the gradients of the synthesis bases and both
 * autoencoders (grads[offs[2]..end)) are final: all-reduce them while p2 computes the analysis
 * weight gradient (rows [0,F) of tensors 0 and 1) and the loss scalars. */
int st_loss_backward_p1(const st_dims* d, const float* params, float* grads, const float* x, const float* knobs,
                        const float* y_true, void* ws, void* stream);
int st_loss_backward_p2(const st_dims* d, float* grads, const float* x, void* ws, float* scalars, void* stream);
/* p2 with a packed copy of the 2F live analysis rows in `stage` [2F][N] (caller-owned): the buffer the data-parallel
 * all-reduce moves instead of the contiguous range spanning the structurally-zero rows; st_unstage_analysis copies the
 * reduced rows back into grads (rows [0,F) of tensors 0 and 1).  The loss scalars are NOT written here: st_dp_clip_adam
 * publishes them together with the norm of the reduced gradient.  No reference counterpart (see st_loss_backward_stage). */
int st_loss_backward_p2_staged(const st_dims* d, float* grads, float* stage, const float* x, void* ws, float* scalars, void* stream);
int st_unstage_analysis(const st_dims* d, float* grads, const float* stage, void* stream);

/* Finer split of the same step for data parallel (no reference counterpart: train.py:259-263 is a disabled
 * nn.DataParallel stub).  Call stage = 0, 1, 2, 3 in order on one stream; after stage s one gradient range is final
 * and can be all-reduced while the next stage runs:
 *   0  forward (nn_proc.py:304-340) + loss (loss_functions.py:26-36) + synthesis backward -> grads[offs[2], offs[4])
 *   1  autoencoder + polar backward                                                       -> grads[offs[4], total)
 *   2  analysis weight gradient, real basis      (cls_fe_dft.py:50-58 autograd)           -> grads[offs[0], offs[0] + F*N)
 *   3  analysis weight gradient, imaginary basis + loss scalars                           -> grads[offs[1], offs[1] + F*N)
 * Rows >= F of the analysis tensors are structurally zero and never need to move. */
int st_loss_backward_stage(const st_dims* d, const float* params, float* grads, const float* x, const float* knobs,
                           const float* y_true, void* ws, float* scalars, int stage, void* stream);

/* Full single-GPU step: st_loss_backward + L1 clip + Adam (train.py:131-151). */
int st_train_step(const st_dims* d, float* params, float* grads, float* m, float* v, const float* x,
                  const float* knobs, const float* y_true, void* ws, float* scalars,
                  float lr, float beta1, float beta2, float eps, int step, void* stream);

/* After an external all-reduce of `grads` (data parallel): recompute the L1 norm of the reduced, scaled gradient (over the
 * STFT tensors, or over everything with st_dims.clip_all), clip and Adam.  grad_scale = 1/world (the loss scale of
 * st_dims.loss_scale, if any, is removed here as well). */
int st_dp_clip_adam(const st_dims* d, float* params, float* grads, float* m, float* v, void* ws,
                    float* scalars, float grad_scale, float lr, float beta1, float beta2, float eps,
                    int step, void* stream);

/* ---- data parallel inside the library: RCCL over xGMI -------------------------------------------
 * Replaces the reference's disabled nn.DataParallel stub (train.py:259-263) by one process per GPU + an explicit exchange
 * step.  The communicator handle is the only state the library owns; RCCL is bound with dlopen at st_dp_init (a process
 * that already carries librccl -- PyTorch-ROCm does -- shares that copy).
 *   st_dp_unique_id   rank 0: fills 128 bytes; the host hands them to every rank (any bootstrap channel).
 *   st_dp_init        collective over all ranks (ncclCommInitRank); creates the communicator stream and ordering events.
 *   st_dp_allreduce   in-place SUM of n floats on the communicator stream, ordered after the work issued so far on `stream`;
 *                     returns at once, so the caller's next kernels overlap the collective.
 *   st_dp_broadcast   same ordering, root's buffer to all (initial parameters).
 *   st_dp_sync        `stream` waits for every collective issued so far.
 *   st_dp_train_step  one whole data-parallel optimisation step driven from C (no host code between the buckets):
 *                     st_loss_backward_p1 -> all-reduce grads[offs[2]..) || st_loss_backward_p2_staged -> all-reduce the
 *                     packed live analysis rows (`stage`, 2*F*N floats, caller-owned) -> st_unstage_analysis ->
 *                     st_dp_clip_adam(1/world).  p == NULL or world == 1 (and bit 0 of force_exchange clear): plain st_train_step.
 *                     force_exchange is a bit set: 1 = run the exchange even with one rank (tests, overhead measurements);
 *                     2 = split the LAST exchange by basis -- the real rows' all-reduce runs under the weight-gradient GEMM of the imaginary
 *                         rows, so F*N values (2.1 MB) stay exposed instead of 2*F*N;
 *                     4 = the last exchange on bfloat16 values (ncclBfloat16; first half of `stage`), honoured only where the autoencoder
 *                         layers already run in 16 bits (ST_PREC_*_ALL): half the exposed bytes.
 *                     The sum of the synthesis weight-gradient slabs runs on the communicator stream, beside the autoencoder backward
 *                     (round 6: from a slab area of its own, so the analysis weight-gradient GEMM waits for nothing).  Round 6: without bit 1 the
 *                     LAST exchange -- exposed whatever stream it runs on -- is issued in line on `stream` behind one join with the communicator
 *                     stream (recorded behind the last hidden collective): two cross-stream hand-offs of ~10 us each less on the critical path
 *                     (one rank, --force-dp: +44 -> +16 us over the plain step); every rank issues the collectives of the communicator in the same order.
 *   st_dp_rccl_version  ncclGetVersion of the bound library (0 if it exports none): evidence for multi-GPU bench lines. */
typedef struct st_dp st_dp;
int st_dp_unique_id(void* id128);
int st_dp_init(const void* id128, int rank, int world, st_dp** out);
int st_dp_destroy(st_dp* p);
int st_dp_rank(const st_dp* p);
int st_dp_world(const st_dp* p);
int st_dp_rccl_version(const st_dp* p);
int st_dp_allreduce(st_dp* p, float* buf, int64_t n, void* stream);
int st_dp_broadcast(st_dp* p, float* buf, int64_t n, int root, void* stream);
int st_dp_sync(st_dp* p, void* stream);
int st_dp_train_step(st_dp* p, const st_dims* d, float* params, float* grads, float* m, float* v, float* stage,
                     const float* x, const float* knobs, const float* y_true, void* ws, float* scalars,
                     float lr, float beta1, float beta2, float eps, int step, int force_exchange, void* stream);

/* ---- the whole optimisation step as ONE HIP graph -----------------------------------------------------------------------
 * st_graph_create captures st_train_step (every kernel of train.py:112-151) on `stream` -- which must not be the legacy
 * default stream -- with the per-iteration quantities on the device: scalars[6] = step counter (0 before the first step;
 * restore it together with m / v when resuming), scalars[7] = learning rate of the step, looked up by the graph's head
 * node in lr_table (device float[n_lr] = the 1-cycle table, learningrate.py:14-52) as lr_sched[max(i - 1, 0)] for the
 * 0-based iteration i (train.py:150).  The graph reads the minibatch from the x / knobs / y_true buffers it was captured
 * with: refill them before each st_graph_launch.  All pointers must stay valid for the graph's lifetime. */
typedef struct st_graph st_graph;
int st_graph_create(const st_dims* d, float* params, float* grads, float* m, float* v, const float* x,
                    const float* knobs, const float* y_true, void* ws, float* scalars,
                    const float* lr_table, int n_lr, float beta1, float beta2, float eps, void* stream, st_graph** out);
int st_graph_launch(st_graph* g, void* stream);
int st_graph_destroy(st_graph* g);

/* ---- diagnostics: the activations st_model.forward(return_acts=True) returns (nn_proc.py:311-338, plotted by utils/viz.py:135) ----
 * st_workspace_offsets: float offsets, inside a workspace carved for d, of the forward state st_model_fwd(save_for_backward = 1) left:
 *   offs8 = { re, im, mag, phs [B][T][F], mag_hat, phs_hat [B][OT][F], AA [B*OT][KP] (an_real | an_imag halves), y_hat [B][y] }.
 * st_ae_acts: the ten activation tensors of ONE autoencoder (nn_proc.py:77-126 with return_acts), [B][F][width] each, concatenated
 *   (widths 64, 32, 16, 16, 16 + K, 16, 16, 32, 64, OT = st_ae_acts_floats(d) floats); v = its [B][T][F] input, ae = its packed parameters,
 *   sf != 0 for the magnitude net's skip-filter output (nn_proc.py:115), 0 for the phase net's bare output (nn_proc.py:117). */
int st_workspace_offsets(const st_dims* d, int64_t* offs8);
size_t st_ae_acts_floats(const st_dims* d);
int st_ae_acts(const st_dims* d, const float* v, const float* knobs, const float* ae, int sf, float* acts, void* stream);

/* ---- device-side data feed -------------------------------------------------------------------
 * audio.compressor_4controls (signaltrain/audio.py:380-426), the effect of the synthetic comp_4c task
 * (SynthAudioDataSet, datasets.py:312-334), for a batch of device-resident windows:
 *   x [B][L] fp32, knobs_wc [B][4] = (threshold dB, ratio, attack s, release s) in WORLD coordinates
 *   (Effect.knobs_wc, audio.py:455), y [B][ysz] = the last ysz samples of the processed window (datasets.py:327-330). */
int st_compressor_4c(const float* x, const float* knobs_wc, float sr, int B, int L, int ysz, float* y, void* stream);
/* One training minibatch of the synthetic comp_4c task made on the device in ONE launch (st_feed.h; replaces
 * SynthAudioDataSet.gen_single_chunk, datasets.py:312-334, over audio.synth_input_sample, audio.py:296-334, chooser set
 * {0,1,2,4,6,7}, and compressor_4controls): per window the test signal, knobs = Beta(0.8, 0.8) - 0.5 (audio.py:20-21,
 * datasets.py:325), the target (last ysz samples of the compressed window) and, if `augment`, the random polarity flip of
 * the pair (datasets.py:27-29).  Counter-based generator: window i of the stream `seed` is the same whatever the batching;
 * `first_window` = index of window 0 of this call.  K must be 4; knob_lo / knob_hi = Effect.knob_ranges (host arrays of 4).
 * chooser: -1 = drawn per window (the training feed); 0,1,2,4,6,7 force one signal family (tests).
 * pink_in: optional caller-made [B][L] unit-peak 1/f noise for windows longer than 8192 samples (the in-LDS FFT's limit); NULL: the library makes it --
 * in LDS up to 8192 samples, by its own four-step inverse FFT through `scratch` for powers of two up to 65536 (BASELINE configs[4]'s window), both
 * keyed by (seed, window index) like everything else of the window.
 * scratch: st_synth_comp4c_scratch_floats(B, L) floats (B * (L + 4) up to 8192 samples; may be NULL there): with it (and L % 64 == 0) the effect's
 * sequential attack / release stage runs one LANE per window in a second, tiny launch (64 windows per wave) instead of one workgroup per window --
 * the form that runs beside the training step.
 * Outputs x [B][L], y [B][ysz], knobs [B][4] (fp32, normalised to [-0.5, 0.5]). */
size_t st_synth_comp4c_scratch_floats(int B, int L);
int st_synth_comp4c(unsigned seed, unsigned long long first_window, int B, int L, int ysz, int K, float sr,
                    const float* knob_lo, const float* knob_hi, int augment, int chooser, const float* pink_in,
                    float* x, float* y, float* knobs, float* scratch, void* stream);

/* Gradient of the loss w.r.t. the (halved) input waveform, for callers with something trainable upstream of the model (the reference's
 * autograd provides it; nn_proc.py:307, cls_fe_dft.py:55-56).  Call right after st_model_bwd on the SAME workspace:
 *   gxh[b][n] = conv-transpose of the analysis output gradient with both bases, cropped by the Conv1d padding   [B][L]
 * The caller multiplies by 1/2 (nn_proc.py:307) and adds g_y_hat on the last y samples (the skip of nn_proc.py:340).
 * scratch: st_model_input_grad_ws_floats(d) floats. */
size_t st_model_input_grad_ws_floats(const st_dims* d);
int st_model_input_grad(const st_dims* d, const float* params, void* ws, float* scratch, float* gxh, void* stream);

/* Gradient w.r.t. the knob settings for arbitrary upstream gradients (same meaning as st_model_bwd's): what the reference's autograd hands to a
 * knobs tensor that requires grad -- nn_proc.py:92-93 repeats the settings over the rows of a window and concatenates them in front of
 * fnn_addknobs of both autoencoders (nn_proc.py:332-333).  g_knobs [B][K].  The exact, SLOW route: one forward + backward per window, whose
 * fnn_addknobs bias gradients are that window's row sums of d a5; the reference's training never asks for this gradient (knobs are data), so
 * the hot kernels carry nothing for it.  grads_scratch: st_param_offsets() floats, overwritten.  The saved-for-backward state of `ws` belongs
 * to the last window afterwards: run st_model_fwd(save_for_backward = 1) again before st_model_bwd.  The per-window passes run in the arithmetic of
 * d->prec like the batch itself (round 5: a single window takes 16-bit layers on every geometry, st_effective_prec). */
int st_model_knob_grad(const st_dims* d, const float* params, float* grads_scratch, const float* x, const float* knobs,
                       const float* g_y_hat, const float* g_mag_hat, const float* g_mag, void* ws, float* g_knobs, void* stream);

/* ---- generic learned-basis front end: SURVEY.md row a15, signaltrain/cls_fe_dct_bases.py ------------------------------
 * Analysis.forward (:129-136)  = Conv1d(1 -> C, kernel KW, stride hop, padding pad, bias) transposed to [B][T][C];
 * Synthesis.forward (:174-179) = ConvTranspose1d(C -> 1, kernel KW, stride hop) with `crop` samples cut from both ends.
 * W is the [C][1][KW] Conv weight seen as [C][KW].  T = st_fe_frames(L, KW, hop, pad).  The *_bwd entries are the
 * autograd of the forward ones (weight, bias and input gradients).  ws: st_fe_ws_floats() floats (for synthesis pass
 * L = the output length + 2*crop - 2*pad equivalent, i.e. call it with the analysis geometry of the same model). */
int st_fe_frames(int L, int KW, int hop, int pad);
size_t st_fe_ws_floats(int B, int L, int C, int KW, int hop, int pad);
int st_fe_analysis_fwd(const float* x, int B, int L, const float* W, const float* bias, int C, int KW, int hop, int pad,
                       float* out, void* stream);
int st_fe_synthesis_fwd(const float* xft, int B, int T, const float* W, int C, int KW, int hop, int crop,
                        float* ws, float* out, void* stream);
int st_fe_analysis_bwd(const float* x, int B, int L, const float* W, int C, int KW, int hop, int pad, const float* g_out,
                       float* ws, float* gW, float* gbias, float* gx, void* stream);
int st_fe_synthesis_bwd(const float* xft, int B, int T, const float* W, int C, int KW, int hop, int crop, const float* g_wave,
                        float* ws, float* gW, float* g_xft, void* stream);

#ifdef __cplusplus
}
#endif
#endif
