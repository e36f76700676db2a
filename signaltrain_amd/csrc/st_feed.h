// st_feed.h -- the comp_4c training feed as ONE kernel per minibatch (gfx950): SURVEY.md 8(f)-1.
//
// Reference: datasets.py:312-334 (SynthAudioDataSet.gen_single_chunk: chooser in {0, 1, 2, 4, 6, 7}, knobs = Beta(0.8, 0.8) - 0.5, target = effect
// output's last y_size samples, random polarity flip of the pair :27-29) over audio.py:85-196 / :296-334 (the test signals) and
// audio.py:380-426 (compressor_4controls).  Round 2 evaluated the signals with a few dozen torch launches per batch (2.9 ms per 256
// windows) and a separate compressor launch; here one workgroup makes one training item end to end:
//   draw   the window's parameters from a counter-based generator keyed by (seed, global window index): no state, no host RNG, any number
//          of windows per launch, reproducible per window whatever the batching;
//   pink   1/f noise as the reference builds it -- inverse FFT of the REAL spectrum (2u - 1) / sqrt(k + 1) -- with a radix-2 FFT in LDS
//          (64 KB for the 8192-sample window; longer windows take the noise from a caller-provided buffer);
//   signal the chosen family evaluated per sample, peak-normalised (normish: two passes, the first only takes the maximum), polarity, 1e-8 noise;
//   effect the 4-control compressor on the finished window (stm::compressor_window: parallel gain computer, sequential attack / release
//          smoother in one lane, parallel apply) -> the last ysz samples.
// Output contract: x [B][L], y [B][ysz], knobs [B][4] in [-0.5, 0.5] (float32), as the reference's collated batch (train.py:104-120 casts y to float).
// Parity is DISTRIBUTIONAL (another generator than numpy's): tests compare per-family peak ranges, the even symmetry and 1/f slope of the noise,
// the box structure, the Beta(0.8, 0.8) law, and the compressor against golden G9 through the same device function.
#pragma once
#include "st_common.h"
#include "st_misc.h"

namespace stf {

__device__ __forceinline__ unsigned mix32(unsigned x)
{
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
__device__ __forceinline__ float u01(const unsigned h) { return (float)(h >> 8) * (1.0f / 16777216.0f); }      // [0, 1)
struct Draw {              // sequential scalar draws of one window (every thread evaluates the same sequence)
    unsigned key, ctr;
    __device__ float u() { return u01(mix32(key + 0x9E3779B9u * (++ctr))); }
    __device__ float sign() { return u() < 0.5f ? -1.f : 1.f; }
    __device__ int randint(int lo, int hi) { const int v = lo + (int)(u() * (float)(hi - lo)); return v < hi ? v : hi - 1; }      // [lo, hi)
    // Beta(a, a), a < 1: Joehnk's method (accept x + y <= 1 with x = u^(1/a), y = v^(1/a); acceptance 0.61 for a = 0.8)
    __device__ float beta(const float a) {
        float x = 0.5f, y = 0.5f;
        for (int it = 0; it < 64; ++it) {
            x = __powf(fmaxf(u(), 1e-30f), 1.0f / a); y = __powf(fmaxf(u(), 1e-30f), 1.0f / a);
            if (x + y <= 1.0f && x + y > 0.f) break;
        }
        return x / (x + y);
    }
};
// per-sample streams: value n of stream s of the window
__device__ __forceinline__ float su01(const unsigned key, const unsigned s, const unsigned n) { return u01(mix32(mix32(key ^ (0xA511E9B3u * (s + 1u))) + 0x9E3779B9u * n)); }

struct FeedArgs {
    float* x; float* y; float* knobs;       // outputs
    const float* pink_in;                   // [B][L] 1/f noise for windows longer than the in-kernel FFT handles (else NULL): unit-peak from the caller, or
    const float* pink_peak;                 // ... unnormalised from pink_long_pass1 / 2 below with its per-window peak here (NULL: pink_in is unit-peak)
    unsigned seed; unsigned long long first;   // global index of window 0 of this launch
    int L, ysz, K; float sr;
    float lo[4], hi[4];                     // knob ranges (Effect.knob_ranges, audio.py:493-510)
    int augment;
    float* gc; float* kw;                   // optional scratch [B][L] + [B][4]: the generator leaves the gain curve and the world-coordinate knobs there and
                                            // stm::comp_smooth_apply_kernel finishes the effect (lane per window); NULL: the effect runs inside this kernel
    int chooser;                            // -1: drawn per window from {0, 1, 2, 4, 6, 7}; else forced (tests); 100 = the bare 1/f noise (tests)
};
constexpr int FFT_MAX = 8192;

// in-place inverse FFT (radix-2, decimation in time; the caller stored the spectrum bit-reversed), N a power of two <= FFT_MAX
__device__ __forceinline__ void ifft_lds(float2* a, const int N, const int logN)
{
    for (int s = 1; s <= logN; ++s) {
        const int half = 1 << (s - 1);
        for (int j = threadIdx.x; j < N / 2; j += 256) {
            const int grp = j >> (s - 1), pos = j & (half - 1);
            const int i0 = (grp << s) + pos, i1 = i0 + half;
            float sn, cs; __sincosf(6.28318530717958648f * (float)pos / (float)(2 * half), &sn, &cs);
            const float2 u = a[i0], v = a[i1];
            const float2 t = make_float2(v.x * cs - v.y * sn, v.x * sn + v.y * cs);
            a[i0] = make_float2(u.x + t.x, u.y + t.y);
            a[i1] = make_float2(u.x - t.x, u.y - t.y);
        }
        __syncthreads();
    }
}
__device__ __forceinline__ float block_max(float v, float* red)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    const float m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    return m;
}

// ---- 1/f noise of LONG windows (L = 2^m > FFT_MAX, e.g. the 65536-sample window of BASELINE configs[4]): the same inverse FFT of the real spectrum
// (2u - 1) / sqrt(k + 1) (audio.py:85-94), as a four-step transform N = N1 * N2 (N1 = 256) through a global scratch -- two launches, no FFT library:
//   y[N2 n1 + n2] = sum_k1 e^{2 pi i k1 n1 / N1} ( e^{2 pi i k1 n2 / N} sum_k2 X[k1 + N1 k2] e^{2 pi i k2 n2 / N2} )
//   pass 1  (k1 fixed, eight k1 per workgroup): the spectrum values are GENERATED (counter-based, stream 7 of the window's key -- the same law and the
//           same per-window reproducibility as the in-LDS transform of the short windows), N2-point transform over k2 in LDS, twiddle, -> scr[n2][k1];
//   pass 2  (n2 fixed, eight n2 per workgroup): 256-point transform over k1 of a contiguous 2 KB row, real part -> pink[N2 n1 + n2], |.| peak -> peak[b]
//           (atomicMax on the bits of a non-negative float); the generator kernel divides by the peak when it mixes the noise in.
// Only windows whose family uses the noise (chooser 1, 7; 100 in tests) do any work: a third of the training stream.  Round 3 took this noise from
// torch.fft (rocFFT) driven by a stateful torch.Generator: 2 ms of full-width GPU time per 2048 windows and not reproducible per window index.
constexpr int PL_N1 = 256, PL_G = 16;      // 16 transforms per workgroup: 128-byte segments in pass 1's transposed store, 64-byte ones in pass 2's
__device__ __forceinline__ int feed_family(Draw& d, const int chooser)
{
    const int ci = d.randint(0, 6);
    return chooser >= 0 ? chooser : (ci < 3 ? ci : (ci == 3 ? 4 : (ci == 4 ? 6 : 7)));
}
__device__ __forceinline__ unsigned feed_key(const unsigned seed, const unsigned long long w)
{
    return mix32(seed ^ mix32((unsigned)w + 1u) ^ mix32((unsigned)(w >> 32) + 0x51ED27u));
}
// G independent in-place inverse FFTs of length n = 2^logn in LDS (a[g * n + i], input stored bit-reversed), 256 threads; tw[k] = e^{2 pi i k / n}, k < n / 2
__device__ __forceinline__ void ifft_batch_lds(float2* a, const float2* tw, const int n, const int logn, const int G)
{
    const int half_total = G * (n >> 1);
    for (int s = 1; s <= logn; ++s) {
        const int half = 1 << (s - 1), tstep = n >> s;
        for (int j = threadIdx.x; j < half_total; j += 256) {
            const int g = j / (n >> 1), jj = j - g * (n >> 1);
            const int grp = jj >> (s - 1), pos = jj & (half - 1);
            const int i0 = g * n + (grp << s) + pos, i1 = i0 + half;
            const float2 w = tw[pos * tstep];
            const float2 u = a[i0], v = a[i1];
            const float2 t = make_float2(v.x * w.x - v.y * w.y, v.x * w.y + v.y * w.x);
            a[i0] = make_float2(u.x + t.x, u.y + t.y);
            a[i1] = make_float2(u.x - t.x, u.y - t.y);
        }
        __syncthreads();
    }
}
__device__ __forceinline__ void twiddle_table(float2* tw, const int n)
{
    for (int k = threadIdx.x; k < n / 2; k += 256) { float sn, cs; sincospif(2.0f * (float)k / (float)n, &sn, &cs); tw[k] = make_float2(cs, sn); }
}
__global__ void __launch_bounds__(256)
pink_long_pass1_kernel(const unsigned seed, const unsigned long long first, const int L, const int chooser, float2* __restrict__ scr, float* __restrict__ peak)
{
    __shared__ float2 lds[PL_G * 256];
    __shared__ float2 tw[128];
    const int b = blockIdx.y, N2 = L / PL_N1;
    const unsigned long long w = first + (unsigned long long)b;
    Draw d{feed_key(seed, w), 0u};
    const unsigned key = d.key;
    const int ch = feed_family(d, chooser);
    if (!(ch == 1 || ch == 7 || ch == 100)) return;                     // workgroup-uniform
    if (blockIdx.x == 0 && threadIdx.x == 0) peak[b] = 0.f;
    int logn = 0; while ((1 << logn) < N2) ++logn;
    twiddle_table(tw, N2);
    const int k1_0 = blockIdx.x * PL_G;
    for (int j = threadIdx.x; j < PL_G * N2; j += 256) {
        const int g = j / N2, k2 = j - g * N2;
        const int k = k1_0 + g + PL_N1 * k2;
        const int kk = k <= L / 2 ? k : L - k;                           // Hermitian extension of a real spectrum
        const float v = (2.f * su01(key, 7u, (unsigned)kk) - 1.f) * rsqrtf((float)kk + 1.f);
        lds[g * N2 + (int)(__brev((unsigned)k2) >> (32 - logn))] = make_float2(v, 0.f);
    }
    __syncthreads();
    ifft_batch_lds(lds, tw, N2, logn, PL_G);
    float2* out = scr + (size_t)b * L;
    for (int j = threadIdx.x; j < PL_G * N2; j += 256) {
        const int n2 = j / PL_G, g = j - n2 * PL_G, k1 = k1_0 + g;
        float sn, cs; sincospif(2.0f * (float)(k1 * n2) / (float)L, &sn, &cs);      // k1 n2 < 2^24: exact in float
        const float2 v = lds[g * N2 + n2];
        out[(size_t)n2 * PL_N1 + k1] = make_float2(v.x * cs - v.y * sn, v.x * sn + v.y * cs);
    }
}
__global__ void __launch_bounds__(256)
pink_long_pass2_kernel(const unsigned seed, const unsigned long long first, const int L, const int chooser, const float2* __restrict__ scr,
                       float* __restrict__ pink, float* __restrict__ peak)
{
    __shared__ float2 lds[PL_G * PL_N1];
    __shared__ float2 tw[PL_N1 / 2];
    __shared__ float red[4];
    const int b = blockIdx.y, N2 = L / PL_N1;
    const unsigned long long w = first + (unsigned long long)b;
    Draw d{feed_key(seed, w), 0u};
    const int ch = feed_family(d, chooser);
    if (!(ch == 1 || ch == 7 || ch == 100)) return;
    twiddle_table(tw, PL_N1);
    const int n2_0 = blockIdx.x * PL_G;
    const float2* in = scr + (size_t)b * L + (size_t)n2_0 * PL_N1;
    for (int j = threadIdx.x; j < PL_G * PL_N1; j += 256) {
        const int g = j >> 8, k1 = j & 255;
        lds[g * PL_N1 + (int)(__brev((unsigned)k1) >> 24)] = in[j];
    }
    __syncthreads();
    ifft_batch_lds(lds, tw, PL_N1, 8, PL_G);
    float m = 0.f;
    float* o = pink + (size_t)b * L;
    for (int j = threadIdx.x; j < PL_G * PL_N1; j += 256) {
        const int n1 = j / PL_G, g = j - n1 * PL_G;
        const float v = lds[g * PL_N1 + n1].x;
        o[(size_t)N2 * n1 + n2_0 + g] = v;
        m = fmaxf(m, fabsf(v));
    }
    m = block_max(m, red);
    if (threadIdx.x == 0) atomicMax(reinterpret_cast<unsigned*>(peak + b), __float_as_uint(m));
}

__global__ void __launch_bounds__(256)
synth_comp4c_kernel(const FeedArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float feed_lds[];      // float2[FFT_MAX] (pink) -- later float[COMP_CH] (compressor)
    __shared__ float red[4];
    __shared__ float carry;
    const int b = blockIdx.x, L = a.L;
    const unsigned long long w = a.first + (unsigned long long)b;
    Draw d{feed_key(a.seed, w), 0u};
    const unsigned key = d.key;
    const float dt = 1.0f / a.sr, tl = (float)(L - 1) * dt;

    // ---- the window's parameters (datasets.py:317, audio.py:296-334)
    const int ch = feed_family(d, a.chooser);      // {0, 1, 2, 4, 6, 7}
    float kn[4], kw[4];
    for (int k = 0; k < 4; ++k) { kn[k] = d.beta(0.8f) - 0.5f; kw[k] = a.lo[k] + (kn[k] + 0.5f) * (a.hi[k] - a.lo[k]); }
    // randsine (audio.py:96-104)
    const int s_n = d.randint(1, 3);
    float s_amp[2], s_frq[2], s_t0[2];
    for (int i = 0; i < 2; ++i) { s_amp[i] = i < s_n ? 0.2f + 0.7f * d.u() : 0.f; s_frq[i] = 5.f + 145.f * d.u(); s_t0[i] = d.u() * tl; }
    // pluck (audio.py:138-148) and its decay envelope (audio.py:126-136)
    const int p_n = d.randint(1, 4);
    float p_amp[3], p_frq[3], p_t0[3];
    for (int i = 0; i < 3; ++i) { const float am = (0.45f * d.u() + 0.5f) * d.sign(); p_amp[i] = i < p_n ? am : 0.f; p_t0[i] = (2.f * d.u() - 1.f) * 0.3f * tl; p_frq[i] = 50.f + 6350.f * d.u(); }
    const float e_t0 = 0.35f * d.u() * tl, e_hi = 0.35f * d.u() + 0.6f, e_lo = 0.1f * d.u() + 0.1f, e_dec = 12.f * d.u();
    // box (audio.py:106-124)
    const float b_h0 = 0.15f * d.u(), b_h1 = 0.35f * d.u() + 0.6f, b_h2 = 0.2f * d.u() + 0.1f;
    const int b_up = (int)(0.3f * d.u() * (float)L);
    int b_dn = b_up + (int)((0.3f + 0.35f * d.u()) * (float)L); if (b_dn > L - 1) b_dn = L - 1;
    const float nrm_u = 0.6f + 0.3f * d.u();                       // normish: U(0.6, 0.9) / peak
    const float c_pink1 = 0.2f * d.u(), c_white1 = 0.2f * d.u(), c_pink7 = 0.3f * d.u() + 0.1f;
    float pol = d.sign();
    if (a.augment) pol *= d.sign();                                  // datasets.py:27-29: the effect is odd in x, so flipping the pair == flipping x first
    const bool want_pink = ch == 1 || ch == 7 || ch == 100;

    // ---- 1/f noise (audio.py:85-94)
    float2* fa = reinterpret_cast<float2*>(feed_lds);
    float pink_peak = 1.f;
    if (want_pink && !a.pink_in) {                                   // workgroup-uniform
        int logN = 0; while ((1 << logN) < L) ++logN;
        for (int k = threadIdx.x; k < L; k += 256) {
            const int kk = k <= L / 2 ? k : L - k;                   // Hermitian extension of a real spectrum: X[N - k] = X[k]
            const float v = (2.f * su01(key, 7u, (unsigned)kk) - 1.f) * rsqrtf((float)kk + 1.f);
            fa[__brev((unsigned)k) >> (32 - logN)] = make_float2(v, 0.f);
        }
        __syncthreads();
        ifft_lds(fa, L, logN);
        float m = 0.f;
        for (int n = threadIdx.x; n < L; n += 256) m = fmaxf(m, fabsf(fa[n].x));
        pink_peak = fmaxf(block_max(m, red), 1e-30f);
    }
    if (a.pink_in && a.pink_peak) pink_peak = fmaxf(a.pink_peak[b], 1e-30f);
    auto sine = [&](const float t) { return s_amp[0] * __cosf(s_frq[0] * (t - s_t0[0])) + s_amp[1] * __cosf(s_frq[1] * (t - s_t0[1])); };
    auto plk = [&](const float t) {
        const float env = t < e_t0 ? e_lo : __expf(-e_dec * (t - e_t0)) * e_hi;
        return (p_amp[0] * __sinf(p_frq[0] * (t - p_t0[0])) + p_amp[1] * __sinf(p_frq[1] * (t - p_t0[1])) + p_amp[2] * __sinf(p_frq[2] * (t - p_t0[2]))) * env;
    };
    auto boxv = [&](const int n) { return n < b_up - 1 ? b_h0 : ((n >= b_up && n < b_dn) ? b_h1 : b_h2); };

    // ---- pass 1: the peak normish divides by (families built on randsine / pluck)
    float scale = 1.f;
    if (ch == 0 || ch == 1 || ch == 2 || ch == 7) {
        float m = 0.f;
        for (int n = threadIdx.x; n < L; n += 256) { const float t = (float)n * dt; m = fmaxf(m, fabsf((ch == 0 || ch == 1) ? sine(t) : plk(t))); }
        scale = nrm_u / fmaxf(block_max(m, red), 1e-30f);
    }
    // ---- pass 2: the window -- four consecutive samples per thread and trip (16-byte loads of the long-window noise, 16-byte stores), and with the
    // lane-per-window form of the effect the static gain curve of each sample right away, from the value in registers.  (Round 3 wrote the window and
    // then re-read it sample by sample in a third loop: without __restrict__ every trip's load waited for the previous trip's store -- 256 dependent
    // memory round trips per thread at the 65536-sample window, most of the generator's 300 us per window.)
    float* __restrict__ xb = a.x + (size_t)b * L;
    float* __restrict__ gcb = a.gc ? a.gc + (size_t)b * L : nullptr;
    const float* __restrict__ pin = a.pink_in ? a.pink_in + (size_t)b * L : nullptr;
    const float inv_peak = 1.0f / pink_peak;
    auto sample = [&](const int n, const float pk) {
        const float t = (float)n * dt;
        float v;
        if (ch == 0) v = sine(t) * scale;
        else if (ch == 1) v = sine(t) * scale + c_pink1 * pk + c_white1 * (2.f * su01(key, 3u, (unsigned)n) - 1.f);
        else if (ch == 2) v = plk(t) * scale;
        else if (ch == 4) v = boxv(n);
        else if (ch == 6) v = boxv(n) * (2.f * su01(key, 3u, (unsigned)n) - 1.f);
        else if (ch == 100) v = pk;
        else v = plk(t) * scale + c_pink7 * pk;
        return ch == 100 ? v : v * pol + su01(key, 5u, (unsigned)n) * 1e-8f;       // audio.py:333
    };
    if ((L & 3) == 0) {
        for (int n4 = threadIdx.x; n4 < L / 4; n4 += 256) {
            const int n = 4 * n4;
            float4 pk = make_float4(0.f, 0.f, 0.f, 0.f);
            if (want_pink) {
                if (pin) { pk = *reinterpret_cast<const float4*>(pin + n); if (a.pink_peak) { pk.x *= inv_peak; pk.y *= inv_peak; pk.z *= inv_peak; pk.w *= inv_peak; } }
                else pk = make_float4(fa[n].x * inv_peak, fa[n + 1].x * inv_peak, fa[n + 2].x * inv_peak, fa[n + 3].x * inv_peak);
            }
            const float4 v = make_float4(sample(n, pk.x), sample(n + 1, pk.y), sample(n + 2, pk.z), sample(n + 3, pk.w));
            *reinterpret_cast<float4*>(xb + n) = v;
            if (gcb) *reinterpret_cast<float4*>(gcb + n) = make_float4(stm::comp_gain_curve(v.x, (double)kw[0], (double)kw[1]), stm::comp_gain_curve(v.y, (double)kw[0], (double)kw[1]),
                                                                       stm::comp_gain_curve(v.z, (double)kw[0], (double)kw[1]), stm::comp_gain_curve(v.w, (double)kw[0], (double)kw[1]));
        }
    } else {
        for (int n = threadIdx.x; n < L; n += 256) {
            const float pk = want_pink ? (pin ? (a.pink_peak ? pin[n] * inv_peak : pin[n]) : fa[n].x * inv_peak) : 0.f;
            const float v = sample(n, pk);
            xb[n] = v;
            if (gcb) gcb[n] = stm::comp_gain_curve(v, (double)kw[0], (double)kw[1]);
        }
    }
    if (threadIdx.x == 0) { for (int k = 0; k < 4; ++k) if (k < a.K) a.knobs[(size_t)b * a.K + k] = kn[k]; }
    if (a.gc) {                                                      // workgroup-uniform: the recurrence + apply run lane-per-window (comp_smooth_kernel, comp_apply_kernel)
        if (threadIdx.x == 0) { for (int k = 0; k < 4; ++k) a.kw[(size_t)b * 4 + k] = kw[k]; }
        return;
    }
    __threadfence_block();
    __syncthreads();                                                 // the window is complete (and the FFT buffer is dead)
    // ---- the effect (audio.py:380-426) on the finished window, inside this workgroup
    const double alphaA = exp(-log(9.0) / ((double)a.sr * (double)kw[2])), alphaR = exp(-log(9.0) / ((double)a.sr * (double)kw[3]));
    stm::compressor_window(xb, a.y + (size_t)b * a.ysz, (double)kw[0], (double)kw[1], alphaA, alphaR, L, a.ysz, feed_lds, &carry);
}

}  // namespace stf
