/* st_audio.c -- host-side helper for the synthetic comp_4c data feed (CPU workers, not the GPU hot path).
 * Replaces the numba-jitted sequential loop of signaltrain/audio.py:380-426 (compressor_4controls):
 * a switched one-pole smoother of the static gain curve, attack coefficient when the gain is falling,
 * release coefficient otherwise.  Arithmetic is done in double like numpy's float64 path. */
#include <math.h>
#include <stddef.h>

void st_compressor_4controls(const float* x, float* y, size_t n, double thresh, double ratio,
                             double attackTime, double releaseTime, double sr)
{
    const double alphaA = exp(-log(9.0) / (sr * attackTime));
    const double alphaR = exp(-log(9.0) / (sr * releaseTime));
    double prev = 0.0;                     /* lin_A[0] = 0 (audio.py:402) */
    if (n) y[0] = x[0];                    /* 10^(0/20) * x[0] */
    for (size_t i = 1; i < n; ++i) {
        float xf = x[i];
        float xdb = (float)(20.0 * log10((double)fabsf(xf) + 1e-8));   /* float32 array in the reference */
        if (xdb < -96.0f) xdb = -96.0f;
        float gc = 0.0f;
        if ((double)xdb > thresh) gc = (float)(thresh + ((double)xdb - thresh) / ratio - (double)xdb);
        double g = gc;
        if (g < prev) prev = (float)((1.0 - alphaA) * g + alphaA * prev);
        else prev = (float)((1.0 - alphaR) * g + alphaR * prev);
        y[i] = (float)pow(10.0, prev / 20.0) * xf;
    }
}
