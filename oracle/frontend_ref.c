/*
 * CPU ORACLE (plain C) -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 * Independent restatement, without PyTorch, of the two ends of the hot path:
 *   - prepare_data's noisy branch: torch.stft(n_fft, hop, win = n_fft, window, center=True,
 *     pad_mode='reflect', onesided) + sqrt-magnitude compression
 *     (reference train_distributed.py:80,83,86,89,91);
 *   - the complex filter-and-sum, reference EaBNet.py:114-117.
 * The DFT is the O(N^2) definition in double precision, so it shares no code path with either
 * torch.stft (pocketfft/MKL) or the HIP FFT.  Pinned by tests/test_oracle_golden.py against the
 * fixtures generated from the reference.  Only tests/ may load this.
 */
#include <math.h>
#include <stddef.h>

static long reflect_index(long i, long pad, long L) {   /* index into the un-padded wave */
    long j = i - pad;
    if (j < 0) j = -j;
    if (j >= L) j = 2 * (L - 1) - j;
    return j;
}

/* wav [B][M][L], window [n_fft] -> out [B][T][F][M][2], T = 1 + L/hop, F = n_fft/2 + 1 */
int oracle_stft_compress(const float* wav, const float* window, float* out, int B, int M, long L, int n_fft, int hop) {
    const double two_pi = 6.283185307179586476925286766559;
    const long T = 1 + L / hop;
    const int F = n_fft / 2 + 1;
    if (L <= n_fft / 2) return 1;
    for (int b = 0; b < B; ++b)
        for (int m = 0; m < M; ++m) {
            const float* x = wav + ((size_t)b * M + m) * L;
            for (long t = 0; t < T; ++t)
                for (int f = 0; f < F; ++f) {
                    double re = 0.0, im = 0.0;
                    for (int n = 0; n < n_fft; ++n) {
                        /* the window multiplies in fp32, as torch.stft does before its FFT */
                        const double v = (double)(window[n] * x[reflect_index(t * hop + n, n_fft / 2, L)]);
                        const double ph = two_pi * (double)(((long)f * n) % n_fft) / (double)n_fft;
                        re += v * cos(ph);
                        im -= v * sin(ph);
                    }
                    /* mag = |X|^0.5, phase kept: X * |X|^-1/2, 0 -> 0 */
                    const double mag = sqrt(re * re + im * im);
                    const double s = mag > 0.0 ? 1.0 / sqrt(mag) : 0.0;
                    float* o = out + ((((size_t)b * T + t) * F + f) * M + m) * 2;
                    o[0] = (float)(re * s);
                    o[1] = (float)(im * s);
                }
        }
    return 0;
}

/* w, x [bins][M][2] -> y_r [bins], y_i [bins]:  Y = sum_m W_m * X_m  (no conjugate) */
void oracle_filter_sum(const float* w, const float* x, float* yr, float* yi, long bins, int M) {
    for (long i = 0; i < bins; ++i) {
        double r = 0.0, q = 0.0;
        for (int m = 0; m < M; ++m) {
            const double wr = w[(i * M + m) * 2], wi = w[(i * M + m) * 2 + 1];
            const double xr = x[(i * M + m) * 2], xi = x[(i * M + m) * 2 + 1];
            r += wr * xr - wi * xi;
            q += wr * xi + wi * xr;
        }
        yr[i] = (float)r;
        yi[i] = (float)q;
    }
}
