// ISTFT back end: (B,2,T,F) estimate -> (B, hop*(T-1)) wave.
// Reference: enhance.py:59-62 / test.py:189-191 / train_distributed.py:128-130
//   esti.permute(0,3,2,1) -> view_as_complex -> torch.istft(n_fft=320, hop=160, win=320, hann)
// i.e. (torch.istft defaults: center, onesided, not normalized, length=None)
//   frame_t = irfft(X[:, t]) * w;   y = overlap_add(frame) / overlap_add(w^2);   trim n_fft/2 per side.
// Any hop with R = ceil(n_fft/hop) <= 8 frames per sample (the reference: 320/160, R = 2; the hop need not divide n_fft): the
// padded position p is covered by the frames t = floor(p/hop), floor(p/hop)-1, ... while t hop + n_fft > p, so there is no
// scatter and no envelope buffer --
//   y[p] = sum_t w[p - t hop] x_t[p - t hop] / sum_t w[p - t hop]^2,   frames outside [0, T) skipped.
//
// One workgroup inverts FFT_SIGS consecutive frames t0 .. t0+7 of one utterance in LDS and emits the positions whose
// covering frames it holds completely: [(t0-1) hop + n_fft, (t0+8) hop); consecutive workgroups start per = FFT_SIGS-(R-1)
// frames apart, so their ranges tile the wave (the next workgroup re-inverts the R-1 shared frames: no inter-workgroup
// dependency).  Real inverse FFT by the even/odd split: with
// E = (X[k] + conj X[N/2-k])/2 and O = (X[k] - conj X[N/2-k])/2 * e^{+2 pi i k/N},
// z = IDFT_{N/2}(E + iO) holds x[2n] + i x[2n+1]; the inverse transform runs as
// conj(FFT(conj .)) on the forward Stockham passes of fft_lds.h.  As in a C2R transform the
// imaginary parts of the DC and Nyquist bins are ignored.
// Bound: HBM (reads 2*F*4 B, writes hop*4 B per frame); at the reference sizes it is launch/latency
// sized (13 MB per 16-utterance batch).
#include "common.h"
#include "fft_lds.h"

#define ISTFT_THREADS 256
#define ISTFT_MAX_NFFT 512

__global__ __launch_bounds__(ISTFT_THREADS) void istft_kernel(const float* __restrict__ spec, const float* __restrict__ window,
                                                              const float* __restrict__ twiddle, float* __restrict__ wav,
                                                              int T, int n_fft, int hop, int chunks, FftPlan plan) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int NH = n_fft / 2, F = NH + 1;
    const int R = (n_fft + hop - 1) / hop;                        // most frames that cover one output sample (2 for the reference)
    float2* tw = reinterpret_cast<float2*>(smem);                 // [n_fft] exp(-2 pi i j / n_fft)
    float2* buf0 = tw + n_fft;                                    // [FFT_SIGS][NH]
    float2* buf1 = buf0 + FFT_SIGS * NH;
    float* win = reinterpret_cast<float*>(buf1 + FFT_SIGS * NH);  // [n_fft]
    float* xs = win + n_fft;                                      // [FFT_SIGS][2][F] staged spectrum rows
    const int tid = threadIdx.x;
    const int b = blockIdx.x / chunks, chunk = blockIdx.x - b * chunks;
    const int t0 = chunk * (FFT_SIGS - (R - 1));                  // consecutive workgroups share R-1 frames

    for (int k = tid; k < n_fft; k += ISTFT_THREADS) {
        const float2 cs = reinterpret_cast<const float2*>(twiddle)[k];     // (cos, sin)(+theta)
        tw[k] = make_float2(cs.x, -cs.y);
        win[k] = window[k];
    }
    // stage re/im rows of the frames (coalesced; the split below reads them mirrored)
    for (int e = tid; e < FFT_SIGS * 2 * F; e += ISTFT_THREADS) {
        const int c = e / (2 * F), r = e - c * 2 * F;
        const int ri = r / F, f = r - ri * F;
        const int t = t0 + c;
        xs[e] = t < T ? spec[(((size_t)b * 2 + ri) * T + t) * F + f] : 0.0f;
    }
    __syncthreads();
    // conj(Z[k]),  Z = E + iO
    for (int e = tid; e < FFT_SIGS * NH; e += ISTFT_THREADS) {
        const int c = e / NH, k = e - c * NH;
        const float* re = xs + c * 2 * F;
        const float* im = re + F;
        float2 xk = make_float2(re[k], im[k]);
        float2 xm = make_float2(re[NH - k], im[NH - k]);
        if (k == 0) xk.y = xm.y = 0.0f;                            // C2R: DC and Nyquist are real
        const float2 E = make_float2(0.5f * (xk.x + xm.x), 0.5f * (xk.y - xm.y));
        const float2 D = make_float2(0.5f * (xk.x - xm.x), 0.5f * (xk.y + xm.y));
        const float2 O = cmul(D, make_float2(tw[k].x, -tw[k].y)); // * e^{+i theta_k}
        buf0[e] = make_float2(E.x - O.y, -(E.y + O.x));
    }
    __syncthreads();
    const float2* y = fft_run(buf0, buf1, tw, NH, n_fft, plan, tid, ISTFT_THREADS);
    // x_c[2j] = Re y_c[j] / NH,  x_c[2j+1] = -Im y_c[j] / NH
    const float inv = 1.0f / (float)NH;
    const float* yf = reinterpret_cast<const float*>(y);
    // Overlap-add over the frames that cover a sample, divided by the squared-window envelope of the frames that exist
    // (torch.istft), then the centre trim.  This workgroup holds frames t0 .. t0+7 and emits the padded positions
    // [p_lo, p_hi) whose covering frames it holds completely (the first workgroup: from 0); highest frame first.
    const int per = FFT_SIGS - (R - 1);
    const int p_lo = chunk == 0 ? 0 : t0 * hop + n_fft - hop;
    const int p_hi = (t0 + per) * hop + n_fft - hop;
    const int out_len = hop * (T - 1);
    for (int e = tid; e < p_hi - p_lo; e += ISTFT_THREADS) {
        const int p = p_lo + e;
        const int j = p - NH;                                      // output sample (centre trim of n_fft/2)
        if (j < 0 || j >= out_len) continue;
        const int th = p / hop;
        float acc = 0.0f, env = 0.0f;
        for (int r = 0; r < R; ++r) {
            const int t = th - r;
            const int idx = p - t * hop;                           // sample of frame t; float index = idx (re/im interleave)
            if (t < 0 || idx >= n_fft) break;
            if (t >= T) continue;
            float a = yf[(t - t0) * n_fft + idx];
            if (idx & 1) a = -a;
            const float w = win[idx];
            acc = fmaf(w, a * inv, acc);
            env = fmaf(w, w, env);
        }
        wav[(size_t)b * out_len + j] = acc / env;
    }
}

extern "C" int eab_istft_f32(const float* spec, const float* window, const float* twiddle, float* wav, int B, int T,
                             int n_fft, int hop, eab_stream_t stream) {
    EAB_CHECK_ARG(spec && window && twiddle && wav);
    EAB_CHECK_ARG(B > 0 && T >= 2 && hop > 0);
    EAB_CHECK_ARG(n_fft >= 4 && n_fft <= ISTFT_MAX_NFFT && (n_fft % 2) == 0);
    // at most FFT_SIGS frames may cover a sample (R = ceil(n_fft / hop) <= 8: 87.5 % overlap); torch.istft wants hop <= win
    if (hop > n_fft || (n_fft + hop - 1) / hop > FFT_SIGS) return EAB_EUNSUPPORTED;
    FftPlan plan;
    if (!fft_plan(n_fft / 2, &plan)) return EAB_EUNSUPPORTED;
    const int R = (n_fft + hop - 1) / hop, per = FFT_SIGS - (R - 1);
    // padded positions [0, n_fft/2 + hop (T-1)) carry the trimmed output; the workgroup of chunk c ends at
    // (c+1) per hop + n_fft - hop
    const long long need = (long long)n_fft / 2 + (long long)hop * (T - 1) - (n_fft - hop);
    const int chunks = need <= 0 ? 1 : (int)((need + (long long)per * hop - 1) / ((long long)per * hop));
    EAB_CHECK_ARG((long long)B * chunks < (1ll << 31));
    const size_t sh = (size_t)(2 * n_fft + 2 * FFT_SIGS * n_fft + n_fft + FFT_SIGS * 2 * (n_fft / 2 + 1)) * sizeof(float);
    hipLaunchKernelGGL(istft_kernel, dim3(B * chunks), dim3(ISTFT_THREADS), sh, eab_stream(stream), spec, window, twiddle,
                       wav, T, n_fft, hop, chunks, plan);
    EAB_RETURN_LAUNCH_STATUS();
}
