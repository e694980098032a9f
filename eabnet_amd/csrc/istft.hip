// ISTFT back end: (B,2,T,F) estimate -> (B, hop*(T-1)) wave.
// Reference: enhance.py:59-62 / test.py:189-191 / train_distributed.py:128-130
//   esti.permute(0,3,2,1) -> view_as_complex -> torch.istft(n_fft=320, hop=160, win=320, hann)
// i.e. (torch.istft defaults: center, onesided, not normalized, length=None)
//   frame_t = irfft(X[:, t]) * w;   y = overlap_add(frame) / overlap_add(w^2);   trim n_fft/2 per side.
// Implemented for hop = n_fft/2 (the reference's 320/160): after the centre trim EVERY output sample
// is covered by exactly two frames, so there is no scatter and no envelope buffer --
//   y[hop k + n] = (w[n+hop] x_k[n+hop] + w[n] x_{k+1}[n]) / (w[n+hop]^2 + w[n]^2),  k < T-1, n < hop.
//
// One workgroup inverts FFT_SIGS consecutive frames of one utterance in LDS and emits the
// FFT_SIGS-1 segments between them (the next workgroup re-inverts the shared frame: 1/7 extra
// reads, no inter-workgroup dependency).  Real inverse FFT by the even/odd split: with
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
                                                              int T, int n_fft, int chunks, FftPlan plan) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int NH = n_fft / 2, F = NH + 1;
    float2* tw = reinterpret_cast<float2*>(smem);                 // [n_fft] exp(-2 pi i j / n_fft)
    float2* buf0 = tw + n_fft;                                    // [FFT_SIGS][NH]
    float2* buf1 = buf0 + FFT_SIGS * NH;
    float* win = reinterpret_cast<float*>(buf1 + FFT_SIGS * NH);  // [n_fft]
    float* xs = win + n_fft;                                      // [FFT_SIGS][2][F] staged spectrum rows
    const int tid = threadIdx.x;
    const int b = blockIdx.x / chunks, chunk = blockIdx.x - b * chunks;
    const int t0 = chunk * (FFT_SIGS - 1);

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
    for (int e = tid; e < (FFT_SIGS - 1) * NH; e += ISTFT_THREADS) {
        const int c = e / NH, n = e - c * NH;
        const int k = t0 + c;
        if (k + 1 >= T) break;                                     // e grows with c: nothing further is live
        const int p0 = n + NH;                                     // sample of frame k; float index = p (re/im interleave)
        float a0 = yf[c * n_fft + p0], a1 = yf[(c + 1) * n_fft + n];
        if (p0 & 1) a0 = -a0;
        if (n & 1) a1 = -a1;
        const float w0 = win[p0], w1 = win[n];
        wav[(size_t)b * NH * (T - 1) + (size_t)k * NH + n] = (w0 * (a0 * inv) + w1 * (a1 * inv)) / (w0 * w0 + w1 * w1);
    }
}

extern "C" int eab_istft_f32(const float* spec, const float* window, const float* twiddle, float* wav, int B, int T,
                             int n_fft, int hop, eab_stream_t stream) {
    EAB_CHECK_ARG(spec && window && twiddle && wav);
    EAB_CHECK_ARG(B > 0 && T >= 2);
    EAB_CHECK_ARG(n_fft >= 4 && n_fft <= ISTFT_MAX_NFFT && (n_fft % 2) == 0);
    if (hop * 2 != n_fft) return EAB_EUNSUPPORTED;                  // the two-frame closed form needs hop = n_fft/2
    FftPlan plan;
    if (!fft_plan(n_fft / 2, &plan)) return EAB_EUNSUPPORTED;
    const int chunks = (T - 1 + FFT_SIGS - 2) / (FFT_SIGS - 1);
    EAB_CHECK_ARG((long long)B * chunks < (1ll << 31));
    const size_t sh = (size_t)(2 * n_fft + 2 * FFT_SIGS * n_fft + n_fft + FFT_SIGS * 2 * (n_fft / 2 + 1)) * sizeof(float);
    hipLaunchKernelGGL(istft_kernel, dim3(B * chunks), dim3(ISTFT_THREADS), sh, eab_stream(stream), spec, window, twiddle,
                       wav, T, n_fft, chunks, plan);
    EAB_RETURN_LAUNCH_STATUS();
}
