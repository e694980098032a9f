// Half-length complex FFT in LDS shared by the STFT front end (stft.hip) and the ISTFT back end
// (istft.hip): Stockham autosort passes of radix 5/4/2 against an exact twiddle table.
#pragma once
#include "common.h"

#define FFT_SIGS 8             // transforms per LDS pass (microphones in the STFT, frames in the ISTFT)
#define FFT_MAX_PASSES 8

struct FftPlan {
    int npass;
    int radix[FFT_MAX_PASSES];
};

__device__ __forceinline__ float2 cmul(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }

// One Stockham pass (decimation in frequency, autosort) of NT-point transforms: sub-length n,
// stride s, radix R:
//   y[q + s(R p + k)] = (sum_j x[q + s(p + j n/R)] W_R^{jk}) * W_n^{pk},   p < n/R, q < s.
// tw[j] = exp(-2 pi i j / TW) with NT | TW, so W_n^x = tw[x * TW/n].
template <int R>
__device__ __forceinline__ void fft_pass(const float2* __restrict__ x, float2* __restrict__ y, const float2* __restrict__ tw,
                                         int NT, int TW, int n, int s, int tid, int nthreads) {
    const int m = n / R, per = NT / R;                        // butterflies per transform
    const int twstep = TW / n;
    for (int e = tid; e < FFT_SIGS * per; e += nthreads) {
        const int c = e / per, bfly = e - c * per;
        const int p = bfly / s, q = bfly - p * s;
        const float2* xi = x + c * NT + q + s * p;
        float2* yo = y + c * NT + q + s * R * p;
        float2 a[R];
#pragma unroll
        for (int j = 0; j < R; ++j) a[j] = xi[s * m * j];
        float2 o[R];
        if (R == 2) {
            o[0] = make_float2(a[0].x + a[1].x, a[0].y + a[1].y);
            o[1] = make_float2(a[0].x - a[1].x, a[0].y - a[1].y);
        } else if (R == 4) {
            const float2 t0 = make_float2(a[0].x + a[2].x, a[0].y + a[2].y), t1 = make_float2(a[0].x - a[2].x, a[0].y - a[2].y);
            const float2 t2 = make_float2(a[1].x + a[3].x, a[1].y + a[3].y), t3 = make_float2(a[1].x - a[3].x, a[1].y - a[3].y);
            o[0] = make_float2(t0.x + t2.x, t0.y + t2.y);
            o[2] = make_float2(t0.x - t2.x, t0.y - t2.y);
            o[1] = make_float2(t1.x + t3.y, t1.y - t3.x);     // t1 - i t3   (W_4 = -i)
            o[3] = make_float2(t1.x - t3.y, t1.y + t3.x);     // t1 + i t3
        } else {                                              // generic small radix (5): direct DFT with table twiddles
#pragma unroll
            for (int k = 0; k < R; ++k) {
                float2 acc = a[0];
#pragma unroll
                for (int j = 1; j < R; ++j) {
                    const float2 t = cmul(a[j], tw[((j * k) % R) * (TW / R)]);
                    acc.x += t.x;
                    acc.y += t.y;
                }
                o[k] = acc;
            }
        }
        yo[0] = o[0];
#pragma unroll
        for (int k = 1; k < R; ++k) yo[s * k] = (m > 1) ? cmul(o[k], tw[p * k * twstep]) : o[k];
    }
}


// All passes of the plan over FFT_SIGS transforms of length NT; returns the buffer holding the result.
__device__ __forceinline__ float2* fft_run(float2* src, float2* dst, const float2* tw, int NT, int TW, const FftPlan& plan,
                                           int tid, int nthreads) {
    int n = NT, s = 1;
    for (int ps = 0; ps < plan.npass; ++ps) {
        const int R = plan.radix[ps];
        if (R == 5) fft_pass<5>(src, dst, tw, NT, TW, n, s, tid, nthreads);
        else if (R == 4) fft_pass<4>(src, dst, tw, NT, TW, n, s, tid, nthreads);
        else fft_pass<2>(src, dst, tw, NT, TW, n, s, tid, nthreads);
        __syncthreads();
        n /= R;
        s *= R;
        float2* tmp = src; src = dst; dst = tmp;
    }
    return src;
}

static inline bool fft_plan(int n, FftPlan* p) {
    p->npass = 0;
    // radix 5 first (its generic butterfly is the most expensive one and runs once), then 4s, then a 2
    while (n % 5 == 0 && p->npass < FFT_MAX_PASSES) { p->radix[p->npass++] = 5; n /= 5; }
    while (n % 4 == 0 && p->npass < FFT_MAX_PASSES) { p->radix[p->npass++] = 4; n /= 4; }
    while (n % 2 == 0 && p->npass < FFT_MAX_PASSES) { p->radix[p->npass++] = 2; n /= 2; }
    return n == 1;
}

