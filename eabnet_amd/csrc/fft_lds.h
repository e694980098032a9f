// Half-length complex FFT in LDS shared by the STFT front end (stft.hip) and the ISTFT back end
// (istft.hip): Stockham autosort passes of radix 5/4/8/2 against an exact twiddle table.
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
        } else if (R == 8) {
            // radix 8, decimation in frequency: even outputs = DFT4 of the sums, odd outputs = DFT4 of the differences rotated by
            // W8^j (W8 = exp(-2 pi i / 8) = (1 - i) / sqrt 2, W8^2 = -i, W8^3 = (-1 - i) / sqrt 2)
            const float h = 0.70710678118654752440f;
            float2 b[8];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                b[j] = make_float2(a[j].x + a[j + 4].x, a[j].y + a[j + 4].y);
                b[j + 4] = make_float2(a[j].x - a[j + 4].x, a[j].y - a[j + 4].y);
            }
            b[5] = make_float2(h * (b[5].x + b[5].y), h * (b[5].y - b[5].x));
            b[6] = make_float2(b[6].y, -b[6].x);
            b[7] = make_float2(h * (b[7].y - b[7].x), -h * (b[7].x + b[7].y));
#pragma unroll
            for (int q = 0; q < 2; ++q) {                     // q = 0: X0, X2, X4, X6;  q = 1: X1, X3, X5, X7
                const float2* x = b + 4 * q;
                const float2 t0 = make_float2(x[0].x + x[2].x, x[0].y + x[2].y), t1 = make_float2(x[0].x - x[2].x, x[0].y - x[2].y);
                const float2 t2 = make_float2(x[1].x + x[3].x, x[1].y + x[3].y), t3 = make_float2(x[1].x - x[3].x, x[1].y - x[3].y);
                o[q] = make_float2(t0.x + t2.x, t0.y + t2.y);
                o[q + 4] = make_float2(t0.x - t2.x, t0.y - t2.y);
                o[q + 2] = make_float2(t1.x + t3.y, t1.y - t3.x);
                o[q + 6] = make_float2(t1.x - t3.y, t1.y + t3.x);
            }
        } else if (R == 5) {
            // radix 5 with the symmetric / antisymmetric pairs (W = exp(-2 pi i / 5) = tw[TW/5], W^2 = tw[2 TW/5]):
            //   X0 = a0 + t1 + t2;  X1,4 = m1 -+ i n1;  X2,3 = m2 -+ i n2   with  t1 = a1 + a4, t2 = a2 + a3, t3 = a1 - a4,
            //   t4 = a2 - a3,  m1 = a0 + c1 t1 + c2 t2,  m2 = a0 + c2 t1 + c1 t2,  n1 = s1 t3 + s2 t4,  n2 = s2 t3 - s1 t4
            // (~40 flops; the direct 5 x 4 complex products it replaces were 120 and twenty table reads)
            const float2 w1 = tw[TW / 5], w2 = tw[2 * (TW / 5)];
            const float c1 = w1.x, s1 = -w1.y, c2 = w2.x, s2 = -w2.y;
            const float2 t1 = make_float2(a[1].x + a[4].x, a[1].y + a[4].y), t2 = make_float2(a[2].x + a[3].x, a[2].y + a[3].y);
            const float2 t3 = make_float2(a[1].x - a[4].x, a[1].y - a[4].y), t4 = make_float2(a[2].x - a[3].x, a[2].y - a[3].y);
            o[0] = make_float2(a[0].x + t1.x + t2.x, a[0].y + t1.y + t2.y);
            const float2 m1 = make_float2(a[0].x + c1 * t1.x + c2 * t2.x, a[0].y + c1 * t1.y + c2 * t2.y);
            const float2 m2 = make_float2(a[0].x + c2 * t1.x + c1 * t2.x, a[0].y + c2 * t1.y + c1 * t2.y);
            const float2 n1 = make_float2(s1 * t3.x + s2 * t4.x, s1 * t3.y + s2 * t4.y);
            const float2 n2 = make_float2(s2 * t3.x - s1 * t4.x, s2 * t3.y - s1 * t4.y);
            o[1] = make_float2(m1.x + n1.y, m1.y - n1.x);
            o[4] = make_float2(m1.x - n1.y, m1.y + n1.x);
            o[2] = make_float2(m2.x + n2.y, m2.y - n2.x);
            o[3] = make_float2(m2.x - n2.y, m2.y + n2.x);
        } else {                                              // generic small radix: direct DFT with table twiddles
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
        else if (R == 8) fft_pass<8>(src, dst, tw, NT, TW, n, s, tid, nthreads);
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
    // (a trailing 4 . 2 becomes one radix-8 pass: one pass and one barrier fewer, and the last pass has no twiddles)
    while (n % 4 == 0 && n != 8 && p->npass < FFT_MAX_PASSES) { p->radix[p->npass++] = 4; n /= 4; }
    if (n == 8) { p->radix[p->npass++] = 8; n = 1; }
    while (n % 2 == 0 && p->npass < FFT_MAX_PASSES) { p->radix[p->npass++] = 2; n /= 2; }
    return n == 1;
}

