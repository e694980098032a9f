// K1+K2+K3: fused STFT front end  (reference train_distributed.py:80-92).
//
// One workgroup per (b, t): all M microphones of one frame, because the output
// tensor (B,T,F,M,2) keeps the M*2 values of a TF bin contiguous -- the
// workgroup's output is ONE contiguous F*M*2 block (coalesced stores), and the
// twiddle pair of (bin f, sample n) is read once and used for all M mics.
//
// Arithmetic: real FFT.  The even and odd samples of ONE microphone share a
// complex transform of half the length (z[n] = x[2n] + i x[2n+1], split by
// conjugate symmetry and one twiddle afterwards) -- never two microphones, so a
// silent microphone gives exactly 0 as in the reference and no rounding noise
// leaks across channels (sqrt-compression would amplify 1e-8 to 1e-4).  The
// n_fft/2-point complex FFT is a Stockham autosort chain of radix-5/4/2 passes
// in LDS (160 = 5*4*8) against an exact (host, fp64-rounded) twiddle table, so the
// kernel does ~6.6 kFLOP per frame and mic and is bound by its HBM traffic
// (5,120 B read + 20,608 B written per frame at M = 8).  Sizes that do not
// factor into {5,4,2} fall back to a direct DFT kernel.
#include "common.h"
#include "fft_lds.h"

#define STFT_MAX_NFFT 512
#define STFT_THREADS 256
#define STFT_MC 8  // mics per pass held in registers

// reflect_pad(x, P)[i] for i in [0, L + 2P): index into the un-padded wave.
__device__ __forceinline__ int reflect_index(int i, int P, int L) {
    int j = i - P;
    if (j < 0) j = -j;
    if (j >= L) j = 2 * (L - 1) - j;
    return j;
}

__global__ __launch_bounds__(STFT_THREADS) void stft_dft_kernel(
    const float* __restrict__ wav, const float* __restrict__ window, const float* __restrict__ twiddle,
    float* __restrict__ out, int M, int L, int n_fft, int hop, int T, int layout) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float2* tw = reinterpret_cast<float2*>(smem);           // [n_fft] (cos, sin)
    float* fr = smem + 2 * n_fft;                           // [n_fft][STFT_MC]
    const int F = n_fft / 2 + 1;
    const int b = blockIdx.x / T, t = blockIdx.x % T;
    const int tid = threadIdx.x;

    for (int k = tid; k < n_fft; k += STFT_THREADS) tw[k] = reinterpret_cast<const float2*>(twiddle)[k];

    for (int m0 = 0; m0 < M; m0 += STFT_MC) {
        __syncthreads();   // previous pass done with fr (and tw visible on first pass)
        // gather + window: fr[n][mm] = window[n] * wav[b][m0+mm][reflect(t*hop + n)]
        for (int e = tid; e < n_fft * STFT_MC; e += STFT_THREADS) {
            int mm = e / n_fft, n = e - mm * n_fft;        // consecutive threads -> consecutive samples
            float v = 0.0f;
            if (m0 + mm < M) {
                int j = reflect_index(t * hop + n, n_fft / 2, L);
                v = window[n] * wav[((size_t)b * M + m0 + mm) * L + j];
            }
            fr[n * STFT_MC + mm] = v;
        }
        __syncthreads();
        for (int f = tid; f < F; f += STFT_THREADS) {
            float re[STFT_MC], im[STFT_MC];
#pragma unroll
            for (int mm = 0; mm < STFT_MC; ++mm) re[mm] = im[mm] = 0.0f;
            int idx = 0;                                    // (f*n) mod n_fft
            for (int n = 0; n < n_fft; ++n) {
                float2 cs = tw[idx];
                const f32x4 x0 = *reinterpret_cast<const f32x4*>(&fr[n * STFT_MC]);
                const f32x4 x1 = *reinterpret_cast<const f32x4*>(&fr[n * STFT_MC + 4]);
                float x[STFT_MC] = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
#pragma unroll
                for (int mm = 0; mm < STFT_MC; ++mm) {
                    re[mm] = fmaf(x[mm], cs.x, re[mm]);     // X = sum x * exp(-i 2 pi f n / N)
                    im[mm] = fmaf(-x[mm], cs.y, im[mm]);
                }
                idx += f;
                if (idx >= n_fft) idx -= n_fft;
            }
#pragma unroll
            for (int mm = 0; mm < STFT_MC; ++mm) {
                if (m0 + mm >= M) break;
                // sqrt-magnitude compression with the phase kept: X * |X|^-1/2, 0 -> 0
                float mag = sqrtf(re[mm] * re[mm] + im[mm] * im[mm]);
                float s = mag > 0.0f ? 1.0f / sqrtf(mag) : 0.0f;
                float yr = re[mm] * s, yi = im[mm] * s;
                if (layout == EAB_STFT_LAYOUT_BTFM2) {
                    size_t o = ((((size_t)b * T + t) * F + f) * M + (m0 + mm)) * 2;
                    *reinterpret_cast<float2*>(&out[o]) = make_float2(yr, yi);
                } else {                                    // (B,2,T,F), M == 1
                    out[(((size_t)b * 2 + 0) * T + t) * F + f] = yr;
                    out[(((size_t)b * 2 + 1) * T + t) * F + f] = yi;
                }
            }
        }
    }
}

// NFFT_CT / HOP_CT != 0: the reference's front end (fft_num 320, hop 160) with every size a compile-time constant -- the
// same passes and the same arithmetic as the run-time plan (bit-identical results), but the index divisions fold into
// multiplies and shifts.
// DUMP: the verification twin behind eab_stft_frames_f32 -- the SAME gather code (both paths: four samples per load for
// interior frames, reflected scalar loads otherwise) with the window factor left out, and the gathered LDS rows stored
// instead of transformed: frames[b][m][t][n] must equal reflect_pad(wav[b][m], n_fft/2)[t*hop + n] bit for bit
// (train_distributed.py:83, torch.stft(center=True, pad_mode="reflect")).
template <int NFFT_CT, int HOP_CT, bool DUMP>
__global__ __launch_bounds__(STFT_THREADS) void stft_fft_kernel(
    const float* __restrict__ wav, const float* __restrict__ window, const float* __restrict__ twiddle,
    float* __restrict__ out, int M, int L, int n_fft_rt, int hop_rt, int T, int layout, FftPlan plan) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int n_fft = NFFT_CT ? NFFT_CT : n_fft_rt, hop = HOP_CT ? HOP_CT : hop_rt;
    const int NH = n_fft / 2, F = NH + 1;
    float2* tw = reinterpret_cast<float2*>(smem);                         // [n_fft]  exp(-2 pi i j / n_fft)
    float2* buf0 = tw + n_fft;                                            // [FFT_SIGS][NH]
    float2* buf1 = buf0 + FFT_SIGS * NH;
    // Workgroups are dealt to the 8 XCDs round-robin and every XCD has its own L2: neighbouring frames share half of their
    // samples, so let each XCD walk one contiguous eighth of the (b, t) sequence -- the overlap is then fetched from HBM once
    // (measured before: 2 x the wave bytes, profiles/r02_final).
    unsigned vblk = blockIdx.x;
    if (gridDim.x >= 64) {
        const unsigned G = gridDim.x, G8 = G >> 3, rem = G & 7, xcd = vblk & 7, idx = vblk >> 3;
        vblk = xcd * G8 + (xcd < rem ? xcd : rem) + idx;
    }
    // b = vblk / T by a float reciprocal with one correction step (exact for vblk < 2^24; the 32-bit division it replaces is
    // ~25 vector instructions per workgroup); larger grids take the division
    int b;
    if (gridDim.x < (1u << 24)) {
        b = (int)((float)vblk * __builtin_amdgcn_rcpf((float)T));
        const int r = (int)vblk - b * T;
        b += (r >= T) - (r < 0);
    } else {
        b = (int)(vblk / (unsigned)T);
    }
    const int t = (int)vblk - b * T;
    const int tid = threadIdx.x;
    if (!DUMP)
        for (int k = tid; k < n_fft; k += STFT_THREADS) {   // table is (cos, sin)(+theta); the passes use exp(-i theta)
            const float2 cs = reinterpret_cast<const float2*>(twiddle)[k];
            tw[k] = make_float2(cs.x, -cs.y);
        }

    for (int m0 = 0; m0 < M; m0 += FFT_SIGS) {
        __syncthreads();
        // gather + window: z_mm[n/2].(re|im) = w[n] x_mm[reflect(t hop + n)]; consecutive threads -> consecutive samples
        const int first = t * hop - NH;                                   // first sample of the frame (reflect padding: may be < 0)
        if (first >= 0 && first + n_fft <= L && ((first | L | n_fft) & 3) == 0) {
            // interior frame (all but the first and last of an utterance): no reflection, four samples per load
            // 32 threads per microphone walk its n_fft / 4 four-sample groups (no index division)
            const int n4 = n_fft >> 2, mm = tid >> 5;
            static_assert(STFT_THREADS == 32 * FFT_SIGS, "one 32-thread group per staged signal");
            for (int q = tid & 31; q < n4; q += 32) {
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (m0 + mm < M) {
                    const f32x4 x4 = *reinterpret_cast<const f32x4*>(&wav[((size_t)b * M + m0 + mm) * L + first + 4 * q]);
                    if (DUMP) {
                        v = x4;
                    } else {
                        const f32x4 w4 = *reinterpret_cast<const f32x4*>(&window[4 * q]);
                        v = f32x4{w4[0] * x4[0], w4[1] * x4[1], w4[2] * x4[2], w4[3] * x4[3]};
                    }
                }
                *reinterpret_cast<f32x4*>(&reinterpret_cast<float*>(buf0)[mm * n_fft + 4 * q]) = v;
            }
        } else {
            for (int e = tid; e < FFT_SIGS * n_fft; e += STFT_THREADS) {
                const int mm = e / n_fft, n = e - mm * n_fft;
                float v = 0.0f;
                if (m0 + mm < M) {
                    const float x = wav[((size_t)b * M + m0 + mm) * L + reflect_index(t * hop + n, NH, L)];
                    v = DUMP ? x : window[n] * x;
                }
                reinterpret_cast<float*>(buf0)[mm * n_fft + n] = v;       // float index 2*(n/2) + (n&1) = n
            }
        }
        __syncthreads();
        if (DUMP) {                                           // out = frames [B][M][T][n_fft]
            for (int e = tid; e < FFT_SIGS * n_fft; e += STFT_THREADS) {
                const int mm = e / n_fft, n = e - mm * n_fft;
                if (m0 + mm < M)
                    out[(((size_t)b * M + m0 + mm) * T + t) * n_fft + n] = reinterpret_cast<const float*>(buf0)[mm * n_fft + n];
            }
            continue;
        }
        const float2* src;
        if (NFFT_CT == 320) {                                 // 160 = 5 * 4 * 8, unrolled with constant sizes
            // (the trailing radix-8 pass instead of 4 . 2: one pass and one barrier fewer, no twiddles in it: 37 -> 34.5 us)
            fft_pass<5>(buf0, buf1, tw, 160, 320, 160, 1, tid, STFT_THREADS);
            __syncthreads();
            fft_pass<4>(buf1, buf0, tw, 160, 320, 32, 5, tid, STFT_THREADS);
            __syncthreads();
            fft_pass<8>(buf0, buf1, tw, 160, 320, 8, 20, tid, STFT_THREADS);
            __syncthreads();
            src = buf1;
        } else {
            src = fft_run(buf0, buf1, tw, NH, n_fft, plan, tid, STFT_THREADS);
        }
        // X[k] = E[k] + W_N^k O[k],  E = (Z[k] + conj Z[NH-k]) / 2,  O = (Z[k] - conj Z[NH-k]) / 2i;  k = 0..NH.
        // Bins k and NH-k are made by ONE thread from the same two values: E[NH-k] = conj E[k], O[NH-k] = conj O[k] and
        // W^(NH-k) = -conj W^k, so X[NH-k] = conj(E[k] - W^k O[k]) -- half the LDS reads and one complex product per pair
        // (k = 0 pairs DC with Nyquist: Z[NH] = Z[0]; NH even: k = NH/2 is its own partner and is stored once).
        const int NP = NH / 2 + 1;                                       // pairs (k, NH-k), k = 0 .. NH/2
        for (int e = tid; e < NP * FFT_SIGS; e += STFT_THREADS) {
            const int k = e / FFT_SIGS, mm = e - k * FFT_SIGS;           // consecutive threads -> consecutive mics
            const int m = m0 + mm;
            if (m >= M) continue;
            const float2 z = src[mm * NH + k];
            const float2 zc = src[mm * NH + (k == 0 ? 0 : NH - k)];
            const float2 E = make_float2(0.5f * (z.x + zc.x), 0.5f * (z.y - zc.y));
            const float2 O = make_float2(0.5f * (z.y + zc.y), 0.5f * (zc.x - z.x));
            const float2 wo = cmul(tw[k], O);
            const float re0 = E.x + wo.x, im0 = E.y + wo.y;              // X[k]
            const float re1 = E.x - wo.x, im1 = wo.y - E.y;              // X[NH-k]
            // sqrt-magnitude compression with the phase kept: X * |X|^-1/2, 0 -> 0
            // |X|^-1/2 = (re^2 + im^2)^-1/4 on the hardware rsq / sqrt (1 ulp each; the IEEE sqrtf and division sequences this
            // replaces were ~40 VALU instructions per bin); p = 0 or denormal -> 0
            const float p0 = re0 * re0 + im0 * im0, p1 = re1 * re1 + im1 * im1;
            const float sc0 = p0 > 1e-37f ? __builtin_amdgcn_sqrtf(__builtin_amdgcn_rsqf(p0)) : 0.0f;
            const float sc1 = p1 > 1e-37f ? __builtin_amdgcn_sqrtf(__builtin_amdgcn_rsqf(p1)) : 0.0f;
            const bool twice = 2 * k != NH;
            if (layout == EAB_STFT_LAYOUT_BTFM2) {
                float* o = &out[(((size_t)b * T + t) * F * M + m) * 2];
                *reinterpret_cast<float2*>(o + (size_t)k * M * 2) = make_float2(re0 * sc0, im0 * sc0);
                if (twice) *reinterpret_cast<float2*>(o + (size_t)(NH - k) * M * 2) = make_float2(re1 * sc1, im1 * sc1);
            } else {                                        // (B,2,T,F), M == 1
                float* o0 = &out[(((size_t)b * 2 + 0) * T + t) * F], *o1 = &out[(((size_t)b * 2 + 1) * T + t) * F];
                o0[k] = re0 * sc0;
                o1[k] = im0 * sc0;
                if (twice) { o0[NH - k] = re1 * sc1; o1[NH - k] = im1 * sc1; }
            }
        }
    }
}

extern "C" int eab_stft_compress_f32(const float* wav, const float* window, const float* twiddle, float* out,
                                     int B, int M, int L, int n_fft, int hop, int layout, eab_stream_t stream) {
    EAB_CHECK_ARG(wav && window && twiddle && out);
    EAB_CHECK_ARG(B > 0 && M > 0 && hop > 0);
    EAB_CHECK_ARG(n_fft >= 2 && n_fft <= STFT_MAX_NFFT && (n_fft % 2) == 0);
    EAB_CHECK_ARG(L > n_fft / 2);                           // reflect padding needs pad < L (as torch.stft)
    EAB_CHECK_ARG(layout == EAB_STFT_LAYOUT_BTFM2 || (layout == EAB_STFT_LAYOUT_B2TF && M == 1));
    const int T = 1 + L / hop;
    EAB_CHECK_ARG((long long)B * T < (1ll << 31));
    FftPlan plan;
    if (fft_plan(n_fft / 2, &plan)) {
        const size_t sh = (size_t)(2 * n_fft + 2 * FFT_SIGS * n_fft) * sizeof(float);
        // the reference front end (fft 320, hop 160) has its own instance: every size a constant, its passes written out
        // (independent of what fft_plan picks for the generic kernel)
        if (n_fft == 320 && hop == 160)
            hipLaunchKernelGGL((stft_fft_kernel<320, 160, false>), dim3(B * T), dim3(STFT_THREADS), sh, eab_stream(stream), wav, window,
                               twiddle, out, M, L, n_fft, hop, T, layout, plan);
        else
            hipLaunchKernelGGL((stft_fft_kernel<0, 0, false>), dim3(B * T), dim3(STFT_THREADS), sh, eab_stream(stream), wav, window,
                               twiddle, out, M, L, n_fft, hop, T, layout, plan);
        EAB_RETURN_LAUNCH_STATUS();
    }
    size_t shmem = (size_t)(2 * n_fft + n_fft * STFT_MC) * sizeof(float);
    hipLaunchKernelGGL(stft_dft_kernel, dim3(B * T), dim3(STFT_THREADS), shmem, eab_stream(stream), wav,
                       window, twiddle, out, M, L, n_fft, hop, T, layout);
    EAB_RETURN_LAUNCH_STATUS();
}

extern "C" int eab_stft_frames_f32(const float* wav, float* frames, int B, int M, int L, int n_fft, int hop,
                                   eab_stream_t stream) {
    EAB_CHECK_ARG(wav && frames && B > 0 && M > 0 && hop > 0);
    EAB_CHECK_ARG(n_fft >= 2 && n_fft <= STFT_MAX_NFFT && (n_fft % 2) == 0 && L > n_fft / 2);
    const int T = 1 + L / hop;
    EAB_CHECK_ARG((long long)B * T < (1ll << 31));
    // the product kernel's own gather (the instance eab_stft_compress_f32 picks for these sizes), stores instead of passes
    FftPlan plan;
    EAB_CHECK_ARG(fft_plan(n_fft / 2, &plan));
    const size_t sh = (size_t)(2 * n_fft + 2 * FFT_SIGS * n_fft) * sizeof(float);
    if (n_fft == 320 && hop == 160)
        hipLaunchKernelGGL((stft_fft_kernel<320, 160, true>), dim3(B * T), dim3(STFT_THREADS), sh, eab_stream(stream), wav,
                           (const float*)nullptr, (const float*)nullptr, frames, M, L, n_fft, hop, T, 0, plan);
    else
        hipLaunchKernelGGL((stft_fft_kernel<0, 0, true>), dim3(B * T), dim3(STFT_THREADS), sh, eab_stream(stream), wav,
                           (const float*)nullptr, (const float*)nullptr, frames, M, L, n_fft, hop, T, 0, plan);
    EAB_RETURN_LAUNCH_STATUS();
}
