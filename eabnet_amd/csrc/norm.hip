// K8: InstanceNorm(affine) statistics finalisation and the fused
// affine + PReLU (+ residual) apply.  Reference: nn.InstanceNorm{1,2}d
// (EaBNet.py:684,686; eps 1e-5, biased variance, statistics over the whole
// utterance), nn.PReLU(c), En_unet_module's residual add (EaBNet.py:386).
#include "common.h"

// grid = (B * nsets, C/64), block = 1024: 16 tile-slices x 64 channels.
#define FIN_SLICES 16
// Partials are Welford triples (n, mean_i, M2_i) per tile (conv_gemm.hip).  Exact merge:
//   N = sum n_i,  mean = sum n_i mean_i / N,  M2 = sum (M2_i + n_i mean_i^2) - N mean^2.
// The three sums run in fp64 over fp32 inputs that are already centred per tile, so the final
// subtraction loses ~1e-16 * mean^2 / var -- unlike fp32 (sum x, sum x^2) partials -- and no
// division is needed per tile.  Fixed summation order => bit-reproducible.
__global__ __launch_bounds__(1024) void in_finalize_kernel(const float* __restrict__ stats, int C, int nsets,
                                                           int stat_tiles, float eps,
                                                           const float* __restrict__ gamma0,
                                                           const float* __restrict__ beta0, float* __restrict__ xf0,
                                                           const float* __restrict__ gamma1,
                                                           const float* __restrict__ beta1, float* __restrict__ xf1,
                                                           float* __restrict__ mr0, float* __restrict__ mr1) {
    __shared__ double red[3][FIN_SLICES][64];
    const int b = blockIdx.x / nsets, s = blockIdx.x % nsets;
    const float* gamma = s == 0 ? gamma0 : gamma1;
    const float* beta = s == 0 ? beta0 : beta1;
    float* xf = s == 0 ? xf0 : xf1;
    const int cl = threadIdx.x & 63, slice = threadIdx.x >> 6;
    const int c = blockIdx.y * 64 + cl;
    double sn = 0.0, sm = 0.0, sq = 0.0;
    auto add = [&](const f32x4 v) {
        const double n = (double)v[0], mu = (double)v[1];
        sn += n;
        sm = fma(n, mu, sm);
        sq += fma(n * mu, mu, (double)v[2]);
    };
    if (c < C) {
        const size_t stride = (size_t)nsets * C * 4;                         // floats between tiles
        const float* p = stats + (((size_t)b * stat_tiles) * nsets + s) * C * 4 + (size_t)c * 4;
        int t = slice;
        for (; t + 3 * FIN_SLICES < stat_tiles; t += 4 * FIN_SLICES) {       // four loads in flight
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(p + (size_t)t * stride);
            const f32x4 v1 = *reinterpret_cast<const f32x4*>(p + (size_t)(t + FIN_SLICES) * stride);
            const f32x4 v2 = *reinterpret_cast<const f32x4*>(p + (size_t)(t + 2 * FIN_SLICES) * stride);
            const f32x4 v3 = *reinterpret_cast<const f32x4*>(p + (size_t)(t + 3 * FIN_SLICES) * stride);
            add(v0); add(v1); add(v2); add(v3);
        }
        for (; t < stat_tiles; t += FIN_SLICES) add(*reinterpret_cast<const f32x4*>(p + (size_t)t * stride));
    }
    red[0][slice][cl] = sn;
    red[1][slice][cl] = sm;
    red[2][slice][cl] = sq;
    __syncthreads();
    if (slice == 0 && c < C) {
        sn = 0.0; sm = 0.0; sq = 0.0;
#pragma unroll
        for (int k = 0; k < FIN_SLICES; ++k) {
            sn += red[0][k][cl];
            sm += red[1][k][cl];
            sq += red[2][k][cl];
        }
        const double mean = sn > 0.0 ? sm / sn : 0.0;
        double var = sn > 0.0 ? sq / sn - mean * mean : 0.0;                 // biased variance, as nn.InstanceNorm
        if (var < 0.0) var = 0.0;
        const double scale = (double)gamma[c] / sqrt(var + (double)eps);
        const double shift = (double)beta[c] - mean * scale;
        *reinterpret_cast<float2*>(&xf[((size_t)b * C + c) * 2]) = make_float2((float)scale, (float)shift);
        float* mr = s == 0 ? mr0 : mr1;                                      // training: (mean, rstd) for the backward pass
        if (mr) *reinterpret_cast<float2*>(&mr[((size_t)b * C + c) * 2]) = make_float2((float)mean, (float)(1.0 / sqrt(var + (double)eps)));
    }
}

extern "C" int eab_in_finalize_f32(const float* stats, int B, int C, int nsets, int stat_tiles, int count, float eps,
                                   const float* gamma0, const float* beta0, float* xf0, const float* gamma1,
                                   const float* beta1, float* xf1, eab_stream_t stream) {
    return eab_in_finalize_mr_f32(stats, B, C, nsets, stat_tiles, count, eps, gamma0, beta0, xf0, gamma1, beta1, xf1, nullptr,
                                  nullptr, stream);
}

extern "C" int eab_in_finalize_mr_f32(const float* stats, int B, int C, int nsets, int stat_tiles, int count, float eps,
                                      const float* gamma0, const float* beta0, float* xf0, const float* gamma1,
                                      const float* beta1, float* xf1, float* mr0, float* mr1, eab_stream_t stream) {
    EAB_CHECK_ARG(stats && B > 0 && C > 0 && stat_tiles > 0 && count > 0);
    EAB_CHECK_ARG(nsets == 1 || nsets == 2);
    EAB_CHECK_ARG(gamma0 && beta0 && xf0);
    EAB_CHECK_ARG(nsets == 1 || (gamma1 && beta1 && xf1));
    hipLaunchKernelGGL(in_finalize_kernel, dim3(B * nsets, (C + 63) / 64), dim3(1024), 0, eab_stream(stream), stats, C,
                       nsets, stat_tiles, eps, gamma0, beta0, xf0, gamma1, beta1, xf1, mr0, mr1);
    EAB_RETURN_LAUNCH_STATUS();
}

// out = prelu(a*sa + ha) [+ prelu(b*sb + hb)], float4 per thread.  blockIdx.y = batch element, grid-stride over
// its float4s in 32-bit arithmetic (the flat 64-bit index needed two emulated 64-bit divisions per float4).
// NA_UNROLL float4s per operand and thread are in flight before the first is used (one per iteration ran at one memory latency
// per grid sweep, 2.6 TB/s); HOIST: C/4 divides the block size, so a thread's four channels never change over its walk and
// their (scale, shift, slope) are loaded once.
#define NA_UNROLL 4
template <bool HOIST, bool TWO>
__global__ __launch_bounds__(256) void norm_act_kernel(const float* __restrict__ a, const float* __restrict__ xfa,
                                                       const float* __restrict__ sla, const float* __restrict__ bb,
                                                       const float* __restrict__ xfb, const float* __restrict__ slb,
                                                       float* __restrict__ out, int P, int C,
                                                       const int* __restrict__ t_pos, int rows_per_t, int Pw) {
    // rows computed per batch element: all P, or the Pw rows of the streaming window starting at row
    // *t_pos * rows_per_t (clipped at P: the last chunk may be shorter than the window)
    const unsigned C4 = (unsigned)C >> 2;
    const unsigned p_lo = t_pos ? (unsigned)*t_pos * (unsigned)rows_per_t : 0u;
    const unsigned rows = t_pos ? ((unsigned)Pw < (unsigned)P - p_lo ? (unsigned)Pw : (unsigned)P - p_lo) : (unsigned)P;
    const unsigned n4 = p_lo < (unsigned)P ? rows * C4 : 0u;
    const unsigned b = blockIdx.y;
    const size_t base = ((size_t)b * P + p_lo) * C4;
    const unsigned stride = gridDim.x * blockDim.x, start = blockIdx.x * blockDim.x + threadIdx.x;
    f32x4 s01, s23, sl, t01, t23, tl;
    auto tables = [&](unsigned r) {
        const int c = (int)(r % C4) * 4;
        const float* xp = xfa + ((size_t)b * C + c) * 2;
        s01 = *reinterpret_cast<const f32x4*>(xp);
        s23 = *reinterpret_cast<const f32x4*>(xp + 4);
        sl = *reinterpret_cast<const f32x4*>(sla + c);
        if (TWO) {
            const float* yp = xfb + ((size_t)b * C + c) * 2;
            t01 = *reinterpret_cast<const f32x4*>(yp);
            t23 = *reinterpret_cast<const f32x4*>(yp + 4);
            tl = *reinterpret_cast<const f32x4*>(slb + c);
        }
    };
    if (HOIST && start < n4) tables(start);
    for (unsigned r0 = start; r0 < n4; r0 += NA_UNROLL * stride) {
        f32x4 va[NA_UNROLL], vb[NA_UNROLL];
#pragma unroll
        for (int u = 0; u < NA_UNROLL; ++u) {
            const unsigned r = r0 + u * stride;
            const size_t i = base + (r < n4 ? r : r0);
            va[u] = reinterpret_cast<const f32x4*>(a)[i];
            if (TWO) vb[u] = reinterpret_cast<const f32x4*>(bb)[i];
        }
#pragma unroll
        for (int u = 0; u < NA_UNROLL; ++u) {
            const unsigned r = r0 + u * stride;
            if (r >= n4) continue;
            if (!HOIST) tables(r);
            f32x4 r4;
            r4[0] = eab_prelu(fmaf(va[u][0], s01[0], s01[1]), sl[0]);
            r4[1] = eab_prelu(fmaf(va[u][1], s01[2], s01[3]), sl[1]);
            r4[2] = eab_prelu(fmaf(va[u][2], s23[0], s23[1]), sl[2]);
            r4[3] = eab_prelu(fmaf(va[u][3], s23[2], s23[3]), sl[3]);
            if (TWO) {
                r4[0] += eab_prelu(fmaf(vb[u][0], t01[0], t01[1]), tl[0]);
                r4[1] += eab_prelu(fmaf(vb[u][1], t01[2], t01[3]), tl[1]);
                r4[2] += eab_prelu(fmaf(vb[u][2], t23[0], t23[1]), tl[2]);
                r4[3] += eab_prelu(fmaf(vb[u][3], t23[2], t23[3]), tl[3]);
            }
            reinterpret_cast<f32x4*>(out)[base + r] = r4;
        }
    }
}

extern "C" int eab_norm_act_win_f32(const float* a, const float* xfa, const float* slopea, const float* b,
                                    const float* xfb, const float* slopeb, float* out, int B, int T, int rows_per_t,
                                    int C, eab_time_window win, eab_stream_t stream) {
    EAB_CHECK_ARG(a && xfa && slopea && out && B > 0 && T > 0 && rows_per_t > 0 && C > 0 && (C % 4) == 0);
    EAB_CHECK_ARG(b == nullptr || (xfb && slopeb));
    EAB_CHECK_ARG(win.pos == nullptr || win.count > 0);
    const long long P = (long long)T * rows_per_t, Pw = (long long)(win.pos ? win.count : T) * rows_per_t;
    EAB_CHECK_ARG(P * (C / 4) < (1ll << 31) && B <= 65535);
    long long gx = (Pw * (C / 4) + 256 * NA_UNROLL - 1) / (256 * NA_UNROLL);
    const long long cap = (256 * 4 + B - 1) / B;       // <= 4 blocks per CU over the whole grid, grid-stride the rest
    if (gx > cap) gx = cap;
    if (gx < 1) gx = 1;
    const dim3 grid((unsigned)gx, (unsigned)B), block(256);
    hipStream_t s = eab_stream(stream);
    const bool hoist = 256 % (C / 4) == 0;
#define NA_LAUNCH(H_, T_) hipLaunchKernelGGL((norm_act_kernel<H_, T_>), grid, block, 0, s, a, xfa, slopea, b, xfb, slopeb, out, (int)P, C, \
                                              win.pos, rows_per_t, (int)Pw)
    if (hoist) {
        if (b) NA_LAUNCH(true, true); else NA_LAUNCH(true, false);
    } else {
        if (b) NA_LAUNCH(false, true); else NA_LAUNCH(false, false);
    }
#undef NA_LAUNCH
    EAB_RETURN_LAUNCH_STATUS();
}

extern "C" int eab_norm_act_f32(const float* a, const float* xfa, const float* slopea, const float* b,
                                const float* xfb, const float* slopeb, float* out, int B, int P, int C,
                                eab_stream_t stream) {
    return eab_norm_act_win_f32(a, xfa, slopea, b, xfb, slopeb, out, B, P, 1, C, eab_time_window{nullptr, 0}, stream);
}
