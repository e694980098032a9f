// Training-side kernels of the hot path: the HBM-bound forward pieces the fused inference
// program never materialises, and every elementwise / reduction step of the backward pass.  The contractions of
// the backward pass are eab_conv_f32 (dgrad = the gather form of the transposed / strided counterpart of each
// forward convolution) and eab_wgrad_f32 (csrc/wgrad.hip).
//
// Reference call-sites (the autograd graph PyTorch builds for train_distributed.py:221-228):
//   NormSwitch IN + PReLU        EaBNet.py:684-686, 187-188, 403-404, 426-427, 545-569
//   GLU                          EaBNet.py:459-460, 489-490
//   S-TCM gate                   EaBNet.py:575-576
//   LayerNorm                    EaBNet.py:598, 608
//   w_dnn ReLU                   EaBNet.py:595
//   filter-and-sum               EaBNet.py:114-117
// All tensors channels-last [B][P][C] fp32 (P = T*F positions); roofline "hbm" for every kernel in this file.
#include "common.h"

// two fp32 -> packed bf16, round to nearest even (v_cvt_pk_bf16_f32): the rounding the bf16 contractions apply to their operands
typedef __bf16 tr_bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned tr_bf2(float x0, float x1) {
    const tr_bf16x2 v = {(__bf16)x0, (__bf16)x1};
    return __builtin_bit_cast(unsigned, v);
}

#define TR_THREADS 256

// ---------------------------------------------------------------------------------------------------
// parameter packing: out[i] = flat[ia[i]] (+ flat[ib[i]]); index < 0 = 0.0.  One launch turns the flat parameter
// vector into every packed operand of the program (forward and dgrad layouts); the same kernel, driven by the
// inverse table, turns the packed gradient arena back into the flat gradient.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(TR_THREADS) void gather_kernel(const float* __restrict__ flat, const int32_t* __restrict__ ia,
                                                            const int32_t* __restrict__ ib, float* __restrict__ out, long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int a = ia[i];
        float v = a >= 0 ? flat[a] : 0.0f;
        if (ib) {
            const int b = ib[i];
            if (b >= 0) v += flat[b];
        }
        out[i] = v;
    }
}

extern "C" int eab_gather_f32(const float* flat, const int32_t* ia, const int32_t* ib, float* out, long long n,
                              eab_stream_t stream) {
    EAB_CHECK_ARG(flat && ia && out && n >= 0);
    if (n == 0) return EAB_OK;
    long long g = (n + TR_THREADS - 1) / TR_THREADS;
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(gather_kernel, dim3((unsigned)g), dim3(TR_THREADS), 0, eab_stream(stream), flat, ia, ib, out, n);
    EAB_RETURN_LAUNCH_STATUS();
}

// ---------------------------------------------------------------------------------------------------
// InstanceNorm statistics of a materialised [B][P][C] tensor (1-D units: P = T), optional PReLU in front
// (S-TCM order, EaBNet.py:545-547): per (b, c) mean / biased variance over P in fp64, then
//   xf[b][c] = (gamma*rstd, beta - mean*gamma*rstd),  mr[b][c] = (mean, rstd).
// grid (B, C/64), block 256 = 64 channels x 4 position lanes.
// ---------------------------------------------------------------------------------------------------
template <bool APPLY, int NPL>
__global__ __launch_bounds__(16 * NPL) void in_stats_kernel(const float* __restrict__ x, const float* __restrict__ slope,
                                                              int P, int C, float eps, const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, float* __restrict__ xf,
                                                              float* __restrict__ mr, float* __restrict__ y, int xC) {
    // thread -> 4 channels (float4) x every 16th position; shifted sums (shift = the channel's first value, so a
    // nearly constant channel loses nothing to cancellation) in fp64, combined through LDS in a fixed order
    __shared__ double red[2][NPL][64];                  // NPL position lanes x 16 float4 channel groups (NPL = 64: 1024 threads,
                                                        // for slabs a lone workgroup walks: its own loads are all the parallelism)
    __shared__ float scsh[2][64];
    const int b = blockIdx.x, cl = threadIdx.x & 15, pl = threadIdx.x >> 4;
    const int c = blockIdx.y * 64 + cl * 4;
    double s[4] = {0.0, 0.0, 0.0, 0.0}, q[4] = {0.0, 0.0, 0.0, 0.0};
    f32x4 k = {0.f, 0.f, 0.f, 0.f}, a = {1.f, 1.f, 1.f, 1.f};
    if (c < C) {
        if (slope) a = *reinterpret_cast<const f32x4*>(slope + c);
        // xC < C: the output channels are several views of the same xC input channels (the two branch norms of an S-TCM
        // see the same tensor through their own PReLU / gain / bias): channel c reads input channel c % xC
        const float* p = x + (size_t)b * P * xC + (c % xC);
        k = *reinterpret_cast<const f32x4*>(p);
#pragma unroll
        for (int j = 0; j < 4; ++j) k[j] = eab_prelu(k[j], a[j]);
        // four rows in flight per thread: a workgroup owns its slab alone, so its own loads are all the memory-level
        // parallelism there is (one row per iteration ran at one memory latency per 16 rows)
        for (int i0 = pl; i0 < P; i0 += 4 * NPL) {
            f32x4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = i0 + NPL * u;
                v[u] = i < P ? *reinterpret_cast<const f32x4*>(p + (size_t)i * xC) : k;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (i0 + NPL * u < P) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const double d = (double)(eab_prelu(v[u][j], a[j]) - k[j]);
                        s[j] += d;
                        q[j] += d * d;
                    }
                }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        red[0][pl][cl * 4 + j] = s[j];
        red[1][pl][cl * 4 + j] = q[j];
    }
    __syncthreads();
    if (threadIdx.x < 64) {
        const int cc = blockIdx.y * 64 + threadIdx.x;
        if (cc < C) {
            double ss = 0.0, qq = 0.0;
            // (eight at a time: the fully unrolled 64-lane form of the 1024-thread instance kept 128 doubles live and spilled)
#pragma unroll 8
            for (int r = 0; r < NPL; ++r) {
                ss += red[0][r][threadIdx.x];
                qq += red[1][r][threadIdx.x];
            }
            const float kk = eab_prelu(x[(size_t)b * P * xC + (cc % xC)], slope ? slope[cc] : 1.0f);
            const double mean_s = ss / P;
            double var = qq / P - mean_s * mean_s;
            if (var < 0.0) var = 0.0;
            const double mean = mean_s + (double)kk;
            const double rstd = 1.0 / sqrt(var + (double)eps);
            const double scale = (double)gamma[cc] * rstd;
            *reinterpret_cast<float2*>(&xf[((size_t)b * C + cc) * 2]) = make_float2((float)scale, (float)((double)beta[cc] - mean * scale));
            *reinterpret_cast<float2*>(&mr[((size_t)b * C + cc) * 2]) = make_float2((float)mean, (float)rstd);
            if (APPLY) {
                scsh[0][threadIdx.x] = (float)scale;
                scsh[1][threadIdx.x] = (float)((double)beta[cc] - mean * scale);
            }
        }
    }
    if (APPLY) {
        // the whole 1-D unit in this launch: y = prelu(x) * scale + shift for the slab this workgroup just reduced (second
        // pass from L2); the same expression and the same rounded (scale, shift) as tr_norm_act_kernel, EAB_XF_PRELU_NORM
        __syncthreads();
        if (c < C) {
            float sc4[4], sh4[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                sc4[j] = scsh[0][cl * 4 + j];
                sh4[j] = scsh[1][cl * 4 + j];
            }
            const float* p = x + (size_t)b * P * xC + (c % xC);
            // several views (xC < C): every view is its own contiguous [B][P][xC] tensor, one behind the other
            const int yC = xC < C ? xC : C;
            float* yp = y + (size_t)(c / yC) * gridDim.x * P * yC + (size_t)b * P * yC + (c % yC);
            for (int i0 = pl; i0 < P; i0 += 4 * NPL) {
                f32x4 v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int i = i0 + NPL * u;
                    v[u] = i < P ? *reinterpret_cast<const f32x4*>(p + (size_t)i * xC) : f32x4{0.f, 0.f, 0.f, 0.f};
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int i = i0 + NPL * u;
                    if (i < P) {
                        f32x4 o;
#pragma unroll
                        for (int j = 0; j < 4; ++j) o[j] = fmaf(eab_prelu(v[u][j], a[j]), sc4[j], sh4[j]);
                        *reinterpret_cast<f32x4*>(yp + (size_t)i * yC) = o;
                    }
                }
            }
        }
    }
}

extern "C" int eab_train_in_stats_f32(const float* x, const float* slope, int B, int P, int C, float eps, const float* gamma,
                                      const float* beta, float* xf, float* mr, eab_stream_t stream) {
    EAB_CHECK_ARG(x && gamma && beta && xf && mr && B > 0 && P > 0 && C > 0 && (C % 4) == 0 && B <= 65535);
    hipLaunchKernelGGL((in_stats_kernel<false, 16>), dim3(B, (C + 63) / 64), dim3(TR_THREADS), 0, eab_stream(stream), x, slope, P, C, eps,
                       gamma, beta, xf, mr, nullptr, C);
    EAB_RETURN_LAUNCH_STATUS();
}

// The whole 1-D unit y = InstanceNorm1d(prelu(x)) of an S-TCM (EaBNet.py:545-547) in one launch: statistics, (xf, mr) for the
// backward pass, and the normalised output.  One workgroup per (b, 64 channels) walks its P positions twice.
extern "C" int eab_train_in1d_f32(const float* x, const float* slope, int B, int P, int C, float eps, const float* gamma,
                                  const float* beta, float* xf, float* mr, float* y, eab_stream_t stream) {
    EAB_CHECK_ARG(x && slope && gamma && beta && xf && mr && y && B > 0 && P > 0 && C > 0 && (C % 4) == 0 && B <= 65535);
    // few workgroups (B * C/64) each walking its whole slab twice: 1024 threads per workgroup when the grid cannot fill the chip
    if ((long long)B * ((C + 63) / 64) < 256 && P >= 256)
        hipLaunchKernelGGL((in_stats_kernel<true, 64>), dim3(B, (C + 63) / 64), dim3(1024), 0, eab_stream(stream), x, slope, P, C, eps,
                           gamma, beta, xf, mr, y, C);
    else
        hipLaunchKernelGGL((in_stats_kernel<true, 16>), dim3(B, (C + 63) / 64), dim3(TR_THREADS), 0, eab_stream(stream), x, slope, P, C,
                           eps, gamma, beta, xf, mr, y, C);
    EAB_RETURN_LAUNCH_STATUS();
}

// Several 1-D units on ONE input: view v = c / xC of the C = k * xC parameter channels normalises prelu(x[b][p][c % xC], slope[c])
// into its own contiguous tensor y + v * B*P*xC (the left and right branch norms of a SqueezedTCM, EaBNet.py:545-547 and
// 559-560, in one launch).
extern "C" int eab_train_in1d_multi_f32(const float* x, const float* slope, int B, int P, int C, int xC, float eps,
                                        const float* gamma, const float* beta, float* xf, float* mr, float* y, eab_stream_t stream) {
    EAB_CHECK_ARG(x && slope && gamma && beta && xf && mr && y && B > 0 && P > 0 && C > 0 && xC > 0 && (xC % 4) == 0 && (C % xC) == 0 &&
                  B <= 65535);
    if ((long long)B * ((C + 63) / 64) < 256 && P >= 256)
        hipLaunchKernelGGL((in_stats_kernel<true, 64>), dim3(B, (C + 63) / 64), dim3(1024), 0, eab_stream(stream), x, slope, P, C, eps,
                           gamma, beta, xf, mr, y, xC);
    else
        hipLaunchKernelGGL((in_stats_kernel<true, 16>), dim3(B, (C + 63) / 64), dim3(TR_THREADS), 0, eab_stream(stream), x, slope, P, C,
                           eps, gamma, beta, xf, mr, y, xC);
    EAB_RETURN_LAUNCH_STATUS();
}

// ---------------------------------------------------------------------------------------------------
// y = f(x) [+ add]   with f = prelu(x*scale + shift)  (EAB_XF_NORM_PRELU, 2-D units)
//                        or prelu(x)*scale + shift    (EAB_XF_PRELU_NORM, S-TCM)
// ---------------------------------------------------------------------------------------------------
// HOIST: C/4 divides the block size, so a thread's channel group never changes over its grid-stride walk and the
// (scale, shift, slope) of its four channels are loaded once.  TR_UNROLL float4s per thread are in flight before the first is used.
#define TR_UNROLL 4
template <int MODE, bool HOIST>
__global__ __launch_bounds__(TR_THREADS) void tr_norm_act_kernel(const float* __restrict__ x, const float* __restrict__ xf,
                                                                 const float* __restrict__ slope, const float* __restrict__ add,
                                                                 float* __restrict__ y, int P, int C, int store_bf16) {
    const unsigned C4 = (unsigned)C >> 2, n4 = (unsigned)P * C4, b = blockIdx.y;
    const size_t base = (size_t)b * n4;
    const unsigned stride = gridDim.x * blockDim.x, start = blockIdx.x * blockDim.x + threadIdx.x;
    // xf == NULL: no norm in front of the PReLU (the plain U-Net's middle encoder layers, EaBNet.py:219-226)
    const f32x4 one0 = {1.f, 0.f, 1.f, 0.f};
    f32x4 s01 = one0, s23 = one0, sl = {0.f, 0.f, 0.f, 0.f};
    auto tables = [&](unsigned r) {
        const int c = (int)(r % C4) * 4;
        if (xf) {
            const float* xp = xf + ((size_t)b * C + c) * 2;
            s01 = *reinterpret_cast<const f32x4*>(xp);
            s23 = *reinterpret_cast<const f32x4*>(xp + 4);
        }
        sl = *reinterpret_cast<const f32x4*>(slope + c);
    };
    if (HOIST) tables(start);
    for (unsigned r0 = start; r0 < n4; r0 += TR_UNROLL * stride) {
        f32x4 v[TR_UNROLL], ad[TR_UNROLL];
#pragma unroll
        for (int u = 0; u < TR_UNROLL; ++u) {
            const unsigned r = r0 + u * stride, rr = r < n4 ? r : r0;
            v[u] = reinterpret_cast<const f32x4*>(x)[base + rr];
            if (add) ad[u] = reinterpret_cast<const f32x4*>(add)[base + rr];
        }
#pragma unroll
        for (int u = 0; u < TR_UNROLL; ++u) {
            const unsigned r = r0 + u * stride;
            if (r >= n4) continue;
            if (!HOIST) tables(r);
            const float sc[4] = {s01[0], s01[2], s23[0], s23[2]}, sh[4] = {s01[1], s01[3], s23[1], s23[3]};
            f32x4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                o[j] = MODE == EAB_XF_NORM_PRELU ? eab_prelu(fmaf(v[u][j], sc[j], sh[j]), sl[j]) : fmaf(eab_prelu(v[u][j], sl[j]), sc[j], sh[j]);
            if (add) o += ad[u];
            if (store_bf16)                       // the tensor is read by bf16 contractions only: store what they would round to
                reinterpret_cast<uint2*>(y)[base + r] = make_uint2(tr_bf2(o[0], o[1]), tr_bf2(o[2], o[3]));
            else
                reinterpret_cast<f32x4*>(y)[base + r] = o;
        }
    }
}

static inline unsigned tr_grid_x(long long n4, int B) {
    long long gx = (n4 + (long long)TR_THREADS * TR_UNROLL - 1) / ((long long)TR_THREADS * TR_UNROLL);
    const long long cap = (256 * 4 + B - 1) / B;
    if (gx > cap) gx = cap;
    return (unsigned)(gx < 1 ? 1 : gx);
}

extern "C" int eab_train_norm_act_f32(const float* x, const float* xf, const float* slope, const float* add, float* y, int B,
                                      int P, int C, int mode, eab_stream_t stream) {
    EAB_CHECK_ARG(x && slope && y && B > 0 && P > 0 && C > 0 && (C % 4) == 0 && B <= 65535);
    const int bf = (mode & EAB_STORE_BF16) ? 1 : 0;                  // y stored as bf16 (2-byte elements)
    mode &= ~EAB_STORE_BF16;
    EAB_CHECK_ARG(mode == EAB_XF_NORM_PRELU || mode == EAB_XF_PRELU_NORM);
    EAB_CHECK_ARG(xf || mode == EAB_XF_NORM_PRELU);                  // xf == NULL: y = prelu(x) [+ add]
    EAB_CHECK_ARG((long long)P * (C / 4) < (1ll << 31));
    const dim3 grid(tr_grid_x((long long)P * (C / 4), B), B), block(TR_THREADS);
    hipStream_t s = eab_stream(stream);
    const bool hoist = TR_THREADS % (C / 4) == 0;
#define TR_NA(MODE_, H_) hipLaunchKernelGGL((tr_norm_act_kernel<MODE_, H_>), grid, block, 0, s, x, xf, slope, add, y, P, C, bf)
    if (mode == EAB_XF_NORM_PRELU) {
        if (hoist) TR_NA(EAB_XF_NORM_PRELU, true); else TR_NA(EAB_XF_NORM_PRELU, false);
    } else {
        if (hoist) TR_NA(EAB_XF_PRELU_NORM, true); else TR_NA(EAB_XF_PRELU_NORM, false);
    }
#undef TR_NA
    EAB_RETURN_LAUNCH_STATUS();
}

// ---------------------------------------------------------------------------------------------------
// Backward of y = f(x) with InstanceNorm statistics taken over P (per b, c).  xh = normalised value.
//   NORM_PRELU: u = g*xh + be, y = prelu(u):  du = dy*(u > 0 ? 1 : a);  dslope += dy*min(u,0)... (u <= 0 ? u : 0)
//               A = sum du, Q = sum du*xh;   dx = rstd*g*(du - A/P - xh*Q/P);   dgamma += Q, dbeta += A
//   PRELU_NORM: p = prelu(x), xh = (p-mean)*rstd, y = g*xh + be:
//               A = sum dy, Q = sum dy*xh;   dp = rstd*g*(dy - A/P - xh*Q/P);   dx = dp*(x > 0 ? 1 : a);
//               dslope += dp*(x <= 0 ? x : 0)  (second pass);  dgamma += Q, dbeta += A
// pass 1 (reduce): sums[b][c][4] += (A, Q, S, 0), S = the NORM_PRELU slope sum; fp32 atomics per block.
// pass 2 (apply):  dx (= or +=), PRELU_NORM slope sum into sums[..][2].
// pass 3 (params): dgamma[c] += sum_b Q, dbeta[c] += sum_b A, dslope[c] += sum_b S.
// ---------------------------------------------------------------------------------------------------
#define NB_ROWS 16      // position lanes per block: block = 16 float4 channel groups (64 channels) x 16 positions
#define NB_UNROLL 4     // rows a thread has in flight
#define NB_SHARDS EAB_NB_SUM_COPIES

__global__ __launch_bounds__(TR_THREADS) void tr_zero_kernel(float* __restrict__ p, long long n4) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n4) reinterpret_cast<f32x4*>(p)[i] = f32x4{0.f, 0.f, 0.f, 0.f};
}

// thread -> 4 channels (float4) x every 16th position of the block's chunk: a wave reads 4 positions x 256 B = 1 KB.
// NB_UNROLL rows per thread are in flight before the first is used (32 KB of loads per workgroup: a streaming kernel with one
// load per thread and iteration ran at one memory latency per 16 rows, 2.6 TB/s), and the block's atomics go to copy
// blockIdx.x % NB_SHARDS of the sums (330 workgroups adding into the 24 words of one cache line cost more than the reads).
template <int MODE>
__global__ __launch_bounds__(TR_THREADS) void norm_bwd_reduce_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                                     const float* __restrict__ mr, const float* __restrict__ gamma,
                                                                     const float* __restrict__ beta, const float* __restrict__ slope,
                                                                     float* __restrict__ sums, int P, int C, int chunk,
                                                                     int xC, const float* __restrict__ dy1) {
    __shared__ f32x4 red[3][NB_ROWS][16];
    const int b = blockIdx.z, cl = threadIdx.x & 15, pl = threadIdx.x >> 4;
    const int c = blockIdx.y * 64 + cl * 4;
    f32x4 A = {0.f, 0.f, 0.f, 0.f}, Q = A, S = A;
    if (c < C) {
        // mr == NULL: no norm (u = x): only the slope sum S is needed
        const f32x4 zero_one = {0.f, 1.f, 0.f, 1.f}, ones = {1.f, 1.f, 1.f, 1.f}, zeros = {0.f, 0.f, 0.f, 0.f};
        const float* mp = &mr[((size_t)b * C + c) * 2];
        const f32x4 m01 = mr ? *reinterpret_cast<const f32x4*>(mp) : zero_one, m23 = mr ? *reinterpret_cast<const f32x4*>(mp + 4) : zero_one;
        const float mean[4] = {m01[0], m01[2], m23[0], m23[2]}, rstd[4] = {m01[1], m01[3], m23[1], m23[3]};
        const f32x4 g = mr ? *reinterpret_cast<const f32x4*>(gamma + c) : ones, be = mr ? *reinterpret_cast<const f32x4*>(beta + c) : zeros,
                    a = *reinterpret_cast<const f32x4*>(slope + c);
        const int p0 = blockIdx.x * chunk, p1 = p0 + chunk < P ? p0 + chunk : P;
        // xC < C: views (see in_stats_kernel): view v's output gradient is its own [B][P][xC] tensor (dy, dy1)
        const size_t xbase = (size_t)b * P * xC + (c % xC);
        const float* dyp = xC < C ? (c < xC ? dy : dy1) + xbase : dy + (size_t)b * P * C + c;
        const int dC = xC < C ? xC : C;
        for (int i0 = p0 + pl; i0 < p1; i0 += NB_ROWS * NB_UNROLL) {
            f32x4 xv[NB_UNROLL], dv[NB_UNROLL];
#pragma unroll
            for (int u = 0; u < NB_UNROLL; ++u) {               // rows past the chunk re-read row i0 with a zero gradient: they add 0
                const int i = i0 + NB_ROWS * u, ii = i < p1 ? i : i0;
                xv[u] = *reinterpret_cast<const f32x4*>(&x[xbase + (size_t)ii * xC]);
                dv[u] = *reinterpret_cast<const f32x4*>(&dyp[(size_t)ii * dC]);
            }
#pragma unroll
            for (int u = 0; u < NB_UNROLL; ++u) {
                if (i0 + NB_ROWS * u >= p1) continue;
                const f32x4 d = dv[u];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (MODE == EAB_XF_NORM_PRELU) {
                        const float xh = (xv[u][j] - mean[j]) * rstd[j], uu = fmaf(g[j], xh, be[j]);
                        const float du = uu > 0.f ? d[j] : a[j] * d[j];
                        A[j] += du;
                        Q[j] = fmaf(du, xh, Q[j]);
                        S[j] += uu > 0.f ? 0.f : d[j] * uu;
                    } else {
                        const float xh = (eab_prelu(xv[u][j], a[j]) - mean[j]) * rstd[j];
                        A[j] += d[j];
                        Q[j] = fmaf(d[j], xh, Q[j]);
                    }
                }
            }
        }
    }
    red[0][pl][cl] = A;
    red[1][pl][cl] = Q;
    red[2][pl][cl] = S;
    __syncthreads();
    if (pl < 3 && c < C) {       // pl = which of (A, Q, S); fixed-order sum over the 16 position lanes, then one atomic per value
        if (pl == 2 && MODE != EAB_XF_NORM_PRELU) return;
        if (pl < 2 && !mr) return;
        f32x4 t = red[pl][0][cl];
#pragma unroll
        for (int k = 1; k < NB_ROWS; ++k) t += red[pl][k][cl];
        float* sh = sums + (size_t)(blockIdx.x % NB_SHARDS) * gridDim.z * C * 4;
#pragma unroll
        for (int j = 0; j < 4; ++j) atomicAdd(&sh[((size_t)b * C + c + j) * 4 + pl], t[j]);
    }
}

// (A, Q, S, -) of (b, c): the sum of the NB_SHARDS copies, in copy order
__device__ __forceinline__ f32x4 nb_sums(const float* __restrict__ sums, int B, int C, int b, int c) {
    f32x4 t = *reinterpret_cast<const f32x4*>(&sums[((size_t)b * C + c) * 4]);
#pragma unroll
    for (int s = 1; s < NB_SHARDS; ++s) t += *reinterpret_cast<const f32x4*>(&sums[(((size_t)s * B + b) * C + c) * 4]);
    return t;
}

template <int MODE>
__global__ __launch_bounds__(TR_THREADS) void norm_bwd_apply_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                                    const float* __restrict__ mr, const float* __restrict__ gamma,
                                                                    const float* __restrict__ beta, const float* __restrict__ slope,
                                                                    float* __restrict__ sums, const float* __restrict__ acc_in,
                                                                    float* __restrict__ dx, float* __restrict__ dgamma,
                                                                    float* __restrict__ dbeta, float* __restrict__ dslope, int P, int C,
                                                                    int chunk, int dx_bf16) {
    __shared__ f32x4 red[NB_ROWS][16];
    const int b = blockIdx.z, B = gridDim.z, cl = threadIdx.x & 15, pl = threadIdx.x >> 4;
    const int c = blockIdx.y * 64 + cl * 4;
    f32x4 S = {0.f, 0.f, 0.f, 0.f};
    if (c < C) {
        // mr == NULL: no norm -- dx = dy * prelu'(x), no projection
        const f32x4 zero_one = {0.f, 1.f, 0.f, 1.f}, ones = {1.f, 1.f, 1.f, 1.f}, zeros = {0.f, 0.f, 0.f, 0.f};
        const float* mp = &mr[((size_t)b * C + c) * 2];
        const f32x4 m01 = mr ? *reinterpret_cast<const f32x4*>(mp) : zero_one, m23 = mr ? *reinterpret_cast<const f32x4*>(mp + 4) : zero_one;
        const float mean[4] = {m01[0], m01[2], m23[0], m23[2]}, rstd[4] = {m01[1], m01[3], m23[1], m23[3]};
        const f32x4 g = mr ? *reinterpret_cast<const f32x4*>(gamma + c) : ones, be = mr ? *reinterpret_cast<const f32x4*>(beta + c) : zeros,
                    a = *reinterpret_cast<const f32x4*>(slope + c);
        const float inv_p = mr ? 1.0f / (float)P : 0.0f;
        float Am[4], Qm[4], k[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const f32x4 v = nb_sums(sums, B, C, b, c + j);
            Am[j] = v[0] * inv_p;
            Qm[j] = v[1] * inv_p;
            k[j] = rstd[j] * g[j];
            // the sums A, Q (and S for NORM_PRELU) of (b, c) are final after the reduce pass -> the parameter gradients ride
            // along here (one workgroup per (b, channel group); B atomics per address)
            if ((dgamma || !mr) && blockIdx.x == 0 && pl == 0) {
                if (mr) {
                    atomicAdd(&dbeta[c + j], v[0]);
                    atomicAdd(&dgamma[c + j], v[1]);
                }
                if (MODE == EAB_XF_NORM_PRELU) atomicAdd(&dslope[c + j], v[2]);
            }
        }
        const int p0 = blockIdx.x * chunk, p1 = p0 + chunk < P ? p0 + chunk : P;
        const size_t base = (size_t)b * P * C + c;
        for (int i0 = p0 + pl; i0 < p1; i0 += NB_ROWS * NB_UNROLL) {
            f32x4 xq[NB_UNROLL], dq[NB_UNROLL], aq[NB_UNROLL];
#pragma unroll
            for (int u = 0; u < NB_UNROLL; ++u) {               // NB_UNROLL rows in flight (rows past the chunk re-read row i0)
                const int i = i0 + NB_ROWS * u;
                const size_t e = base + (size_t)(i < p1 ? i : i0) * C;
                xq[u] = *reinterpret_cast<const f32x4*>(&x[e]);
                dq[u] = *reinterpret_cast<const f32x4*>(&dy[e]);
                if (acc_in) aq[u] = *reinterpret_cast<const f32x4*>(&acc_in[e]);
            }
#pragma unroll
            for (int u = 0; u < NB_UNROLL; ++u) {
                const int i = i0 + NB_ROWS * u;
                if (i >= p1) continue;
                const size_t e = base + (size_t)i * C;
                const f32x4 xv = xq[u], d = dq[u];
                f32x4 r;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (MODE == EAB_XF_NORM_PRELU) {
                        const float xh = (xv[j] - mean[j]) * rstd[j], uu = fmaf(g[j], xh, be[j]);
                        const float du = uu > 0.f ? d[j] : a[j] * d[j];
                        r[j] = k[j] * (du - Am[j] - xh * Qm[j]);
                    } else {
                        const float xh = (eab_prelu(xv[j], a[j]) - mean[j]) * rstd[j];
                        const float dp = k[j] * (d[j] - Am[j] - xh * Qm[j]);
                        r[j] = xv[j] > 0.f ? dp : a[j] * dp;
                        S[j] += xv[j] > 0.f ? 0.f : dp * xv[j];
                    }
                }
                if (acc_in) r += aq[u];
                if (dx_bf16)                               // read by bf16 contractions only (wgrad, dgrad): half the bytes
                    *reinterpret_cast<uint2*>(reinterpret_cast<char*>(dx) + e * 2) = make_uint2(tr_bf2(r[0], r[1]), tr_bf2(r[2], r[3]));
                else
                    *reinterpret_cast<f32x4*>(&dx[e]) = r;
            }
        }
    }
    if (MODE == EAB_XF_PRELU_NORM) {
        red[pl][cl] = S;
        __syncthreads();
        if (pl == 0 && c < C) {
            f32x4 t = red[0][cl];
#pragma unroll
            for (int kk = 1; kk < NB_ROWS; ++kk) t += red[kk][cl];
            // PRELU_NORM: the slope sum only exists after this pass.  merged (dgamma != NULL): straight into dslope (the host
            // picks that when a slope address sees few workgroups), else into copy 0 of the sums for the params kernel
#pragma unroll
            for (int j = 0; j < 4; ++j) atomicAdd(dgamma ? &dslope[c + j] : &sums[((size_t)b * C + c + j) * 4 + 2], t[j]);
        }
    }
}

// Apply pass of the multi-view form (C = NV * xC: NV norms on views of the same xC-channel tensor, PRELU_NORM order): the
// input gradient is the SUM over the views, written once:  dx[c] = (acc_in) + sum_v dx_v[c].  Parameter gradients always ride
// along (this form is only used for the S-TCN's small slabs).  Thread -> 4 input channels x every 16th position.
__global__ __launch_bounds__(TR_THREADS) void norm_bwd_apply_multi_kernel(const float* __restrict__ dy, const float* __restrict__ dy1,
                                                                          const float* __restrict__ x,
                                                                          const float* __restrict__ mr, const float* __restrict__ gamma,
                                                                          const float* __restrict__ slope, const float* __restrict__ sums,
                                                                          const float* __restrict__ acc_in, float* __restrict__ dx,
                                                                          float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                                          float* __restrict__ dslope, int P, int C, int xC, int chunk) {
    __shared__ f32x4 red[2][NB_ROWS][16];
    const int b = blockIdx.z, cl = threadIdx.x & 15, pl = threadIdx.x >> 4;
    const int c = blockIdx.y * 64 + cl * 4;                        // input channel
    const int NV = C / xC;                                         // 2 (host-checked: <= 2)
    f32x4 S[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    if (c < xC) {
        float mean[2][4], k[2][4], Am[2][4], Qm[2][4];
        f32x4 a[2];
        const float inv_p = 1.0f / (float)P;
        for (int v = 0; v < NV; ++v) {
            const int cv = c + v * xC;
            const float* mp = &mr[((size_t)b * C + cv) * 2];
            const f32x4 m01 = *reinterpret_cast<const f32x4*>(mp), m23 = *reinterpret_cast<const f32x4*>(mp + 4);
            const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + cv);
            a[v] = *reinterpret_cast<const f32x4*>(slope + cv);
            const float mn[4] = {m01[0], m01[2], m23[0], m23[2]}, rs[4] = {m01[1], m01[3], m23[1], m23[3]};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x4 sv = nb_sums(sums, gridDim.z, C, b, cv + j);
                mean[v][j] = mn[j];
                k[v][j] = rs[j] * g[j];
                Am[v][j] = sv[0] * inv_p;
                Qm[v][j] = sv[1] * inv_p * rs[j];                // xh = (p - mean) * rstd folded: xh * Qm = (p - mean) * (Qm * rstd)
                if (blockIdx.x == 0 && pl == 0) {
                    atomicAdd(&dbeta[cv + j], sv[0]);
                    atomicAdd(&dgamma[cv + j], sv[1]);
                }
            }
        }
        const int p0 = blockIdx.x * chunk, p1 = p0 + chunk < P ? p0 + chunk : P;
        const size_t xbase = (size_t)b * P * xC + c;
        for (int i = p0 + pl; i < p1; i += NB_ROWS) {
            const f32x4 xv = *reinterpret_cast<const f32x4*>(&x[xbase + (size_t)i * xC]);
            f32x4 r = {0.f, 0.f, 0.f, 0.f};
            if (acc_in) r = *reinterpret_cast<const f32x4*>(&acc_in[xbase + (size_t)i * xC]);
            for (int v = 0; v < NV; ++v) {
                const f32x4 d = *reinterpret_cast<const f32x4*>(&(v == 0 ? dy : dy1)[xbase + (size_t)i * xC]);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float pc = eab_prelu(xv[j], a[v][j]) - mean[v][j];
                    const float dp = k[v][j] * (d[j] - Am[v][j] - pc * Qm[v][j]);
                    r[j] += xv[j] > 0.f ? dp : a[v][j] * dp;
                    S[v][j] += xv[j] > 0.f ? 0.f : dp * xv[j];
                }
            }
            *reinterpret_cast<f32x4*>(&dx[xbase + (size_t)i * xC]) = r;
        }
    }
    red[0][pl][cl] = S[0];
    red[1][pl][cl] = S[1];
    __syncthreads();
    if (pl < NV && c < xC) {
        f32x4 t = red[pl][0][cl];
#pragma unroll
        for (int kk = 1; kk < NB_ROWS; ++kk) t += red[pl][kk][cl];
#pragma unroll
        for (int j = 0; j < 4; ++j) atomicAdd(&dslope[c + pl * xC + j], t[j]);
    }
}

__global__ __launch_bounds__(64) void norm_bwd_params_kernel(const float* __restrict__ sums, int B, int C, float* __restrict__ dgamma,
                                                             float* __restrict__ dbeta, float* __restrict__ dslope) {
    const int c = blockIdx.x * 64 + threadIdx.x;
    if (c >= C) return;
    float A = 0.f, Q = 0.f, S = 0.f;
    for (int b = 0; b < B; ++b) {
        const f32x4 v = nb_sums(sums, B, C, b, c);
        A += v[0];
        Q += v[1];
        S += v[2];
    }
    dgamma[c] += Q;
    dbeta[c] += A;
    dslope[c] += S;
}

static inline int nb_chunk(int P, int B, int C) {
    // positions per block: ~1024 blocks over the grid (four per CU, each with NB_UNROLL rows per thread in flight), at least 64
    // positions each, a whole number of unrolled iterations
    const long long blocks_bc = (long long)B * ((C + 63) / 64);
    long long want = (1024 + blocks_bc - 1) / blocks_bc;
    if (want < 1) want = 1;
    long long chunk = (P + want - 1) / want;
    if (chunk < 64) chunk = 64;
    const int step = NB_ROWS * NB_UNROLL;
    return (int)((chunk + step - 1) / step * step);
}

extern "C" int eab_train_norm_bwd_f32(const float* dy, const float* x, const float* mr, const float* gamma, const float* beta,
                                      const float* slope, float* sums, const float* acc_in, float* dx, float* dgamma,
                                      float* dbeta, float* dslope, int B, int P, int C, int mode, eab_stream_t stream) {
    // mr == NULL (then gamma, beta, dgamma, dbeta are not touched): no norm in front of the PReLU, NORM_PRELU order only
    EAB_CHECK_ARG(dy && x && slope && sums && dx && dslope);
    EAB_CHECK_ARG(mr ? (gamma && beta && dgamma && dbeta) : true);
    const bool zeroed = (mode & EAB_NB_SUMS_ZEROED) != 0;          // the caller guarantees sums == 0 on entry
    const int dx_bf16 = (mode & EAB_STORE_BF16) ? 1 : 0;           // dx stored as bf16 (needs acc_in == NULL)
    mode &= ~(EAB_NB_SUMS_ZEROED | EAB_STORE_BF16);
    EAB_CHECK_ARG(!dx_bf16 || acc_in == nullptr);
    EAB_CHECK_ARG(mr || mode == EAB_XF_NORM_PRELU);
    if (!mr) dgamma = dbeta = nullptr;
    EAB_CHECK_ARG(B > 0 && P > 0 && C > 0 && (C % 4) == 0 && B <= 65535 && (mode == EAB_XF_NORM_PRELU || mode == EAB_XF_PRELU_NORM));
    hipStream_t s = eab_stream(stream);
    const int chunk = nb_chunk(P, B, C);
    dim3 grid((P + chunk - 1) / chunk, (C + 63) / 64, B);
    // (a kernel, not hipMemsetAsync: memset nodes of a captured graph were not reliably ordered on this stack)
    if (!zeroed)
        hipLaunchKernelGGL(tr_zero_kernel, dim3((NB_SHARDS * B * C + TR_THREADS - 1) / TR_THREADS), dim3(TR_THREADS), 0, s, sums,
                           (long long)NB_SHARDS * B * C);
    if (mode == EAB_XF_NORM_PRELU)
        hipLaunchKernelGGL(norm_bwd_reduce_kernel<EAB_XF_NORM_PRELU>, grid, dim3(TR_THREADS), 0, s, dy, x, mr, gamma, beta, slope, sums, P,
                           C, chunk, C, nullptr);
    else
        hipLaunchKernelGGL(norm_bwd_reduce_kernel<EAB_XF_PRELU_NORM>, grid, dim3(TR_THREADS), 0, s, dy, x, mr, gamma, beta, slope, sums, P,
                           C, chunk, C, nullptr);
    // parameter gradients inside the apply pass: always for NORM_PRELU (all three sums are final after the reduce pass); for
    // PRELU_NORM when its slope sums can go straight into dslope (<= 256 workgroups per address: the S-TCN's slabs)
    const bool merged = mode == EAB_XF_NORM_PRELU || (long long)grid.x * B <= 256;
    if (mode == EAB_XF_NORM_PRELU)
        hipLaunchKernelGGL(norm_bwd_apply_kernel<EAB_XF_NORM_PRELU>, grid, dim3(TR_THREADS), 0, s, dy, x, mr, gamma, beta, slope, sums,
                           acc_in, dx, merged ? dgamma : nullptr, dbeta, dslope, P, C, chunk, dx_bf16);
    else
        hipLaunchKernelGGL(norm_bwd_apply_kernel<EAB_XF_PRELU_NORM>, grid, dim3(TR_THREADS), 0, s, dy, x, mr, gamma, beta, slope, sums,
                           acc_in, dx, merged ? dgamma : nullptr, dbeta, dslope, P, C, chunk, dx_bf16);
    if (!merged) hipLaunchKernelGGL(norm_bwd_params_kernel, dim3((C + 63) / 64), dim3(64), 0, s, sums, B, C, dgamma, dbeta, dslope);
    EAB_RETURN_LAUNCH_STATUS();
}

// Backward of the two one-dimensional units (PRELU_NORM order) that read the same xC-channel tensor x (the branch norms of a
// SqueezedTCM, forward: eab_train_in1d_multi_f32): dy0 / dy1 = the views' output gradients, each [B][P][xC]; mr, gamma, slope,
// sums and the parameter gradients are [..][2 xC] (view-major); dx [B][P][xC] = (acc_in) + both input gradients.
// sums must be zero on entry.  Two launches.
extern "C" int eab_train_norm_bwd_multi_f32(const float* dy0, const float* dy1, const float* x, const float* mr, const float* gamma,
                                            const float* slope, float* sums, const float* acc_in, float* dx, float* dgamma,
                                            float* dbeta, float* dslope, int B, int P, int xC, eab_stream_t stream) {
    EAB_CHECK_ARG(dy0 && dy1 && x && mr && gamma && slope && sums && dx && dgamma && dbeta && dslope);
    EAB_CHECK_ARG(B > 0 && P > 0 && xC > 0 && (xC % 4) == 0 && B <= 65535);
    const int C = 2 * xC;
    const float* const dy = dy0;
    const float* const beta = gamma;            // (not read in the PRELU_NORM order)
    const int chunk = nb_chunk(P, B, C);
    hipStream_t s = eab_stream(stream);
    hipLaunchKernelGGL(norm_bwd_reduce_kernel<EAB_XF_PRELU_NORM>, dim3((P + chunk - 1) / chunk, (C + 63) / 64, B), dim3(TR_THREADS), 0, s,
                       dy, x, mr, gamma, beta, slope, sums, P, C, chunk, xC, dy1);
    hipLaunchKernelGGL(norm_bwd_apply_multi_kernel, dim3((P + chunk - 1) / chunk, (xC + 63) / 64, B), dim3(TR_THREADS), 0, s, dy, dy1, x,
                       mr, gamma, slope, sums, acc_in, dx, dgamma, dbeta, dslope, P, C, xC, chunk);
    EAB_RETURN_LAUNCH_STATUS();
}

// ---------------------------------------------------------------------------------------------------
// GLU backward.  Forward (eab_conv_f32, EAB_EPI_GLU with glu_dump): y[c] = a[c]*s[c], s = sigmoid(gate); the dump
// holds a and s in the PACKED column order of the convolution (column r: half = (r%64)/32 (0 = value, 1 = gate),
// channel c = (r/64)*32 + r%32).  dz (same packed order, N = 2*Cout columns) = (dy*s | dy*a*s*(1-s)).
// ---------------------------------------------------------------------------------------------------
// (LG4 >= 0: N / 4 = 1 << LG4 -- every width the networks use -- so the row of a float4 is a shift; the emulated 64-bit
// division of the generic form costs more VALU work than the rest of the loop body)
template <int GENERIC>
__global__ __launch_bounds__(TR_THREADS) void glu_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ dump,
                                                             float* __restrict__ dz, long long rows, int N, int lg4, int dz_bf16) {
    const int Cout = N >> 1, N4 = N >> 2;
    const long long n4 = rows * N4, stride = (long long)gridDim.x * blockDim.x;
    for (long long i0 = (long long)blockIdx.x * blockDim.x + threadIdx.x; i0 < n4; i0 += TR_UNROLL * stride) {
        f32x4 d[TR_UNROLL], me[TR_UNROLL], ot[TR_UNROLL];
        long long at[TR_UNROLL];
        int half[TR_UNROLL];
#pragma unroll
        for (int u = 0; u < TR_UNROLL; ++u) {                     // TR_UNROLL float4s of each operand in flight
            const long long i = i0 + u * stride < n4 ? i0 + u * stride : i0;
            const long long row = GENERIC ? i / N4 : i >> lg4;
            const int r = (int)(i - row * N4) * 4;                 // first packed column of this float4 (same half, same 32-group)
            const int c = (r >> 6) * 32 + (r & 31);
            half[u] = (r & 63) >> 5;
            const int mate = half[u] ? r - 32 : r + 32;            // packed column of the partner (gate <-> value)
            at[u] = row * N + r;
            d[u] = *reinterpret_cast<const f32x4*>(&dy[row * Cout + c]);
            me[u] = *reinterpret_cast<const f32x4*>(&dump[row * N + r]);
            ot[u] = *reinterpret_cast<const f32x4*>(&dump[row * N + mate]);
        }
#pragma unroll
        for (int u = 0; u < TR_UNROLL; ++u) {
            if (i0 + u * stride >= n4) continue;
            f32x4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = half[u] ? d[u][j] * ot[u][j] * me[u][j] * (1.0f - me[u][j]) : d[u][j] * ot[u][j];
            if (dz_bf16)
                *reinterpret_cast<uint2*>(reinterpret_cast<char*>(dz) + at[u] * 2) = make_uint2(tr_bf2(o[0], o[1]), tr_bf2(o[2], o[3]));
            else
                *reinterpret_cast<f32x4*>(&dz[at[u]]) = o;
        }
    }
}

extern "C" int eab_glu_bwd_ex_f32(const float* dy, const float* dump, float* dz, long long rows, int N, int flags,
                                  eab_stream_t stream) {
    EAB_CHECK_ARG(dy && dump && dz && rows > 0 && N > 0 && (N % 64) == 0 && (flags & ~EAB_STORE_BF16) == 0);
    const int dz_bf16 = (flags & EAB_STORE_BF16) ? 1 : 0;            // dz stored as bf16 (read by bf16 wgrad / dgrad only)
    long long g = (rows * (N / 4) + TR_THREADS * TR_UNROLL - 1) / (TR_THREADS * TR_UNROLL);
    if (g > 2048) g = 2048;
    const int N4 = N / 4;
    if ((N4 & (N4 - 1)) == 0)
        hipLaunchKernelGGL(glu_bwd_kernel<0>, dim3((unsigned)g), dim3(TR_THREADS), 0, eab_stream(stream), dy, dump, dz, rows, N,
                           __builtin_ctz(N4), dz_bf16);
    else
        hipLaunchKernelGGL(glu_bwd_kernel<1>, dim3((unsigned)g), dim3(TR_THREADS), 0, eab_stream(stream), dy, dump, dz, rows, N, 0,
                           dz_bf16);
    EAB_RETURN_LAUNCH_STATUS();
}

extern "C" int eab_glu_bwd_f32(const float* dy, const float* dump, float* dz, long long rows, int N, eab_stream_t stream) {
    return eab_glu_bwd_ex_f32(dy, dump, dz, rows, N, 0, stream);
}

// ---------------------------------------------------------------------------------------------------
// S-TCM gate z = a * sigmoid(r) (EaBNet.py:575) and its backward; flat float4 kernels.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(TR_THREADS) void gate_fwd_kernel(const float* __restrict__ a, const float* __restrict__ r,
                                                              float* __restrict__ z, long long n4) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
        const f32x4 av = reinterpret_cast<const f32x4*>(a)[i], rv = reinterpret_cast<const f32x4*>(r)[i];
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = av[j] * eab_sigmoid(rv[j]);
        reinterpret_cast<f32x4*>(z)[i] = o;
    }
}

__global__ __launch_bounds__(TR_THREADS) void gate_bwd_kernel(const float* __restrict__ dz, const float* __restrict__ a,
                                                              const float* __restrict__ r, float* __restrict__ da,
                                                              float* __restrict__ dr, long long n4) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
        const f32x4 d = reinterpret_cast<const f32x4*>(dz)[i], av = reinterpret_cast<const f32x4*>(a)[i],
                    rv = reinterpret_cast<const f32x4*>(r)[i];
        f32x4 oa, orr;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float s = eab_sigmoid(rv[j]);
            oa[j] = d[j] * s;
            orr[j] = d[j] * av[j] * s * (1.0f - s);
        }
        reinterpret_cast<f32x4*>(da)[i] = oa;
        reinterpret_cast<f32x4*>(dr)[i] = orr;
    }
}

static inline unsigned flat_grid(long long n4) {
    long long g = (n4 + TR_THREADS - 1) / TR_THREADS;
    if (g > 8192) g = 8192;
    return (unsigned)(g < 1 ? 1 : g);
}

extern "C" int eab_gate_fwd_f32(const float* a, const float* r, float* z, long long n, eab_stream_t stream) {
    EAB_CHECK_ARG(a && r && z && n > 0 && (n % 4) == 0);
    hipLaunchKernelGGL(gate_fwd_kernel, dim3(flat_grid(n / 4)), dim3(TR_THREADS), 0, eab_stream(stream), a, r, z, n / 4);
    EAB_RETURN_LAUNCH_STATUS();
}

extern "C" int eab_gate_bwd_f32(const float* dz, const float* a, const float* r, float* da, float* dr, long long n,
                                eab_stream_t stream) {
    EAB_CHECK_ARG(dz && a && r && da && dr && n > 0 && (n % 4) == 0);
    hipLaunchKernelGGL(gate_bwd_kernel, dim3(flat_grid(n / 4)), dim3(TR_THREADS), 0, eab_stream(stream), dz, a, r, da, dr, n / 4);
    EAB_RETURN_LAUNCH_STATUS();
}

// dst = src, 16 bytes per thread.  Used for host -> device uploads from PINNED host memory (device-visible): a copy
// KERNEL instead of hipMemcpyAsync, because DMA-engine copies submitted between compute kernels cost ~5 ms each on
// the measured stack (tools/diag_pd.py) while the same bytes through a kernel take 0.6 ms.
__global__ __launch_bounds__(TR_THREADS) void copy_kernel(const float* __restrict__ src, float* __restrict__ dst, long long n4, long long n) {
    const long long i0 = (long long)blockIdx.x * blockDim.x + threadIdx.x, stride = (long long)gridDim.x * blockDim.x;
    for (long long i = i0; i < n4; i += stride) reinterpret_cast<f32x4*>(dst)[i] = reinterpret_cast<const f32x4*>(src)[i];
    for (long long i = 4 * n4 + i0; i < n; i += stride) dst[i] = src[i];
}

extern "C" int eab_copy_f32(const float* src, float* dst, long long n, eab_stream_t stream) {
    EAB_CHECK_ARG(src && dst && n >= 0 && ((uintptr_t)src & 15) == 0 && ((uintptr_t)dst & 15) == 0);
    if (n == 0) return EAB_OK;
    // 128 workgroups keep ~0.5 MB of reads in flight -- several times what a PCIe link needs at its latency -- and leave the
    // rest of the chip to the batches that compute while this one uploads (2048 workgroups queued behind their kernels)
    long long g = (n / 4 + TR_THREADS - 1) / TR_THREADS;
    if (g > 128) g = 128;
    if (g < 1) g = 1;
    hipLaunchKernelGGL(copy_kernel, dim3((unsigned)g), dim3(TR_THREADS), 0, eab_stream(stream), src, dst, n / 4, n);
    EAB_RETURN_LAUNCH_STATUS();
}

// out = a + b   /   dx = y > 0 ? dy : 0
__global__ __launch_bounds__(TR_THREADS) void add_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                         float* __restrict__ out, long long n4) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x)
        reinterpret_cast<f32x4*>(out)[i] = reinterpret_cast<const f32x4*>(a)[i] + reinterpret_cast<const f32x4*>(b)[i];
}

__global__ __launch_bounds__(TR_THREADS) void relu_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                              float* __restrict__ dx, long long n4) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
        const f32x4 d = reinterpret_cast<const f32x4*>(dy)[i], v = reinterpret_cast<const f32x4*>(y)[i];
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = v[j] > 0.0f ? d[j] : 0.0f;
        reinterpret_cast<f32x4*>(dx)[i] = o;
    }
}

extern "C" int eab_add_f32(const float* a, const float* b, float* out, long long n, eab_stream_t stream) {
    EAB_CHECK_ARG(a && b && out && n > 0 && (n % 4) == 0);
    hipLaunchKernelGGL(add_kernel, dim3(flat_grid(n / 4)), dim3(TR_THREADS), 0, eab_stream(stream), a, b, out, n / 4);
    EAB_RETURN_LAUNCH_STATUS();
}

extern "C" int eab_relu_bwd_f32(const float* dy, const float* y, float* dx, long long n, eab_stream_t stream) {
    EAB_CHECK_ARG(dy && y && dx && n > 0 && (n % 4) == 0);
    hipLaunchKernelGGL(relu_bwd_kernel, dim3(flat_grid(n / 4)), dim3(TR_THREADS), 0, eab_stream(stream), dy, y, dx, n / 4);
    EAB_RETURN_LAUNCH_STATUS();
}

// ---------------------------------------------------------------------------------------------------
// column sums of a [rows][N] matrix (bias gradients): out[n] += sum_r x[r][n]
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(TR_THREADS) void colsum_kernel(const float* __restrict__ x, float* __restrict__ out, long long rows,
                                                            int N, long long chunk) {
    __shared__ float red[4][64];
    const int cl = threadIdx.x & 63, pl = threadIdx.x >> 6;
    const int c = blockIdx.y * 64 + cl;
    float s = 0.f;
    const long long r0 = (long long)blockIdx.x * chunk, r1 = r0 + chunk < rows ? r0 + chunk : rows;
    if (c < N)
        for (long long r = r0 + pl; r < r1; r += 4) s += x[r * N + c];
    red[pl][cl] = s;
    __syncthreads();
    if (pl == 0 && c < N) atomicAdd(&out[c], red[0][cl] + red[1][cl] + red[2][cl] + red[3][cl]);
}

extern "C" int eab_colsum_f32(const float* x, float* out, long long rows, int N, eab_stream_t stream) {
    EAB_CHECK_ARG(x && out && rows > 0 && N > 0);
    const int cb = (N + 63) / 64;
    long long blocks = 1024 / cb;
    if (blocks < 1) blocks = 1;
    long long chunk = (rows + blocks - 1) / blocks;
    if (chunk < 64) chunk = 64;
    chunk = (chunk + 3) / 4 * 4;
    hipLaunchKernelGGL(colsum_kernel, dim3((unsigned)((rows + chunk - 1) / chunk), cb), dim3(TR_THREADS), 0, eab_stream(stream), x,
                       out, rows, N, chunk);
    EAB_RETURN_LAUNCH_STATUS();
}

// ---------------------------------------------------------------------------------------------------
// filter-and-sum backward w.r.t. the beam-forming weights (EaBNet.py:114-117):
//   Yr = sum_m Wr Xr - Wi Xi,  Yi = sum_m Wr Xi + Wi Xr   =>   dWr = dYr Xr + dYi Xi,  dWi = dYi Xr - dYr Xi
//   dout [B][2][T][F], x [B][T][F][M][2] -> dw [B][T][F][ld] (first 2M columns, the rest zero)
// ---------------------------------------------------------------------------------------------------
// Four lanes per TF bin (lane p: microphones p, p+4, ...), 32-bit bin arithmetic with one division per bin: see
// filter_sum_kernel (csrc/filter_sum.hip) -- one thread per bin with four 64-bit divisions was the first version of these too.
__global__ __launch_bounds__(TR_THREADS) void filter_sum_ld_kernel(const float* __restrict__ w, const float* __restrict__ x,
                                                                   float* __restrict__ y, unsigned TF, int M, int ld, unsigned bins) {
    const unsigned p = threadIdx.x & 3, stride = gridDim.x * (TR_THREADS / 4);
    for (unsigned base = blockIdx.x * (TR_THREADS / 4); base < bins; base += stride) {   // workgroup-uniform trip count
        const unsigned bin = base + (threadIdx.x >> 2);
        const bool valid = bin < bins;
        const float2* wp = reinterpret_cast<const float2*>(w + (size_t)bin * ld);
        const float2* xp = reinterpret_cast<const float2*>(x) + (size_t)bin * M;
        float yr = 0.0f, yi = 0.0f;
        if (valid) {
            for (int m = p; m < M; m += 4) {
                const float2 a = wp[m], c = xp[m];
                yr += a.x * c.x - a.y * c.y;
                yi += a.x * c.y + a.y * c.x;
            }
        }
        yr += __shfl_xor(yr, 1); yi += __shfl_xor(yi, 1);
        yr += __shfl_xor(yr, 2); yi += __shfl_xor(yi, 2);
        if (valid && p == 0) {
            const unsigned b = bin / TF, pos = bin - b * TF;
            y[(size_t)(2 * b) * TF + pos] = yr;
            y[(size_t)(2 * b + 1) * TF + pos] = yi;
        }
    }
}

__global__ __launch_bounds__(TR_THREADS) void filter_sum_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ x,
                                                                    float* __restrict__ dw, unsigned TF, int M, int ld, unsigned bins) {
    const unsigned p = threadIdx.x & 3;
    for (unsigned bin = blockIdx.x * (TR_THREADS / 4) + (threadIdx.x >> 2); bin < bins; bin += gridDim.x * (TR_THREADS / 4)) {
        const unsigned b = bin / TF, pos = bin - b * TF;
        const float dr = dout[(size_t)(2 * b) * TF + pos], di = dout[(size_t)(2 * b + 1) * TF + pos];
        const float2* xp = reinterpret_cast<const float2*>(x) + (size_t)bin * M;
        float2* wp = reinterpret_cast<float2*>(dw + (size_t)bin * ld);
        for (int m = p; m < M; m += 4) {
            const float2 c = xp[m];
            wp[m] = make_float2(dr * c.x + di * c.y, di * c.x - dr * c.y);
        }
        for (int m = M + p; m < ld / 2; m += 4) wp[m] = make_float2(0.0f, 0.0f);   // padding columns of the row
    }
}

static inline unsigned fs_grid(long long bins) {
    const long long tiles = (bins + TR_THREADS / 4 - 1) / (TR_THREADS / 4);
    return (unsigned)(tiles < 256 * 8 ? tiles : 256 * 8);
}

// w rows of `ld` floats (the first 2M used): the training program keeps the beam-forming weights in a 64-column tile
extern "C" int eab_filter_sum_ld_f32(const float* w, const float* x, float* y, int B, int T, int F, int M, int ld, eab_stream_t stream) {
    EAB_CHECK_ARG(w && x && y && B > 0 && T > 0 && F > 0 && M > 0 && ld >= 2 * M && (ld % 2) == 0);
    const long long bins = (long long)B * T * F;
    EAB_CHECK_ARG(bins < (1ll << 30));
    hipLaunchKernelGGL(filter_sum_ld_kernel, dim3(fs_grid(bins)), dim3(TR_THREADS), 0, eab_stream(stream), w, x, y,
                       (unsigned)((long long)T * F), M, ld, (unsigned)bins);
    EAB_RETURN_LAUNCH_STATUS();
}

extern "C" int eab_filter_sum_bwd_f32(const float* dout, const float* x, float* dw, int B, int T, int F, int M, int ld,
                                      eab_stream_t stream) {
    EAB_CHECK_ARG(dout && x && dw && B > 0 && T > 0 && F > 0 && M > 0 && ld >= 2 * M && (ld % 2) == 0);
    const long long bins = (long long)B * T * F;
    EAB_CHECK_ARG(bins < (1ll << 30));
    hipLaunchKernelGGL(filter_sum_bwd_kernel, dim3(fs_grid(bins)), dim3(TR_THREADS), 0, eab_stream(stream), dout, x, dw,
                       (unsigned)((long long)T * F), M, ld, (unsigned)bins);
    EAB_RETURN_LAUNCH_STATUS();
}

// ---------------------------------------------------------------------------------------------------
// LayerNorm(64) over the channel axis of [rows][64] (EaBNet.py:598,608), forward with saved (mean, rstd), and
// backward.  16 lanes x float4 per row (DPP row reductions); dgamma / dbeta through LDS + atomics.
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ float tr_row_sum(float v) {      // sum over the 16 lanes of a DPP row
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));
    return v;
}

__global__ __launch_bounds__(TR_THREADS) void layernorm_fwd_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                                   const float* __restrict__ b, float eps, float* __restrict__ y,
                                                                   float* __restrict__ mr, long long rows) {
    const int lc = (threadIdx.x & 15) * 4;
    const f32x4 g4 = *reinterpret_cast<const f32x4*>(g + lc), b4 = *reinterpret_cast<const f32x4*>(b + lc);
    for (long long r = (long long)blockIdx.x * 16 + (threadIdx.x >> 4); r < rows; r += (long long)gridDim.x * 16) {
        f32x4 v = *reinterpret_cast<const f32x4*>(&x[r * 64 + lc]);
        const float mean = tr_row_sum((v[0] + v[1]) + (v[2] + v[3])) * (1.0f / 64.0f);
        const f32x4 d = {v[0] - mean, v[1] - mean, v[2] - mean, v[3] - mean};
        const float q = tr_row_sum((d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]));
        const float rstd = 1.0f / sqrtf(q * (1.0f / 64.0f) + eps);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = d[j] * rstd * g4[j] + b4[j];
        *reinterpret_cast<f32x4*>(&y[r * 64 + lc]) = v;
        if (lc == 0) *reinterpret_cast<float2*>(&mr[r * 2]) = make_float2(mean, rstd);
    }
}

__global__ __launch_bounds__(TR_THREADS) void layernorm_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                                   const float* __restrict__ mr, const float* __restrict__ g,
                                                                   float* __restrict__ dx, float* __restrict__ dg,
                                                                   float* __restrict__ db, long long rows) {
    __shared__ float red[2][16][64];
    const int lc = (threadIdx.x & 15) * 4, rl = threadIdx.x >> 4;
    const f32x4 g4 = *reinterpret_cast<const f32x4*>(g + lc);
    f32x4 sg = {0.f, 0.f, 0.f, 0.f}, sb = {0.f, 0.f, 0.f, 0.f};
    // every thread of a 16-lane row takes the same trip count (DPP reductions need all 16 lanes)
    for (long long r = (long long)blockIdx.x * 16 + rl; r < rows; r += (long long)gridDim.x * 16) {
        const f32x4 d = *reinterpret_cast<const f32x4*>(&dy[r * 64 + lc]), v = *reinterpret_cast<const f32x4*>(&x[r * 64 + lc]);
        const float2 m = *reinterpret_cast<const float2*>(&mr[r * 2]);
        f32x4 xh, dh;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            xh[j] = (v[j] - m.x) * m.y;
            dh[j] = d[j] * g4[j];
            sg[j] += d[j] * xh[j];
            sb[j] += d[j];
        }
        const float A = tr_row_sum((dh[0] + dh[1]) + (dh[2] + dh[3])) * (1.0f / 64.0f);
        const float Q = tr_row_sum((dh[0] * xh[0] + dh[1] * xh[1]) + (dh[2] * xh[2] + dh[3] * xh[3])) * (1.0f / 64.0f);
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = m.y * (dh[j] - A - xh[j] * Q);
        *reinterpret_cast<f32x4*>(&dx[r * 64 + lc]) = o;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        red[0][rl][lc + j] = sg[j];
        red[1][rl][lc + j] = sb[j];
    }
    __syncthreads();
    if (threadIdx.x < 128) {
        const int k = threadIdx.x >> 6, c = threadIdx.x & 63;
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) s += red[k][i][c];
        atomicAdd(k == 0 ? &dg[c] : &db[c], s);
    }
}

extern "C" int eab_layernorm64_fwd_f32(const float* x, const float* g, const float* b, float eps, float* y, float* mr, long long rows,
                                       eab_stream_t stream) {
    EAB_CHECK_ARG(x && g && b && y && mr && rows > 0);
    long long grid = (rows + 15) / 16;
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(layernorm_fwd_kernel, dim3((unsigned)grid), dim3(TR_THREADS), 0, eab_stream(stream), x, g, b, eps, y, mr, rows);
    EAB_RETURN_LAUNCH_STATUS();
}

extern "C" int eab_layernorm64_bwd_f32(const float* dy, const float* x, const float* mr, const float* g, float* dx, float* dg,
                                       float* db, long long rows, eab_stream_t stream) {
    EAB_CHECK_ARG(dy && x && mr && g && dx && dg && db && rows > 0);
    long long grid = (rows + 15) / 16;
    if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(layernorm_bwd_kernel, dim3((unsigned)grid), dim3(TR_THREADS), 0, eab_stream(stream), dy, x, mr, g, dx, dg, db, rows);
    EAB_RETURN_LAUNCH_STATUS();
}
