// Cumulative LayerNorm ("cLN", reference CumulativeLayerNorm1d / 2d, EaBNet.py:696-769 -- the norm NormSwitch means to
// build for norm_type="cLN" but cannot: it passes the string dim_size as num_features, :689,691).  Per batch element
// and frame t the statistics run over all channels, all frequency bins and all frames <= t:
//     cum_mean[t] = sum_{t'<=t} sum_{c,f} x / (C F (t+1)),   cum_var[t] = E[x^2] - cum_mean^2,   y = gain_c (x - mean)/std + bias_c
// so the norm is causal and streams exactly: the running sums are the only state.
//   eab_cln_stats_f32 : frame sums (one workgroup per (b, t), fp64) -> scan over t (one workgroup per b) -> mr[b][t] = (mean, rstd)
//   eab_cln_apply_f32 : y = prelu(gain (x-mean) rstd + bias) [+ add]      (2-D units)      or
//                       y = gain (prelu(x) - mean) rstd + bias            (S-TCM order: the statistics are those of prelu(x))
//   eab_gate_rows_f32 : z = a * sigmoid(r) on the rows of the window (S-TCM gate, EaBNet.py:575)
// All three take the streaming window (eab_time_window).  Roofline "hbm".
#include "common.h"

#define CL_THREADS 256

__device__ __forceinline__ double cl_block_sum(double v, double* red) {
    // 256 threads -> one value (fixed order: wave shuffles then 4 partials)
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    const double s = (red[0] + red[1]) + (red[2] + red[3]);
    __syncthreads();
    return s;
}

// sums[b][t] = (sum v, sum v^2) over the row of P = F*C values, v = x or prelu(x, slope[c])
__global__ __launch_bounds__(CL_THREADS) void cln_frame_sums_kernel(const float* __restrict__ x, const float* __restrict__ slope, int T,
                                                                    int P, int C, const int* __restrict__ t_pos, int t_count,
                                                                    double* __restrict__ sums) {
    __shared__ double red[4];
    const int rows = t_pos ? t_count : T;
    const int b = blockIdx.x / rows, t = (t_pos ? *t_pos : 0) + blockIdx.x % rows;
    if (t >= T) return;
    const float* row = x + ((size_t)b * T + t) * P;
    double s = 0.0, q = 0.0;
    for (int i = threadIdx.x * 4; i < P; i += CL_THREADS * 4) {
        f32x4 v = *reinterpret_cast<const f32x4*>(row + i);
        if (slope) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(slope + (i % C));
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = eab_prelu(v[j], a[j]);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            s += (double)v[j];
            q += (double)v[j] * (double)v[j];
        }
    }
    s = cl_block_sum(s, red);
    q = cl_block_sum(q, red);
    if (threadIdx.x == 0) {
        sums[((size_t)b * T + t) * 2] = s;
        sums[((size_t)b * T + t) * 2 + 1] = q;
    }
}

// per batch element: running sums over t -> (mean, rstd); state[b] = (sum, sum of squares) up to the last frame done
__global__ __launch_bounds__(64) void cln_scan_kernel(const double* __restrict__ sums, int T, int P, float eps, const int* __restrict__ t_pos,
                                                      int t_count, double* __restrict__ state, float* __restrict__ mr) {
    if (threadIdx.x != 0) return;
    const int b = blockIdx.x;
    const int t_lo = t_pos ? *t_pos : 0;
    const int t_hi = t_pos ? (t_lo + t_count < T ? t_lo + t_count : T) : T;
    double cs = 0.0, cq = 0.0;
    if (state && t_lo > 0) {
        cs = state[b * 2];
        cq = state[b * 2 + 1];
    }
    for (int t = t_lo; t < t_hi; ++t) {
        cs += sums[((size_t)b * T + t) * 2];
        cq += sums[((size_t)b * T + t) * 2 + 1];
        const double cnt = (double)P * (double)(t + 1);
        const double mean = cs / cnt;
        double var = (cq - 2.0 * mean * cs) / cnt + mean * mean;        // the reference's expression (EaBNet.py:731, 764)
        *reinterpret_cast<float2*>(&mr[((size_t)b * T + t) * 2]) = make_float2((float)mean, (float)(1.0 / sqrt(var + (double)eps)));
    }
    if (state) {
        state[b * 2] = cs;
        state[b * 2 + 1] = cq;
    }
}

extern "C" int eab_cln_stats_f32(const float* x, const float* slope, int B, int T, int P, int C, float eps, double* sums,
                                 double* state, float* mr, eab_time_window win, eab_stream_t stream) {
    EAB_CHECK_ARG(x && sums && mr && B > 0 && T > 0 && P > 0 && C > 0 && (P % C) == 0 && (C % 4) == 0);
    EAB_CHECK_ARG(win.pos == nullptr || (win.count > 0 && state));
    const int rows = win.pos ? win.count : T;
    EAB_CHECK_ARG((long long)B * rows < (1ll << 31));
    hipLaunchKernelGGL(cln_frame_sums_kernel, dim3(B * rows), dim3(CL_THREADS), 0, eab_stream(stream), x, slope, T, P, C, win.pos,
                       win.count, sums);
    hipLaunchKernelGGL(cln_scan_kernel, dim3(B), dim3(64), 0, eab_stream(stream), sums, T, P, eps, win.pos, win.count, state, mr);
    EAB_RETURN_LAUNCH_STATUS();
}

__global__ __launch_bounds__(CL_THREADS) void cln_apply_kernel(const float* __restrict__ x, const float* __restrict__ mr,
                                                               const float* __restrict__ gain, const float* __restrict__ bias,
                                                               const float* __restrict__ slope, const float* __restrict__ add,
                                                               float* __restrict__ y, int T, int P, int C, int mode,
                                                               const int* __restrict__ t_pos, int t_count) {
    const int rows = t_pos ? t_count : T;
    const int b = blockIdx.y, t_lo = t_pos ? *t_pos : 0;
    const unsigned P4 = (unsigned)P >> 2, n4 = (unsigned)rows * P4;
    for (unsigned r = blockIdx.x * blockDim.x + threadIdx.x; r < n4; r += gridDim.x * blockDim.x) {
        const int t = t_lo + (int)(r / P4);
        if (t >= T) break;
        const size_t i = (((size_t)b * T + t) * P4 + r % P4);
        const int c = (int)((r % P4) * 4 % C);
        const float2 m = *reinterpret_cast<const float2*>(&mr[((size_t)b * T + t) * 2]);
        const f32x4 v = reinterpret_cast<const f32x4*>(x)[i];
        const f32x4 g = *reinterpret_cast<const f32x4*>(gain + c), be = *reinterpret_cast<const f32x4*>(bias + c),
                    a = *reinterpret_cast<const f32x4*>(slope + c);
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            o[j] = mode == EAB_XF_NORM_PRELU ? eab_prelu(fmaf((v[j] - m.x) * m.y, g[j], be[j]), a[j])
                                             : fmaf((eab_prelu(v[j], a[j]) - m.x) * m.y, g[j], be[j]);
        if (add) o += reinterpret_cast<const f32x4*>(add)[i];
        reinterpret_cast<f32x4*>(y)[i] = o;
    }
}

extern "C" int eab_cln_apply_f32(const float* x, const float* mr, const float* gain, const float* bias, const float* slope,
                                 const float* add, float* y, int B, int T, int P, int C, int mode, eab_time_window win,
                                 eab_stream_t stream) {
    EAB_CHECK_ARG(x && mr && gain && bias && slope && y && B > 0 && T > 0 && P > 0 && C > 0 && (C % 4) == 0 && (P % C) == 0 && B <= 65535);
    EAB_CHECK_ARG(mode == EAB_XF_NORM_PRELU || mode == EAB_XF_PRELU_NORM);
    EAB_CHECK_ARG(win.pos == nullptr || win.count > 0);
    const long long n4 = (long long)(win.pos ? win.count : T) * (P / 4);
    EAB_CHECK_ARG(n4 < (1ll << 31));
    long long gx = (n4 + CL_THREADS - 1) / CL_THREADS;
    const long long cap = (256 * 8 + B - 1) / B;
    if (gx > cap) gx = cap;
    if (gx < 1) gx = 1;
    hipLaunchKernelGGL(cln_apply_kernel, dim3((unsigned)gx, (unsigned)B), dim3(CL_THREADS), 0, eab_stream(stream), x, mr, gain, bias,
                       slope, add, y, T, P, C, mode, win.pos, win.count);
    EAB_RETURN_LAUNCH_STATUS();
}

// One frame of a streaming step (win.count == 1): statistics, running sums and apply of a unit in ONE launch, one workgroup
// per utterance.  Each part is the code of the stand-alone kernel it replaces -- the same thread-to-element map and block
// reduction as cln_frame_sums_kernel, the scan step of cln_scan_kernel, the expression of cln_apply_kernel -- so the frame's
// bits are those of the three-launch form (and of the offline pass).
__global__ __launch_bounds__(CL_THREADS) void cln_step_kernel(const float* __restrict__ x, const float* __restrict__ st_slope,
                                                              double* __restrict__ sums, double* __restrict__ state,
                                                              float* __restrict__ mr, const float* __restrict__ gain,
                                                              const float* __restrict__ bias, const float* __restrict__ slope,
                                                              const float* __restrict__ add, float* __restrict__ y, int T, int P,
                                                              int C, int mode, float eps, const int* __restrict__ t_pos) {
    __shared__ double red[4];
    __shared__ float2 m_sh;
    const int b = blockIdx.x, t = *t_pos;
    if (t >= T) return;                                            // (uniform)
    const float* row = x + ((size_t)b * T + t) * P;
    double s = 0.0, q = 0.0;
    for (int i = threadIdx.x * 4; i < P; i += CL_THREADS * 4) {
        f32x4 v = *reinterpret_cast<const f32x4*>(row + i);
        if (st_slope) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(st_slope + (i % C));
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = eab_prelu(v[j], a[j]);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            s += (double)v[j];
            q += (double)v[j] * (double)v[j];
        }
    }
    s = cl_block_sum(s, red);
    q = cl_block_sum(q, red);
    if (threadIdx.x == 0) {
        sums[((size_t)b * T + t) * 2] = s;
        sums[((size_t)b * T + t) * 2 + 1] = q;
        double cs = 0.0, cq = 0.0;
        if (t > 0) {
            cs = state[b * 2];
            cq = state[b * 2 + 1];
        }
        cs += s;
        cq += q;
        const double cnt = (double)P * (double)(t + 1);
        const double mean = cs / cnt;
        double var = (cq - 2.0 * mean * cs) / cnt + mean * mean;        // the reference's expression (EaBNet.py:731, 764)
        const float2 m = make_float2((float)mean, (float)(1.0 / sqrt(var + (double)eps)));
        *reinterpret_cast<float2*>(&mr[((size_t)b * T + t) * 2]) = m;
        state[b * 2] = cs;
        state[b * 2 + 1] = cq;
        m_sh = m;
    }
    __syncthreads();
    const float2 m = m_sh;
    const size_t base4 = ((size_t)b * T + t) * ((unsigned)P >> 2);
    for (unsigned r = threadIdx.x; r < ((unsigned)P >> 2); r += CL_THREADS) {
        const size_t i = base4 + r;
        const int c = (int)(r * 4 % C);
        const f32x4 v = reinterpret_cast<const f32x4*>(x)[i];
        const f32x4 g = *reinterpret_cast<const f32x4*>(gain + c), be = *reinterpret_cast<const f32x4*>(bias + c),
                    a = *reinterpret_cast<const f32x4*>(slope + c);
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            o[j] = mode == EAB_XF_NORM_PRELU ? eab_prelu(fmaf((v[j] - m.x) * m.y, g[j], be[j]), a[j])
                                             : fmaf((eab_prelu(v[j], a[j]) - m.x) * m.y, g[j], be[j]);
        if (add) o += reinterpret_cast<const f32x4*>(add)[i];
        reinterpret_cast<f32x4*>(y)[i] = o;
    }
}

extern "C" int eab_cln_step_f32(const float* x, const float* stat_slope, double* sums, double* state, float* mr, const float* gain,
                                const float* bias, const float* slope, const float* add, float* y, int B, int T, int P, int C,
                                int mode, float eps, eab_time_window win, eab_stream_t stream) {
    EAB_CHECK_ARG(x && sums && state && mr && gain && bias && slope && y && B > 0 && T > 0 && P > 0 && C > 0);
    EAB_CHECK_ARG((P % C) == 0 && (C % 4) == 0 && win.pos != nullptr && win.count == 1);
    EAB_CHECK_ARG(mode == EAB_XF_NORM_PRELU || mode == EAB_XF_PRELU_NORM);
    hipLaunchKernelGGL(cln_step_kernel, dim3((unsigned)B), dim3(CL_THREADS), 0, eab_stream(stream), x, stat_slope, sums, state, mr,
                       gain, bias, slope, add, y, T, P, C, mode, eps, win.pos);
    EAB_RETURN_LAUNCH_STATUS();
}

__global__ __launch_bounds__(CL_THREADS) void gate_rows_kernel(const float* __restrict__ a, const float* __restrict__ r,
                                                               float* __restrict__ z, int T, int row4, const int* __restrict__ t_pos,
                                                               int t_count) {
    const int rows = t_pos ? t_count : T;
    const int b = blockIdx.y, t_lo = t_pos ? *t_pos : 0;
    const unsigned n4 = (unsigned)rows * (unsigned)row4;
    for (unsigned k = blockIdx.x * blockDim.x + threadIdx.x; k < n4; k += gridDim.x * blockDim.x) {
        const int t = t_lo + (int)(k / row4);
        if (t >= T) break;
        const size_t i = ((size_t)b * T + t) * row4 + k % row4;
        const f32x4 av = reinterpret_cast<const f32x4*>(a)[i], rv = reinterpret_cast<const f32x4*>(r)[i];
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = av[j] * eab_sigmoid(rv[j]);
        reinterpret_cast<f32x4*>(z)[i] = o;
    }
}

extern "C" int eab_gate_rows_f32(const float* a, const float* r, float* z, int B, int T, int row_floats, eab_time_window win,
                                 eab_stream_t stream) {
    EAB_CHECK_ARG(a && r && z && B > 0 && T > 0 && row_floats > 0 && (row_floats % 4) == 0 && B <= 65535);
    EAB_CHECK_ARG(win.pos == nullptr || win.count > 0);
    const long long n4 = (long long)(win.pos ? win.count : T) * (row_floats / 4);
    EAB_CHECK_ARG(n4 < (1ll << 31));
    long long gx = (n4 + CL_THREADS - 1) / CL_THREADS;
    if (gx > 1024) gx = 1024;
    hipLaunchKernelGGL(gate_rows_kernel, dim3((unsigned)gx, (unsigned)B), dim3(CL_THREADS), 0, eab_stream(stream), a, r, z, T,
                       row_floats / 4, win.pos, win.count);
    EAB_RETURN_LAUNCH_STATUS();
}

// ---------------------------------------------------------------------------------------------------
// Backward of the cumulative LayerNorm unit (training; forward = eab_cln_stats_f32 + eab_cln_apply_f32 on the whole
// utterance).  With v = x (NORM_PRELU) or prelu(x) (PRELU_NORM), n_t = P (t+1), S_t / Q_t the running sums of v / v^2,
//   mu_t = S_t / n_t,  var_t = Q_t / n_t - mu_t^2,  r_t = (var_t + eps)^-1/2,  y = g_c (v - mu_t) r_t + be_c
// and dh = (d loss / d y) g_c (NORM_PRELU: d loss / d u through the PReLU that follows):
//   a_t = sum_p dh,   b_t = sum_p dh (v - mu_t),   c_t = -1/2 r_t^3 b_t                     (pass 1, fp64 per row)
//   A_t = sum_{t' >= t} (-r a - 2 mu c)_{t'} / n_{t'},   Bq_t = sum_{t' >= t} c_{t'} / n_{t'}     (pass 2, reverse scan per b)
//   dv[t][p] = dh r_t + A_t + 2 v Bq_t;   dx = dv (NORM_PRELU) or dv prelu'(x) (PRELU_NORM)       (pass 3)
// Parameter gradients per row and channel (summed over the rows by eab_colsum_f32): part[row][0][c] = sum_f d xh (gain),
// [1][c] = sum_f d (bias), [2][c] = the PReLU slope sum (NORM_PRELU: from pass 1; PRELU_NORM: from pass 3).
// Reference: autograd of CumulativeLayerNorm1d / 2d (EaBNet.py:713-733, 752-769) + nn.PReLU.
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ void cl_channel_reduce(f32x4 v, f32x4* red, int C, float* out) {
    // threads tid, tid + C/4, tid + 2 C/4, ... hold the same four channels (the row is walked with a stride of 1024 values,
    // 1024 % C == 0): fixed-order sum, one store per channel
    __syncthreads();
    red[threadIdx.x] = v;
    __syncthreads();
    const int G = C >> 2;
    if ((int)threadIdx.x < G) {
        f32x4 s = red[threadIdx.x];
        for (int k = threadIdx.x + G; k < CL_THREADS; k += G) s += red[k];
        *reinterpret_cast<f32x4*>(out + threadIdx.x * 4) = s;
    }
}

__global__ __launch_bounds__(CL_THREADS) void cln_bwd_rows_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                                  const float* __restrict__ mr, const float* __restrict__ gain,
                                                                  const float* __restrict__ bias, const float* __restrict__ slope,
                                                                  int P, int C, int mode, double* __restrict__ rs,
                                                                  float* __restrict__ part) {
    __shared__ double red[4];
    __shared__ f32x4 cred[CL_THREADS];
    const size_t row = blockIdx.x;
    const float2 m = *reinterpret_cast<const float2*>(&mr[row * 2]);
    double a = 0.0, bb = 0.0;
    f32x4 G = {0.f, 0.f, 0.f, 0.f}, Bt = G, S = G;
    for (int i = threadIdx.x * 4; i < P; i += CL_THREADS * 4) {
        const int c = i % C;
        const f32x4 v = *reinterpret_cast<const f32x4*>(x + row * P + i), d = *reinterpret_cast<const f32x4*>(dy + row * P + i);
        const f32x4 g = *reinterpret_cast<const f32x4*>(gain + c), be = *reinterpret_cast<const f32x4*>(bias + c),
                    al = *reinterpret_cast<const f32x4*>(slope + c);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float dd, vv;
            if (mode == EAB_XF_NORM_PRELU) {
                const float u = fmaf((v[j] - m.x) * m.y, g[j], be[j]);
                dd = u > 0.f ? d[j] : al[j] * d[j];
                S[j] += u > 0.f ? 0.f : d[j] * u;
                vv = v[j];
            } else {
                dd = d[j];
                vv = eab_prelu(v[j], al[j]);
            }
            const float cen = vv - m.x;
            G[j] = fmaf(dd, cen * m.y, G[j]);
            Bt[j] += dd;
            const double dh = (double)dd * (double)g[j];
            a += dh;
            bb += dh * (double)cen;
        }
    }
    a = cl_block_sum(a, red);
    bb = cl_block_sum(bb, red);
    if (threadIdx.x == 0) {
        rs[row * 2] = a;
        rs[row * 2 + 1] = bb;
    }
    float* pr = part + row * 3 * C;
    cl_channel_reduce(G, cred, C, pr);
    cl_channel_reduce(Bt, cred, C, pr + C);
    if (mode == EAB_XF_NORM_PRELU) cl_channel_reduce(S, cred, C, pr + 2 * C);
}

__global__ __launch_bounds__(64) void cln_bwd_scan_kernel(const double* __restrict__ rs, const float* __restrict__ mr, int T, int P,
                                                          float* __restrict__ ab) {
    if (threadIdx.x != 0) return;
    const size_t base = (size_t)blockIdx.x * T;
    double A = 0.0, Bq = 0.0;
    for (int t = T - 1; t >= 0; --t) {
        const double n = (double)P * (double)(t + 1);
        const double mu = (double)mr[(base + t) * 2], r = (double)mr[(base + t) * 2 + 1];
        const double c = -0.5 * r * r * r * rs[(base + t) * 2 + 1];
        A += (-r * rs[(base + t) * 2] - 2.0 * mu * c) / n;
        Bq += c / n;
        *reinterpret_cast<float2*>(&ab[(base + t) * 2]) = make_float2((float)A, (float)(2.0 * Bq));
    }
}

__global__ __launch_bounds__(CL_THREADS) void cln_bwd_apply_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                                   const float* __restrict__ mr, const float* __restrict__ ab,
                                                                   const float* __restrict__ gain, const float* __restrict__ bias,
                                                                   const float* __restrict__ slope, const float* __restrict__ acc_in,
                                                                   float* __restrict__ dx, int P, int C, int mode,
                                                                   float* __restrict__ part) {
    __shared__ f32x4 cred[CL_THREADS];
    const size_t row = blockIdx.x;
    const float2 m = *reinterpret_cast<const float2*>(&mr[row * 2]);
    const float2 q = *reinterpret_cast<const float2*>(&ab[row * 2]);
    f32x4 S = {0.f, 0.f, 0.f, 0.f};
    for (int i = threadIdx.x * 4; i < P; i += CL_THREADS * 4) {
        const int c = i % C;
        const size_t e = row * P + i;
        const f32x4 v = *reinterpret_cast<const f32x4*>(x + e), d = *reinterpret_cast<const f32x4*>(dy + e);
        const f32x4 g = *reinterpret_cast<const f32x4*>(gain + c), be = *reinterpret_cast<const f32x4*>(bias + c),
                    al = *reinterpret_cast<const f32x4*>(slope + c);
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (mode == EAB_XF_NORM_PRELU) {
                const float u = fmaf((v[j] - m.x) * m.y, g[j], be[j]);
                const float du = u > 0.f ? d[j] : al[j] * d[j];
                o[j] = fmaf(du * g[j], m.y, fmaf(v[j], q.y, q.x));
            } else {
                const float pv = eab_prelu(v[j], al[j]);
                const float dv = fmaf(d[j] * g[j], m.y, fmaf(pv, q.y, q.x));
                o[j] = v[j] > 0.f ? dv : al[j] * dv;
                S[j] += v[j] > 0.f ? 0.f : dv * v[j];
            }
        }
        if (acc_in) o += *reinterpret_cast<const f32x4*>(acc_in + e);
        *reinterpret_cast<f32x4*>(dx + e) = o;
    }
    if (mode == EAB_XF_PRELU_NORM) cl_channel_reduce(S, cred, C, part + row * 3 * C + 2 * C);
}

extern "C" int eab_train_cln_bwd_f32(const float* dy, const float* x, const float* mr, const float* gain, const float* bias,
                                     const float* slope, double* rowsums, float* ab, float* part, const float* acc_in, float* dx,
                                     int B, int T, int P, int C, int mode, eab_stream_t stream) {
    EAB_CHECK_ARG(dy && x && mr && gain && bias && slope && rowsums && ab && part && dx);
    EAB_CHECK_ARG(B > 0 && T > 0 && P > 0 && C > 0 && (P % C) == 0 && (mode == EAB_XF_NORM_PRELU || mode == EAB_XF_PRELU_NORM));
    EAB_CHECK_ARG((long long)B * T < (1ll << 31) && (long long)B * T * P < (1ll << 40));
    if ((CL_THREADS * 4) % C != 0) return EAB_EUNSUPPORTED;      // a thread keeps its four channels along the row
    hipStream_t s = eab_stream(stream);
    hipLaunchKernelGGL(cln_bwd_rows_kernel, dim3((unsigned)(B * T)), dim3(CL_THREADS), 0, s, dy, x, mr, gain, bias, slope, P, C, mode,
                       rowsums, part);
    hipLaunchKernelGGL(cln_bwd_scan_kernel, dim3((unsigned)B), dim3(64), 0, s, rowsums, mr, T, P, ab);
    hipLaunchKernelGGL(cln_bwd_apply_kernel, dim3((unsigned)(B * T)), dim3(CL_THREADS), 0, s, dy, x, mr, ab, gain, bias, slope, acc_in,
                       dx, P, C, mode, part);
    EAB_RETURN_LAUNCH_STATUS();
}
