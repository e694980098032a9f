// K13: complex filter-and-sum  Y = sum_m W_m * X_m  (reference EaBNet.py:114-117),
// and K12b+K13 fused: second Linear of w_dnn + filter-and-sum (EaBNet.py:596,613-117).
#include "common.h"

// one thread per TF bin; W and X rows are M*2 contiguous floats (64 B at M=8).
__global__ __launch_bounds__(256) void filter_sum_kernel(const float* __restrict__ w, const float* __restrict__ x,
                                                         float* __restrict__ y, int T, int F, int M, long long bins) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < bins;
         i += (long long)gridDim.x * blockDim.x) {
        const float2* wp = reinterpret_cast<const float2*>(w) + i * M;
        const float2* xp = reinterpret_cast<const float2*>(x) + i * M;
        float yr = 0.0f, yi = 0.0f;
        for (int m = 0; m < M; ++m) {
            float2 a = wp[m], c = xp[m];
            yr += a.x * c.x - a.y * c.y;
            yi += a.x * c.y + a.y * c.x;
        }
        long long f = i % F, bt = i / F, t = bt % T, b = bt / T;
        y[((b * 2 + 0) * T + t) * F + f] = yr;
        y[((b * 2 + 1) * T + t) * F + f] = yi;
    }
}

extern "C" int eab_filter_sum_f32(const float* w, const float* x, float* y, int B, int T, int F, int M,
                                  eab_stream_t stream) {
    EAB_CHECK_ARG(w && x && y && B > 0 && T > 0 && F > 0 && M > 0);
    long long bins = (long long)B * T * F;
    int grid = (int)((bins + 255) / 256 < 8192 ? (bins + 255) / 256 : 8192);
    hipLaunchKernelGGL(filter_sum_kernel, dim3(grid), dim3(256), 0, eab_stream(stream), w, x, y, T, F, M, bins);
    EAB_RETURN_LAUNCH_STATUS();
}

// ---------------------------------------------------------------------------
// bfw_filter_sum: 64 TF bins per workgroup.  The 64x64 activation tile and the
// (2M)x64 weight matrix are staged in LDS; 4 lanes share a bin, lane p of the
// quad computes mics p, p+4, ... (both re and im weights), multiplies with the
// bin's X and the quad is reduced with two xor-shuffles.
// MLP = true: the tile staged from HBM is the LSTM output h and the first
// Linear + ReLU of LSTM_BF.w_dnn (EaBNet.py:594-595,612) runs here too --
// y1 = relu(h W1^T + b1) as one 64x64x64 fp32-MFMA product per workgroup
// (2x2 waves, 32 MFMAs each) -- so the 264 MB y1 tensor of a 16-utterance
// batch is neither written nor read back.
// ---------------------------------------------------------------------------
#define BFW_ROWS 64
#define BFW_K 64
#define BFW_MAXM 32

template <bool MLP>
__global__ __launch_bounds__(256) void bfw_filter_sum_kernel(
    const float* __restrict__ y1, const float* __restrict__ w2, const float* __restrict__ b2,
    const float* __restrict__ x, float* __restrict__ out, float* __restrict__ bfw, int T, int F, int M,
    long long bins, const int* __restrict__ t_pos, int t_count, const float* __restrict__ w1,
    const float* __restrict__ b1) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* ytile = smem;                            // [BFW_ROWS][BFW_K + 4]
    float* wl = smem + BFW_ROWS * (BFW_K + 4);      // [2M][BFW_K + 4]
    float* w1l = wl + 2 * M * (BFW_K + 4);          // MLP: [64][BFW_K + 4] first-layer weights
    const int tid = threadIdx.x;
    // `bins` counts the TF bins computed.  Streaming window: index j runs over [B][t_count][F] and maps to
    // bin (b*T + *t_pos)*F + rem; rows past the utterance end are dropped.
    const long long per_b = (long long)(t_pos ? t_count : T) * F, p_lo = t_pos ? (long long)*t_pos * F : 0;
    auto to_bin = [&](long long j) -> long long {
        if (j >= bins) return -1;
        const long long b = j / per_b, rem = j - b * per_b;
        return p_lo + rem < (long long)T * F ? b * T * F + p_lo + rem : -1;
    };
    // stage weights and activations (float4, coalesced)
    for (int e = tid; e < 2 * M * (BFW_K / 4); e += 256) {
        int r = e / (BFW_K / 4), c4 = e % (BFW_K / 4);
        *reinterpret_cast<f32x4*>(&wl[r * (BFW_K + 4) + c4 * 4]) =
            *reinterpret_cast<const f32x4*>(&w2[(size_t)r * BFW_K + c4 * 4]);
    }
    if (MLP) {
        for (int e = tid; e < BFW_K * (BFW_K / 4); e += 256) {
            int r = e / (BFW_K / 4), c4 = e % (BFW_K / 4);
            *reinterpret_cast<f32x4*>(&w1l[r * (BFW_K + 4) + c4 * 4]) =
                *reinterpret_cast<const f32x4*>(&w1[(size_t)r * BFW_K + c4 * 4]);
        }
    }
    // the weights stay in LDS while the workgroup walks over its 64-bin tiles (grid-stride)
    const long long ntiles = (bins + BFW_ROWS - 1) / BFW_ROWS;
    for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const long long row0 = tile * BFW_ROWS;
    __syncthreads();                                 // previous tile fully consumed (and weights staged)
    for (int e = tid; e < BFW_ROWS * (BFW_K / 4); e += 256) {
        int r = e / (BFW_K / 4), c4 = e % (BFW_K / 4);
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        const long long bn = to_bin(row0 + r);
        if (bn >= 0) v = *reinterpret_cast<const f32x4*>(&y1[(size_t)bn * BFW_K + c4 * 4]);
        *reinterpret_cast<f32x4*>(&ytile[r * (BFW_K + 4) + c4 * 4]) = v;
    }
    if (MLP) {
        __syncthreads();
        // y1[row][n] = relu(b1[n] + sum_k h[row][k] W1[n][k]): wave (wm, wn) owns the 32x32 block
        // (rows 32 wm.., columns 32 wn..); fragment idiom of conv_gemm.hip (lane reads floats
        // [8g + 4h, +4) of its row, four k-steps per read, same k permutation on both operands)
        const int lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1, li = lane & 31, lh = lane >> 5;
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
        const float* ar = &ytile[(wm * 32 + li) * (BFW_K + 4) + 4 * lh];
        const float* br = &w1l[(wn * 32 + li) * (BFW_K + 4) + 4 * lh];
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(ar + 8 * g);
            const f32x4 bq = *reinterpret_cast<const f32x4*>(br + 8 * g);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], bq[j], acc, 0, 0, 0);
        }
        __syncthreads();                             // every wave has read the h tile
        const float bias = b1[wn * 32 + li];
#pragma unroll
        for (int i = 0; i < 16; ++i)                 // accumulator i: row 8 (i/4) + 4 lh + i%4, column li
            ytile[(wm * 32 + 8 * (i >> 2) + 4 * lh + (i & 3)) * (BFW_K + 4) + wn * 32 + li] = fmaxf(acc[i] + bias, 0.0f);
    }
    __syncthreads();
    const int r = tid >> 2, p = tid & 3;
    const long long bin = to_bin(row0 + r);
    const bool valid = bin >= 0;
    const float* yr_ = &ytile[r * (BFW_K + 4)];
    float accr = 0.0f, acci = 0.0f;
    for (int m = p; m < M; m += 4) {
        const float* wr_ = &wl[(2 * m) * (BFW_K + 4)];
        const float* wi_ = &wl[(2 * m + 1) * (BFW_K + 4)];
        float wr = b2[2 * m], wi = b2[2 * m + 1];
#pragma unroll 4
        for (int k = 0; k < BFW_K; k += 4) {
            f32x4 a = *reinterpret_cast<const f32x4*>(&yr_[k]);
            f32x4 u = *reinterpret_cast<const f32x4*>(&wr_[k]);
            f32x4 v = *reinterpret_cast<const f32x4*>(&wi_[k]);
            wr += a[0] * u[0] + a[1] * u[1] + a[2] * u[2] + a[3] * u[3];
            wi += a[0] * v[0] + a[1] * v[1] + a[2] * v[2] + a[3] * v[3];
        }
        if (valid) {
            float2 xv = reinterpret_cast<const float2*>(x)[bin * M + m];
            accr += wr * xv.x - wi * xv.y;
            acci += wr * xv.y + wi * xv.x;
            if (bfw) reinterpret_cast<float2*>(bfw)[bin * M + m] = make_float2(wr, wi);
        }
    }
    accr += __shfl_xor(accr, 1); acci += __shfl_xor(acci, 1);
    accr += __shfl_xor(accr, 2); acci += __shfl_xor(acci, 2);
    if (valid && p == 0) {
        long long f = bin % F, bt = bin / F, t = bt % T, b = bt / T;
        out[((b * 2 + 0) * T + t) * F + f] = accr;
        out[((b * 2 + 1) * T + t) * F + f] = acci;
    }
    }   // tile loop
}

extern "C" int eab_bfw_filter_sum_f32(const float* y1, const float* w2, const float* b2, const float* x, float* out,
                                      float* bfw, int B, int T, int F, int M, eab_stream_t stream) {
    return eab_mlp_bfw_filter_sum_f32(y1, nullptr, nullptr, w2, b2, x, out, bfw, B, T, F, M, eab_time_window{nullptr, 0},
                                      stream);
}

extern "C" int eab_bfw_filter_sum_win_f32(const float* y1, const float* w2, const float* b2, const float* x, float* out,
                                          float* bfw, int B, int T, int F, int M, eab_time_window win,
                                          eab_stream_t stream) {
    return eab_mlp_bfw_filter_sum_f32(y1, nullptr, nullptr, w2, b2, x, out, bfw, B, T, F, M, win, stream);
}

extern "C" int eab_mlp_bfw_filter_sum_f32(const float* y1, const float* w1, const float* b1, const float* w2,
                                          const float* b2, const float* x, float* out, float* bfw, int B, int T, int F,
                                          int M, eab_time_window win, eab_stream_t stream) {
    EAB_CHECK_ARG(y1 && w2 && b2 && x && out && B > 0 && T > 0 && F > 0 && M > 0 && M <= BFW_MAXM);
    EAB_CHECK_ARG((w1 == nullptr) == (b1 == nullptr));
    EAB_CHECK_ARG(win.pos == nullptr || win.count > 0);
    long long bins = (long long)B * (win.pos ? win.count : T) * F;
    long long grid = (bins + BFW_ROWS - 1) / BFW_ROWS;
    if (grid > 256 * 4) grid = 256 * 4;              // four resident workgroups per CU walk the tiles
    size_t shmem = (size_t)(BFW_ROWS + 2 * M + (w1 ? BFW_K : 0)) * (BFW_K + 4) * sizeof(float);
    if (w1)
        hipLaunchKernelGGL(bfw_filter_sum_kernel<true>, dim3((unsigned)grid), dim3(256), shmem, eab_stream(stream), y1, w2,
                           b2, x, out, bfw, T, F, M, bins, win.pos, win.count, w1, b1);
    else
        hipLaunchKernelGGL(bfw_filter_sum_kernel<false>, dim3((unsigned)grid), dim3(256), shmem, eab_stream(stream), y1, w2,
                           b2, x, out, bfw, T, F, M, bins, win.pos, win.count, w1, b1);
    EAB_RETURN_LAUNCH_STATUS();
}
