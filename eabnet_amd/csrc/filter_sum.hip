// K13: complex filter-and-sum  Y = sum_m W_m * X_m  (reference EaBNet.py:114-117),
// and K12b+K13 fused: second Linear of w_dnn + filter-and-sum (EaBNet.py:596,613-117).
#include "common.h"

typedef float f32x2 __attribute__((ext_vector_type(2)));

// Four lanes per TF bin: lane p multiplies microphones p, p+4, ... and the quad is reduced with two xor-shuffles, so a wave
// reads 16 bins x M x 8 B = one contiguous block of W and of X per load (one thread per bin -- the first version -- had every
// lane walk its own 64-byte row, and paid four emulated 64-bit divisions per bin for the (b, t, f) of the store).
// 32-bit bin arithmetic (host check), one division per stored bin.
__global__ __launch_bounds__(256) void filter_sum_kernel(const float* __restrict__ w, const float* __restrict__ x,
                                                         float* __restrict__ y, unsigned TF, int M, unsigned bins) {
    const unsigned p = threadIdx.x & 3, stride = gridDim.x * 64u;
    for (unsigned base = blockIdx.x * 64u; base < bins; base += stride) {       // workgroup-uniform trip count
        const unsigned bin = base + (threadIdx.x >> 2);
        const bool valid = bin < bins;
        const float2* wp = reinterpret_cast<const float2*>(w) + (size_t)bin * M;
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
        if (valid && p == 0) {                             // y[b][ri][t][f]: t * F + f = bin - b * TF
            const unsigned b = bin / TF, pos = bin - b * TF;
            y[(size_t)(2 * b) * TF + pos] = yr;
            y[(size_t)(2 * b + 1) * TF + pos] = yi;
        }
    }
}

extern "C" int eab_filter_sum_f32(const float* w, const float* x, float* y, int B, int T, int F, int M,
                                  eab_stream_t stream) {
    EAB_CHECK_ARG(w && x && y && B > 0 && T > 0 && F > 0 && M > 0);
    const long long bins = (long long)B * T * F;
    EAB_CHECK_ARG(bins < (1ll << 30));                     // 32-bit bin arithmetic in the kernel
    const long long tiles = (bins + 63) / 64;
    const int grid = (int)(tiles < 256 * 8 ? tiles : 256 * 8);
    hipLaunchKernelGGL(filter_sum_kernel, dim3(grid), dim3(256), 0, eab_stream(stream), w, x, y, (unsigned)((long long)T * F), M,
                       (unsigned)bins);
    EAB_RETURN_LAUNCH_STATUS();
}

// ---------------------------------------------------------------------------
// bfw_filter_sum: 64 TF bins per workgroup.  The 64x64 activation tile and the
// (2M)x64 weight matrix are staged in LDS; the per-bin beam-forming weights are one
// 64 x 2M x 64 product on v_mfma_f32_16x16x4_f32 (a wave per 16 bins), written back
// into the tile; then 4 lanes share a bin, lane p of the quad multiplies mics p, p+4, ...
// with the bin's X and the quad is reduced with two xor-shuffles.
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
    // (32-bit index arithmetic throughout: the host checks B * T * F < 2^31)
    const int per_b = (t_pos ? t_count : T) * F, p_lo = t_pos ? *t_pos * F : 0;
    const int nbins = (int)bins;
    // Index j = b * per_b + rem.  (b, rem) of a tile's first row is carried along the grid-stride walk and a row's bin follows
    // by addition: no integer division anywhere (the 64-bit divisions of the first version -- four per thread and tile in the
    // fetch, four more in the store -- were several hundred VALU instructions per tile, which this kernel's matrix pipe pays for).
    const int TF = T * F;
    auto advance = [&](int& b_, int& rem_, int by) {
        rem_ += by;
        while (rem_ >= per_b) {
            rem_ -= per_b;
            ++b_;
        }
    };
    // bin of row r of the tile that starts at (b_, rem_), index j0 + r; false: past the end
    auto row_bin = [&](int b_, int rem_, int j0, int r, int& bb, int& pos) -> bool {
        bb = b_;
        int rr = rem_ + r;
        while (rr >= per_b) {                       // at most once unless a batch element has fewer than 64 bins
            rr -= per_b;
            ++bb;
        }
        pos = p_lo + rr;
        return j0 + r < nbins && pos < TF;
    };
    // Whole-utterance call (no streaming window) on utterances of at least one tile: index j IS the bin, a tile's rows are
    // consecutive bins and (b, t F + f) of a row needs at most one carry -- straight-line address arithmetic instead of the
    // loops above (which the compiler keeps as branches around every row of the fetch and of the store)
    const bool flat = t_pos == nullptr && TF >= BFW_ROWS;
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
    const int ntiles = (nbins + BFW_ROWS - 1) / BFW_ROWS;
    // the activation tile of the NEXT step of the grid-stride walk is fetched into registers while this one is multiplied
    // (4 float4 per thread): the HBM round trip used to sit in front of every tile's MFMAs
    constexpr int TPT = BFW_ROWS * (BFW_K / 4) / 256;
    f32x4 pre[TPT];
    auto fetch_tile = [&](int tl, int b_, int rem_) {
#pragma unroll
        for (int k = 0; k < TPT; ++k) {
            const int e = tid + k * 256, r = e / (BFW_K / 4), c4 = e % (BFW_K / 4);
            pre[k] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (flat) {
                const int idx = tl * BFW_ROWS + r;                  // (tl >= ntiles: idx >= nbins)
                if (idx < nbins) pre[k] = *reinterpret_cast<const f32x4*>(&y1[(size_t)idx * BFW_K + c4 * 4]);
                continue;
            }
            int bb, pos;
            const bool ok = tl < ntiles && row_bin(b_, rem_, tl * BFW_ROWS, r, bb, pos);
            if (ok) pre[k] = *reinterpret_cast<const f32x4*>(&y1[(size_t)(bb * TF + pos) * BFW_K + c4 * 4]);
        }
    };
    int tb = 0, trem = 0;                            // (b, rem) of this tile's first row
    advance(tb, trem, (int)blockIdx.x * BFW_ROWS);
    int nb = tb, nrem = trem;                        // ... and of the next tile of the walk
    fetch_tile((int)blockIdx.x, tb, trem);
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int row0 = tile * BFW_ROWS;
    tb = nb;
    trem = nrem;
    advance(nb, nrem, (int)gridDim.x * BFW_ROWS);
    __syncthreads();                                 // previous tile fully consumed (and weights staged)
#pragma unroll
    for (int k = 0; k < TPT; ++k) {
        const int e = tid + k * 256, r = e / (BFW_K / 4), c4 = e % (BFW_K / 4);
        *reinterpret_cast<f32x4*>(&ytile[r * (BFW_K + 4) + c4 * 4]) = pre[k];
    }
    fetch_tile(tile + (int)gridDim.x, nb, nrem);
    // this lane's bin and its microphones' X values (mics p, p+4, p+8, p+12; more are fetched in the tail): requested here so that
    // the HBM round trip runs under the two matrix products instead of behind them
    const int r = tid >> 2, p = tid & 3;
    int ob, opos;
    bool valid;
    if (flat) {
        ob = tb;
        opos = trem + r;
        if (opos >= TF) {
            opos -= TF;
            ++ob;
        }
        valid = row0 + r < nbins;
    } else {
        valid = row_bin(tb, trem, row0, r, ob, opos);
    }
    const size_t bin = (size_t)(ob * TF + opos);
    float2 xpre[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int m = p + 4 * k;
        xpre[k] = make_float2(0.0f, 0.0f);
        if (valid && m < M) xpre[k] = reinterpret_cast<const float2*>(x)[bin * M + m];
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
    {
        // Second Linear of w_dnn (EaBNet.py:596): bfw[row][n] = b2[n] + sum_k y1[row][k] W2[n][k], n < 2M, on
        // v_mfma_f32_16x16x4_f32 (it ran on the vector unit before: 256 FMAs + ~100 LDS reads per thread and tile made this
        // HBM-side kernel VALU-bound).  Wave w owns rows 16w .. 16w+15 of the tile -- the rows its own lanes reduce below, so
        // the result goes back into the wave's rows of `ytile` without a workgroup barrier.  Lane (i = l & 15, kq = l >> 4)
        // reads floats [8 m + 2 kq, +2) of its row: conflict-free ds_read_b64 (row stride 68 floats = 2 * 17 eight-byte slots).
        const int lane = tid & 63, wave = tid >> 6, li = lane & 15, kq = lane >> 4;
        const int NB = (2 * M + 15) >> 4;             // 16-column blocks of the 2M outputs (<= 4)
        const float* ar = &ytile[(wave * 16 + li) * (BFW_K + 4) + 2 * kq];
        f32x2 af[8];
#pragma unroll
        for (int m = 0; m < 8; ++m) af[m] = *reinterpret_cast<const f32x2*>(ar + 8 * m);
        f32x4 acc2[4];
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) {
            acc2[cb] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (cb < NB) {                            // workgroup-uniform
                const int n = cb * 16 + li;
                const float* br = &wl[(n < 2 * M ? n : 2 * M - 1) * (BFW_K + 4) + 2 * kq];   // columns past 2M: a copy, never stored
#pragma unroll
                for (int m = 0; m < 8; ++m) {
                    const f32x2 bq = *reinterpret_cast<const f32x2*>(br + 8 * m);
                    acc2[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[m][0], bq[0], acc2[cb], 0, 0, 0);
                    acc2[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[m][1], bq[1], acc2[cb], 0, 0, 0);
                }
            }
        }
        // C layout: column n = cb*16 + li, rows 4 kq + r.  (All of this wave's reads of its ytile rows are complete: the
        // MFMAs that consumed them have produced acc2.)
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) {
            const int n = cb * 16 + li;
            if (cb < NB && n < 2 * M) {
                const float bias2 = b2[n];
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) ytile[(wave * 16 + 4 * kq + r4) * (BFW_K + 4) + n] = acc2[cb][r4] + bias2;
            }
        }
    }
    // per bin: 4 lanes share a row, lane p takes microphones p, p+4, ..; (wr, wi) come from the wave's own rows of ytile
    const float* yr_ = &ytile[r * (BFW_K + 4)];
    float accr = 0.0f, acci = 0.0f;
    for (int m = p, k = 0; m < M; m += 4, ++k) {
        const float2 wv = *reinterpret_cast<const float2*>(&yr_[2 * m]);
        const float wr = wv.x, wi = wv.y;
        if (valid) {
            const float2 xv = k < 4 ? xpre[k < 4 ? k : 0] : reinterpret_cast<const float2*>(x)[bin * M + m];
            accr += wr * xv.x - wi * xv.y;
            acci += wr * xv.y + wi * xv.x;
            if (bfw) reinterpret_cast<float2*>(bfw)[bin * M + m] = make_float2(wr, wi);
        }
    }
    accr += __shfl_xor(accr, 1); acci += __shfl_xor(acci, 1);
    accr += __shfl_xor(accr, 2); acci += __shfl_xor(acci, 2);
    if (valid && p == 0) {                           // out[b][ri][t][f], t * F + f = opos
        out[(size_t)(ob * 2 + 0) * TF + opos] = accr;
        out[(size_t)(ob * 2 + 1) * TF + opos] = acci;
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
    EAB_CHECK_ARG((long long)B * T * F < (1ll << 30));   // 32-bit bin arithmetic in the kernel ((2 b + 1) T F must fit too)
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
