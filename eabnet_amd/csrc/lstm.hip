// K10+K11: LayerNorm(64) + one LSTM(64->64) layer, persistent over time.
// Reference: LSTM_BF.forward, EaBNet.py:608-611 (nn.LSTM, batch_first, zero
// initial state, gate order i,f,g,o, biases b_ih + b_hh).
//
// The B*F sequences are independent; the time loop is strictly sequential.
// One workgroup (4 waves) owns 16 sequences for the whole utterance:
//   * wave w owns hidden units [16w, 16w+16): its four 16-column MFMA tiles are
//     the i, f, g, o pre-activations of the SAME units, so the cell update is
//     lane-local in the accumulator layout (col = lane&15 = unit,
//     row = 4*(lane>>4)+r = sequence);
//   * the [x_t | h_{t-1}] -> gates weights (K = 128, the wave's 64 columns) stay
//     in 128 VGPRs for all T steps as MFMA B operands (v_mfma_f32_16x16x4_f32);
//   * per step the A operand [16 seq][128] comes from a double-buffered LDS tile:
//     x_{t+1} is loaded (one contiguous 4 KB block of the channels-last tensor),
//     layer-normalised with 16-lane shuffles and stored while step t computes;
//     h_t is written back by each lane; ONE barrier per step;
//   * h_t leaves as one coalesced 4 KB store read back from that LDS tile.
// Bound: the fp32 matrix pipe (128 MFMA x 32 cycles per step per SIMD).
#include "common.h"

#define LS_H 64
#define LS_K 128
#define LS_SEQ 16
#define LS_LD (LS_K + 4)   // odd 16-byte-slot stride (33): conflict-free b128 fragment reads

__global__ __launch_bounds__(256) void lstm64_kernel(const float* __restrict__ x, const float* __restrict__ ln_g,
                                                     const float* __restrict__ ln_b, float ln_eps,
                                                     const float* __restrict__ wcat, const float* __restrict__ bias,
                                                     float* __restrict__ h_out, int T, int F, int S) {
    __shared__ __attribute__((aligned(16))) float xh[2][LS_SEQ * LS_LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ln = lane & 15, lk = lane >> 4;
    const int s0 = blockIdx.x * LS_SEQ;

    // ---- stationary weights: wreg[g][4j+s] = Wcat[g*64 + 16w + ln][16j + 4*lk + s]
    float wreg[4][32];
    float bia[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int row = g * LS_H + wave * 16 + ln;
        bia[g] = bias[row];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(&wcat[(size_t)row * LS_K + 16 * j + 4 * lk]);
#pragma unroll
            for (int s = 0; s < 4; ++s) wreg[g][4 * j + s] = v[s];
        }
    }

    // ---- loader role: thread -> (sequence ls, channels lc..lc+3)
    const int ls = tid >> 4, lc = (tid & 15) * 4;
    const int sg = s0 + ls;                       // global sequence = b*F + f
    const bool sv = sg < S;
    const int sb = sv ? sg / F : 0, sf = sv ? sg - sb * F : 0;
    const size_t seq_off = ((size_t)sb * T * F + sf) * LS_H + lc;   // + t*F*64
    const size_t t_stride = (size_t)F * LS_H;
    f32x4 g4 = {1.f, 1.f, 1.f, 1.f}, b4 = {0.f, 0.f, 0.f, 0.f};
    if (ln_g) {
        g4 = *reinterpret_cast<const f32x4*>(ln_g + lc);
        b4 = *reinterpret_cast<const f32x4*>(ln_b + lc);
    }

    auto load_x = [&](int t) -> f32x4 {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (sv) v = *reinterpret_cast<const f32x4*>(x + seq_off + (size_t)t * t_stride);
        return v;
    };
    auto norm_store = [&](f32x4 v, int buf) {
        if (ln_g) {
            // LayerNorm over the 64 channels = 16 lanes x 4, two-pass in registers
            float s = (v[0] + v[1]) + (v[2] + v[3]);
#pragma unroll
            for (int m = 1; m < 16; m <<= 1) s += __shfl_xor(s, m);
            const float mean = s * (1.0f / 64.0f);
            f32x4 dlt = {v[0] - mean, v[1] - mean, v[2] - mean, v[3] - mean};
            float q = (dlt[0] * dlt[0] + dlt[1] * dlt[1]) + (dlt[2] * dlt[2] + dlt[3] * dlt[3]);
#pragma unroll
            for (int m = 1; m < 16; m <<= 1) q += __shfl_xor(q, m);
            const float rstd = 1.0f / sqrtf(q * (1.0f / 64.0f) + ln_eps);
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = dlt[j] * rstd * g4[j] + b4[j];
        }
        *reinterpret_cast<f32x4*>(&xh[buf][ls * LS_LD + lc]) = v;
    };

    // h_{-1} = 0, c_{-1} = 0
    *reinterpret_cast<f32x4*>(&xh[0][ls * LS_LD + LS_H + lc]) = f32x4{0.f, 0.f, 0.f, 0.f};
    norm_store(load_x(0), 0);
    float cst[4] = {0.f, 0.f, 0.f, 0.f};
    __syncthreads();

    for (int t = 0; t < T; ++t) {
        const int cur = t & 1, nxt = cur ^ 1;
        f32x4 xn = {0.f, 0.f, 0.f, 0.f};
        if (t + 1 < T) xn = load_x(t + 1);            // in flight during the MFMAs

        f32x4 acc[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) acc[g] = f32x4{bia[g], bia[g], bia[g], bia[g]};
        const float* arow = &xh[cur][ln * LS_LD + 4 * lk];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(arow + 16 * j);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], wreg[g][4 * j + s], acc[g], 0, 0, 0);
        }
        // cell update: lane holds unit u = 16*wave + ln for sequences 4*lk + r
        const int u = wave * 16 + ln;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float ig = eab_sigmoid(acc[0][r]);
            const float fg = eab_sigmoid(acc[1][r]);
            const float gg = eab_tanh(acc[2][r]);
            const float og = eab_sigmoid(acc[3][r]);
            cst[r] = fg * cst[r] + ig * gg;
            const float h = og * eab_tanh(cst[r]);
            xh[nxt][(4 * lk + r) * LS_LD + LS_H + u] = h;
        }
        if (t + 1 < T) norm_store(xn, nxt);
        __syncthreads();
        // coalesced write-back of h_t from the tile the next step reads
        if (sv) {
            const f32x4 hv = *reinterpret_cast<const f32x4*>(&xh[nxt][ls * LS_LD + LS_H + lc]);
            *reinterpret_cast<f32x4*>(h_out + seq_off + (size_t)t * t_stride) = hv;
        }
    }
}

extern "C" int eab_lstm64_f32(const float* x, const float* ln_g, const float* ln_b, float ln_eps, const float* wcat,
                              const float* bias, float* h_out, int B, int T, int F, eab_stream_t stream) {
    EAB_CHECK_ARG(x && wcat && bias && h_out && B > 0 && T > 0 && F > 0);
    EAB_CHECK_ARG((ln_g == nullptr) == (ln_b == nullptr));
    const long long S = (long long)B * F;
    EAB_CHECK_ARG(S < (1ll << 30));
    const int grid = (int)((S + LS_SEQ - 1) / LS_SEQ);
    hipLaunchKernelGGL(lstm64_kernel, dim3(grid), dim3(256), 0, eab_stream(stream), x, ln_g, ln_b, ln_eps, wcat, bias,
                       h_out, T, F, (int)S);
    EAB_RETURN_LAUNCH_STATUS();
}
