// K10+K11: LayerNorm(64) + one LSTM(64->64) layer, persistent over time.
// Reference: LSTM_BF.forward, EaBNet.py:608-611 (nn.LSTM, batch_first, zero
// initial state, gate order i,f,g,o, biases b_ih + b_hh).
//
// The B*F sequences are independent; the time loop is strictly sequential.
// One workgroup (4 waves, one per SIMD) owns 16 sequences for the whole utterance:
//   * wave w owns hidden units [16w, 16w+16): its four 16-column MFMA tiles are
//     the i, f, g, o pre-activations of the SAME units, so the cell update is
//     lane-local in the accumulator layout (col = lane&15 = unit,
//     row = 4*(lane>>4)+r = sequence);
//   * the gate weights of the wave's 64 columns ([x | h] -> K = 128) stay in 128
//     VGPRs for all T steps as MFMA B operands (v_mfma_f32_16x16x4_f32);
//   * software pipeline over time.  The input half  W_x . x_{t+1}  does not depend
//     on the recurrence; its 64 MFMAs fill the matrix pipe while the recurrence
//     is latency bound.  A wave issues in order, so the order is pinned:
//         H  : 64 MFMAs  acc = accx + W_h h_{t-1}          (critical path)
//              + LayerNorm of x_{t+2} on the VALU (DPP row reductions, no LDS)
//         X1 : first 32 MFMAs of accx' = b + W_x x_{t+1}, cell update of step t
//              interleaved by the compiler (exp/rcp sigmoid, tanh)
//         EX : h_t -> LDS, ONE barrier, h_t fragments -> registers
//         X2 : last 32 MFMAs of accx' -- they run while the barrier resolves
//     so the pipe holds 128 MFMA x 32 cycles per step back to back.
//   * x is fetched from HBM four steps ahead (one contiguous 4 KB block of the
//     channels-last tensor per workgroup); h_t leaves as one coalesced 4 KB store.
// Bound: fp32 matrix pipe.
#include "common.h"
#include <cstdlib>
#include <type_traits>

#define LS_H 64
#define LS_SEQ 16
#define LS_LD (LS_H + 4)   // 272-byte rows: odd 16-byte-slot stride -> conflict-free b128 fragment reads

// sigmoid / tanh on the hardware exp and rcp (1 ulp class): abs error ~2e-7,
// three orders of magnitude inside the 1e-4 parity bar, and short enough to hide
// under the input-half MFMAs.
__device__ __forceinline__ float ls_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float ls_tanh(float x) { return fmaf(2.0f, ls_sigmoid(2.0f * x), -1.0f); }

// sum over the 16 lanes of a DPP row; every lane ends with the total
__device__ __forceinline__ float ls_row_sum(float v) {
    auto dpp = [](float x, auto ctrl) {
        return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), decltype(ctrl)::value, 0xF, 0xF, true));
    };
    v += dpp(v, std::integral_constant<int, 0xB1>{});    // quad_perm [1,0,3,2]
    v += dpp(v, std::integral_constant<int, 0x4E>{});    // quad_perm [2,3,0,1]
    v += dpp(v, std::integral_constant<int, 0x141>{});   // row_half_mirror
    v += dpp(v, std::integral_constant<int, 0x140>{});   // row_mirror
    return v;
}

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define LS_OOB 0x80000000u   // byte offset outside every legal tensor (host checks < 2^31): loads give 0, stores drop

template <bool LN, bool DUMP>
__global__ __launch_bounds__(256) void lstm64_kernel(const float* __restrict__ x, const float* __restrict__ ln_g,
                                                     const float* __restrict__ ln_b, float ln_eps,
                                                     const float* __restrict__ wcat, const float* __restrict__ bias,
                                                     float* __restrict__ h_out, int T, int F, int S,
                                                     const int* __restrict__ t_pos, int t_count,
                                                     float* __restrict__ c_state, float* __restrict__ gates) {
    __shared__ __attribute__((aligned(16))) float xs[2][LS_SEQ * LS_LD];
    __shared__ __attribute__((aligned(16))) float hs[2][LS_SEQ * LS_LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ln = lane & 15, lk = lane >> 4;
    const int s0 = blockIdx.x * LS_SEQ;
    // streaming (eab_time_window): steps [t_lo, t_hi) only; h_{t_lo-1} comes back from h_out, c from c_state
    const int t_lo = t_pos ? *t_pos : 0;
    const int t_hi = t_pos ? (t_lo + t_count < T ? t_lo + t_count : T) : T;

    // ---- stationary weights: w?[g][4j+s] = Wcat[g*64 + 16w + ln][(x:0 | h:64) + 16j + 4*lk + s]
    float wx[4][16], wh[4][16];
    float bia[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int row = g * LS_H + wave * 16 + ln;
        bia[g] = bias[row];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const f32x4 vx = *reinterpret_cast<const f32x4*>(&wcat[(size_t)row * 128 + 16 * j + 4 * lk]);
            const f32x4 vh = *reinterpret_cast<const f32x4*>(&wcat[(size_t)row * 128 + 64 + 16 * j + 4 * lk]);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                wx[g][4 * j + s] = vx[s];
                wh[g][4 * j + s] = vh[s];
            }
        }
    }

    // ---- loader role: thread -> (sequence ls, channels lc..lc+3)
    const int ls = tid >> 4, lc = (tid & 15) * 4;
    const int sg = s0 + ls;                       // global sequence = b*F + f
    const bool sv = sg < S;
    const int sb = sv ? sg / F : 0, sf = sv ? sg - sb * F : 0;
    // bounds-checked buffer accesses instead of exec-mask branches (tail sequences, t >= T)
    const unsigned total_bytes = (unsigned)S * (unsigned)T * (LS_H * 4u);
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, total_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rh = __builtin_amdgcn_make_buffer_rsrc(h_out, 0, total_bytes, 0x00020000);
    const unsigned seq_off = (unsigned)((((size_t)sb * T * F + sf) * LS_H + lc) * 4);   // bytes, + t*t_stride
    const unsigned t_stride = (unsigned)F * LS_H * 4u;
    f32x4 g4 = {1.f, 1.f, 1.f, 1.f}, b4 = {0.f, 0.f, 0.f, 0.f};
    if (LN) {
        g4 = *reinterpret_cast<const f32x4*>(ln_g + lc);
        b4 = *reinterpret_cast<const f32x4*>(ln_b + lc);
    }

    auto load_x = [&](int t) -> f32x4 {
        const unsigned off = (sv && t < t_hi) ? seq_off + (unsigned)t * t_stride : LS_OOB;
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, off, 0, 0));
    };
    auto norm_store = [&](f32x4 v, int buf) {
        if (LN) {
            // LayerNorm over the 64 channels = 16 lanes x 4, two-pass in registers
            const float mean = ls_row_sum((v[0] + v[1]) + (v[2] + v[3])) * (1.0f / 64.0f);
            f32x4 dlt = {v[0] - mean, v[1] - mean, v[2] - mean, v[3] - mean};
            const float q = ls_row_sum((dlt[0] * dlt[0] + dlt[1] * dlt[1]) + (dlt[2] * dlt[2] + dlt[3] * dlt[3]));
            const float rstd = 1.0f / sqrtf(q * (1.0f / 64.0f) + ln_eps);
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = dlt[j] * rstd * g4[j] + b4[j];
        }
        *reinterpret_cast<f32x4*>(&xs[buf][ls * LS_LD + lc]) = v;
    };
    // fragments of a 16 x 64 LDS tile: lane (m = ln, kk = lk) holds floats [16j + 4kk, +4), j = 0..3
    auto frags = [&](const float* tile, f32x4 (&a)[4]) {
        const float* arow = tile + ln * LS_LD + 4 * lk;
#pragma unroll
        for (int j = 0; j < 4; ++j) a[j] = *reinterpret_cast<const f32x4*>(arow + 16 * j);
    };
    // acc[g] += A[:, 16j..16j+15] . W[g]  for j in [J0, J1)   -- 16 MFMAs per j
    auto mma = [&](const f32x4 (&a)[4], const float (&w)[4][16], f32x4 (&acc)[4], auto j0, auto j1) {
#pragma unroll
        for (int j = decltype(j0)::value; j < decltype(j1)::value; ++j)
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j][s], w[g][4 * j + s], acc[g], 0, 0, 0);
    };
    using I0 = std::integral_constant<int, 0>;
    using I2 = std::integral_constant<int, 2>;
    using I4 = std::integral_constant<int, 4>;

    // prologue: h_{t_lo-1}, c_{t_lo-1} (zero at the start of the utterance), x_{t_lo} and x_{t_lo+1}
    // normalised in LDS, accx = b + W_x x_{t_lo}
    *reinterpret_cast<f32x4*>(&hs[0][ls * LS_LD + lc]) = __builtin_bit_cast(
        f32x4, __builtin_amdgcn_raw_buffer_load_b128(rh, (sv && t_lo > 0) ? seq_off + (unsigned)(t_lo - 1) * t_stride : LS_OOB, 0, 0));
    norm_store(load_x(t_lo), 0);
    norm_store(load_x(t_lo + 1), 1);
    const int u = wave * 16 + ln;
    float cst[4] = {0.f, 0.f, 0.f, 0.f};
    if (c_state && t_lo > 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (s0 + 4 * lk + r < S) cst[r] = c_state[(size_t)(s0 + 4 * lk + r) * LS_H + u];
    }
    __syncthreads();
    f32x4 accx[4], hf[4], xf[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) accx[g] = f32x4{bia[g], bia[g], bia[g], bia[g]};
    frags(xs[0], xf);
    mma(xf, wx, accx, I0{}, I4{});
    frags(hs[0], hf);
    // every wave has taken its x_{t_lo} fragments before any wave's first step overwrites xs[0] with x_{t_lo+2}
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    f32x4 xq = load_x(t_lo + 2), xr = load_x(t_lo + 3); // x_{t+2}, x_{t+3}: raw, in registers

    for (int t = t_lo; t < t_hi; ++t) {
        const int cur = (t - t_lo) & 1, nxt = cur ^ 1;
        const f32x4 xn = load_x(t + 4);                 // four steps ahead: two full steps of HBM latency cover

        // ---- H: recurrent half (critical path) + LayerNorm of x_{t+2} on the VALU
        f32x4 acc[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) acc[g] = accx[g];
        mma(hf, wh, acc, I0{}, I4{});
        norm_store(xq, cur);                            // x_{t+2} replaces x_t (consumed one step ago)
        frags(xs[nxt], xf);                             // x_{t+1}
        if (LN) {
            // spread the LayerNorm's ~90 VALU ops through the 64 MFMAs instead of behind them
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);   // MFMA
                __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);   // VALU
            }
        }
        __builtin_amdgcn_sched_barrier(0);

        // ---- X1: first half of the next step's input MFMAs.  The cell update of step t
        // (lane holds unit u for sequences 4*lk + r) is cut into 32 slices, one per MFMA
        // gap: a 16x16x4 MFMA holds the issue port 8 of its 32 cycles, a plain VALU op
        // costs 4, exp/rcp 8 (MI355X_MICROARCH cycle table), so <= 2 of each fit a gap.
        // n = 4*gate + r;  stage A: e = exp(-k x)   (k = 2 for the tanh gate)
        //                  stage B: s = 1/(1+e)     (tanh gate: 2s - 1)
        //                  stage C: c = f c + i g, e_c = exp(-2c);  stage D: h = o (2/(1+e_c) - 1)
#pragma unroll
        for (int g = 0; g < 4; ++g) accx[g] = f32x4{bia[g], bia[g], bia[g], bia[g]};
        float ev[16], hval[4];
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            {
                const int j = i >> 4, k = (i >> 2) & 3, g = i & 3;
                accx[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(xf[j][k], wx[g][4 * j + k], accx[g], 0, 0, 0);
            }
            if (i < 8) {
#pragma unroll
                for (int n = 2 * i; n < 2 * i + 2; ++n) {
                    const float v = acc[n >> 2][n & 3];
                    ev[n] = __expf((n >> 2) == 2 ? -2.0f * v : -v);
                }
            } else if (i < 16) {
#pragma unroll
                for (int n = 2 * (i - 8); n < 2 * (i - 8) + 2; ++n) {
                    const float sg = __builtin_amdgcn_rcpf(1.0f + ev[n]);
                    ev[n] = (n >> 2) == 2 ? fmaf(2.0f, sg, -1.0f) : sg;
                }
            } else if (i < 20) {
                const int r = i - 16;
                cst[r] = fmaf(ev[4 + r], cst[r], ev[r] * ev[8 + r]);
                hval[r] = __expf(-2.0f * cst[r]);
            } else if (i < 24) {
                const int r = i - 20;
                hval[r] = ev[12 + r] * fmaf(2.0f, __builtin_amdgcn_rcpf(1.0f + hval[r]), -1.0f);
            }
            __builtin_amdgcn_sched_barrier(0);
        }

        if (DUMP) {
            // training: the activated gates i, f, g, o and the cell state c_t of every (sequence, step), layout
            // gates[seq][t][5][64] -- what the reverse-time kernel (csrc/lstm_bwd.hip) needs; 64 B per 16 lanes
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int sq = s0 + 4 * lk + r;
                if (sq < S) {
                    float* gp = gates + (((size_t)sq * T + t) * 5) * LS_H + u;
#pragma unroll
                    for (int g = 0; g < 4; ++g) gp[g * LS_H] = ev[4 * g + r];
                    gp[4 * LS_H] = cst[r];
                }
            }
        }
        // ---- EX: exchange h_t through LDS
#pragma unroll
        for (int r = 0; r < 4; ++r) hs[nxt][(4 * lk + r) * LS_LD + u] = hval[r];
        // raw barrier: only the LDS writes have to land.  __syncthreads() would also wait
        // vmcnt(0), i.e. drain the x prefetch and the previous h store every step.
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        frags(hs[nxt], hf);
        f32x4 hv = {0.f, 0.f, 0.f, 0.f};
        hv = *reinterpret_cast<const f32x4*>(&hs[nxt][ls * LS_LD + lc]);
        __builtin_amdgcn_sched_barrier(0);

        // ---- X2: second half of the input MFMAs covers the barrier and the LDS latency
        mma(xf, wx, accx, I2{}, I4{});
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, hv), rh,
                                               sv ? seq_off + (unsigned)t * t_stride : LS_OOB, 0, 0);   // coalesced 4 KB
        xq = xr;
        xr = xn;
    }
    if (c_state) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (s0 + 4 * lk + r < S) c_state[(size_t)(s0 + 4 * lk + r) * LS_H + u] = cst[r];
    }
}

// ---------------------------------------------------------------------------------------------------
// Small-batch variant: 4*G sequences per workgroup on v_mfma_f32_4x4x1_16b_f32.
// The 16x16x4 kernel above needs 16 sequences per workgroup, so B*F <= 2048 sequences leave most CUs
// idle (one 4-s utterance: 11 workgroups) while its step time stays 128 x 32 MFMA cycles.  The 4x4x1
// form (16 independent 4x4 blocks: 4 sequences x 64 gate columns per instruction, measured 10 cycles
// each with >= 6 independent accumulation chains, tools/probe_mfma_4x4.hip) cuts the granule to 4
// sequences: a step is 128*G MFMAs of 10 cycles, and four times as many workgroups share the work
// (G = 1 is what the dispatcher uses: whole-network step at B = 1 / 4 / 8 / 12: 3.85 / 4.82 / 6.42 / 7.52 ms
// against 4.65 / 5.69 / 6.79 / 7.83 ms with the 16-sequence kernel; at B = 16 the latter wins, 9.10 vs 9.73).
//   lane l of wave w: hidden unit u = 16w + l/4, gate j = l%4 (i, f, g, o); B operand = that gate
//   column of [W_x | W_h] (128 VGPRs, stationary); A operand = the 4 sequences of a group, identical
//   in all 16 blocks (read from LDS as a broadcast); result VGPR s = sequence s of the group.
//   The four gates of a unit sit in one quad: quad_perm DPP broadcasts hand every lane i, f, g, o and
//   the quad updates the cell redundantly (lanes are free; lane j stores sequence j).
// Same x prefetch, LayerNorm, streaming window and state hand-off as lstm64_kernel.
template <bool LN, int G, bool DUMP = false>
__global__ __launch_bounds__(256) void lstm64_q_kernel(const float* __restrict__ x, const float* __restrict__ ln_g,
                                                       const float* __restrict__ ln_b, float ln_eps,
                                                       const float* __restrict__ wcat, const float* __restrict__ bias,
                                                       float* __restrict__ h_out, int T, int F, int S,
                                                       const int* __restrict__ t_pos, int t_count,
                                                       float* __restrict__ c_state, float* __restrict__ gates = nullptr) {
    constexpr int NSEQ = 4 * G;
    constexpr int NCH = G == 1 ? 8 : (G == 2 ? 4 : 2);       // accumulation chains per group (>= 6 in flight overall)
    __shared__ __attribute__((aligned(16))) float xs[2][NSEQ * LS_LD];
    __shared__ __attribute__((aligned(16))) float hs[2][NSEQ * LS_LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int gate = lane & 3, u = wave * 16 + (lane >> 2);
    const int s0 = blockIdx.x * NSEQ;
    const int t_lo = t_pos ? *t_pos : 0;
    const int t_hi = t_pos ? (t_lo + t_count < T ? t_lo + t_count : T) : T;

    // stationary weights: this lane's gate column of [W_x | W_h]
    float wx[64], wh[64];
    const int wrow = gate * LS_H + u;
    const float bia = bias[wrow];
#pragma unroll
    for (int k4 = 0; k4 < 16; ++k4) {
        const f32x4 vx = *reinterpret_cast<const f32x4*>(&wcat[(size_t)wrow * 128 + 4 * k4]);
        const f32x4 vh = *reinterpret_cast<const f32x4*>(&wcat[(size_t)wrow * 128 + 64 + 4 * k4]);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            wx[4 * k4 + kk] = vx[kk];
            wh[4 * k4 + kk] = vh[kk];
        }
    }
    // activation of this lane's gate: sigmoid(v) = 1/(1+exp(-v)), tanh(v) = 2/(1+exp(-2v)) - 1
    const float act_k = gate == 2 ? -2.0f : -1.0f, act_a = gate == 2 ? 2.0f : 1.0f, act_b = gate == 2 ? -1.0f : 0.0f;

    // loader role: thread -> (sequence ls, channels lc..lc+3); only the first NSEQ rows are live
    const int ls = tid >> 4, lc = (tid & 15) * 4;
    const int sg = s0 + ls;
    const bool sv = ls < NSEQ && sg < S;
    const int sb = sv ? sg / F : 0, sf = sv ? sg - sb * F : 0;
    const unsigned total_bytes = (unsigned)S * (unsigned)T * (LS_H * 4u);
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, total_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rh = __builtin_amdgcn_make_buffer_rsrc(h_out, 0, total_bytes, 0x00020000);
    const unsigned seq_off = (unsigned)((((size_t)sb * T * F + sf) * LS_H + lc) * 4);
    const unsigned t_stride = (unsigned)F * LS_H * 4u;
    f32x4 g4 = {1.f, 1.f, 1.f, 1.f}, b4 = {0.f, 0.f, 0.f, 0.f};
    if (LN) {
        g4 = *reinterpret_cast<const f32x4*>(ln_g + lc);
        b4 = *reinterpret_cast<const f32x4*>(ln_b + lc);
    }
    auto load_x = [&](int t) -> f32x4 {
        const unsigned off = (sv && t < t_hi) ? seq_off + (unsigned)t * t_stride : LS_OOB;
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, off, 0, 0));
    };
    auto norm_store = [&](f32x4 v, int buf) {
        if (LN) {
            const float mean = ls_row_sum((v[0] + v[1]) + (v[2] + v[3])) * (1.0f / 64.0f);
            f32x4 dlt = {v[0] - mean, v[1] - mean, v[2] - mean, v[3] - mean};
            const float q = ls_row_sum((dlt[0] * dlt[0] + dlt[1] * dlt[1]) + (dlt[2] * dlt[2] + dlt[3] * dlt[3]));
            const float rstd = 1.0f / sqrtf(q * (1.0f / 64.0f) + ln_eps);
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = dlt[j] * rstd * g4[j] + b4[j];
        }
        if (ls < NSEQ) *reinterpret_cast<f32x4*>(&xs[buf][ls * LS_LD + lc]) = v;
    };
    auto quad = [](float v, auto ctrl) {   // quad_perm broadcast of one lane of the quad
        return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), decltype(ctrl)::value, 0xF, 0xF, true));
    };

    // prologue: h_{t_lo-1}, c_{t_lo-1}, x_{t_lo} normalised in LDS; x_{t_lo+1}, x_{t_lo+2} raw in registers
    if (ls < NSEQ)
        *reinterpret_cast<f32x4*>(&hs[0][ls * LS_LD + lc]) = __builtin_bit_cast(
            f32x4, __builtin_amdgcn_raw_buffer_load_b128(rh, (sv && t_lo > 0) ? seq_off + (unsigned)(t_lo - 1) * t_stride : LS_OOB, 0, 0));
    norm_store(load_x(t_lo), 0);
    float cst[G][4];
#pragma unroll
    for (int q = 0; q < G; ++q)
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int sq = s0 + 4 * q + s;
            cst[q][s] = (c_state && t_lo > 0 && sq < S) ? c_state[(size_t)sq * LS_H + u] : 0.0f;
        }
    f32x4 xq = load_x(t_lo + 1), xr = load_x(t_lo + 2);
    __syncthreads();

    for (int t = t_lo; t < t_hi; ++t) {
        const int cur = (t - t_lo) & 1, nxt = cur ^ 1;
        const f32x4 xn = load_x(t + 3);
        // ---- gates: pre[q][s] = b + W_x x_t + W_h h_{t-1} for the 4 sequences of every group
        f32x4 acc[G][NCH];
#pragma unroll
        for (int q = 0; q < G; ++q)
#pragma unroll
            for (int c = 0; c < NCH; ++c) acc[q][c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int part = 0; part < 2; ++part) {
            const float* src = part == 0 ? xs[cur] : hs[cur];
#pragma unroll
            for (int k4 = 0; k4 < 16; ++k4) {
#pragma unroll
                for (int q = 0; q < G; ++q) {
                    const f32x4 a = *reinterpret_cast<const f32x4*>(&src[(4 * q + gate) * LS_LD + 4 * k4]);
                    constexpr int dummy = 0;
                    (void)dummy;
                    const int c = (part * 16 + k4) % NCH;
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk)
                        acc[q][c] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[kk], part == 0 ? wx[4 * k4 + kk] : wh[4 * k4 + kk],
                                                                       acc[q][c], 0, 0, 0);
                }
            }
        }
        if (G == 1) {
            // the compiler's own order kept one LDS read ahead of its four MFMAs (40 cycles of multiply per ~100 cycles of read
            // latency: the matrix pipe idled most of the step); six reads ahead instead
            __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);        // DS read
#pragma unroll
            for (int i = 0; i < 26; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);    // MFMA
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 24, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        norm_store(xq, nxt);                            // x_{t+1}: its buffer was last read one step ago
        // ---- activations, cell update (redundant in the quad), h_t -> LDS
#pragma unroll
        for (int q = 0; q < G; ++q) {
            f32x4 pre = acc[q][0];
#pragma unroll
            for (int c = 1; c < NCH; ++c) pre += acc[q][c];
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const float sgm = __builtin_amdgcn_rcpf(1.0f + __expf(act_k * (pre[s] + bia)));
                const float a = fmaf(act_a, sgm, act_b);
                const float gi = quad(a, std::integral_constant<int, 0x00>{}), gf = quad(a, std::integral_constant<int, 0x55>{});
                const float gg = quad(a, std::integral_constant<int, 0xAA>{}), go = quad(a, std::integral_constant<int, 0xFF>{});
                cst[q][s] = fmaf(gf, cst[q][s], gi * gg);
                const float hv = go * ls_tanh(cst[q][s]);
                if (s == gate) hs[nxt][(4 * q + s) * LS_LD + u] = hv;
                if (DUMP) {
                    // training: gates[seq][t][5][64] = i, f, g, o, c (the layout csrc/lstm_bwd.hip reads): every lane its own
                    // activated gate, lane s of the quad the cell state of sequence s
                    const int sq = s0 + 4 * q + s;
                    if (sq < S) {
                        float* gp = gates + (((size_t)sq * T + t) * 5) * LS_H + u;
                        gp[gate * LS_H] = a;
                        if (s == gate) gp[4 * LS_H] = cst[q][s];
                    }
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (ls < NSEQ) {
            const f32x4 hv4 = *reinterpret_cast<const f32x4*>(&hs[nxt][ls * LS_LD + lc]);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, hv4), rh,
                                                   sv ? seq_off + (unsigned)t * t_stride : LS_OOB, 0, 0);
        }
        xq = xr;
        xr = xn;
    }
    if (c_state && (lane & 3) == 0) {
#pragma unroll
        for (int q = 0; q < G; ++q)
#pragma unroll
            for (int s = 0; s < 4; ++s)
                if (s0 + 4 * q + s < S) c_state[(size_t)(s0 + 4 * q + s) * LS_H + u] = cst[q][s];
    }
}

template <int G>
static void lstm64_q_launch(bool ln, int grid, hipStream_t st, const float* x, const float* ln_g, const float* ln_b,
                            float ln_eps, const float* wcat, const float* bias, float* h_out, int T, int F, int S,
                            const int* t_pos, int t_count, float* cs) {
    if (ln)
        hipLaunchKernelGGL((lstm64_q_kernel<true, G>), dim3(grid), dim3(256), 0, st, x, ln_g, ln_b, ln_eps, wcat, bias,
                           h_out, T, F, S, t_pos, t_count, cs);
    else
        hipLaunchKernelGGL((lstm64_q_kernel<false, G>), dim3(grid), dim3(256), 0, st, x, ln_g, ln_b, ln_eps, wcat, bias,
                           h_out, T, F, S, t_pos, t_count, cs);
}

int eab_lstm64_h3_launch(const float* x, const float* ln_g, const float* ln_b, float ln_eps, const float* wcat,
                         const float* bias, float* h_out, int T, int F, int S, int precision, const int* t_pos, int t_count,
                         float* c_state, hipStream_t stream);   // lstm_h3.hip

extern "C" int eab_lstm64_f32(const float* x, const float* ln_g, const float* ln_b, float ln_eps, const float* wcat,
                              const float* bias, float* h_out, int B, int T, int F, eab_stream_t stream) {
    return eab_lstm64_prec_f32(x, ln_g, ln_b, ln_eps, wcat, bias, h_out, B, T, F, EAB_PREC_F32, stream);
}

extern "C" int eab_lstm64_prec_f32(const float* x, const float* ln_g, const float* ln_b, float ln_eps,
                                   const float* wcat, const float* bias, float* h_out, int B, int T, int F,
                                   int precision, eab_stream_t stream) {
    return eab_lstm64_stream_f32(x, ln_g, ln_b, ln_eps, wcat, bias, h_out, nullptr, B, T, F, precision,
                                 eab_time_window{nullptr, 0}, stream);
}

extern "C" int eab_lstm64_stream_f32(const float* x, const float* ln_g, const float* ln_b, float ln_eps,
                                     const float* wcat, const float* bias, float* h_out, float* c_state, int B, int T,
                                     int F, int precision, eab_time_window win, eab_stream_t stream) {
    EAB_CHECK_ARG(x && wcat && bias && h_out && B > 0 && T > 0 && F > 0);
    EAB_CHECK_ARG(win.pos == nullptr || (win.count > 0 && c_state));
    EAB_CHECK_ARG(precision == EAB_PREC_F32 || precision == EAB_PREC_F16X3 || precision == EAB_PREC_BF16);
    EAB_CHECK_ARG((ln_g == nullptr) == (ln_b == nullptr));
    const long long S = (long long)B * F;
    EAB_CHECK_ARG(S * T * LS_H * 4 < (1ll << 31));          // 31-bit byte offsets in the buffer descriptors
    float* cs = win.pos ? c_state : nullptr;
    if (precision != EAB_PREC_F32)      // streaming state (h in h_out, c in c_state) is carried in fp32 in every mode
        return eab_lstm64_h3_launch(x, ln_g, ln_b, ln_eps, wcat, bias, h_out, T, F, (int)S, precision, win.pos, win.count, cs,
                                    eab_stream(stream));
    // up to 2048 sequences (12 four-second utterances): 4-sequence workgroups on the 4x4x1 MFMA (measured faster
    // than the 16-sequence kernel up to there, even at two workgroups per CU; larger groups never won)
    if (S <= 2048) {
        lstm64_q_launch<1>(ln_g != nullptr, (int)((S + 3) / 4), eab_stream(stream), x, ln_g, ln_b, ln_eps, wcat, bias, h_out,
                           T, F, (int)S, win.pos, win.count, cs);
        EAB_RETURN_LAUNCH_STATUS();
    }
    const int grid = (int)((S + LS_SEQ - 1) / LS_SEQ);
    if (ln_g)
        hipLaunchKernelGGL((lstm64_kernel<true, false>), dim3(grid), dim3(256), 0, eab_stream(stream), x, ln_g, ln_b, ln_eps,
                           wcat, bias, h_out, T, F, (int)S, win.pos, win.count, cs, nullptr);
    else
        hipLaunchKernelGGL((lstm64_kernel<false, false>), dim3(grid), dim3(256), 0, eab_stream(stream), x, ln_g, ln_b, ln_eps,
                           wcat, bias, h_out, T, F, (int)S, win.pos, win.count, cs, nullptr);
    EAB_RETURN_LAUNCH_STATUS();
}

// Training forward: the same layer (no LayerNorm inside: the training program materialises it, its output is an
// operand of the weight gradient) that also stores the activated gates and cell states, gates [B*F][T][5][64].
int eab_lstm64_bf_train_launch(const float* x, const float* wcat, const float* bias, float* h_out, float* gates, int T, int F, int S,
                               hipStream_t stream);     // lstm_h3.hip

extern "C" int eab_lstm64_train_fwd_f32(const float* x, const float* wcat, const float* bias, float* h_out, float* gates, int B,
                                        int T, int F, eab_stream_t stream) {
    return eab_lstm64_train_fwd_prec_f32(x, wcat, bias, h_out, gates, B, T, F, EAB_PREC_F32, stream);
}

// precision EAB_PREC_BF16: x_t, h_{t-1} and the weights rounded to bf16, one product on the 16-bit matrix cores, fp32
// accumulation, activations, cell state and stored gates (the arithmetic of torch.autocast(bfloat16) on nn.LSTM)
extern "C" int eab_lstm64_train_fwd_prec_f32(const float* x, const float* wcat, const float* bias, float* h_out, float* gates, int B,
                                             int T, int F, int precision, eab_stream_t stream) {
    EAB_CHECK_ARG(x && wcat && bias && h_out && gates && B > 0 && T > 0 && F > 0);
    EAB_CHECK_ARG(precision == EAB_PREC_F32 || precision == EAB_PREC_BF16);
    const long long S = (long long)B * F;
    EAB_CHECK_ARG(S * T * LS_H * 4 < (1ll << 31));
    if (precision == EAB_PREC_BF16) return eab_lstm64_bf_train_launch(x, wcat, bias, h_out, gates, T, F, (int)S, eab_stream(stream));
    if (S <= 2048) {        // same split as the inference dispatcher: 4-sequence workgroups while they fit the chip in two rounds
        hipLaunchKernelGGL((lstm64_q_kernel<false, 1, true>), dim3((unsigned)((S + 3) / 4)), dim3(256), 0, eab_stream(stream), x,
                           nullptr, nullptr, 0.0f, wcat, bias, h_out, T, F, (int)S, nullptr, 0, nullptr, gates);
        EAB_RETURN_LAUNCH_STATUS();
    }
    const int grid = (int)((S + LS_SEQ - 1) / LS_SEQ);
    hipLaunchKernelGGL((lstm64_kernel<false, true>), dim3(grid), dim3(256), 0, eab_stream(stream), x, nullptr, nullptr, 0.0f, wcat,
                       bias, h_out, T, F, (int)S, nullptr, 0, nullptr, gates);
    EAB_RETURN_LAUNCH_STATUS();
}

extern "C" int eab_lstm64_bf16(const float* x, const float* ln_g, const float* ln_b, float ln_eps, const float* wcat,
                               const float* bias, float* h_out, int B, int T, int F, eab_stream_t stream) {
    return eab_lstm64_prec_f32(x, ln_g, ln_b, ln_eps, wcat, bias, h_out, B, T, F, EAB_PREC_BF16, stream);
}
