// K10+K11 in reduced-precision arithmetic: LayerNorm(64) + one LSTM(64->64) layer on the 16-bit matrix cores.
//   EAB_PREC_F16X3: v_mfma_f32_16x16x32_f16, every fp32 operand split x = hi + lo and every product taken as
//                   lo*hi + hi*lo + hi*hi with fp32 accumulation (include/eabnet_hip.h);
//   EAB_PREC_BF16 : v_mfma_f32_16x16x32_bf16, operands rounded to bf16, ONE product, fp32 accumulation and fp32
//                   cell state / activations (the arithmetic of torch.autocast(bfloat16) on nn.LSTM).
// Both also run the streaming window (eab_time_window): state h_{t-1} from h_out, c from c_state.
// Reference: LSTM_BF.forward, EaBNet.py:608-611.
//
// Same ownership as the fp32 kernel (csrc/lstm.hip): one workgroup = 16 sequences for all T
// steps, wave w = hidden units 16w..16w+15 (i,f,g,o columns of the same units), weights
// register-resident (128 VGPRs of fp16 hi/lo fragments).  What changes is the bound: the 48
// MFMAs of a step take ~770 matrix-pipe cycles instead of 4096, and f16 MFMAs co-execute with
// VALU, so a step is bound by the recurrence chain
//     H-MFMAs (24) -> cell update (exp/rcp) -> h_t hi/lo -> LDS -> barrier -> fragments
// and the input half (24 MFMAs of step t+1) is issued behind the barrier where it covers the
// LDS round trip.
#include "common.h"
#include <type_traits>

#define LH_SEQ 16
#define LH_ROW 272          // bytes: 2 k-blocks x (32 hi + 32 lo halves) + 16 pad (odd 16-byte-slot stride)
#define LH_OOB 0x80000000u

typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef __fp16 h16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float lh_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float lh_tanh(float x) { return fmaf(2.0f, lh_sigmoid(2.0f * x), -1.0f); }

__device__ __forceinline__ float lh_row_sum(float v) {     // sum over the 16 lanes of a DPP row
    auto dpp = [](float x, auto ctrl) {
        return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), decltype(ctrl)::value, 0xF, 0xF, true));
    };
    v += dpp(v, std::integral_constant<int, 0xB1>{});
    v += dpp(v, std::integral_constant<int, 0x4E>{});
    v += dpp(v, std::integral_constant<int, 0x141>{});
    v += dpp(v, std::integral_constant<int, 0x140>{});
    return v;
}

template <bool BF>
__device__ __forceinline__ void lh_split2(float x0, float x1, unsigned& hi, unsigned& lo) {
    if (BF) {
        const bf16x2 h = {(__bf16)x0, (__bf16)x1};
        const bf16x2 l = {(__bf16)(x0 - (float)h[0]), (__bf16)(x1 - (float)h[1])};
        hi = __builtin_bit_cast(unsigned, h);
        lo = __builtin_bit_cast(unsigned, l);
    } else {
        const h16x2 h = __builtin_amdgcn_cvt_pkrtz(x0, x1);
        const h16x2 l = __builtin_amdgcn_cvt_pkrtz(x0 - (float)h[0], x1 - (float)h[1]);
        hi = __builtin_bit_cast(unsigned, h);
        lo = __builtin_bit_cast(unsigned, l);
    }
}

// 16-bit pair (hi, lo) -> fp32 hi + lo
template <bool BF>
__device__ __forceinline__ float lh_join(unsigned short hi, unsigned short lo) {
    if (BF) return __builtin_bit_cast(float, (unsigned)hi << 16) + __builtin_bit_cast(float, (unsigned)lo << 16);
    return (float)__builtin_bit_cast(_Float16, hi) + (float)__builtin_bit_cast(_Float16, lo);
}

// NS = 8 / 4 (bf16 training): 8 or 4 sequences per workgroup in tile rows 4q + {0, 1} / 4q, so that every lane still owns
// sequences (two / one instead of four: half / a quarter of the exp / rcp chain of the cell update per step, two / four times the
// workgroups); the other rows are padding the MFMAs carry along.  Per layer at 966 sequences: NS 16 / 8 / 4 = 638 / 534 / 454 us,
// at 2,576: 594 / 570 / 542.
template <bool LN, bool BF, bool DUMP = false, int NS = LH_SEQ>
__global__ __launch_bounds__(256) void lstm64_h3_kernel(const float* __restrict__ x, const float* __restrict__ ln_g,
                                                        const float* __restrict__ ln_b, float ln_eps,
                                                        const float* __restrict__ wcat, const float* __restrict__ bias,
                                                        float* __restrict__ h_out, int T, int F, int S,
                                                        const int* __restrict__ t_pos, int t_count, float* __restrict__ c_state,
                                                        float* __restrict__ gates = nullptr) {
    using v8 = std::conditional_t<BF, bf16x8, h16x8>;
    using e16 = std::conditional_t<BF, __bf16, _Float16>;
    __shared__ __attribute__((aligned(16))) char xs[2][LH_SEQ * LH_ROW];
    __shared__ __attribute__((aligned(16))) char hs[2][LH_SEQ * LH_ROW];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ln = lane & 15, lk = lane >> 4;
    static_assert(NS == LH_SEQ || ((NS == 8 || NS == 4) && BF), "the 8- and 4-sequence forms exist for the bf16 kernel only");
    constexpr int NR = NS / 4;                                // sequences per lane
    // tile row -> sequence of this workgroup (or -1: padding)
    auto row_seq = [](int row) { return NS == LH_SEQ ? row : ((row & 3) < NR ? (row >> 2) * NR + (row & 3) : -1); };
    const int s0 = blockIdx.x * NS;
    // streaming (eab_time_window): steps [t_lo, t_hi) only
    const int t_lo = t_pos ? *t_pos : 0;
    const int t_hi = t_pos ? (t_lo + t_count < T ? t_lo + t_count : T) : T;

    // ---- stationary weights as 16-bit hi/lo B fragments (bf16 mode: hi only is multiplied):
    //      w?[g][kb] = W[g*64 + 16w + ln][(x:0 | h:64) + 32*kb + 8*lk + j], j = 0..7
    v8 wxh[4][2], wxl[4][2], whh[4][2], whl[4][2];
    float bia[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int row = g * 64 + wave * 16 + ln;
        bia[g] = bias[row];
#pragma unroll
        for (int half = 0; half < 2; ++half)
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                const float* p = &wcat[(size_t)row * 128 + half * 64 + 32 * kb + 8 * lk];
                const f32x4 v0 = *reinterpret_cast<const f32x4*>(p), v1 = *reinterpret_cast<const f32x4*>(p + 4);
                v8 hi, lo;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float w = j < 4 ? v0[j] : v1[j - 4];
                    hi[j] = (e16)w;
                    lo[j] = (e16)(w - (float)hi[j]);
                }
                if (half == 0) { wxh[g][kb] = hi; wxl[g][kb] = lo; } else { whh[g][kb] = hi; whl[g][kb] = lo; }
            }
    }

    // ---- loader role: thread -> (sequence ls, channels lc..lc+3)
    const int ls = tid >> 4, lc = (tid & 15) * 4;
    const int sg = s0 + row_seq(ls);
    const bool sv = row_seq(ls) >= 0 && sg < S;
    const int sb = sv ? sg / F : 0, sf = sv ? sg - sb * F : 0;
    const unsigned total_bytes = (unsigned)S * (unsigned)T * 256u;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, total_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rh = __builtin_amdgcn_make_buffer_rsrc(h_out, 0, total_bytes, 0x00020000);
    const unsigned seq_off = (unsigned)((((size_t)sb * T * F + sf) * 64 + lc) * 4);
    const unsigned t_stride = (unsigned)F * 256u;
    f32x4 g4 = {1.f, 1.f, 1.f, 1.f}, b4 = {0.f, 0.f, 0.f, 0.f};
    if (LN) {
        g4 = *reinterpret_cast<const f32x4*>(ln_g + lc);
        b4 = *reinterpret_cast<const f32x4*>(ln_b + lc);
    }
    // byte offset of channel lc inside a row: k-block lc/32, then hi at +0 / lo at +64
    const int lrow = ls * LH_ROW + (lc >> 5) * 128 + (lc & 31) * 2;

    auto load_x = [&](int t) -> f32x4 {
        const unsigned off = (sv && t < t_hi) ? seq_off + (unsigned)t * t_stride : LH_OOB;
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, off, 0, 0));
    };
    auto norm_store = [&](f32x4 v, int buf) {
        if (LN) {
            const float mean = lh_row_sum((v[0] + v[1]) + (v[2] + v[3])) * (1.0f / 64.0f);
            f32x4 dlt = {v[0] - mean, v[1] - mean, v[2] - mean, v[3] - mean};
            const float q = lh_row_sum((dlt[0] * dlt[0] + dlt[1] * dlt[1]) + (dlt[2] * dlt[2] + dlt[3] * dlt[3]));
            const float rstd = 1.0f / sqrtf(q * (1.0f / 64.0f) + ln_eps);
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = dlt[j] * rstd * g4[j] + b4[j];
        }
        unsigned h01, l01, h23, l23;
        lh_split2<BF>(v[0], v[1], h01, l01);
        lh_split2<BF>(v[2], v[3], h23, l23);
        *reinterpret_cast<uint2*>(&xs[buf][lrow]) = make_uint2(h01, h23);
        if (!BF) *reinterpret_cast<uint2*>(&xs[buf][lrow + 64]) = make_uint2(l01, l23);
    };
    // A fragments of a 16 x 64 tile: lane (m = ln, kq = lk) holds k = 32*kb + 8*kq + j
    auto frags = [&](const char* tile, v8 (&ah)[2], v8 (&al)[2]) {
        const char* p = tile + ln * LH_ROW + lk * 16;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            ah[kb] = *reinterpret_cast<const v8*>(p + kb * 128);
            if (!BF) al[kb] = *reinterpret_cast<const v8*>(p + kb * 128 + 64);
        }
    };
    auto mma = [&](const v8 (&ah)[2], const v8 (&al)[2], const v8 (&wh)[4][2], const v8 (&wl)[4][2], f32x4 (&acc)[4]) {
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                if constexpr (BF) {
                    acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[kb], wh[g][kb], acc[g], 0, 0, 0);
                } else {
                    acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[kb], wh[g][kb], acc[g], 0, 0, 0);
                    acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[kb], wl[g][kb], acc[g], 0, 0, 0);
                    acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[kb], wh[g][kb], acc[g], 0, 0, 0);
                }
            }
    };

    // prologue: h_{t_lo-1} (0 at the start of the utterance; hi and lo), c_{t_lo-1}, x_{t_lo} / x_{t_lo+1} in LDS,
    // accx = b + W_x x_{t_lo}
    for (int e = tid; e < 2 * LH_SEQ * LH_ROW / 4; e += 256) reinterpret_cast<unsigned*>(hs[0])[e] = 0u;     // (both buffers)
    __syncthreads();
    if (t_lo > 0) {
        const f32x4 hp = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                                       rh, sv ? seq_off + (unsigned)(t_lo - 1) * t_stride : LH_OOB, 0, 0));
        unsigned h01, l01, h23, l23;
        lh_split2<BF>(hp[0], hp[1], h01, l01);
        lh_split2<BF>(hp[2], hp[3], h23, l23);
        *reinterpret_cast<uint2*>(&hs[0][lrow]) = make_uint2(h01, h23);
        *reinterpret_cast<uint2*>(&hs[0][lrow + 64]) = make_uint2(l01, l23);
    }
    norm_store(load_x(t_lo), 0);
    norm_store(load_x(t_lo + 1), 1);
    const int uq = wave * 16 + ln;
    float cst[4] = {0.f, 0.f, 0.f, 0.f};
    if (c_state && t_lo > 0) {
#pragma unroll
        for (int r = 0; r < NR; ++r)
            if (s0 + row_seq(4 * lk + r) < S) cst[r] = c_state[(size_t)(s0 + row_seq(4 * lk + r)) * 64 + uq];
    }
    __syncthreads();
    f32x4 accx[4];
    v8 xh[2], xl[2], hh[2], hl[2];
#pragma unroll
    for (int g = 0; g < 4; ++g) accx[g] = f32x4{bia[g], bia[g], bia[g], bia[g]};
    frags(xs[0], xh, xl);
    mma(xh, xl, wxh, wxl, accx);
    frags(hs[0], hh, hl);
    // every wave has taken its x_0 fragments before any wave's first step overwrites xs[0] with x_2
    // (a wave that leads by the 24 recurrent MFMAs of step 0 would otherwise clobber rows a slower wave
    // has not read yet: a cross-wave write-after-read race that shows when other kernels share the CU)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    f32x4 xq = load_x(t_lo + 2), xr = load_x(t_lo + 3);

    const int u = wave * 16 + ln;
    const int hcol = (u >> 5) * 128 + (u & 31) * 2;         // byte offset of unit u inside a row (hi; lo at +64)
    // bf16 mode writes h_t to HBM straight from the registers (exact fp32: a streamed chunk then restarts from the
    // very value the offline run rounded): byte offsets of (sequence 4*lk + r, unit u)
    unsigned hdir[4];
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const int sq = s0 + row_seq(4 * lk + r);
        const int b = sq < S ? sq / F : 0, f = sq < S ? sq - b * F : 0;
        hdir[r] = sq < S ? (unsigned)((((size_t)b * T * F + f) * 64 + u) * 4) : LH_OOB;
    }
    // training (DUMP): gates[seq][t][5][64] = activated i, f, g, o and the cell state (the layout csrc/lstm_bwd.hip reads)
    float* gdump[4];
    if (DUMP) {
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const int sq = s0 + row_seq(4 * lk + r);
            gdump[r] = sq < S ? gates + ((size_t)sq * T * 5) * 64 + u : nullptr;
        }
    }
    for (int t = t_lo; t < t_hi; ++t) {
        const int cur = (t - t_lo) & 1, nxt = cur ^ 1;
        const f32x4 xn = load_x(t + 4);

        // ---- recurrence: acc = accx + W_h h_{t-1}, then the cell update
        f32x4 acc[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) acc[g] = accx[g];
        mma(hh, hl, whh, whl, acc);
        // x_{t+1} fragments are taken BEFORE this step's barrier (as csrc/lstm.hip does): xs[nxt] is overwritten
        // with x_{t+3} by the next step's norm_store, which a wave that runs ahead reaches without passing
        // another barrier -- reading it behind the barrier was a cross-wave write-after-read race
        frags(xs[nxt], xh, xl);
        norm_store(xq, cur);                            // x_{t+2} replaces x_t (its fragments were read one barrier ago)
        __builtin_amdgcn_sched_barrier(0);
        char* hrow = &hs[nxt][hcol];
#pragma unroll
        for (int r = 0; r < NR; ++r) {                  // lane holds unit u for the sequences of tile rows 4*lk + r
            const float ig = lh_sigmoid(acc[0][r]);
            const float fg = lh_sigmoid(acc[1][r]);
            const float gg = lh_tanh(acc[2][r]);
            const float og = lh_sigmoid(acc[3][r]);
            cst[r] = fmaf(fg, cst[r], ig * gg);
            const float h = og * lh_tanh(cst[r]);
            if (DUMP && gdump[r]) {
                float* gp = gdump[r] + (size_t)t * 320;
                gp[0] = ig;
                gp[64] = fg;
                gp[128] = gg;
                gp[192] = og;
                gp[256] = cst[r];
            }
            const e16 hi = (e16)h;
            *reinterpret_cast<e16*>(hrow + (4 * lk + r) * LH_ROW) = hi;
            if (BF) {
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, h), rh,
                                                      hdir[r] != LH_OOB ? hdir[r] + (unsigned)t * t_stride : LH_OOB, 0, 0);
            } else {
                const e16 lo = (e16)(h - (float)hi);
                *reinterpret_cast<e16*>(hrow + (4 * lk + r) * LH_ROW + 64) = lo;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // LDS only: keep the x prefetch in flight
        frags(hs[nxt], hh, hl);
        // coalesced write-back of h_t (hi + lo reproduces h to 2^-21)
        const uint2 ph = *reinterpret_cast<const uint2*>(&hs[nxt][lrow]);
        const uint2 pl = *reinterpret_cast<const uint2*>(&hs[nxt][lrow + 64]);
        __builtin_amdgcn_sched_barrier(0);

        // ---- input half of the next step: independent of the recurrence, covers the LDS round trip
#pragma unroll
        for (int g = 0; g < 4; ++g) accx[g] = f32x4{bia[g], bia[g], bia[g], bia[g]};
        mma(xh, xl, wxh, wxl, accx);
        if (!BF) {
            const f32x4 hv = {lh_join<BF>(ph.x & 0xFFFF, pl.x & 0xFFFF), lh_join<BF>(ph.x >> 16, pl.x >> 16),
                              lh_join<BF>(ph.y & 0xFFFF, pl.y & 0xFFFF), lh_join<BF>(ph.y >> 16, pl.y >> 16)};
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, hv), rh,
                                                   sv ? seq_off + (unsigned)t * t_stride : LH_OOB, 0, 0);
        }
        xq = xr;
        xr = xn;
    }
    if (c_state) {
#pragma unroll
        for (int r = 0; r < NR; ++r)
            if (s0 + row_seq(4 * lk + r) < S) c_state[(size_t)(s0 + row_seq(4 * lk + r)) * 64 + u] = cst[r];
    }
}

// training forward in bf16 (eab_lstm64_train_fwd_prec_f32, csrc/lstm.hip): the bf16 kernel that also stores the gates
int eab_lstm64_bf_train_launch(const float* x, const float* wcat, const float* bias, float* h_out, float* gates, int T, int F, int S,
                               hipStream_t stream) {
    // up to 4096 sequences (the training batch of configs[3] is 966): 4 sequences per workgroup -- the step is the cell update's
    // exp / rcp chain per lane, not the 16 MFMAs, and four times the workgroups quarter it (measured up to 2,576 sequences)
    if (S <= 4096)
        hipLaunchKernelGGL((lstm64_h3_kernel<false, true, true, 4>), dim3((S + 3) / 4), dim3(256), 0, stream, x, nullptr, nullptr, 0.0f,
                           wcat, bias, h_out, T, F, S, nullptr, 0, nullptr, gates);
    else
        hipLaunchKernelGGL((lstm64_h3_kernel<false, true, true>), dim3((S + LH_SEQ - 1) / LH_SEQ), dim3(256), 0, stream, x, nullptr,
                           nullptr, 0.0f, wcat, bias, h_out, T, F, S, nullptr, 0, nullptr, gates);
    EAB_RETURN_LAUNCH_STATUS();
}

// dispatcher shared with csrc/lstm.hip
int eab_lstm64_h3_launch(const float* x, const float* ln_g, const float* ln_b, float ln_eps, const float* wcat,
                         const float* bias, float* h_out, int T, int F, int S, int precision, const int* t_pos, int t_count,
                         float* c_state, hipStream_t stream) {
    const int grid = (S + LH_SEQ - 1) / LH_SEQ;
#define LH_GO(LN_, BF_)                                                                                                     \
    hipLaunchKernelGGL((lstm64_h3_kernel<LN_, BF_>), dim3(grid), dim3(256), 0, stream, x, ln_g, ln_b, ln_eps, wcat, bias, h_out, \
                       T, F, S, t_pos, t_count, c_state)
    if (precision == EAB_PREC_BF16) {
        if (ln_g) LH_GO(true, true); else LH_GO(false, true);
    } else {
        if (ln_g) LH_GO(true, false); else LH_GO(false, false);
    }
#undef LH_GO
    EAB_RETURN_LAUNCH_STATUS();
}
