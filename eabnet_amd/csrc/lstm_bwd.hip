// Backward through time of one LSTM(64->64) layer (what autograd does for nn.LSTM in LSTM_BF, EaBNet.py:610-611,
// under train_distributed.py:228), persistent over the reversed time axis.
//
// Same ownership as the forward kernel (csrc/lstm.hip): one workgroup = 16 sequences for all T steps, wave w =
// hidden units 16w..16w+15; lane (ln = lane&15, lk = lane>>4) holds unit u = 16w+ln of sequences 4*lk + r.
//   per step t = T-1 .. 0:
//     dh  = dh_out[t] + dh_rec                         (upstream gradient + recurrence)
//     do' = dh * tanh(c_t) * o(1-o)
//     dc  = dc_carry + dh * o * (1 - tanh(c_t)^2)
//     di' = dc * g * i(1-i),  dg' = dc * i * (1-g^2),  df' = dc * c_{t-1} * f(1-f),  dc_carry = dc * f
//     dgates_t (16 x 256, gate-major) -> LDS -> HBM [B][T][F][256] (the operand of dx = dgates W_ih, dW = dgates^T [x | h_{t-1}])
//     dh_rec  = dgates_t . W_hh  as 64 v_mfma_f32_16x16x4_f32 per wave (K = 256; W_hh^T stationary in 64 VGPRs)
// The activated gates and cell states come from the training forward (eab_lstm64_train_fwd_f32, gates [S][T][5][64]).
// Bound: fp32 matrix pipe (64 MFMAs x 32 cycles per step), half of the forward's.
#include "common.h"

#define LB_H 64
#define LB_SEQ 16
#define LB_LD (4 * LB_H + 4)      // dgates tile row: 256 floats + pad (odd 16-byte-slot stride)
#define LB_OOB 0x80000000u
#define LB_PF 3                    // steps of operands in flight in the 16-sequence reverse-time kernel (28 registers per step)
// ... and in the 4-sequence kernel (7 registers per step).  Measured at 966 sequences, us per layer: 1 step of lead 870, 2: 997,
// 3: 644, 4: 633, 5: 552, 8: 533, 12: 1843 (84 loads exceed the 6-bit vmcnt).  The latency to cover is ~3.5 us, far above an HBM
// read: on gfx9 loads and the dgates stores of earlier steps retire in issue order, so a load waits for the stores' acknowledgements.
#define LBQ_PF 5

#define LBB_ROW (4 * LB_H * 2 + 16) // the same tile as bf16: 512 bytes + pad (odd 16-byte-slot stride)

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 lb_bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ float lb_tanh(float x) { return fmaf(2.0f, __builtin_amdgcn_rcpf(1.0f + __expf(-2.0f * x)), -1.0f); }

// BF (the bf16 training programs): dgates_t and W_hh rounded to bf16, dh_rec on 8 v_mfma_f32_16x16x32_bf16 per wave instead of
// 64 fp32 MFMAs of 32 cycles -- the recurrent product torch.autocast(bfloat16) runs for nn.LSTM's backward; the elementwise
// chain, the carries and the dgates written for the weight / input gradients stay fp32.
template <bool BF>
__global__ __launch_bounds__(256) void lstm64_bwd_kernel(const float* __restrict__ gates, const float* __restrict__ dh_out,
                                                         const float* __restrict__ wcat, float* __restrict__ dgates, int T, int F,
                                                         int S) {
    __shared__ __attribute__((aligned(16))) float dg[2][LB_SEQ * LB_LD];
    __shared__ __attribute__((aligned(16))) char dgb[BF ? 2 : 1][BF ? LB_SEQ * LBB_ROW : 16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ln = lane & 15, lk = lane >> 4;
    const int s0 = blockIdx.x * LB_SEQ;
    const int u = wave * 16 + ln;

    // stationary operand: whT[4j+s] = W_hh[k = 16j + 4lk + s][u] = wcat[k][64 + u]
    // (BF: B fragments of the 16x16x32 form: whb[kb][j] = W_hh[k = 32kb + 8lk + j][u])
    float whT[BF ? 1 : 64];
    lb_bf16x8 whb[BF ? 8 : 1];
    if constexpr (BF) {
#pragma unroll
        for (int kb = 0; kb < 8; ++kb)
#pragma unroll
            for (int j = 0; j < 8; ++j) whb[kb][j] = (__bf16)wcat[(size_t)(32 * kb + 8 * lk + j) * 128 + 64 + u];
    } else {
#pragma unroll
        for (int j = 0; j < 16; ++j)
#pragma unroll
            for (int s = 0; s < 4; ++s) whT[4 * j + s] = wcat[(size_t)(16 * j + 4 * lk + s) * 128 + 64 + u];
    }

    // element role: unit u of sequences sq[r] = s0 + 4*lk + r
    size_t goff[4];       // float offset of gates[sq][0][0][u]
    size_t hoff[4];       // float offset of dh_out[b][0][f][u]
    bool ok[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int sq = s0 + 4 * lk + r;
        ok[r] = sq < S;
        const int b = ok[r] ? sq / F : 0, f = ok[r] ? sq - b * F : 0;
        goff[r] = ((size_t)(ok[r] ? sq : 0) * T * 5) * LB_H + u;
        hoff[r] = (((size_t)b * T) * F + f) * LB_H + u;
    }
    const size_t g_t = (size_t)5 * LB_H, h_t = (size_t)F * LB_H;         // strides per time step

    // writer role: thread -> sequence ws = tid>>4, 16 floats starting at column (tid&15)*16 of the 256-wide row
    const int ws = tid >> 4, wc = (tid & 15) * 16;
    const int wsq = s0 + ws;
    const bool wok = wsq < S;
    const int wb = wok ? wsq / F : 0, wf = wok ? wsq - wb * F : 0;
    const size_t dg_base = (((size_t)wb * T) * F + wf) * (4 * LB_H) + wc;
    const size_t dg_t = (size_t)F * 4 * LB_H;

    float dhr[4] = {0.f, 0.f, 0.f, 0.f}, dcc[4] = {0.f, 0.f, 0.f, 0.f};
    // operands of the next LB_PF steps in registers: every step touches new cache lines of gates / dh_out, and one step of
    // lead (rounds 2-3) made the step time the memory latency
    float pf[LB_PF][7][4];
    auto load = [&](int t, float (&q)[7][4]) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const bool v = ok[r] && t >= 0;
            const float* gp = gates + goff[r] + (size_t)(t >= 0 ? t : 0) * g_t;
            q[0][r] = v ? gp[0] : 0.f;
            q[1][r] = v ? gp[LB_H] : 0.f;
            q[2][r] = v ? gp[2 * LB_H] : 0.f;
            q[3][r] = v ? gp[3 * LB_H] : 0.f;
            q[4][r] = v ? gp[4 * LB_H] : 0.f;
            q[5][r] = (v && t > 0) ? *(gp - LB_H) : 0.f;        // c_{t-1}: slot 4 of step t-1 sits 64 floats below slot 0 of step t
            q[6][r] = v ? dh_out[hoff[r] + (size_t)(t >= 0 ? t : 0) * h_t] : 0.f;
        }
    };
#pragma unroll
    for (int k = 0; k < LB_PF; ++k) load(T - 1 - k, pf[k]);

    for (int t0 = T - 1; t0 >= 0; t0 -= LB_PF) {
#pragma unroll
      for (int k = 0; k < LB_PF; ++k) {
        const int t = t0 - k;
        if (t < 0) break;
        const int buf = t & 1;
        float gi[4], gf[4], gg[4], go[4], ct[4], cp[4], du[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            gi[r] = pf[k][0][r]; gf[r] = pf[k][1][r]; gg[r] = pf[k][2][r]; go[r] = pf[k][3][r];
            ct[r] = pf[k][4][r]; cp[r] = pf[k][5][r]; du[r] = pf[k][6][r];
        }
        load(t - LB_PF, pf[k]);                                   // this slot's next use, LB_PF steps ahead
        // ---- elementwise: gate pre-activation gradients of step t
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float dh = du[r] + dhr[r];
            const float tc = lb_tanh(ct[r]);
            const float d_o = dh * tc * go[r] * (1.0f - go[r]);
            const float dc = dcc[r] + dh * go[r] * (1.0f - tc * tc);
            const float d_i = dc * gg[r] * gi[r] * (1.0f - gi[r]);
            const float d_g = dc * gi[r] * (1.0f - gg[r] * gg[r]);
            const float d_f = dc * cp[r] * gf[r] * (1.0f - gf[r]);
            dcc[r] = dc * gf[r];
            float* row = &dg[buf][(4 * lk + r) * LB_LD + u];
            row[0] = d_i;
            row[LB_H] = d_f;
            row[2 * LB_H] = d_g;
            row[3 * LB_H] = d_o;
            if constexpr (BF) {
                __bf16* rb = reinterpret_cast<__bf16*>(&dgb[buf][(4 * lk + r) * LBB_ROW]) + u;
                rb[0] = (__bf16)d_i;
                rb[LB_H] = (__bf16)d_f;
                rb[2 * LB_H] = (__bf16)d_g;
                rb[3 * LB_H] = (__bf16)d_o;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        // ---- dh_rec = dgates_t . W_hh   (four accumulation chains)
        f32x4 acc[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
        if constexpr (BF) {
            // A fragments: lane (m = ln, kq = lk) holds k = 32*kb + 8*kq + j
            const char* ab = &dgb[buf][ln * LBB_ROW + lk * 16];
#pragma unroll
            for (int kb = 0; kb < 8; ++kb)
                acc[kb & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const lb_bf16x8*>(ab + kb * 64), whb[kb], acc[kb & 3],
                                                                     0, 0, 0);
        } else {
            const float* arow = &dg[buf][ln * LB_LD + 4 * lk];
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const f32x4 a = *reinterpret_cast<const f32x4*>(arow + 16 * j);
#pragma unroll
                for (int s = 0; s < 4; ++s) acc[j & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], whT[4 * j + s], acc[j & 3], 0, 0, 0);
            }
        }
        // ---- coalesced write-back of the dgates tile: 1 KB per sequence row
        if (wok) {
            const float* src = &dg[buf][ws * LB_LD + wc];
            float* dst = dgates + dg_base + (size_t)t * dg_t;
#pragma unroll
            for (int q = 0; q < 4; ++q) *reinterpret_cast<f32x4*>(dst + 4 * q) = *reinterpret_cast<const f32x4*>(src + 4 * q);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) dhr[r] = (acc[0][r] + acc[1][r]) + (acc[2][r] + acc[3][r]);
      }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Small-batch variant (up to 2048 sequences; the training batch of BASELINE configs[3] is 6 x 161 = 966): 4 sequences
// per workgroup on v_mfma_f32_4x4x1_16b_f32, the reverse-time twin of lstm64_q_kernel.  The 16-sequence kernel above is a
// latency chain (elementwise -> LDS -> barrier -> 64 dependent-chain MFMAs of 32 cycles) on 61 of 256 CUs at that size;
// here a step is 64 MFMAs of ~10 cycles per wave and four times as many workgroups run side by side.
//   element role: thread (s = tid/64, u = tid%64) owns hidden unit u of sequence s: dh, dc carry, the four gate gradients;
//   matrix role : dh_rec[s][u'] = sum_k dgates[s][k] W_hh[k][u'], K = 256 split over the 4 waves (wave w: k in [64w, 64w+64));
//                 lane = output unit u' (its W_hh column slice stationary in 64 VGPRs), A operand = the 4 sequences
//                 (broadcast from LDS), accumulator VGPR r = sequence r; the four K-slices meet in LDS.
// ---------------------------------------------------------------------------------------------------------------
#define LBQ_LD (4 * LB_H + 4)

__global__ __launch_bounds__(256) void lstm64_bwd_q_kernel(const float* __restrict__ gates, const float* __restrict__ dh_out,
                                                           const float* __restrict__ wcat, float* __restrict__ dgates, int T, int F,
                                                           int S) {
    __shared__ __attribute__((aligned(16))) float dg[2][4 * LBQ_LD];
    __shared__ __attribute__((aligned(16))) float part[4][4][LB_H];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int es = tid >> 6, eu = tid & 63;                       // element role (es happens to equal the wave index)
    const int sq = blockIdx.x * 4 + es;
    const bool ok = sq < S;
    const int b = ok ? sq / F : 0, f = ok ? sq - b * F : 0;
    const size_t goff = ((size_t)(ok ? sq : 0) * T * 5) * LB_H + eu;
    const size_t hoff = (((size_t)b * T) * F + f) * LB_H + eu;
    const size_t g_t = (size_t)5 * LB_H, h_t = (size_t)F * LB_H;
    const size_t dg_base = (((size_t)b * T) * F + f) * (4 * LB_H) + (size_t)eu * 4;      // writer: float4 eu of row es
    const size_t dg_t = (size_t)F * 4 * LB_H;

    float wh[64];                                                  // W_hh[64*wave + kk][lane]
#pragma unroll
    for (int kk = 0; kk < 64; ++kk) wh[kk] = wcat[(size_t)(64 * wave + kk) * 128 + 64 + lane];

    float dhr = 0.f, dcc = 0.f;
    float pf[LBQ_PF][7];                                            // (i, f, g, o, c_t, c_{t-1}, dh_out) of the next LBQ_PF steps
    // buffer loads with an out-of-range offset for "nothing to load" (returns 0): no branch around a load, so the number of
    // loads outstanding at every point of the unrolled loop is a compile-time fact and the waits before a step's operands are
    // counted ones (with predicated global loads the compiler fell back to vmcnt(0) once per unrolled round)
    const __amdgpu_buffer_rsrc_t rs_g = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(gates), 0,
                                                                         (unsigned)((size_t)S * T * 5 * LB_H * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_h = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(dh_out), 0,
                                                                         (unsigned)((size_t)S * T * LB_H * 4), 0x00020000);
    const unsigned gb = (unsigned)(goff * 4), hb = (unsigned)(hoff * 4), g_tb = (unsigned)(g_t * 4), h_tb = (unsigned)(h_t * 4);
    auto load = [&](int t, float (&q)[7]) {
        const bool v = ok && t >= 0;
        const unsigned go_ = v ? gb + (unsigned)t * g_tb : LB_OOB;
        auto ld = [&](const __amdgpu_buffer_rsrc_t& rs, unsigned off) {
            return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, off, 0, 0));
        };
        q[0] = ld(rs_g, go_);
        q[1] = ld(rs_g, v ? go_ + LB_H * 4 : LB_OOB);
        q[2] = ld(rs_g, v ? go_ + 2 * LB_H * 4 : LB_OOB);
        q[3] = ld(rs_g, v ? go_ + 3 * LB_H * 4 : LB_OOB);
        q[4] = ld(rs_g, v ? go_ + 4 * LB_H * 4 : LB_OOB);
        q[5] = ld(rs_g, (v && t > 0) ? go_ - LB_H * 4 : LB_OOB);   // c_{t-1}: slot 4 of step t-1 sits 64 floats below slot 0 of step t
        q[6] = ld(rs_h, v ? hb + (unsigned)t * h_tb : LB_OOB);
    };
#pragma unroll
    for (int k = 0; k < LBQ_PF; ++k) load(T - 1 - k, pf[k]);

    for (int t0 = T - 1; t0 >= 0; t0 -= LBQ_PF) {
#pragma unroll
      for (int k = 0; k < LBQ_PF; ++k) {
        const int t = t0 - k;
        if (t < 0) break;
        const int buf = t & 1;
        const float gi = pf[k][0], gf = pf[k][1], gg = pf[k][2], go = pf[k][3], ct = pf[k][4], cp = pf[k][5], du = pf[k][6];
        load(t - LBQ_PF, pf[k]);
        {   // same expressions as the 16-sequence kernel
            const float dh = du + dhr;
            const float tc = lb_tanh(ct);
            const float d_o = dh * tc * go * (1.0f - go);
            const float dc = dcc + dh * go * (1.0f - tc * tc);
            const float d_i = dc * gg * gi * (1.0f - gi);
            const float d_g = dc * gi * (1.0f - gg * gg);
            const float d_f = dc * cp * gf * (1.0f - gf);
            dcc = dc * gf;
            float* row = &dg[buf][es * LBQ_LD + eu];
            row[0] = d_i;
            row[LB_H] = d_f;
            row[2 * LB_H] = d_g;
            row[3 * LB_H] = d_o;
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        f32x4 acc[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
        const float* arow = &dg[buf][(lane & 3) * LBQ_LD + 64 * wave];
        // all sixteen A reads first (64 registers; one wave per SIMD has them): read-by-read the compiler kept one LDS read ahead
        // of its four MFMAs -- 40 cycles of multiply per ~100 cycles of read latency
        f32x4 ar[16];
#pragma unroll
        for (int k4 = 0; k4 < 16; ++k4) ar[k4] = *reinterpret_cast<const f32x4*>(arow + 4 * k4);
#pragma unroll
        for (int k4 = 0; k4 < 16; ++k4)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
                acc[(4 * k4 + kk) & 7] = __builtin_amdgcn_mfma_f32_4x4x1f32(ar[k4][kk], wh[4 * k4 + kk], acc[(4 * k4 + kk) & 7], 0, 0, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 16, 0);          // (the scheduler would sink the reads back to their uses)
        __builtin_amdgcn_sched_group_barrier(0x008, 64, 0);
        __builtin_amdgcn_sched_barrier(0);
        const f32x4 p = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
#pragma unroll
        for (int r = 0; r < 4; ++r) part[wave][r][lane] = p[r];
        if (ok)     // coalesced write-back of the dgates rows: 1 KB per sequence
            *reinterpret_cast<f32x4*>(dgates + dg_base + (size_t)t * dg_t) = *reinterpret_cast<const f32x4*>(&dg[buf][es * LBQ_LD + eu * 4]);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        dhr = (part[0][es][eu] + part[1][es][eu]) + (part[2][es][eu] + part[3][es][eu]);
      }
    }
}

extern "C" int eab_lstm64_bwd_f32(const float* gates, const float* dh_out, const float* wcat, float* dgates, int B, int T, int F,
                                  eab_stream_t stream) {
    return eab_lstm64_bwd_prec_f32(gates, dh_out, wcat, dgates, B, T, F, EAB_PREC_F32, stream);
}

extern "C" int eab_lstm64_bwd_prec_f32(const float* gates, const float* dh_out, const float* wcat, float* dgates, int B, int T, int F,
                                       int precision, eab_stream_t stream) {
    EAB_CHECK_ARG(gates && dh_out && wcat && dgates && B > 0 && T > 0 && F > 0);
    EAB_CHECK_ARG(precision == EAB_PREC_F32 || precision == EAB_PREC_BF16);
    const long long S = (long long)B * F;
    EAB_CHECK_ARG(S * T * 5 * LB_H < (1ll << 40));
    // bf16: a step of the 16-sequence form is 8 short MFMAs instead of 64 long ones (742 vs 1004 us per layer at 2576 sequences);
    // up to 2048 sequences the 4-sequence fp32 kernel below is faster than either (655 vs 926 us at 966: four times the
    // workgroups, a quarter of the elementwise chain per lane) and exact, so small batches take it in every mode
    if (precision == EAB_PREC_BF16 && S > 2048) {
        hipLaunchKernelGGL(lstm64_bwd_kernel<true>, dim3((unsigned)((S + LB_SEQ - 1) / LB_SEQ)), dim3(256), 0, eab_stream(stream), gates,
                           dh_out, wcat, dgates, T, F, (int)S);
        EAB_RETURN_LAUNCH_STATUS();
    }
    if (S <= 2048 && S * T * 5 * LB_H * 4 < (1ll << 32)) {       // (32-bit byte offsets in the 4-sequence kernel's buffer descriptors)
        hipLaunchKernelGGL(lstm64_bwd_q_kernel, dim3((unsigned)((S + 3) / 4)), dim3(256), 0, eab_stream(stream), gates, dh_out, wcat,
                           dgates, T, F, (int)S);
        EAB_RETURN_LAUNCH_STATUS();
    }
    const int grid = (int)((S + LB_SEQ - 1) / LB_SEQ);
    hipLaunchKernelGGL(lstm64_bwd_kernel<false>, dim3(grid), dim3(256), 0, eab_stream(stream), gates, dh_out, wcat, dgates, T, F, (int)S);
    EAB_RETURN_LAUNCH_STATUS();
}
