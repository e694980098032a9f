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

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float lb_tanh(float x) { return fmaf(2.0f, __builtin_amdgcn_rcpf(1.0f + __expf(-2.0f * x)), -1.0f); }

__global__ __launch_bounds__(256) void lstm64_bwd_kernel(const float* __restrict__ gates, const float* __restrict__ dh_out,
                                                         const float* __restrict__ wcat, float* __restrict__ dgates, int T, int F,
                                                         int S) {
    __shared__ __attribute__((aligned(16))) float dg[2][LB_SEQ * LB_LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ln = lane & 15, lk = lane >> 4;
    const int s0 = blockIdx.x * LB_SEQ;
    const int u = wave * 16 + ln;

    // stationary operand: whT[4j+s] = W_hh[k = 16j + 4lk + s][u] = wcat[k][64 + u]
    float whT[64];
#pragma unroll
    for (int j = 0; j < 16; ++j)
#pragma unroll
        for (int s = 0; s < 4; ++s) whT[4 * j + s] = wcat[(size_t)(16 * j + 4 * lk + s) * 128 + 64 + u];

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
    // values of step t in registers (loaded one step ahead)
    float gi[4], gf[4], gg[4], go[4], ct[4], cp[4], du[4];
    auto load = [&](int t, float (&i_)[4], float (&f_)[4], float (&g_)[4], float (&o_)[4], float (&c_)[4], float (&p_)[4],
                    float (&d_)[4]) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const bool v = ok[r] && t >= 0;
            const float* gp = gates + goff[r] + (size_t)(t >= 0 ? t : 0) * g_t;
            i_[r] = v ? gp[0] : 0.f;
            f_[r] = v ? gp[LB_H] : 0.f;
            g_[r] = v ? gp[2 * LB_H] : 0.f;
            o_[r] = v ? gp[3 * LB_H] : 0.f;
            c_[r] = v ? gp[4 * LB_H] : 0.f;
            p_[r] = (v && t > 0) ? *(gp - LB_H) : 0.f;          // c_{t-1}: slot 4 of step t-1 sits 64 floats below slot 0 of step t
            d_[r] = v ? dh_out[hoff[r] + (size_t)(t >= 0 ? t : 0) * h_t] : 0.f;
        }
    };
    load(T - 1, gi, gf, gg, go, ct, cp, du);

    for (int t = T - 1; t >= 0; --t) {
        const int buf = t & 1;
        float ni[4], nf[4], ng[4], no[4], nc[4], np[4], nd[4];
        load(t - 1, ni, nf, ng, no, nc, np, nd);                      // next step's operands in flight
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
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        // ---- dh_rec = dgates_t . W_hh   (four accumulation chains)
        f32x4 acc[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
        const float* arow = &dg[buf][ln * LB_LD + 4 * lk];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(arow + 16 * j);
#pragma unroll
            for (int s = 0; s < 4; ++s) acc[j & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], whT[4 * j + s], acc[j & 3], 0, 0, 0);
        }
        // ---- coalesced write-back of the dgates tile: 1 KB per sequence row
        if (wok) {
            const float* src = &dg[buf][ws * LB_LD + wc];
            float* dst = dgates + dg_base + (size_t)t * dg_t;
#pragma unroll
            for (int q = 0; q < 4; ++q) *reinterpret_cast<f32x4*>(dst + 4 * q) = *reinterpret_cast<const f32x4*>(src + 4 * q);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            dhr[r] = (acc[0][r] + acc[1][r]) + (acc[2][r] + acc[3][r]);
            gi[r] = ni[r]; gf[r] = nf[r]; gg[r] = ng[r]; go[r] = no[r]; ct[r] = nc[r]; cp[r] = np[r]; du[r] = nd[r];
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
    float gi, gf, gg, go, ct, cp, du;
    auto load = [&](int t, float& i_, float& f_, float& g_, float& o_, float& c_, float& p_, float& d_) {
        const bool v = ok && t >= 0;
        const float* gp = gates + goff + (size_t)(t >= 0 ? t : 0) * g_t;
        i_ = v ? gp[0] : 0.f;
        f_ = v ? gp[LB_H] : 0.f;
        g_ = v ? gp[2 * LB_H] : 0.f;
        o_ = v ? gp[3 * LB_H] : 0.f;
        c_ = v ? gp[4 * LB_H] : 0.f;
        p_ = (v && t > 0) ? *(gp - LB_H) : 0.f;                     // c_{t-1}
        d_ = v ? dh_out[hoff + (size_t)(t >= 0 ? t : 0) * h_t] : 0.f;
    };
    load(T - 1, gi, gf, gg, go, ct, cp, du);

    for (int t = T - 1; t >= 0; --t) {
        const int buf = t & 1;
        float ni, nf, ng, no, nc, np, nd;
        load(t - 1, ni, nf, ng, no, nc, np, nd);
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
#pragma unroll
        for (int k4 = 0; k4 < 16; ++k4) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(arow + 4 * k4);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
                acc[(4 * k4 + kk) & 7] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[kk], wh[4 * k4 + kk], acc[(4 * k4 + kk) & 7], 0, 0, 0);
        }
        const f32x4 p = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
#pragma unroll
        for (int r = 0; r < 4; ++r) part[wave][r][lane] = p[r];
        if (ok)     // coalesced write-back of the dgates rows: 1 KB per sequence
            *reinterpret_cast<f32x4*>(dgates + dg_base + (size_t)t * dg_t) = *reinterpret_cast<const f32x4*>(&dg[buf][es * LBQ_LD + eu * 4]);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        dhr = (part[0][es][eu] + part[1][es][eu]) + (part[2][es][eu] + part[3][es][eu]);
        gi = ni; gf = nf; gg = ng; go = no; ct = nc; cp = np; du = nd;
    }
}

extern "C" int eab_lstm64_bwd_f32(const float* gates, const float* dh_out, const float* wcat, float* dgates, int B, int T, int F,
                                  eab_stream_t stream) {
    EAB_CHECK_ARG(gates && dh_out && wcat && dgates && B > 0 && T > 0 && F > 0);
    const long long S = (long long)B * F;
    EAB_CHECK_ARG(S * T * 5 * LB_H < (1ll << 40));
    if (S <= 2048) {
        hipLaunchKernelGGL(lstm64_bwd_q_kernel, dim3((unsigned)((S + 3) / 4)), dim3(256), 0, eab_stream(stream), gates, dh_out, wcat,
                           dgates, T, F, (int)S);
        EAB_RETURN_LAUNCH_STATUS();
    }
    const int grid = (int)((S + LB_SEQ - 1) / LB_SEQ);
    hipLaunchKernelGGL(lstm64_bwd_kernel, dim3(grid), dim3(256), 0, eab_stream(stream), gates, dh_out, wcat, dgates, T, F, (int)S);
    EAB_RETURN_LAUNCH_STATUS();
}
