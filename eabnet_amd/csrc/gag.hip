// GaGNet post-filter glue (reference GaGNet.py): the two elementwise kernels around the conv
// launches of a GlanceGazeModule.  Both are HBM-bound streaming passes over TF bins.
//
//  * gag_pack_kernel: GaGNet.forward (GaGNet.py:80-85) feeds the encoder cat([inpt, pre_x], 1) and
//    every module cat(feat_x, pre_x.view(B, 2F, T)).  The planar (B,2,T,F) inputs are re-laid once:
//    enc_in [B][T][F][4] = (in_r, in_i, pre_r, pre_i) for the channels-last conv kernel, and
//    pre [B][T][LD] with channel f*2+ri (zero padded to a float4 multiple) as the second K source
//    of the gated 1x1 in-convs (their weight columns are permuted to this order on the host).
//  * gag_crm_kernel: GlanceGazeModule.forward tail (GaGNet.py:127-133).  The reference goes through
//    polar form: mag = |pre|, phase = atan2 -> (mag*gain)*(cos, sin)(phase) + residual, which is
//    pre * gain + residual (cos(atan2(i,r)) = r/mag; a zero bin gives 0 either way).  Writes the
//    next module's pre [B][T][LD] and the stage output in planar [B][2][T][F] (= the reference's
//    (B,2,F,T) tensor seen through permute(0,1,3,2), and EaBNetWithPostNet's "esti_stft" as is).
#include "common.h"

#define GAG_THREADS 256

// Both kernels: one (b, t) row per workgroup; with a streaming window (eab_time_window) the grid covers
// [B][count] rows starting at frame *t_pos and rows past the utterance end exit.
__global__ __launch_bounds__(GAG_THREADS) void gag_pack_kernel(const float* __restrict__ inpt, const float* __restrict__ pre_x,
                                                               float* __restrict__ enc_in, float* __restrict__ pre,
                                                               int T, int F, int ld, const int* __restrict__ t_pos, int t_count) {
    const int Tw = t_pos ? t_count : T;
    const int b = blockIdx.x / Tw, t = (t_pos ? *t_pos : 0) + (blockIdx.x - b * Tw);
    if (t >= T) return;
    const int bt = b * T + t;
    const size_t plane = (size_t)T * F;
    const float* ir = inpt + ((size_t)b * 2) * plane + (size_t)t * F;
    const float* pr = pre_x + ((size_t)b * 2) * plane + (size_t)t * F;
    for (int f = threadIdx.x; f < ld / 2; f += GAG_THREADS) {
        const bool ok = f < F;
        const float a0 = ok ? ir[f] : 0.0f, a1 = ok ? ir[plane + f] : 0.0f;
        const float p0 = ok ? pr[f] : 0.0f, p1 = ok ? pr[plane + f] : 0.0f;
        if (ok) *reinterpret_cast<f32x4*>(&enc_in[((size_t)bt * F + f) * 4]) = f32x4{a0, a1, p0, p1};
        *reinterpret_cast<float2*>(&pre[(size_t)bt * ld + 2 * f]) = make_float2(p0, p1);
    }
}

__device__ __forceinline__ float gag_act(float v, int act) {
    if (act == EAB_ACT_SIGMOID) return 1.0f / (1.0f + __expf(-v));
    if (act == EAB_ACT_TANH) return tanhf(v);
    return fmaxf(v, 0.0f);
}

__global__ __launch_bounds__(GAG_THREADS) void gag_crm_kernel(const float* __restrict__ pre, const float* __restrict__ g,
                                                              const float* __restrict__ r, const float* __restrict__ i,
                                                              float* __restrict__ pre_out, float* __restrict__ planar,
                                                              int T, int F, int ld, int lin_ld, int act,
                                                              const int* __restrict__ t_pos, int t_count) {
    const int Tw = t_pos ? t_count : T;
    const int b = blockIdx.x / Tw, t = (t_pos ? *t_pos : 0) + (blockIdx.x - b * Tw);
    if (t >= T) return;
    const int bt = b * T + t;
    const size_t plane = (size_t)T * F;
    float* o_r = planar + ((size_t)b * 2) * plane + (size_t)t * F;
    for (int f = threadIdx.x; f < ld / 2; f += GAG_THREADS) {
        float2 y = make_float2(0.0f, 0.0f);
        if (f < F) {
            const float2 p = *reinterpret_cast<const float2*>(&pre[(size_t)bt * ld + 2 * f]);
            const float gain = gag_act(g[(size_t)bt * lin_ld + f], act);
            y = make_float2(p.x * gain + r[(size_t)bt * lin_ld + f], p.y * gain + i[(size_t)bt * lin_ld + f]);
            o_r[f] = y.x;
            o_r[plane + f] = y.y;
        }
        *reinterpret_cast<float2*>(&pre_out[(size_t)bt * ld + 2 * f]) = y;
    }
}

extern "C" int eab_gag_pack_f32(const float* inpt, const float* pre_x, float* enc_in, float* pre, int B, int T, int F,
                                int ld, eab_time_window win, eab_stream_t stream) {
    EAB_CHECK_ARG(inpt && pre_x && enc_in && pre && B > 0 && T > 0 && F > 0);
    EAB_CHECK_ARG(ld >= 2 * F && (ld % 4) == 0 && (long long)B * T < (1ll << 31));
    EAB_CHECK_ARG(win.pos == nullptr || win.count > 0);
    hipLaunchKernelGGL(gag_pack_kernel, dim3(B * (win.pos ? win.count : T)), dim3(GAG_THREADS), 0, eab_stream(stream), inpt,
                       pre_x, enc_in, pre, T, F, ld, win.pos, win.count);
    EAB_RETURN_LAUNCH_STATUS();
}

extern "C" int eab_gag_crm_f32(const float* pre, const float* g, const float* r, const float* i, float* pre_out,
                               float* planar, int B, int T, int F, int ld, int lin_ld, int act, eab_time_window win,
                               eab_stream_t stream) {
    EAB_CHECK_ARG(pre && g && r && i && pre_out && planar && B > 0 && T > 0 && F > 0);
    EAB_CHECK_ARG(ld >= 2 * F && (ld % 4) == 0 && lin_ld >= F && (long long)B * T < (1ll << 31));
    EAB_CHECK_ARG(act == EAB_ACT_SIGMOID || act == EAB_ACT_TANH || act == EAB_ACT_RELU);
    EAB_CHECK_ARG(win.pos == nullptr || win.count > 0);
    hipLaunchKernelGGL(gag_crm_kernel, dim3(B * (win.pos ? win.count : T)), dim3(GAG_THREADS), 0, eab_stream(stream), pre, g,
                       r, i, pre_out, planar, T, F, ld, lin_ld, act, win.pos, win.count);
    EAB_RETURN_LAUNCH_STATUS();
}

// Backward of gag_crm_kernel (training of the post-filter): y = pre * gain(g) + (r, i) per TF bin, y feeding both the stage
// output (planar) and the next module's pre.
//   dy   = d planar[b][:][t][f] + d pre_out[b][t][2f..]        (either may be NULL)
//   dg   = (dy.re pre.re + dy.im pre.im) * gain'(g),  dr = dy.re,  di = dy.im        [B][T][lin_ld], padding columns zeroed
//   dpre = (acc_in) + dy * gain                                 [B][T][ld]            (NULL: not wanted)
__global__ __launch_bounds__(GAG_THREADS) void gag_crm_bwd_kernel(const float* __restrict__ pre, const float* __restrict__ g,
                                                                  const float* __restrict__ dplanar, const float* __restrict__ dpre_out,
                                                                  const float* __restrict__ acc_in, float* __restrict__ dg,
                                                                  float* __restrict__ dr, float* __restrict__ di,
                                                                  float* __restrict__ dpre, int T, int F, int ld, int lin_ld, int act) {
    const int b = blockIdx.x / T, t = blockIdx.x - b * T;
    const int bt = blockIdx.x;
    const size_t plane = (size_t)T * F;
    const float* d_r = dplanar ? dplanar + ((size_t)b * 2) * plane + (size_t)t * F : nullptr;
    const int nf = ld / 2 > lin_ld ? ld / 2 : lin_ld;
    for (int f = threadIdx.x; f < nf; f += GAG_THREADS) {
        float2 dy = make_float2(0.0f, 0.0f), dp = make_float2(0.0f, 0.0f);
        float dgv = 0.0f;
        if (f < F) {
            if (d_r) dy = make_float2(d_r[f], d_r[plane + f]);
            if (dpre_out) {
                const float2 q = *reinterpret_cast<const float2*>(&dpre_out[(size_t)bt * ld + 2 * f]);
                dy.x += q.x;
                dy.y += q.y;
            }
            const float2 p = *reinterpret_cast<const float2*>(&pre[(size_t)bt * ld + 2 * f]);
            const float gv = g[(size_t)bt * lin_ld + f];
            const float gain = gag_act(gv, act);
            const float dact = act == EAB_ACT_SIGMOID ? gain * (1.0f - gain) : act == EAB_ACT_TANH ? 1.0f - gain * gain : (gv > 0.0f ? 1.0f : 0.0f);
            dgv = (dy.x * p.x + dy.y * p.y) * dact;
            dp = make_float2(dy.x * gain, dy.y * gain);
        }
        if (f < lin_ld) {
            dg[(size_t)bt * lin_ld + f] = dgv;
            dr[(size_t)bt * lin_ld + f] = dy.x;
            di[(size_t)bt * lin_ld + f] = dy.y;
        }
        if (dpre && 2 * f < ld) {
            if (acc_in) {
                const float2 q = *reinterpret_cast<const float2*>(&acc_in[(size_t)bt * ld + 2 * f]);
                dp.x += q.x;
                dp.y += q.y;
            }
            *reinterpret_cast<float2*>(&dpre[(size_t)bt * ld + 2 * f]) = dp;
        }
    }
}

extern "C" int eab_gag_crm_bwd_f32(const float* pre, const float* g, const float* dplanar, const float* dpre_out, const float* acc_in,
                                   float* dg, float* dr, float* di, float* dpre, int B, int T, int F, int ld, int lin_ld, int act,
                                   eab_stream_t stream) {
    EAB_CHECK_ARG(pre && g && dg && dr && di && (dplanar || dpre_out) && B > 0 && T > 0 && F > 0);
    EAB_CHECK_ARG(ld >= 2 * F && (ld % 4) == 0 && lin_ld >= F && (long long)B * T < (1ll << 31));
    EAB_CHECK_ARG(act == EAB_ACT_SIGMOID || act == EAB_ACT_TANH || act == EAB_ACT_RELU);
    hipLaunchKernelGGL(gag_crm_bwd_kernel, dim3(B * T), dim3(GAG_THREADS), 0, eab_stream(stream), pre, g, dplanar, dpre_out, acc_in, dg,
                       dr, di, dpre, T, F, ld, lin_ld, act);
    EAB_RETURN_LAUNCH_STATUS();
}
