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
