/*
 * eabnet_hip.h -- C ABI of libeabnet_hip.so, the MI355X (gfx950) implementation
 * of EaBNet's per-frame beamforming hot path.
 *
 * The reference (Ezreal11/EaBNet) has no FFI layer: its "operator API" is the
 * Python class EaBNet and the free function prepare_data, and every device op
 * is an ATen call.  Each entry point below therefore names the reference
 * call-site(s) it replaces (file:line into the reference tree).  The Python
 * side (eabnet_amd/) binds these with ctypes; INTEGRATION.md shows the stub.
 *
 * Conventions
 *   - all pointers are DEVICE pointers to fp32 unless stated otherwise; the
 *     caller (PyTorch) owns every buffer, the library allocates nothing;
 *   - `stream` is a hipStream_t passed as void* (0 = the null stream); all work
 *     is enqueued on it, no entry point synchronises, all are graph-capturable;
 *   - return value: 0 on success, otherwise an EAB_E* code / hipError_t + 1000;
 *     eab_error_string() describes it.  Arguments are validated BEFORE any
 *     launch (shape/limit violations never reach a kernel);
 *   - activation layout inside the library is channels-last:
 *     [B][T][F][C] fp32, C contiguous.  The two boundary tensors keep the
 *     reference's layouts: input (B,T,F,M,2), output (B,2,T,F).
 */
#ifndef EABNET_HIP_H
#define EABNET_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EAB_ABI_VERSION 8

#define EAB_OK          0
#define EAB_EINVAL      1   /* bad argument (shape, alignment, limit) */
#define EAB_EUNSUPPORTED 2  /* valid but not built (e.g. N not in {64,128,256}) */
#define EAB_EHIP_BASE   1000 /* 1000 + hipError_t */

typedef void* eab_stream_t;

int         eab_abi_version(void);
const char* eab_error_string(int code);

/* Time window for streaming inference (causal programs without data-dependent statistics, i.e.
 * norm_type = "BN" in eval mode; SURVEY §8f N4 / BASELINE config 5).  The activations of the whole
 * utterance [B][T][..] stay resident; when `pos` is non-NULL an op computes only the time rows
 *     t in [*pos, min(*pos + count, T))
 * of its output and leaves every other row as it is -- older rows are the "state" later chunks read.
 * `pos` points to DEVICE memory, so ONE captured hipGraph is replayed for successive chunks after a
 * 4-byte update of the position; the launch geometry depends on `count` only. */
typedef struct eab_time_window {
    const int32_t* pos;
    int32_t count;
} eab_time_window;

/* --------------------------------------------------------------------------
 * K1+K2+K3  STFT front end, fused.
 * Replaces train_distributed.py:80,83,86,89,91 (noisy branch) and
 * :81,84,87,90,92 (target branch): view(B*M,L) -> torch.stft(n_fft, hop,
 * win=n_fft, window, center=True, reflect, onesided) -> permute -> sqrt-magnitude
 * compression X*|X|^-1/2.
 *   wav     [B][M][L]
 *   window  [n_fft]          the caller's torch.hann_window(win) (values are
 *                            taken, not recomputed: ATen's fp32 window is not
 *                            the double-rounded formula)
 *   twiddle [n_fft][2]       (cos, sin)(2 pi k / n_fft), k = 0..n_fft-1
 *   out     layout 0: [B][T][F][M][2]   (noisy_stft handed to EaBNet.forward)
 *           layout 1: [B][2][T][F]      (target_stft; requires M == 1)
 *   T = 1 + L/hop, F = n_fft/2 + 1.  Limits: n_fft <= 512, even; L > n_fft/2.
 * ------------------------------------------------------------------------ */
#define EAB_STFT_LAYOUT_BTFM2 0
#define EAB_STFT_LAYOUT_B2TF  1
int eab_stft_compress_f32(const float* wav, const float* window, const float* twiddle,
                          float* out, int B, int M, int L, int n_fft, int hop,
                          int layout, eab_stream_t stream);

/* Verification twin of the framing step (the bit-exact contract for "STFT frame indexing",
 * train_distributed.py:83 torch.stft(center=True, pad_mode="reflect")): the product kernel of
 * eab_stft_compress_f32 itself, compiled to store the rows it gathered (un-windowed) instead of
 * transforming them -- both of its gather paths are the ones under test:
 *     frames[b][m][t][k] = reflect_pad(wav[b][m], n_fft/2)[t*hop + k].
 *   wav [B][M][L] -> frames [B][M][T][n_fft];  n_fft/2 must factor into {2,4,5,8} as for the FFT path */
int eab_stft_frames_f32(const float* wav, float* frames, int B, int M, int L, int n_fft, int hop,
                        eab_stream_t stream);

/* --------------------------------------------------------------------------
 * ISTFT back end.  Replaces enhance.py:59-62, test.py:189-191,
 * train_distributed.py:128-130:
 *     esti.permute(0,3,2,1) -> view_as_complex -> torch.istft(fft_num, win_shift, win_size, hann)
 * (centre-trimmed, window-envelope normalised overlap-add of the irfft frames; imaginary parts of
 * the DC / Nyquist bins ignored as in any C2R transform).  Applied, as in the reference, to the
 * compressed-domain estimate without decompression.
 *   spec    [B][2][T][F]  estimate, F = n_fft/2+1, T >= 2
 *   window  [n_fft], twiddle [n_fft][2] as for eab_stft_compress_f32
 *   wav     [B][hop*(T-1)]
 * Any hop <= n_fft with ceil(n_fft / hop) <= 8 frames per sample (the reference's 320/160: 2; the hop need not divide
 * n_fft), n_fft/2 = 2^a 5^b, else EAB_EUNSUPPORTED.  A window shorter than n_fft is passed zero-padded to n_fft (centred), as torch.istft pads it.
 * ------------------------------------------------------------------------ */
int eab_istft_f32(const float* spec, const float* window, const float* twiddle, float* wav, int B, int T,
                  int n_fft, int hop, eab_stream_t stream);

/* --------------------------------------------------------------------------
 * GaGNet post-filter glue (reference GaGNet.py).  The convolutions of the post-filter
 * run on eab_conv_f32; these are the two elementwise passes around them.
 *
 * eab_gag_pack_f32: GaGNet.forward's two concatenations (GaGNet.py:80-85, 189-190, 249-250) as one
 * re-layout of the planar inputs:
 *   inpt, pre_x [B][2][T][F] -> enc_in [B][T][F][4] = (in_r, in_i, pre_r, pre_i)
 *                               pre    [B][T][ld], channel f*2+ri, zero padded (ld % 4 == 0, ld >= 2F)
 * eab_gag_crm_f32: GlanceGazeModule.forward tail (GaGNet.py:127-133), coarse filtering by the gain plus
 * the complex residual:  y = pre * act(g) + (r, i)   (the reference's polar form, algebraically)
 *   pre [B][T][ld];  g, r, i [B][T][lin_ld] (first F entries used; biases already added)
 *   -> pre_out [B][T][ld] (next module's pre)  and  planar [B][2][T][F] (the stage output)
 * ------------------------------------------------------------------------ */
/* (both take a streaming window like every other op: win.pos == NULL = the whole utterance) */
#define EAB_ACT_SIGMOID 0
#define EAB_ACT_TANH    1
#define EAB_ACT_RELU    2
int eab_gag_pack_f32(const float* inpt, const float* pre_x, float* enc_in, float* pre, int B, int T, int F,
                     int ld, eab_time_window win, eab_stream_t stream);
int eab_gag_crm_f32(const float* pre, const float* g, const float* r, const float* i, float* pre_out,
                    float* planar, int B, int T, int F, int ld, int lin_ld, int act, eab_time_window win,
                    eab_stream_t stream);
/* Backward of eab_gag_crm_f32 (training of the post-filter; autograd of GaGNet.py:127-133 in its pre*gain + residual form):
 * dplanar [B][2][T][F] = gradient of the stage output, dpre_out [B][T][ld] = gradient of the next module's pre (either may be
 * NULL); dg, dr, di [B][T][lin_ld] (padding columns zeroed); dpre [B][T][ld] = (acc_in) + the gradient w.r.t. pre (NULL: skip). */
int eab_gag_crm_bwd_f32(const float* pre, const float* g, const float* dplanar, const float* dpre_out, const float* acc_in,
                        float* dg, float* dr, float* di, float* dpre, int B, int T, int F, int ld, int lin_ld, int act,
                        eab_stream_t stream);

/* --------------------------------------------------------------------------
 * Training loss, value and gradient in one pass (first piece).  Replaces com_mag_mse_loss
 * (EaBNet.py:627-640) and each stage of stagewise_com_mag_mse_loss (GaGNet.py:601-619):
 *     mask[b,t,f] = t < frames[b],  n = sum(mask)
 *     loss = 0.5 * ( sum mask (|e|-|l|)^2 / n + sum mask |e-l|^2 / (2n) )
 *   esti, label [B][2][T][F];  frames: HOST array of B frame counts (B <= 64; passed by value to the kernel)
 *   partial: device scratch of 2*partial_blocks floats;  loss: device scalar
 *   grad: NULL or [B][2][T][F] receiving d loss / d esti (zero outside the masks)
 * ------------------------------------------------------------------------ */
int eab_com_mag_mse_loss_f32(const float* esti, const float* label, const int32_t* frames, int B, int T, int F,
                             float* partial, int partial_blocks, float* loss, float* grad, eab_stream_t stream);

/* --------------------------------------------------------------------------
 * K13  complex filter-and-sum, stand-alone.   Replaces EaBNet.py:114-117.
 *   w, x [B][T][F][M][2] -> y [B][2][T][F];  Y = sum_m W_m * X_m (no conjugate)
 * ------------------------------------------------------------------------ */
int eab_filter_sum_f32(const float* w, const float* x, float* y, int B, int T, int F, int M,
                       eab_stream_t stream);

/* --------------------------------------------------------------------------
 * K4-K7, K9, K12(first Linear)  gather-GEMM convolution on fp32 MFMA.
 * One launch computes, for every output position (b, t, fo = o*ostride+ophase),
 *     acc[n] = bias[n] + sum_{tap j, channel c} W[n][j][c] * f(src[b][t+dt_j][o*istride+ioff_j][c])
 * with zeros where t+dt_j < 0 or the frequency index leaves [0, Fin), f an
 * optional fused InstanceNorm-affine/PReLU, src an optional channel
 * concatenation of two tensors, followed by a fused epilogue and optional
 * per-(b,channel) sum / sum-of-squares partials for the NEXT InstanceNorm.
 * Covers
 *   GateConv2d           EaBNet.py:449-450,459-460   (pad + Conv2d + GLU)
 *   GateConvTranspose2d  EaBNet.py:478-480,489-490   (two launches: even / odd fo)
 *   Conv2dunit conv      EaBNet.py:402               Deconv2dunit deconv  :423,425
 *   torch.cat skips      EaBNet.py:275,277,502       (src0|src1)
 *   SqueezedTCM convs    EaBNet.py:549,558,564,570,575,577
 *   w_dnn[0] + ReLU      EaBNet.py:594-595
 * ------------------------------------------------------------------------ */
#define EAB_MAX_TAPS 16

#define EAB_XF_NONE       0   /* f(x) = x                                   */
#define EAB_XF_NORM_PRELU 1   /* f(x) = prelu(x*scale + shift)   (2-D units) */
#define EAB_XF_PRELU_NORM 2   /* f(x) = prelu(x)*scale + shift   (S-TCM)     */

#define EAB_EPI_LINEAR  0     /* out = acc                                   */
#define EAB_EPI_GLU     1     /* out[c] = acc[c] * sigmoid(acc[N/2 + c])     */
#define EAB_EPI_RELU    2     /* out = max(acc, 0)                           */
#define EAB_EPI_MULSIG  3     /* out = aux * sigmoid(acc)    (S-TCM gate)    */
#define EAB_EPI_ADD     4     /* out = acc + aux             (residual)      */
#define EAB_EPI_DUALGATE 5    /* S-TCM left*sigmoid(right) in ONE launch: N = 2*Cout, single source src0;
                               * columns [0,Cout) see src0 through transform 0 (xf0/slope0), columns
                               * [Cout,2Cout) through transform 1 (xf1/slope1) of the SAME tensor;
                               * out[c] = acc[c] * sigmoid(acc[Cout + c]); rows interleaved as for GLU */
#define EAB_EPI_PHASE2  6     /* Both output-column phases of a stride-2 transposed convolution (Deconv2dunit,
                               * EaBNet.py:423,425) in ONE launch on ONE staged input patch (EAB_KORDER_CHUNK only):
                               * N = 2*Cout; rows interleaved as for GLU with "value" = phase 0 and "gate" = phase 1.
                               * Row (t, o), o < No = ceil(Fout/2), of the launch writes
                               *     out[t][2o][c]   = acc[c]          (phase 0)
                               *     out[t][2o+1][c] = acc[Cout + c]   (phase 1; only while 2o+1 < Fout)
                               * ostride = 2, ophase = 0, istride = 1.  p2_mask1 bit j = tap j also feeds phase 1 (the
                               * phase-1 rows of `w` are zero for the other taps and the kernel skips those products);
                               * bit 0 must be set.  The statistics partial of a tile covers both phases. */

typedef struct eab_conv_desc {
    /* sources, channels-last [B][T][Fin][C*]; src1 == NULL when there is no concat */
    const float* src0;
    const float* src1;
    const float* xf0;       /* [B][C0][2] (scale, shift) or NULL */
    const float* xf1;       /* [B][C1][2] or NULL */
    const float* slope0;    /* [C0] PReLU slope or NULL (required with xf*) */
    const float* slope1;
    int32_t C0, C1;         /* C1 = 0 without src1; with src1, C0 % 16 == 0 */
    int32_t xf_mode;        /* EAB_XF_* applied to sources that have a table */
    /* weights, packed by the host: w[n][u*16 + i], unit u = tap*UPT + c/16,
     * UPT = ceil((C0+C1)/16), zero padded; Kpad = ntaps*UPT*16.
     * For EAB_EPI_GLU rows are interleaved in 32-row groups:
     * packed row r -> original row (r%64/32)*(N/2) + (r/64)*32 + r%32. */
    const float* w;
    const float* bias;      /* [N] in packed row order, or NULL */
    int32_t N, Kpad;
    /* geometry */
    int32_t B, T, Fin, Fout;
    int32_t No;             /* output columns per time row computed by this launch */
    int32_t ostride, ophase, istride;
    int32_t ntaps;
    int32_t dt[EAB_MAX_TAPS];    /* time offset of the tap; > 0 (look-ahead, non-causal S-TCM) only with EAB_KORDER_TAP */
    int32_t ioff[EAB_MAX_TAPS];
    /* epilogue */
    int32_t epi;
    const float* aux;       /* [B][T][Fout][Cout], EAB_EPI_MULSIG / EAB_EPI_ADD */
    float* dst;             /* [B][T][Fout][Cout] */
    float* dst_acc;         /* optional running sum: dst_acc += out (same layout) */
    int32_t Cout;           /* N/2 for GLU, else N */
    /* statistics for the consumer's InstanceNorm: per tile, per set s, per channel c the
     * Welford triple (n, mean, M2 = sum (g - mean)^2, 0) of g_s(out) over the tile's valid rows,
     * g_s = PReLU(stat_slope[s]) or id:
     * stats[((b*stat_tiles + stat_tile0 + tile)*nsets + s)*Cout + c][4] */
    float* stats;
    int32_t nsets;          /* 0, 1 or 2 */
    const float* stat_slope0;
    const float* stat_slope1;
    int32_t stat_tiles;     /* tiles per batch element over ALL launches feeding this norm */
    int32_t stat_tile0;     /* first tile index of this launch */
    int32_t bm;             /* rows per tile: 64 or 128 (host's choice, see eab_conv_tiles); 16, 32 or 64 with EAB_KORDER_FRAG */
    /* optional in-kernel InstanceNorm finalisation (replaces eab_in_finalize_f32 + xf0[/xf1]
     * when the producer wrote few tiles): fin_stats = the producer's partials
     * [B][fin_tiles][fin_nsets][C0][4]; set 0 -> transform 0 with (fin_gamma0, fin_beta0),
     * set 1 -> transform 1 (EAB_EPI_DUALGATE only).  xf0/xf1 must be NULL then. */
    const float* fin_stats;
    const float* fin_gamma0;
    const float* fin_beta0;
    const float* fin_gamma1;
    const float* fin_beta1;
    int32_t fin_tiles, fin_nsets, fin_count;
    float fin_eps;
    /* arithmetic of the contraction.  EAB_PREC_F32: exact fp32 MFMA (fmaf chain).
     * EAB_PREC_F16X3: every fp32 operand x is split x = hi + lo (hi = x truncated to
     * fp16, lo = fp16(x - hi)) and a*b is taken as hi*hi + hi*lo + lo*hi on the f16
     * matrix cores with fp32 accumulation (~22-bit products; requires |x| < 65504).
     * Then `w` holds, per row n and per group of 4 consecutive k, 4 fp16 hi followed
     * by 4 fp16 lo (same byte size as the fp32 matrix). */
    int32_t precision;
    /* order of the K units inside `w`: EAB_KORDER_TAP (unit = tap*UPT + chunk, the gather
     * pipeline) or EAB_KORDER_CHUNK (unit = chunk*ntaps + tap): selects the patch pipeline,
     * which stages the input patch of one 16-channel chunk once in LDS and lets every tap
     * read it at a shifted position.  Requires eab_conv_patch_positions(d) <= 352. */
    int32_t korder;
    /* streaming: restrict the launch to a window of time rows (see eab_time_window below);
     * needs stats == NULL, fin_stats == NULL and causal taps (dt <= 0) */
    eab_time_window win;
    /* Fused InstanceNorm finalisation (replaces the eab_in_finalize_f32 launch behind this convolution): when
     * fz_counter != NULL, the workgroup that writes the LAST of the stat_tiles partials of batch element b (over all
     * launches feeding this norm; arrival counted in fz_counter[b], which must be zero before the first launch and is
     * re-armed to zero by that workgroup) merges them -- fp64, fixed tile order, so the result does not depend on which
     * workgroup came last -- and writes fz_xf<s>[b][c] = (scale, shift) for s < nsets with (fz_gamma<s>, fz_beta<s>). */
    int32_t* fz_counter;
    const float* fz_gamma0;
    const float* fz_beta0;
    float* fz_xf0;
    const float* fz_gamma1;
    const float* fz_beta1;
    float* fz_xf1;
    float fz_eps;
    /* training (EAB_EPI_GLU only): optional [B][T][Fout][N] dump of the gated epilogue's two factors in the packed
     * column order -- value columns hold acc+bias, gate columns hold sigmoid(acc+bias) -- read by eab_glu_bwd_f32 */
    float* glu_dump;
    /* EAB_KORDER_FRAG only (small-tile kernel, csrc/conv_st.hip): the second output-column phase of a transposed
     * convolution served by the SAME launch.  ph1_No > 0: per batch element the tiles of phase 0 (No, ophase, w, Kpad,
     * ntaps, dt, ioff above) are followed by the tiles of phase 1 described here; both phases share sources, bias,
     * epilogue, dst and the statistics array (stat_tile0 must be 0: partial tile index = phase-0 tiles, then phase-1
     * tiles; stat_tiles = their sum).  ph1_No == 0: single-phase launch. */
    const float* ph1_w;
    int32_t ph1_No, ph1_ophase, ph1_ntaps, ph1_Kpad;
    int32_t ph1_dt[EAB_MAX_TAPS];
    int32_t ph1_ioff[EAB_MAX_TAPS];
    /* EAB_KORDER_FRAG, EAB_PREC_F32, 1-D launches (No = Fin = Fout = 1) with N = 256 only: a SECOND 1x1 convolution applied
     * to the rows this launch has just produced, in the same kernel -- the out_conv of one S-TCM and the in_conv of the next
     * (EaBNet.py:575-577 then :572-573 of the following block): nothing but the residual stream lies between them.
     *   f2_dst[b][t][n2] = sum_c f2_w[n2][c] * dst[b][t][c]     (dst = this launch's output after its epilogue)
     * f2_w: fragment-order weights [f2_N = 64][256]; f2_stats / f2_nsets / f2_stat_slope*: InstanceNorm partials of f2_dst
     * exactly like `stats` (tile index = this launch's tile index, f2_stat_tiles tiles per batch element).  NULL f2_w = none. */
    const float* f2_w;
    float* f2_dst;
    float* f2_stats;
    const float* f2_stat_slope0;
    const float* f2_stat_slope1;
    int32_t f2_N, f2_nsets, f2_stat_tiles;
    /* EAB_EPI_PHASE2 only: which taps also feed the phase-1 columns (bit j = tap j); 0 otherwise */
    int32_t p2_mask1;
    /* EAB_PREC_BF16, EAB_KORDER_TAP, xf_mode NONE, Fin > 1, C0 % 16 == C1 % 16 == 0: bit 0 / bit 1 = src0 / src1 is STORED as
     * bf16 ([B][T][Fin][C] of 2-byte elements) -- the bf16 training programs' normalised activations and convolution-output
     * gradients (eab_train_norm_act_f32 / eab_train_norm_bwd_f32 / eab_glu_bwd_ex_f32 with EAB_STORE_BF16).  The precision
     * rounds fp32 operands to bf16 on their way into LDS anyway, so results are identical to the fp32-stored tensor; the
     * gather moves half the bytes and converts nothing.  0 = fp32 sources. */
    int32_t src_bf16;
} eab_conv_desc;

#define EAB_PREC_F32   0
#define EAB_PREC_F16X3 1
/* EAB_PREC_BF16 (BASELINE configs[3]/[4], "bf16 mixed precision"): tensors and `w` stay fp32 in memory (plain fp32
 * layout, no host-side packing); both operands are rounded to bf16 (nearest even) on their way into LDS and
 * multiplied on the bf16 matrix cores (v_mfma_f32_32x32x16_bf16) with fp32 accumulation, bias, norms and
 * activations in fp32 -- the arithmetic torch.autocast(bfloat16) gives the reference's convolutions.  NOT inside the
 * 1e-4 parity bar: its measured error is stated per test. */
#define EAB_PREC_BF16  2
#define EAB_KORDER_TAP   0
#define EAB_KORDER_CHUNK 1
/* EAB_KORDER_FRAG selects the small-tile kernel (csrc/conv_st.hip; EAB_PREC_F32 or EAB_PREC_BF16, `w` fp32 in both) for latency-bound launches: the
 * S-TCN's 1-D convolutions and 64-column unit convolutions on few frequency bins.  bm = 16, 32 or 64 rows per tile;
 * N = 64, 128 or 256; C0, C1 in {64, 128, 256}; epilogues LINEAR / RELU / ADD / DUALGATE.  `w` holds the weights in
 * MFMA-fragment order, unit order as EAB_KORDER_TAP (k = tap*UPT*16 + channel):
 *     w[((nb*(Kpad/16) + m2)*64 + lane)*4 + j] = W[16*nb + (lane & 15)][16*m2 + 8*(j >> 1) + 2*(lane >> 4) + (j & 1)]
 * with rows in natural order, except EAB_EPI_DUALGATE: row block 2*w + g (g = 0 value, 1 gate) holds the original rows
 * g*Cout + 16*w .. +15, so a wave owns value and gate of the same 16 channels. */
#define EAB_KORDER_FRAG  2

/* number of tiles per batch element a launch with this geometry produces */
int eab_conv_tiles(int T, int No, int bm);
int eab_conv_f32(const eab_conv_desc* d, eab_stream_t stream);
/* the same launch with the contraction on the bf16 matrix cores whatever d->precision says (BASELINE configs[3]/[4]);
 * fp32 tensors and weights in memory, see EAB_PREC_BF16 */
int eab_conv_bf16(const eab_conv_desc* d, eab_stream_t stream);

/* --------------------------------------------------------------------------
 * K8 (statistics half)  InstanceNorm finalisation.  Replaces the reduction in
 * nn.InstanceNorm{1,2}d(affine=True), EaBNet.py:684,686 (eps 1e-5, biased var).
 * Merges the (n, mean, M2) partials written by eab_conv_f32 (Chan's formula in fp64, fixed
 * order => bit-reproducible, no E[x^2]-E[x]^2 cancellation; `count` is informational: the
 * partials carry their own counts) and emits, per set s, xf_s[b][c] = (scale, shift) with
 *   scale = gamma_s[c] / sqrt(var + eps), shift = beta_s[c] - mean*scale.
 * ------------------------------------------------------------------------ */
int eab_in_finalize_f32(const float* stats, int B, int C, int nsets, int stat_tiles,
                        int count /* T*F positions per (b,c) */, float eps,
                        const float* gamma0, const float* beta0, float* xf0,
                        const float* gamma1, const float* beta1, float* xf1,
                        eab_stream_t stream);

/* --------------------------------------------------------------------------
 * K8 (apply half) + K14 residual.  out = prelu(a*sa+ha) [+ prelu(b*sb+hb)].
 * Replaces the affine+PReLU of EaBNet.py:187-188,356-357 and the residual add
 * of En_unet_module.forward, EaBNet.py:386, where the sum must be materialised.
 *   a, b, out [B][P][C] (P = T*F positions); b may be NULL.
 * ------------------------------------------------------------------------ */
int eab_norm_act_f32(const float* a, const float* xfa, const float* slopea,
                     const float* b, const float* xfb, const float* slopeb,
                     float* out, int B, int P, int C, eab_stream_t stream);

/* --------------------------------------------------------------------------
 * K10+K11  LayerNorm + one LSTM layer over time for B*F independent sequences.
 * Replaces EaBNet.py:608 (permute + LayerNorm(64)) and one nn.LSTM(64->64,
 * batch_first, zero initial state) of EaBNet.py:610-611.
 *   x      [B][T][F][64]   sequence (b,f) reads x[b][t][f][:]
 *   ln_g/b [64] or NULL    (NULL: no LayerNorm -- second layer)
 *   wcat   [256][128]      row g*64+u (gate order i,f,g,o) = [W_ih[row][0:64] | W_hh[row][0:64]]
 *   bias   [256]           b_ih + b_hh
 *   h_out  [B][T][F][64]
 * Hidden size and input size are fixed at 64 (reference default hid_node).
 * ------------------------------------------------------------------------ */
int eab_lstm64_f32(const float* x, const float* ln_g, const float* ln_b, float ln_eps,
                   const float* wcat, const float* bias, float* h_out,
                   int B, int T, int F, eab_stream_t stream);
/* same with the arithmetic selectable: EAB_PREC_F32 (as above) or EAB_PREC_F16X3 (operands
 * split in fp16 hi+lo on the fly, three f16 MFMAs per product, fp32 accumulate; wcat stays fp32) */
int eab_lstm64_prec_f32(const float* x, const float* ln_g, const float* ln_b, float ln_eps,
                        const float* wcat, const float* bias, float* h_out,
                        int B, int T, int F, int precision, eab_stream_t stream);
/* bf16 variant (operands rounded to bf16, v_mfma_f32_16x16x32_bf16, fp32 accumulate and cell state) */
int eab_lstm64_bf16(const float* x, const float* ln_g, const float* ln_b, float ln_eps,
                    const float* wcat, const float* bias, float* h_out, int B, int T, int F, eab_stream_t stream);

/* --------------------------------------------------------------------------
 * K12(second Linear)+K13  beam-forming weights + filter-and-sum, fused.
 * Replaces w_dnn[2] (EaBNet.py:596,613) and EaBNet.py:114-117.  The per-bin
 * weights never reach HBM.
 *   y1 [B][T][F][64]  (ReLU output of w_dnn[0]), w2 [2M][64], b2 [2M],
 *   x [B][T][F][M][2] (the network input), out [B][2][T][F].
 *   bfw: optional [B][T][F][M][2] dump of the weights (NULL in production).
 * ------------------------------------------------------------------------ */
int eab_bfw_filter_sum_f32(const float* y1, const float* w2, const float* b2, const float* x,
                           float* out, float* bfw, int B, int T, int F, int M,
                           eab_stream_t stream);

/* --------------------------------------------------------------------------
 * Program runner: the whole EaBNet.forward (EaBNet.py:88-117) as ONE call.
 * The host builds the op list once per (weights, B, T); replaying it costs one
 * FFI crossing and no Python per layer, and can be captured in a hipGraph.
 * ------------------------------------------------------------------------ */
#define EAB_OP_CONV        1
#define EAB_OP_IN_FINALIZE 2
#define EAB_OP_NORM_ACT    3
#define EAB_OP_LSTM64      4
#define EAB_OP_BFW_FS      5
#define EAB_OP_MEMSET0     6
#define EAB_OP_GAG_PACK    7
#define EAB_OP_GAG_CRM     8
#define EAB_OP_CONV_CHAIN  9   /* eab_conv_st_chain_run: p = {dev_descs, dev_codes}, i = {n, B, lds_bytes, bf16} */

/* geometry of a weight gradient (eab_wgrad_f32, documented with the training entry points below) */
typedef struct eab_wgrad_desc {
    const float* dz;        /* [B][T][Fz][N] */
    const float* src0;      /* [B][T][Fin][C0] */
    const float* src1;      /* [B][T][Fin][C1] or NULL */
    float* dw;              /* [N][Kpad] */
    float* dbias;           /* optional [N]: += column sums of dz over the rows of this launch (bias gradient) */
    int32_t N, C0, C1, Kpad;
    int32_t B, T, Fin, Fz, No, ostride, ophase, istride;
    int32_t ntaps;
    int32_t dt[EAB_MAX_TAPS];
    int32_t ioff[EAB_MAX_TAPS];
    int32_t rows_per_wg;    /* set by the library */
    int32_t precision;      /* EAB_PREC_F32 (exact fp32 products) or EAB_PREC_BF16 (operands rounded to bf16, fp32 accumulation) */
    int32_t bf16_mask;      /* EAB_PREC_BF16 only: bit 0 / 1 / 2 = dz / src0 / src1 is STORED as bf16 (2-byte elements, same
                             * shape; needs N % 4 == 0 and C % 16 == 0 for the flagged tensors).  The products are those of
                             * the fp32-stored tensor (it would be rounded to bf16 here); dbias then sums the stored values. */
} eab_wgrad_desc;

typedef struct eab_op {
    int32_t kind;
    int32_t i[8];
    float   f[2];
    const void* p[12];
    eab_time_window win;    /* every kind except CONV (which carries it in conv.win) and IN_FINALIZE */
    eab_conv_desc conv;     /* EAB_OP_CONV only */
    eab_wgrad_desc wgrad;   /* EAB_OP_WGRAD only */
} eab_op;
/* field use per kind:
 *  IN_FINALIZE i = {B, C, nsets, stat_tiles, count}       f = {eps}
 *              p = {stats, gamma0, beta0, xf0, gamma1, beta1, xf1}
 *  NORM_ACT    i = {B, P, C, T}  p = {a, xfa, slopea, b, xfb, slopeb, out}   (T only read when windowed)
 *  LSTM64      i = {B, T, F, precision}  f = {ln_eps}  p = {x, ln_g, ln_b, wcat, bias, h_out, c_state}
 *  BFW_FS      i = {B, T, F, M}  p = {y1 | h, w2, b2, x, out, bfw, w1, b1}   (w1, b1 NULL: y1 given)
 *  MEMSET0     p = {ptr}  i = {bytes_lo, bytes_hi, B, T, row_floats}   (B, T, row_floats only when windowed)
 *  GAG_PACK    i = {B, T, F, ld}  p = {inpt, pre_x, enc_in, pre}
 *  GAG_CRM     i = {B, T, F, ld, lin_ld, act}  p = {pre, g, r, i, pre_out, planar}
 */
int eab_run_program(const eab_op* ops, int n_ops, eab_stream_t stream);

/* A chain of small-tile launches (EAB_KORDER_FRAG) in ONE launch: the 1-D convolutions of the S-TCN in a frame-synchronous
 * streaming step (reference SqueezedTCM chain, EaBNet.py:506-578, with BatchNorm in eval mode).  Every descriptor must give
 * each utterance exactly one 16-row tile (win.count * No <= 16), carry no statistics of any kind, and launch k+1 may read, of
 * launch k's output, only the rows of the same utterance.  One workgroup per utterance then runs the launches back to back
 * (results bit-identical to eab_conv_f32 on each descriptor in turn).
 *   eab_conv_st_chain_plan: host-side check of descs[0..n) (every argument check of eab_conv_f32, no launch);
 *     codes[k] = the kernel form of launch k, *lds_bytes = dynamic LDS of the chain, *bf16 = its precision (uniform).
 *     EAB_EUNSUPPORTED: not a chain -- launch the descriptors one by one.
 *   eab_conv_st_chain_run: dev_descs / dev_codes = the same arrays in DEVICE memory (the caller uploads them once per
 *     binding of the program); B = utterances = workgroups. */
int eab_conv_st_chain_plan(const eab_conv_desc* descs, int n, int* codes, int* lds_bytes, int* bf16);
int eab_conv_st_chain_run(const eab_conv_desc* dev_descs, const int* dev_codes, int n, int B, int lds_bytes, int bf16,
                          eab_stream_t stream);

/* Windowed twins of the non-conv ops (same semantics restricted to the rows of `win`; win.pos == NULL
 * = the whole utterance).  eab_lstm64_stream_f32 additionally carries the recurrent state: the hidden
 * state of row *pos-1 is read back from h_out, the cell state lives in c_state [B*F][64] (read when
 * *pos > 0, always written).  eab_zero_rows_f32 clears rows of a [B][T][row_floats] tensor. */
int eab_norm_act_win_f32(const float* a, const float* xfa, const float* slopea, const float* b,
                         const float* xfb, const float* slopeb, float* out, int B, int T, int rows_per_t, int C,
                         eab_time_window win, eab_stream_t stream);
int eab_lstm64_stream_f32(const float* x, const float* ln_g, const float* ln_b, float ln_eps, const float* wcat,
                          const float* bias, float* h_out, float* c_state, int B, int T, int F, int precision,
                          eab_time_window win, eab_stream_t stream);
int eab_bfw_filter_sum_win_f32(const float* y1, const float* w2, const float* b2, const float* x, float* out,
                               float* bfw, int B, int T, int F, int M, eab_time_window win, eab_stream_t stream);
int eab_zero_rows_f32(float* ptr, int B, int T, int row_floats, eab_time_window win, eab_stream_t stream);

/* K12 complete + K13: the whole w_dnn MLP of LSTM_BF (EaBNet.py:594-596,612-613) and the filter-and-sum in one
 * kernel:  y1 = relu(h W1^T + b1) (fp32 MFMA per 64-bin tile, never written to HBM), W = y1 W2^T + b2,
 * out = sum_m W_m X_m.   h [B][T][F][64] (the second LSTM's output), w1 [64][64], b1 [64]; the rest as
 * eab_bfw_filter_sum_win_f32.  w1 == b1 == NULL: y1 is read from `h` as is (the pointwise cnn / miso heads). */
int eab_mlp_bfw_filter_sum_f32(const float* h, const float* w1, const float* b1, const float* w2, const float* b2,
                               const float* x, float* out, float* bfw, int B, int T, int F, int M,
                               eab_time_window win, eab_stream_t stream);

/* ==========================================================================================================
 * Training: what autograd executes for train_distributed.py:221-228 (net(...), loss.backward())
 * on the reference, as hand-written kernels.  The training forward materialises every normalised activation
 * (convolutions take plain sources), keeps what the backward needs, and the backward is a second static op list.
 * Contractions: dgrad = eab_conv_f32 on the gradient with re-packed weights (the strided convolution's dgrad is
 * the gather form of a transposed convolution and vice versa), wgrad = eab_wgrad_f32.
 * ========================================================================================================== */

/* Parameter packing, one launch for the whole program: out[i] = flat[ia[i]] (+ flat[ib[i]] when ib != NULL); a
 * negative index contributes 0.  Forward: flat parameter vector -> every packed operand; backward: packed
 * gradient arena -> flat gradient (inverse table). */
int eab_gather_f32(const float* flat, const int32_t* ia, const int32_t* ib, float* out, long long n, eab_stream_t stream);

/* nn.InstanceNorm1d statistics of a materialised [B][P][C] tensor, optionally of prelu(x, slope) (S-TCM order,
 * EaBNet.py:545-547,559-560): xf[b][c] = (gamma*rstd, beta - mean*gamma*rstd), mr[b][c] = (mean, rstd). */
int eab_train_in_stats_f32(const float* x, const float* slope, int B, int P, int C, float eps, const float* gamma,
                           const float* beta, float* xf, float* mr, eab_stream_t stream);
/* The whole 1-D unit of an S-TCM in one launch: the statistics above (slope required) AND y = prelu(x)*scale + shift,
 * y [B][P][C] (EaBNet.py:545-547: nn.PReLU -> NormSwitch 1-D).  One workgroup per (b, 64 channels), two passes. */
int eab_train_in1d_f32(const float* x, const float* slope, int B, int P, int C, float eps, const float* gamma,
                       const float* beta, float* xf, float* mr, float* y, eab_stream_t stream);
/* Several such units on ONE input in one launch: output channel c (of C = k * xC) normalises prelu(x[..][c % xC], slope[c]) -- the
 * left and right branch norms of a SqueezedTCM (EaBNet.py:545-547, 559-560); view v = c / xC is written to its own contiguous
 * tensor y + v * B*P*xC; xf, mr [B][C][2]. */
int eab_train_in1d_multi_f32(const float* x, const float* slope, int B, int P, int C, int xC, float eps, const float* gamma,
                             const float* beta, float* xf, float* mr, float* y, eab_stream_t stream);
/* eab_in_finalize_f32 that also emits mr0 / mr1 [B][C][2] = (mean, rstd) for the backward pass (NULL = skip) */
int eab_in_finalize_mr_f32(const float* stats, int B, int C, int nsets, int stat_tiles, int count, float eps,
                           const float* gamma0, const float* beta0, float* xf0, const float* gamma1,
                           const float* beta1, float* xf1, float* mr0, float* mr1, eab_stream_t stream);
/* y = f(x) [+ add], f = prelu(x*scale+shift) (mode EAB_XF_NORM_PRELU) or prelu(x)*scale+shift (EAB_XF_PRELU_NORM);
 * x, add, y [B][P][C] */
int eab_train_norm_act_f32(const float* x, const float* xf, const float* slope, const float* add, float* y, int B, int P,
                           int C, int mode, eab_stream_t stream);
/* Backward of y = f(x) through the InstanceNorm statistics (per (b, c) over P) and the PReLU:
 *   dx = (acc_in ? acc_in : 0) + d loss / d x;   dgamma[c], dbeta[c], dslope[c] += their gradients.
 * sums: scratch [EAB_NB_SUM_COPIES][B][C][4] (the reduce pass spreads its atomics over the copies: hundreds of workgroups adding
 * into the 24 words of one cache line cost more than reading the tensors); OR-ing EAB_NB_SUMS_ZEROED into `mode` promises it is
 * zero on entry (saves the zero-fill launch).
 * mr == NULL (EAB_XF_NORM_PRELU only; eab_train_norm_act_f32: xf == NULL): no norm in front of the PReLU -- y = prelu(x),
 * dx = dy * prelu'(x), only dslope is accumulated (the plain U-Net's middle encoder layers, EaBNet.py:219-226).
 * Two launches (reduce, apply; the parameter gradients ride in the apply pass).  Reference: autograd of EaBNet.py:684-686 + nn.PReLU. */
#define EAB_NB_SUMS_ZEROED 0x100
#define EAB_NB_SUM_COPIES 8
/* OR-ed into the `mode` of eab_train_norm_act_f32 (y) / eab_train_norm_bwd_f32 (dx; acc_in must be NULL) and passed as `flags`
 * to eab_glu_bwd_ex_f32 (dz): the OUTPUT tensor is stored as bf16 (2-byte elements, same shape, round to nearest even).  The
 * bf16 training programs set it for tensors that only bf16 contractions read (eab_conv_desc.src_bf16, eab_wgrad_desc.bf16_mask):
 * those round their operands to bf16 anyway, so the numbers do not change -- the bytes halve. */
#define EAB_STORE_BF16 0x200
int eab_train_norm_bwd_f32(const float* dy, const float* x, const float* mr, const float* gamma, const float* beta,
                           const float* slope, float* sums, const float* acc_in, float* dx, float* dgamma, float* dbeta,
                           float* dslope, int B, int P, int C, int mode, eab_stream_t stream);
/* The same for the TWO units of the EAB_XF_PRELU_NORM order that read ONE xC-channel tensor x (forward:
 * eab_train_in1d_multi_f32): dy0 / dy1 [B][P][xC] = the views' output gradients; mr, gamma, slope, sums, dgamma, dbeta, dslope
 * [..][2 xC] view-major (sums: EAB_NB_SUM_COPIES copies, as above); dx [B][P][xC] = (acc_in) + both input gradients.  sums must be
 * zero on entry. */
int eab_train_norm_bwd_multi_f32(const float* dy0, const float* dy1, const float* x, const float* mr, const float* gamma,
                                 const float* slope, float* sums, const float* acc_in, float* dx, float* dgamma, float* dbeta,
                                 float* dslope, int B, int P, int xC, eab_stream_t stream);
/* GLU backward (EaBNet.py:459-460, 489-490): dy [rows][N/2], dump [rows][N] (eab_conv_desc.glu_dump) -> dz [rows][N]
 * in the packed column order of the forward convolution */
int eab_glu_bwd_f32(const float* dy, const float* dump, float* dz, long long rows, int N, eab_stream_t stream);
/* the same with flags: EAB_STORE_BF16 = dz is stored as bf16 (read by bf16 weight / data gradients only) */
int eab_glu_bwd_ex_f32(const float* dy, const float* dump, float* dz, long long rows, int N, int flags, eab_stream_t stream);
/* S-TCM gate z = a*sigmoid(r) (EaBNet.py:575) and its backward; n floats, n % 4 == 0 */
int eab_gate_fwd_f32(const float* a, const float* r, float* z, long long n, eab_stream_t stream);
int eab_gate_bwd_f32(const float* dz, const float* a, const float* r, float* da, float* dr, long long n, eab_stream_t stream);
int eab_add_f32(const float* a, const float* b, float* out, long long n, eab_stream_t stream);
/* dst = src as a kernel; src may be PINNED host memory (the upload of prepare_data, train_distributed.py:76-77) */
int eab_copy_f32(const float* src, float* dst, long long n, eab_stream_t stream);
int eab_relu_bwd_f32(const float* dy, const float* y, float* dx, long long n, eab_stream_t stream);
/* out[n] += sum over rows of x[row][n]  (bias gradients) */
int eab_colsum_f32(const float* x, float* out, long long rows, int N, eab_stream_t stream);
/* filter-and-sum with the weights in rows of `ld` floats (first 2M used; the training program keeps them in a
 * 64-column tile), and d loss / d W (EaBNet.py:114-117): dout [B][2][T][F], x [B][T][F][M][2] -> dw [B][T][F][ld]
 * (columns >= 2M zeroed) */
int eab_filter_sum_ld_f32(const float* w, const float* x, float* y, int B, int T, int F, int M, int ld, eab_stream_t stream);
int eab_filter_sum_bwd_f32(const float* dout, const float* x, float* dw, int B, int T, int F, int M, int ld, eab_stream_t stream);
/* LayerNorm(64) (EaBNet.py:598,608) materialised with its (mean, rstd) per row, and its backward (dg, db accumulated) */
int eab_layernorm64_fwd_f32(const float* x, const float* g, const float* b, float eps, float* y, float* mr, long long rows,
                            eab_stream_t stream);
int eab_layernorm64_bwd_f32(const float* dy, const float* x, const float* mr, const float* g, float* dx, float* dg,
                            float* db, long long rows, eab_stream_t stream);
/* one nn.LSTM(64->64) layer (EaBNet.py:610-611) that also stores the activated gates and cell states,
 * gates [B*F][T][5][64] = (i, f, g, o, c); and the reverse-time pass: dh_out [B][T][F][64] (gradient w.r.t. every
 * h_t) -> dgates [B][T][F][256] (gradient w.r.t. the gate pre-activations, row g*64+u), from which
 * dx = dgates W_ih (eab_conv_f32), dW_ih / dW_hh / db (eab_wgrad_f32 with taps dt = 0 / -1, eab_colsum_f32). */
int eab_lstm64_train_fwd_f32(const float* x, const float* wcat, const float* bias, float* h_out, float* gates, int B, int T,
                             int F, eab_stream_t stream);
int eab_lstm64_bwd_f32(const float* gates, const float* dh_out, const float* wcat, float* dgates, int B, int T, int F,
                       eab_stream_t stream);
/* the same with `precision` EAB_PREC_F32 or EAB_PREC_BF16 (the bf16 training programs, BASELINE configs[3]): the recurrent
 * products -- [x_t | h_{t-1}] W in the forward, dgates_t W_hh in the backward -- take their operands rounded to bf16 on the
 * 16-bit matrix cores with fp32 accumulation, as torch.autocast(bfloat16) runs nn.LSTM; activations, cell state, carries and
 * every stored tensor stay fp32. */
int eab_lstm64_train_fwd_prec_f32(const float* x, const float* wcat, const float* bias, float* h_out, float* gates, int B, int T,
                                  int F, int precision, eab_stream_t stream);
int eab_lstm64_bwd_prec_f32(const float* gates, const float* dh_out, const float* wcat, float* dgates, int B, int T, int F,
                            int precision, eab_stream_t stream);

/* Weight gradient on fp32 MFMA:
 *   dw[n][tap*UPT*16 + c] += sum_{b,t,o} dz[b][t][o*ostride+ophase][n] * src[b][t+dt_tap][o*istride+ioff_tap][c]
 * with the forward geometry of eab_conv_desc (plain sources, optional concatenation), tap-major columns,
 * Kpad = ntaps*UPT*16, UPT = ceil((C0+C1)/16).  Accumulates with atomics: the caller zeroes dw. */
/* (eab_wgrad_desc is declared next to eab_op above) */
int eab_wgrad_f32(const eab_wgrad_desc* d, eab_stream_t stream);
/* Up to 24 weight gradients of IDENTICAL geometry (every field but the five pointers) in one launch: descs[k] is found
 * `stride_bytes` after descs[k-1] (so the descriptors may sit inside an eab_op array).  eab_wgrad_batchable tells how many
 * of descs[0..n) can join descs[0].  eab_run_program batches consecutive EAB_OP_WGRAD ops this way. */
int eab_wgrad_batchable(const eab_wgrad_desc* descs, int n, int stride_bytes);
int eab_wgrad_batch_f32(const eab_wgrad_desc* descs, int n, int stride_bytes, eab_stream_t stream);

/* op kinds of the training programs (eab_run_program); field use:
 *  GATHER       p = {flat, ia, ib, out}                 i = {n_lo, n_hi}
 *  IN_STATS     p = {x, slope, gamma, beta, xf, mr, y}  i = {B, P, C, xC}      f = {eps}     (y != NULL: eab_train_in1d_f32; xC > 0: _multi_)
 *  IN_FINALIZE  as before, plus p[7], p[8] = mr0, mr1
 *  TR_NORM_ACT  p = {x, xf, slope, add, y}              i = {B, P, C, mode}
 *  NORM_BWD     p = {dy, x, mr, gamma, beta, slope, sums, acc_in, dx, dgamma, dbeta, dslope}   i = {B, P, C, mode, xC}
 *               (xC > 0: eab_train_norm_bwd_multi_f32 with p[4] = dy1 instead of beta)
 *  GLU_BWD      p = {dy, dump, dz}                      i = {rows_lo, rows_hi, N}
 *  GATE_FWD     p = {a, r, z}        GATE_BWD p = {dz, a, r, da, dr}       i = {n_lo, n_hi}
 *  ADD          p = {a, b, out}      RELU_BWD p = {dy, y, dx}              i = {n_lo, n_hi}
 *  COLSUM       p = {x, out}                            i = {rows_lo, rows_hi, N}
 *  FILTER_SUM   p = {w, x, y}        FS_BWD   p = {dout, x, dw}            i = {B, T, F, M, ld}
 *  LN_FWD       p = {x, g, b, y, mr} f = {eps}          i = {rows_lo, rows_hi}
 *  LN_BWD       p = {dy, x, mr, g, dx, dg, db}          i = {rows_lo, rows_hi}
 *  LSTM_TRAIN   p = {x, wcat, bias, h_out, gates}       i = {B, T, F}
 *  LSTM_BWD     p = {gates, dh_out, wcat, dgates}       i = {B, T, F}
 *  GAG_CRM_BWD  p = {pre, g, dplanar, dpre_out, acc_in, dg, dr, di, dpre}   i = {B, T, F, ld, lin_ld, act}
 *  WGRAD        the `wgrad` member */
#define EAB_OP_GATHER      16
#define EAB_OP_IN_STATS    17
#define EAB_OP_TR_NORM_ACT 18
#define EAB_OP_NORM_BWD    19
#define EAB_OP_GLU_BWD     20
#define EAB_OP_GATE_FWD    21
#define EAB_OP_GATE_BWD    22
#define EAB_OP_ADD         23
#define EAB_OP_RELU_BWD    24
#define EAB_OP_COLSUM      25
#define EAB_OP_FILTER_SUM  26
#define EAB_OP_FS_BWD      27
#define EAB_OP_LN_FWD      28
#define EAB_OP_LN_BWD      29
#define EAB_OP_LSTM_TRAIN  30
#define EAB_OP_LSTM_BWD    31
#define EAB_OP_WGRAD       32

/* --------------------------------------------------------------------------
 * Cumulative LayerNorm: CumulativeLayerNorm1d / 2d, EaBNet.py:696-769, the causal norm
 * NormSwitch(norm_type="cLN") is meant to build (its constructor passes the string dim_size as num_features,
 * EaBNet.py:689,691; this is the class with the channel count passed).  x [B][T][P] with P = F*C values per frame,
 * channel = index % C.
 *   eab_cln_stats_f32: mr[b][t] = (cum_mean, 1/sqrt(cum_var + eps)) over all channels, bins and frames <= t, of x or --
 *     slope != NULL -- of prelu(x, slope[c]) (S-TCM order).  sums: scratch [B][T][2] doubles; state: [B][2] doubles,
 *     the running sums carried between streaming chunks (required with a window, optional otherwise).
 *   eab_cln_apply_f32: y = prelu(gain_c (x-mean) rstd + bias_c, slope_c) [+ add]   (mode EAB_XF_NORM_PRELU)
 *                      y = gain_c (prelu(x, slope_c) - mean) rstd + bias_c         (mode EAB_XF_PRELU_NORM)
 *   eab_gate_rows_f32: z = a * sigmoid(r) (S-TCM gate, EaBNet.py:575) on [B][T][row_floats]
 * ------------------------------------------------------------------------ */
int eab_cln_stats_f32(const float* x, const float* slope, int B, int T, int P, int C, float eps, double* sums, double* state,
                      float* mr, eab_time_window win, eab_stream_t stream);
/* Backward of the unit (training; whole utterance): dx = (acc_in ? acc_in : 0) + d loss / d x for y = eab_cln_apply_f32(x, mr,
 * gain, bias, slope) in `mode`, dy = d loss / d y.  Scratch: rowsums [B][T][2] doubles, ab [B][T][2] floats; part [B*T][3][C]
 * receives the per-row parameter-gradient sums (gain | bias | slope), to be summed over the rows by eab_colsum_f32 with
 * N = 3 C.  Three launches (row sums, reverse scan over t, apply).  C must divide 1024.  Reference: autograd of
 * CumulativeLayerNorm1d / 2d, EaBNet.py:713-733, 752-769. */
int eab_train_cln_bwd_f32(const float* dy, const float* x, const float* mr, const float* gain, const float* bias,
                          const float* slope, double* rowsums, float* ab, float* part, const float* acc_in, float* dx,
                          int B, int T, int P, int C, int mode, eab_stream_t stream);
int eab_cln_apply_f32(const float* x, const float* mr, const float* gain, const float* bias, const float* slope,
                      const float* add, float* y, int B, int T, int P, int C, int mode, eab_time_window win,
                      eab_stream_t stream);
int eab_gate_rows_f32(const float* a, const float* r, float* z, int B, int T, int row_floats, eab_time_window win,
                      eab_stream_t stream);
/* One frame of a streaming step (win.count == 1): eab_cln_stats_f32 followed by eab_cln_apply_f32 on the same x in ONE launch
 * (one workgroup per utterance: frame sums -> running sums and mr[b][t] -> normalised frame).  Bit-identical to the two calls
 * (same reduction tree, same expressions); stat_slope = the statistics' PReLU slope (EAB_XF_PRELU_NORM) or NULL. */
int eab_cln_step_f32(const float* x, const float* stat_slope, double* sums, double* state, float* mr, const float* gain,
                     const float* bias, const float* slope, const float* add, float* y, int B, int T, int P, int C, int mode,
                     float eps, eab_time_window win, eab_stream_t stream);
/* program ops:  CLN_STATS  p = {x, slope, sums, state, mr}              i = {B, T, P, C}        f = {eps}
 *               CLN_APPLY  p = {x, mr, gain, bias, slope, add, y}       i = {B, T, P, C, mode}
 *               GATE_ROWS  p = {a, r, z}                                i = {B, T, row_floats}       (all three windowed) */
#define EAB_OP_CLN_STATS 33
#define EAB_OP_CLN_APPLY 34
#define EAB_OP_GATE_ROWS 35
#define EAB_OP_GAG_CRM_BWD 36
#define EAB_OP_CLN_STEP 38  /* eab_cln_step_f32: p = {x, stat_slope, sums, state, mr, gain, bias, slope, add, y}, i = {B, T, P, C, mode}, f = {eps} */
#define EAB_OP_CLN_BWD 37   /* eab_train_cln_bwd_f32: p = {dy, x, mr, gain, bias, slope, rowsums, ab, part, acc_in, dx}, i = {B, T, P, C, mode} */

/* struct-layout handshake for foreign-function mirrors of the structs above */
int eab_sizeof_conv_desc(void);
int eab_sizeof_op(void);
int eab_sizeof_wgrad_desc(void);

#ifdef __cplusplus
}
#endif
#endif /* EABNET_HIP_H */
