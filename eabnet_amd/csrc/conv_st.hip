// Small-tile convolution for the latency-bound layers of the path: the S-TCN's 1-D convolutions (reference
// EaBNet.py:549,558,564,570,575,577: 54 launches per forward on [B][T][256|64] tensors) and the 64-column unit
// convolutions / transposed convolutions of the inner U-Nets on few frequency bins (EaBNet.py:402,423,425).
//
// Why a second kernel next to conv_gemm.hip.  Those launches carry 0.2 .. 2 GFLOP each: a few microseconds of matrix
// work for the whole chip.  conv_gemm_kernel walks K in stages (global load -> LDS -> barrier -> MFMA, one stage of
// prefetch), so a workgroup pays one memory round trip PER STAGE (4 .. 20 of them), on 64- or 128-row tiles that give
// the S-TCN 112 workgroups for 256 CUs.  Here a tile is 16 / 32 / 64 output rows and a workgroup pays ONE round trip:
//   * A (activations): every tap's gathered 16-channel units of the tile's rows -- the WHOLE K extent -- are fetched
//     in one burst, transformed once (fused InstanceNorm-affine / PReLU of the producer) and laid out K-major in LDS;
//   * B (weights) never touches LDS: the host packs them in MFMA-fragment order (EAB_KORDER_FRAG), so a wave's
//     b128 loads are fully coalesced 1-KB reads of exactly the operands its v_mfma_f32_16x16x4_f32 needs, prefetched
//     several K steps ahead straight from L2 into registers; they are issued before the A burst;
//   * one barrier, then nothing but ds_read_b64 + MFMA; latency is hidden by several small workgroups per CU.
// Wave w of the 4 owns output columns [16*NCB*w, 16*NCB*(w+1)) for ALL rows of the tile, so the InstanceNorm partials
// of a column are merged inside one wave (no LDS reduction, no second barrier).
//
// MFMA operand maps (MI355X guide, fragment layout): lane l = (i = l & 15, kq = l >> 4) supplies A[row i][k = kq] and
// B[k = kq][col i]; C/D: col = l & 15, row = 4*(l >> 4) + reg.  K is walked in steps of 16 ("m2"): lane (i, kq) reads
// floats [16 m2 + 8 ms + 2 kq, +2) of row i for ms = 0, 1 as ONE ds_read_b64 each and feeds them to two MFMAs, i.e.
// MFMA (m2, ms, e) contracts k = 16 m2 + 8 ms + 2 kq + e over kq = 0..3; B is packed to the same assignment.  LDS rows
// are Kpad + 4 floats: (Kpad + 4)/2 = 2*odd 8-byte slots, so the 32 lanes of a ds_read_b64 group (16 rows x 2 kq) hit
// 32 different slots -- conflict free.
//
// Every output row is one fmaf chain in a fixed k order, independent of the tile it falls into: streamed frames stay
// bit-identical to the offline pass (eab_time_window), as with conv_gemm_kernel.
#include "common.h"
#include <atomic>
#include <type_traits>

#define ST_THREADS 256
#define ST_OOB 0x80000000u
#define ST_XFC 128          // max channels of a source that carries a fused transform
#define ST_PLAIN 0
#define ST_DUAL 1           // EAB_EPI_DUALGATE: value / gate columns see the same source through two transforms
#define ST_GLU 2            // EAB_EPI_GLU: value / gate columns of ONE convolution (GateConv2d / GateConvTranspose2d)

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float st_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }

// q / n for 0 <= q < 2^22 via the fp32 reciprocal, exact after one correction (as in conv_gemm.hip)
__device__ __forceinline__ int st_div(int q, int n, float inv_n) {
    int t = (int)((float)q * inv_n);
    if (t * n > q) --t;
    if ((t + 1) * n <= q) ++t;
    return t;
}

// Welford/Chan merge, (0,*,*) neutral
__device__ __forceinline__ void st_merge(float& n, float& mean, float& m2, float nb, float meanb, float m2b) {
    // branch-free; counts are small integers, so rcp (1 ulp) instead of an IEEE division chain is exact enough for a
    // quantity that only feeds a normalisation -- and the same bits in every workgroup, which is what matters
    const float nt = n + nb;
    const float fb = nt > 0.0f ? nb * __builtin_amdgcn_rcpf(nt) : 0.0f;
    const float delta = meanb - mean;
    mean = fmaf(delta, fb, mean);
    m2 = m2 + m2b + delta * delta * n * fb;
    n = nt;
}

typedef __bf16 st_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 st_bf16x2 __attribute__((ext_vector_type(2)));
// two fp32 -> packed bf16 (round to nearest even: v_cvt_pk_bf16_f32)
__device__ __forceinline__ unsigned st_bf2(float x0, float x1) {
    const st_bf16x2 v = {(__bf16)x0, (__bf16)x1};
    return __builtin_bit_cast(unsigned, v);
}

// buffer descriptor from wave-uniform inputs, made provably uniform for the compiler (MI355X guide, T20)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t st_rsrc(const float* p, unsigned bytes) {
    const unsigned long long a = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    const unsigned nb = __builtin_amdgcn_readfirstlane(bytes);
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<float*>(((unsigned long long)hi << 32) | lo), 0, nb, 0x00020000);
}

template <int XF>
__device__ __forceinline__ f32x4 st_xform(f32x4 v, f32x4 sh01, f32x4 sh23, f32x4 sl) {
    const float sc[4] = {sh01[0], sh01[2], sh23[0], sh23[2]};
    const float sf[4] = {sh01[1], sh01[3], sh23[1], sh23[3]};
    f32x4 r;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (XF == EAB_XF_NORM_PRELU)
            r[j] = eab_prelu(fmaf(v[j], sc[j], sf[j]), sl[j]);
        else
            r[j] = fmaf(eab_prelu(v[j], sl[j]), sc[j], sf[j]);
    }
    return r;
}

// dynamic LDS: [NA][BM][LD] floats of A, then the tables
struct StTables {
    float xft[2][ST_XFC][2];
    float xsl[2][ST_XFC];
    int2 tap[EAB_MAX_TAPS];     // (dt, ioff) of the launch's (phase's) taps
    double fin[3][ST_THREADS];   // slice partials of the in-kernel InstanceNorm finalisation
};

// workgroups per CU the register budget is cut for: two, except where 160 registers of resident weights (N = 128, K up to
// 320) plus a 32-row tile's gather and transform state do not fit 256 registers (no instantiation may spill)
template <int RB, int NCB, int MODE, int XF, bool BF>
constexpr int st_wgs_per_cu() {
    return (MODE == ST_DUAL || (RB == 2 && NCB == 2 && MODE == ST_PLAIN && (XF != EAB_XF_NONE || BF))) ? 1 : 2;
}

template <int RB, int NCB, int MODE, int XF, bool BF>
__global__ __launch_bounds__(ST_THREADS, (st_wgs_per_cu<RB, NCB, MODE, XF, BF>())) void conv_st_kernel(const eab_conv_desc d) {
    const unsigned block = blockIdx.x, grid = gridDim.x;
#include "conv_st_body.inc"
}

// the same tile body on a descriptor in device memory (one step of conv_st_chain_kernel)
template <int RB, int NCB, int MODE, int XF, bool BF>
__device__ __forceinline__ void conv_st_chain_step(const eab_conv_desc* __restrict__ dp, const unsigned block, const unsigned grid) {
    const eab_conv_desc& d = *dp;
#include "conv_st_body.inc"
}

// ---- a chain of launches in ONE launch (streaming S-TCN) -------------------------------------------------------------------
// A frame-synchronous streaming step gives every 1-D convolution of the S-TCN ONE 16-row tile per utterance, and launch k+1
// reads nothing but what the same utterance's tile of launch k wrote (BatchNorm in eval mode is a static table: no
// statistics cross utterances or tiles).  So the 37 launches of the 18 S-TCMs need no grid-wide ordering at all: one
// workgroup per utterance walks the descriptors (device memory, uploaded when the program is bound) and runs the SAME tile
// body for each -- bit-identical to the separate launches by construction -- with a workgroup barrier and a workgroup-scope
// release / acquire pair between steps (the tile's stores must be visible to its own later loads through the vector L1).
// Every wave runs the loop to n: no early exit, the grid drains.
#define ST_CHAIN_IN 0      // <1, 1, PLAIN, XF_NONE>: first in_conv (and every in_conv of the bf16 form)
#define ST_CHAIN_LR 1      // <1, 2, DUAL, XF_PRELU_NORM>: the gated branch pair
#define ST_CHAIN_OUT 2     // <1, 4, PLAIN, XF_PRELU_NORM>: out_conv (+ fused in_conv of the next S-TCM, fp32)
template <bool BF>
__global__ __launch_bounds__(ST_THREADS, 1) void conv_st_chain_kernel(const eab_conv_desc* __restrict__ descs,
                                                                       const int* __restrict__ codes, int n) {
    for (int k = 0; k < n; ++k) {
        const eab_conv_desc* d = descs + k;
        const int code = __builtin_amdgcn_readfirstlane(codes[k]);
        if (code == ST_CHAIN_IN)
            conv_st_chain_step<1, 1, ST_PLAIN, EAB_XF_NONE, BF>(d, blockIdx.x, gridDim.x);
        else if (code == ST_CHAIN_LR)
            conv_st_chain_step<1, 2, ST_DUAL, EAB_XF_PRELU_NORM, BF>(d, blockIdx.x, gridDim.x);
        else
            conv_st_chain_step<1, 4, ST_PLAIN, EAB_XF_PRELU_NORM, BF>(d, blockIdx.x, gridDim.x);
        // workgroup scope: the waves of a workgroup share the CU's vector L1 (write-through), so ordering the stores before
        // the barrier and the loads after it is all that is needed -- an agent-scope pair would write back and invalidate L2
        // (the weights of the next steps with it) 37 times per launch
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
}

// eab_conv_st_chain_plan: validation without a launch -- st_launch_p reports the instantiation it would have launched
struct StPlan {
    int code;        // ST_CHAIN_* or -1 (an instantiation the chain kernel does not carry)
    size_t lds;
    bool bf;
    long long tiles_per_b;
};
static thread_local StPlan* st_plan = nullptr;

// the current device, as an index into the per-device "attribute set" flags (-1: none / out of range)
#define ST_MAX_DEVICES 64
static int st_device() {
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= ST_MAX_DEVICES) return -1;
    return dev;
}

// LDS bytes of a launch
static size_t st_lds_bytes(const eab_conv_desc* d, int bm, bool dual) {
    const bool bf = d->precision == EAB_PREC_BF16;
    return (size_t)(dual ? 2 : 1) * bm * (bf ? (d->Kpad + 8) * 2 : (d->Kpad + 4) * 4) + sizeof(StTables) +
           (d->f2_w ? (size_t)bm * (256 + 4) * sizeof(float) : 0);
}

template <int RB, int NCB, int MODE, int XF, bool BF>
static int st_launch_p(const eab_conv_desc* d, hipStream_t s) {
    constexpr int BM = 16 * RB;
    const int Tw = d->win.pos ? d->win.count : d->T;
    const long long tiles = ((long long)Tw * d->No + BM - 1) / BM + (d->ph1_No > 0 ? ((long long)Tw * d->ph1_No + BM - 1) / BM : 0);
    const size_t lds = st_lds_bytes(d, BM, MODE == ST_DUAL);
    if (lds > 160 * 1024) return EAB_EUNSUPPORTED;
    if (st_plan) {                                       // planning a chain: report, do not launch
        st_plan->code = RB != 1                                                      ? -1
                        : (NCB == 1 && MODE == ST_PLAIN && XF == EAB_XF_NONE)        ? ST_CHAIN_IN
                        : (NCB == 2 && MODE == ST_DUAL && XF == EAB_XF_PRELU_NORM)   ? ST_CHAIN_LR
                        : (NCB == 4 && MODE == ST_PLAIN && XF == EAB_XF_PRELU_NORM)  ? ST_CHAIN_OUT
                                                                                     : -1;
        st_plan->lds = lds;
        st_plan->bf = BF;
        st_plan->tiles_per_b = tiles;
        return EAB_OK;
    }
    // per instantiation AND device: allow more than the default 64 KB of dynamic LDS (the attribute belongs to the
    // function on one device; a second GPU in the process needs its own call)
    static std::atomic<bool> attr_set[ST_MAX_DEVICES];
    const int dev = st_device();
    if (dev < 0) return EAB_EINVAL;
    if (!attr_set[dev].load(std::memory_order_acquire)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_st_kernel<RB, NCB, MODE, XF, BF>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return eab_hip_status(e);
        attr_set[dev].store(true, std::memory_order_release);
    }
    hipLaunchKernelGGL((conv_st_kernel<RB, NCB, MODE, XF, BF>), dim3((unsigned)(d->B * tiles)), dim3(ST_THREADS), lds, s, *d);
    EAB_RETURN_LAUNCH_STATUS();
}

template <int RB, int NCB, int MODE, int XF>
static int st_launch(const eab_conv_desc* d, hipStream_t s) {
    return d->precision == EAB_PREC_BF16 ? st_launch_p<RB, NCB, MODE, XF, true>(d, s) : st_launch_p<RB, NCB, MODE, XF, false>(d, s);
}

template <int NCB, int MODE, int XF>
static int st_pick_rb(const eab_conv_desc* d, hipStream_t s) {
    switch (d->bm) {
        case 16: return st_launch<1, NCB, MODE, XF>(d, s);
        case 32: return st_launch<2, NCB, MODE, XF>(d, s);
        case 64:
            if constexpr (NCB == 1) return st_launch<4, NCB, MODE, XF>(d, s);
            return EAB_EUNSUPPORTED;
        default: return EAB_EUNSUPPORTED;
    }
}

// eab_conv_f32 with d->korder == EAB_KORDER_FRAG lands here (arguments common to both kernels are checked there)
int eab_conv_st(const eab_conv_desc* d, hipStream_t s) {
    EAB_CHECK_ARG(d->precision == EAB_PREC_F32 || d->precision == EAB_PREC_BF16);
    EAB_CHECK_ARG(d->precision == EAB_PREC_F32 || (d->Kpad % 32 == 0 && (d->ph1_No == 0 || d->ph1_Kpad % 32 == 0)));
    EAB_CHECK_ARG(d->bm == 16 || d->bm == 32 || d->bm == 64);
    EAB_CHECK_ARG((d->C0 == 64 || d->C0 == 128 || d->C0 == 256) && (d->C1 == 0 || d->C1 == 64 || d->C1 == 128 || d->C1 == 256));
    EAB_CHECK_ARG(d->epi == EAB_EPI_LINEAR || d->epi == EAB_EPI_RELU || d->epi == EAB_EPI_ADD || d->epi == EAB_EPI_DUALGATE ||
                  d->epi == EAB_EPI_GLU);
    EAB_CHECK_ARG(d->fz_counter == nullptr);
#ifndef EAB_ST_STAMPS
    // this kernel writes no GLU factor dump (the training forward's gated convolutions run on conv_gemm_kernel): refuse
    // rather than leave the buffer eab_glu_bwd_f32 reads unwritten
    if (d->glu_dump != nullptr) return EAB_EUNSUPPORTED;
#endif
    const bool dual = d->epi == EAB_EPI_DUALGATE;
    if (d->ph1_No > 0) {                                 // second output-column phase of a transposed convolution
        EAB_CHECK_ARG(d->ph1_w && d->ph1_ntaps > 0 && d->ph1_ntaps <= EAB_MAX_TAPS && d->ph1_Kpad > 0 && d->ph1_Kpad <= d->Kpad);
        const int upt = (d->C0 + d->C1 + 15) / 16;
        EAB_CHECK_ARG(d->ph1_Kpad == d->ph1_ntaps * upt * 16);
        EAB_CHECK_ARG(d->ph1_ophase >= 0 && d->ph1_ophase < d->ostride && (d->ph1_No - 1) * d->ostride + d->ph1_ophase < d->Fout);
        EAB_CHECK_ARG(d->stats == nullptr || d->stat_tile0 == 0);
        for (int j = 0; j < d->ph1_ntaps; ++j) EAB_CHECK_ARG(d->ph1_dt[j] < (1 << 20) && d->ph1_dt[j] > -(1 << 20));
        if (d->win.pos)
            for (int j = 0; j < d->ph1_ntaps; ++j) EAB_CHECK_ARG(d->ph1_dt[j] <= 0);
        EAB_CHECK_ARG((long long)d->T * d->ph1_No < (1ll << 22));
    }
    if (d->f2_w) {                                       // fused second 1x1 convolution on this launch's output rows
        EAB_CHECK_ARG(d->precision == EAB_PREC_F32 && d->N == 256 && !dual && d->f2_N == 64 && d->f2_dst);
        EAB_CHECK_ARG(d->No == 1 && d->Fin == 1 && d->Fout == 1 && d->ph1_No == 0 && d->ostride == 1 && d->ophase == 0);
        EAB_CHECK_ARG(d->f2_nsets >= 0 && d->f2_nsets <= 2 && (d->f2_nsets == 0) == (d->f2_stats == nullptr));
        EAB_CHECK_ARG(d->f2_stats == nullptr || (d->win.pos == nullptr && d->f2_stat_tiles >= eab_conv_tiles(d->T, 1, d->bm)));
    }
    const bool has_xf = d->xf_mode != EAB_XF_NONE && (d->xf0 || d->xf1 || d->fin_stats);
    const int xf = has_xf ? d->xf_mode : EAB_XF_NONE;
    if (xf != EAB_XF_NONE) EAB_CHECK_ARG(d->C0 <= ST_XFC && d->C1 <= ST_XFC);
    if (d->fin_stats) EAB_CHECK_ARG(d->fin_nsets * d->C0 <= ST_THREADS);
    // a wave's weights are held in registers: the whole K extent (bf16 form, N = 256) or passes of 320
    const bool glu = d->epi == EAB_EPI_GLU;
    if (d->Kpad > (d->N == 256 ? 64 : (glu && d->precision == EAB_PREC_F32) ? 1024 : 320)) return EAB_EUNSUPPORTED;
    if (d->N == 128 && d->bm == 64) return EAB_EUNSUPPORTED;
    if (dual) {
        if (d->N != 128 || xf != EAB_XF_PRELU_NORM) return EAB_EUNSUPPORTED;
        return st_pick_rb<2, ST_DUAL, EAB_XF_PRELU_NORM>(d, s);
    }
    if (glu) {                                           // gated convolution on materialised sources, exact fp32
        if (d->N != 128 || xf != EAB_XF_NONE || d->precision != EAB_PREC_F32) return EAB_EUNSUPPORTED;
        return st_pick_rb<2, ST_GLU, EAB_XF_NONE>(d, s);
    }
#define ST_DISPATCH_XF(NCB_)                                                   \
    (xf == EAB_XF_NONE        ? st_pick_rb<NCB_, ST_PLAIN, EAB_XF_NONE>(d, s) \
     : xf == EAB_XF_NORM_PRELU ? st_pick_rb<NCB_, ST_PLAIN, EAB_XF_NORM_PRELU>(d, s) \
                               : st_pick_rb<NCB_, ST_PLAIN, EAB_XF_PRELU_NORM>(d, s))
    if (d->N == 64) return ST_DISPATCH_XF(1);
    if (d->N == 128) return ST_DISPATCH_XF(2);
    if (d->N == 256) return ST_DISPATCH_XF(4);
#undef ST_DISPATCH_XF
    return EAB_EUNSUPPORTED;
}


// ---- chain entry points (include/eabnet_hip.h) -------------------------------------------------------------------------------
extern "C" int eab_conv_st_chain_plan(const eab_conv_desc* descs, int n, int* codes, int* lds_bytes, int* bf16) {
    EAB_CHECK_ARG(descs && n > 0 && codes && lds_bytes && bf16);
    size_t lds = 0;
    int bf = -1;
    for (int k = 0; k < n; ++k) {
        const eab_conv_desc* d = &descs[k];
        if (d->korder != EAB_KORDER_FRAG || d->win.pos == nullptr || d->B != descs[0].B) return EAB_EUNSUPPORTED;
        // no statistics, no cross-tile hand-over of any kind, and nothing diagnostic
        if (d->stats || d->fin_stats || d->fz_counter || d->f2_stats || d->glu_dump || d->ph1_No > 0) return EAB_EUNSUPPORTED;
        StPlan plan = {-1, 0, false, 0};
        st_plan = &plan;
        const int rc = eab_conv_f32(d, nullptr);         // every argument check of a real launch
        st_plan = nullptr;
        if (rc != EAB_OK) return rc;
        if (plan.code < 0 || plan.tiles_per_b != 1) return EAB_EUNSUPPORTED;
        if (bf >= 0 && bf != (int)plan.bf) return EAB_EUNSUPPORTED;
        bf = plan.bf;
        codes[k] = plan.code;
        lds = plan.lds > lds ? plan.lds : lds;
    }
    *lds_bytes = (int)lds;
    *bf16 = bf;
    return EAB_OK;
}

extern "C" int eab_conv_st_chain_run(const eab_conv_desc* dev_descs, const int* dev_codes, int n, int B, int lds_bytes, int bf16,
                                     eab_stream_t stream) {
    EAB_CHECK_ARG(dev_descs && dev_codes && n > 0 && B > 0 && lds_bytes > 0 && lds_bytes <= 160 * 1024);
    static std::atomic<bool> attr_set[2][ST_MAX_DEVICES];
    const void* fn = bf16 ? reinterpret_cast<const void*>(&conv_st_chain_kernel<true>)
                          : reinterpret_cast<const void*>(&conv_st_chain_kernel<false>);
    const int dev = st_device();
    if (dev < 0) return EAB_EINVAL;
    if (!attr_set[bf16 ? 1 : 0][dev].load(std::memory_order_acquire)) {
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return eab_hip_status(e);
        attr_set[bf16 ? 1 : 0][dev].store(true, std::memory_order_release);
    }
    if (bf16)
        hipLaunchKernelGGL(conv_st_chain_kernel<true>, dim3((unsigned)B), dim3(ST_THREADS), (size_t)lds_bytes, eab_stream(stream),
                           dev_descs, dev_codes, n);
    else
        hipLaunchKernelGGL(conv_st_chain_kernel<false>, dim3((unsigned)B), dim3(ST_THREADS), (size_t)lds_bytes, eab_stream(stream),
                           dev_descs, dev_codes, n);
    EAB_RETURN_LAUNCH_STATUS();
}
