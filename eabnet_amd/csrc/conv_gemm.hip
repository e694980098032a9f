// K4-K7, K9, K12a: gather-GEMM convolution on fp32 MFMA (v_mfma_f32_32x32x2_f32).
//
// GEMM view of every convolution on the path (reference call-sites in
// include/eabnet_hip.h):
//   rows    = output positions (b, t, o) of ONE batch element per tile (tiles
//             never straddle b: the InstanceNorm partials are per (b, channel)),
//             flattened q = t*No + o, BM consecutive q per workgroup;
//   columns = output channels N (64 / 128 / 256);
//   K       = taps x input channels, walked in "units" of 16 channels of one tap:
//             a unit of one row is 64 contiguous bytes of a channels-last source.
//
// Workgroup = 4 waves (2x2), wave tile (32*MI) x (32*NI); KU units (16*KU of K)
// per pipeline stage.  LDS tiles are [rows][16*KU+4] floats: the +4 (one
// ds_read_b128 width) makes the 16-byte-slot stride odd, so the 64 lanes'
// fragment reads are conflict free (MI355X_MICROARCH §LDS).  Lane (i = l&31,
// h = l>>5) reads floats [8g+4h, 8g+4h+4) of its row as ONE ds_read_b128 and feeds
// them to four successive MFMA k-steps; A and B use the same k permutation, so
// the sum is unchanged.
//
// Straight-line main loop: gathers are raw buffer loads whose descriptor spans
// ONE batch element of the source; a tap that falls outside the tensor (t+dt < 0,
// f outside [0,Fin), rows past the tile end) gets an out-of-range offset and the
// hardware returns 0 -- no exec-mask branches, so the compiler can keep the
// stage-(s+1) loads in flight across the MFMAs of stage s.  The producer's
// InstanceNorm affine + PReLU (compile-time XF) is applied registers->LDS.
//
// fp32-in MFMA is exact fp32 (fmaf chain), 64 FLOP/clk/SIMD: one MFMA occupies
// its SIMD for 64 cycles while needing one A and one B VGPR, so LDS and staging
// traffic are far below their limits: roofline "mfma", fp32 dense 157.3 TFLOP/s.
#include "common.h"
#include <type_traits>

#define CG_THREADS 256
#define CG_OOB 0x80000000u   // > any legal byte offset inside one batch element (host checks < 2^31)

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

#define CG_PLAIN 0
#define CG_GLU 1     // value/gate columns of ONE GEMM (GateConv2d / GateConvTranspose2d)
#define CG_DUAL 2    // value/gate columns see the SAME source through two transforms (S-TCM branches)
#define CG_PH2 3     // EAB_EPI_PHASE2: the two output-column phases of a stride-2 transposed convolution from ONE staged input
                     // patch: column pairs (phase 0, phase 1) of a channel share a lane like (value, gate) of the GLU form
#define CG_XFC 128   // max channels of a source that carries a fused transform

#define CG_PMAX 352  // input positions of one patch (PATCH mode): 352 * 80 B = 27.5 KB, three workgroups per CU

template <int MI, int NI, int KU, int MODE, bool PATCH>
struct CgSmem {
    static constexpr int BM = 64 * MI, BN = 64 * NI, LDK = 16 * KU + 4;
    static constexpr int NA = MODE == CG_DUAL ? 2 : 1;
    static constexpr int ATILE = BM * LDK;                       // one A tile (floats)
    // gather mode: NA x 2 (double-buffered) A tiles of BM output rows;
    // patch mode: ONE tile of CG_PMAX input positions, re-read by every tap of a channel chunk
    float a[PATCH ? CG_PMAX * LDK : NA * 2 * ATILE];
    float b[2][BN * LDK];
    float xft[2][CG_XFC][2];     // (scale, shift) per channel: table 0 = src0 / left, 1 = src1 / right
    float xsl[2][CG_XFC];        // PReLU slopes
    int dt[EAB_MAX_TAPS];
    int ioff[EAB_MAX_TAPS];
    int last;                    // fused finalisation: this workgroup wrote the last partial of its batch element
};

template <int XF>
__device__ __forceinline__ f32x4 cg_xform(f32x4 v, f32x4 sh01, f32x4 sh23, f32x4 sl) {
    // sh01 = (scale0, shift0, scale1, shift1), sh23 likewise for channels 2,3
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

// Welford/Chan merge of two partial (count, mean, M2) statistics; (0,*,*) is the neutral element
__device__ __forceinline__ void cg_merge(float& n, float& mean, float& m2, float nb, float meanb, float m2b) {
    const float nt = n + nb;
    if (nt > 0.0f) {
        const float delta = meanb - mean;
        const float fb = nb / nt;
        mean = fmaf(delta, fb, mean);
        m2 = m2 + m2b + delta * delta * n * fb;
    }
    n = nt;
}

__device__ __forceinline__ float cg_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }

// q / n for 0 <= q < 2^22 via the fp32 reciprocal, exact after one correction
__device__ __forceinline__ int cg_div(int q, int n, float inv_n) {
    int t = (int)((float)q * inv_n);
    if (t * n > q) --t;
    if ((t + 1) * n <= q) ++t;
    return t;
}

typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef __fp16 h16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

// two fp32 -> packed bf16 (round to nearest even: v_cvt_pk_bf16_f32)
__device__ __forceinline__ unsigned cg_bf2(float x0, float x1) {
    const bf16x2 v = {(__bf16)x0, (__bf16)x1};
    return __builtin_bit_cast(unsigned, v);
}

// x = hi + lo with hi = x truncated to fp16 and lo = fp16(x - hi), two elements at a time
// (v_cvt_pkrtz_f16_f32; fp16 subnormals are honoured by the f16 MFMA, probed on gfx950).
__device__ __forceinline__ void cg_split2(float x0, float x1, unsigned& hi, unsigned& lo) {
    const h16x2 h = __builtin_amdgcn_cvt_pkrtz(x0, x1);
    const h16x2 l = __builtin_amdgcn_cvt_pkrtz(x0 - (float)h[0], x1 - (float)h[1]);
    hi = __builtin_bit_cast(unsigned, h);
    lo = __builtin_bit_cast(unsigned, l);
}

// WIDE (bf16 products, gather pipeline, one unit per stage, no fused transform): every source is STORED as bf16 and walked in
// 32-channel units -- a compile-time property of the launch, so that the loads of a stage stay one straight-line burst (a
// run-time flag around them measured slower: the weight-gradient kernel lost 30 % to exactly that)
template <int MI, int NI, int KU, int MODE, int XF, bool VEC, int PREC, bool PATCH, bool WIDE = false>
__global__ __launch_bounds__(CG_THREADS, (MI == 2 && NI == 2 && KU == 1 && MODE == CG_GLU) ? 3 : (MODE == CG_PH2 ? 2 : 1)) void conv_gemm_kernel(const eab_conv_desc d) {
    // (workgroups per CU the register budget is cut for: three for the gated 128x128 tiles -- 168 registers, no spills --, two
    // for the phase-pair form, whose fused transform and second store stream do not fit 168)
    static_assert(MODE != CG_PH2 || (PATCH && NI == 2), "phase-pair form: patch pipeline, one column block per phase and wave");
    static_assert(!PATCH || (KU == 1 && MODE != CG_DUAL && VEC), "patch mode: one unit per stage, single transform");
    constexpr bool H3 = PREC == EAB_PREC_F16X3;
    constexpr bool BF = PREC == EAB_PREC_BF16;       // fp32 in memory, operands rounded to bf16 on their way into LDS, ONE bf16 MFMA
    constexpr bool PH2 = MODE == CG_PH2;            // (phase 0, phase 1) column pair per lane; own epilogue
    constexpr bool GLU = MODE != CG_PLAIN;          // paired columns per lane: gated epilogue (value tile, gate tile) / phase pair
    constexpr bool DUAL = MODE == CG_DUAL;
    using Smem = CgSmem<MI, NI, KU, MODE, PATCH>;
    constexpr int BM = Smem::BM, BN = Smem::BN, LDK = Smem::LDK;
    __shared__ __attribute__((aligned(16))) Smem sm;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 31, lh = lane >> 5;

    // rows per batch element: all of them, or the streaming window [t_lo, t_lo + count) (eab_time_window)
    const int t_lo = d.win.pos ? *d.win.pos : 0;
    const int t_hi = d.win.pos ? (t_lo + d.win.count < d.T ? t_lo + d.win.count : d.T) : d.T;
    const int Q = t_hi * d.No;                      // first row NOT computed
    const int tiles_per_b = ((d.win.pos ? d.win.count : d.T) * d.No + BM - 1) / BM;
    // Workgroups are dealt to the 8 XCDs round-robin by blockIdx.x, and every XCD has its own L2: let each XCD walk one
    // contiguous eighth of the (b, tile) sequence, so that the k_t halo row two neighbouring tiles share (and the weights they
    // both stream) is fetched into ONE L2 instead of two (HBM traffic of the dominant launches 299 -> see profiles/).
    unsigned vblk = blockIdx.x;
    if (gridDim.x >= 64) {
        const unsigned G = gridDim.x, G8 = G >> 3, rem = G & 7, xcd = vblk & 7, idx = vblk >> 3;
        vblk = xcd * G8 + (xcd < rem ? xcd : rem) + idx;
    }
    const int b = (int)(vblk / (unsigned)tiles_per_b);
    const int tile = (int)vblk - b * tiles_per_b;
    const int q0 = t_lo * d.No + tile * BM;
    const int n_blk = blockIdx.y * BN;
    const float inv_no = 1.0f / (float)d.No;

    // constant indices keep the descriptor in the kernarg segment (a lane-indexed
    // read would force a scratch copy of the struct)
#pragma unroll
    for (int j = 0; j < EAB_MAX_TAPS; ++j)
        if (tid == j) {
            sm.dt[j] = d.dt[j];
            sm.ioff[j] = d.ioff[j];
        }

    if (XF != EAB_XF_NONE) {
        // (scale, shift, slope) tables of this batch element -> LDS.  Either the host ran
        // eab_in_finalize_f32 (xf0/xf1) or the producer left few enough tiles that every
        // consumer workgroup reduces them itself (fin_stats; fp64, fixed order, so all
        // workgroups and the stand-alone kernel agree bit for bit).
        const int k = tid >> 7, c = tid & (CG_XFC - 1);
        const int Ck = (k == 0 || DUAL) ? d.C0 : d.C1;
        const float* xfk = k == 0 ? d.xf0 : d.xf1;
        const float* slk = k == 0 ? d.slope0 : d.slope1;
        float sc = 1.0f, sh = 0.0f, sl = 1.0f;
        if (c < Ck) {
            if (d.fin_stats && k < d.fin_nsets) {
                const float* gm = k == 0 ? d.fin_gamma0 : d.fin_gamma1;
                const float* bt = k == 0 ? d.fin_beta0 : d.fin_beta1;
                double sn = 0.0, sm = 0.0, sq = 0.0;            // same exact merge as in_finalize_kernel
                for (int t = 0; t < d.fin_tiles; ++t) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(
                        &d.fin_stats[((((size_t)b * d.fin_tiles + t) * d.fin_nsets + k) * d.C0 + c) * 4]);
                    const double n = (double)v[0], mu = (double)v[1];
                    sn += n;
                    sm = fma(n, mu, sm);
                    sq += fma(n * mu, mu, (double)v[2]);
                }
                const double mean = sn > 0.0 ? sm / sn : 0.0;
                double var = sn > 0.0 ? sq / sn - mean * mean : 0.0;
                if (var < 0.0) var = 0.0;
                const double scale = (double)gm[c] / sqrt(var + (double)d.fin_eps);
                sc = (float)scale;
                sh = (float)((double)bt[c] - mean * scale);
                sl = slk[c];
            } else if (xfk) {
                const float2 v = *reinterpret_cast<const float2*>(&xfk[((size_t)b * Ck + c) * 2]);
                sc = v.x;
                sh = v.y;
                sl = slk[c];
            }
        }
        sm.xft[k][c][0] = sc;
        sm.xft[k][c][1] = sh;
        sm.xsl[k][c] = sl;
    }

    const int Ctot = d.C0 + d.C1;
    const int UPT = (Ctot + 15) >> 4;
    const int NU = d.ntaps * UPT;
    // bf16-STORED sources (src_bf16, LEAN gather of the bf16 precision) are walked in units of 32 channels: a thread's 16-byte
    // load is eight channels, the four threads of a row cover 64 bytes -- a whole sector, where four-channel loads of a bf16
    // tensor leave half of every 64-byte segment unused -- and a stage carries two MFMA k-steps instead of one: half the
    // stages, half the barriers.  The k order (and so every bit of the result) is that of the 16-channel walk.
    static_assert(!WIDE || (PREC == EAB_PREC_BF16 && KU == 1 && VEC && !PATCH && XF == EAB_XF_NONE && MODE != CG_DUAL), "WIDE: see above");
    constexpr bool w0 = WIDE, w1 = WIDE;
    const int NS = (w0 || w1) ? d.ntaps * ((w0 ? d.C0 >> 5 : (d.C0 + 15) >> 4) + (d.C1 > 0 ? (w1 ? d.C1 >> 5 : (d.C1 + 15) >> 4) : 0))
                              : (NU + KU - 1) / KU;              // pipeline stages

    // one buffer descriptor per source, spanning this workgroup's batch element
    // (src_bf16, bf16 products only: a source stored as bf16 [B][T][Fin][C] -- the training programs' normalised activations and
    // convolution-output gradients, which this precision rounds to bf16 on their way into LDS anyway: same operands, half the
    // bytes, no conversion.  esz = bytes per element of the source.)
    constexpr unsigned esz0 = WIDE ? 2u : 4u, esz1 = WIDE ? 2u : 4u;
    const unsigned bytes0 = (unsigned)d.T * d.Fin * d.C0 * esz0;
    const unsigned bytes1 = (unsigned)d.T * d.Fin * d.C1 * esz1;
    const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(d.src0)) + (size_t)b * bytes0, 0, bytes0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(d.src1 ? d.src1 : d.src0)) + (d.src1 ? (size_t)b * bytes1 : 0), 0,
        d.src1 ? bytes1 : 0u, 0x00020000);

    // ---- per-thread staging coordinates ------------------------------------------
    const int srow = tid >> 2;      // 0..63
    const int skq = tid & 3;        // which float4 of a 16-wide unit
    int a_t[MI], a_f0[MI], a_tf[MI];
    bool a_ok[MI];
#pragma unroll
    for (int p = 0; p < MI; ++p) {
        const int q = q0 + srow + 64 * p;
        a_ok[p] = q < Q;
        const int t = a_ok[p] ? cg_div(q, d.No, inv_no) : 0;
        const int o = a_ok[p] ? q - t * d.No : 0;
        a_t[p] = t;
        a_f0[p] = o * d.istride;
        a_tf[p] = t * d.Fin + o * d.istride;
    }
    const float* wrow[NI];
#pragma unroll
    for (int p = 0; p < NI; ++p) wrow[p] = d.w + (size_t)(n_blk + srow + 64 * p) * d.Kpad + skq * 4;

    f32x16 acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.0f;

    // LDS writes must be visible, global loads may stay in flight across the barrier
    // (__syncthreads() would add s_waitcnt vmcnt(0) and drain the prefetch).
    auto lds_barrier = [] { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };

    __syncthreads();   // tap tables (and transform tables) visible

    if constexpr (PATCH) {
        // ------------------------------------------------------------------------------
        // PATCH pipeline.  K is walked chunk-major (16 channels of ALL taps, then the next
        // 16 channels; the host packs the weights in that order).  The input patch of one
        // chunk -- every position (t', f') any row of the tile touches through any tap, plus
        // zeroed halo cells -- is fetched, transformed and (f16x3) split ONCE, and each tap's
        // MFMAs read their A fragments from it at a per-lane base + a per-tap shift.  Only the
        // weights stream per stage.  Versus the gather pipeline this removes the ntaps-fold
        // re-fetch / re-transform / re-split of A.
        // ------------------------------------------------------------------------------
        constexpr int PP = (CG_PMAX + 63) / 64;
        int dt_min = 0, io_min = 0, io_max = 0;
#pragma unroll
        for (int j = 0; j < EAB_MAX_TAPS; ++j)
            if (j < d.ntaps) {
                dt_min = d.dt[j] < dt_min ? d.dt[j] : dt_min;
                io_min = d.ioff[j] < io_min ? d.ioff[j] : io_min;
                io_max = d.ioff[j] > io_max ? d.ioff[j] : io_max;
            }
        const int halo_lo = -io_min;
        const int hi_need = (d.No - 1) * d.istride + io_max - (d.Fin - 1);
        const int Fp = d.Fin + halo_lo + (hi_need > 0 ? hi_need : 0);
        const float inv_fp = 1.0f / (float)Fp;
        const int t_first = cg_div(q0, d.No, inv_no);
        const int t_last = cg_div((q0 + BM < Q ? q0 + BM : (Q > q0 ? Q : q0 + 1)) - 1, d.No, inv_no);
        const int P = (t_last - t_first + 1 - dt_min) * Fp;          // <= CG_PMAX (checked on the host)

        int p_tf[PP];
        bool p_ok[PP];
#pragma unroll
        for (int pp = 0; pp < PP; ++pp) {
            const int pidx = srow + 64 * pp;
            const int tr = cg_div(pidx, Fp, inv_fp);
            const int t_in = t_first + dt_min + tr, fi = pidx - tr * Fp - halo_lo;
            p_ok[pp] = pidx < P && t_in >= 0 && t_in < d.T && fi >= 0 && fi < d.Fin;
            p_tf[pp] = t_in * d.Fin + fi;
        }
        int fa[MI];                                                  // float index of this lane's fragment for a zero tap shift
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            int q = q0 + (wm * MI + mi) * 32 + li;
            q = q < Q ? q : (Q > q0 ? Q - 1 : q0);                   // tile tail: any row of the tile's patch (masked in the epilogue)
            const int t = cg_div(q, d.No, inv_no), o = q - t * d.No;
            fa[mi] = ((t - t_first - dt_min) * Fp + o * d.istride + halo_lo) * LDK + 4 * lh;
        }
        const int b_base = (wn * NI * 32 + li) * LDK + 4 * lh;

        f32x4 rp[PP];
        f32x4 rbw0[NI], rbw1[NI];        // weights of stages u+1 and u+2 (two named sets, static indexing)
        int p_tc = 0;
        auto patch_fetch = [&](int chunk) {
            const int c0 = chunk << 4;
            const bool second = (d.C1 > 0) && (c0 >= d.C0);
            const int Cs = second ? d.C1 : d.C0;
            const int c = (second ? c0 - d.C0 : c0) + skq * 4;
            const bool cok = c < Cs;
            p_tc = ((second ? 1 : 0) << 8) | (cok ? c : 0);
#pragma unroll
            for (int pp = 0; pp < PP; ++pp) {
                const unsigned off = (p_ok[pp] && cok) ? (unsigned)((p_tf[pp] * Cs + c) * 4) : CG_OOB;
                const u32x4 v = second ? __builtin_amdgcn_raw_buffer_load_b128(rs1, off, 0, 0)
                                       : __builtin_amdgcn_raw_buffer_load_b128(rs0, off, 0, 0);
                rp[pp] = __builtin_bit_cast(f32x4, v);
            }
        };
        auto patch_stash = [&]() {
            f32x4 sh01, sh23, sl;
            if (XF != EAB_XF_NONE) {
                const int cc = p_tc & 0xFF, tb = p_tc >> 8;
                sh01 = *reinterpret_cast<const f32x4*>(&sm.xft[tb][cc][0]);
                sh23 = *reinterpret_cast<const f32x4*>(&sm.xft[tb][cc + 2][0]);
                sl = *reinterpret_cast<const f32x4*>(&sm.xsl[tb][cc]);
            }
#pragma unroll
            for (int pp = 0; pp < PP; ++pp) {
                const int pidx = srow + 64 * pp;
                if (pidx >= CG_PMAX) continue;
                f32x4 v = rp[pp];
                if (XF != EAB_XF_NONE) {
                    const f32x4 x = cg_xform<XF>(v, sh01, sh23, sl);
                    v = p_ok[pp] ? x : f32x4{0.f, 0.f, 0.f, 0.f};     // halo / causal zeros stay exactly 0
                }
                float* arow = &sm.a[pidx * LDK];
                if (BF) {
                    *reinterpret_cast<uint2*>(reinterpret_cast<char*>(arow) + skq * 8) = make_uint2(cg_bf2(v[0], v[1]), cg_bf2(v[2], v[3]));
                } else if (H3) {
                    unsigned h01, l01, h23, l23;
                    cg_split2(v[0], v[1], h01, l01);
                    cg_split2(v[2], v[3], h23, l23);
                    *reinterpret_cast<uint2*>(reinterpret_cast<char*>(arow) + skq * 8) = make_uint2(h01, h23);
                    *reinterpret_cast<uint2*>(reinterpret_cast<char*>(arow) + 32 + skq * 8) = make_uint2(l01, l23);
                } else {
                    *reinterpret_cast<f32x4*>(arow + skq * 4) = v;
                }
            }
        };
        auto b_fetch = [&](int u, f32x4 (&rbw)[NI]) {
            const int uu = u < NU ? u : 0;                           // past the end: harmless reload of unit 0
#pragma unroll
            for (int p = 0; p < NI; ++p) rbw[p] = *reinterpret_cast<const f32x4*>(wrow[p] + (size_t)uu * 16);
        };
        auto b_stash = [&](int buf, const f32x4 (&rbw)[NI]) {
#pragma unroll
            for (int p = 0; p < NI; ++p) {
                float* brow = &sm.b[buf][(srow + 64 * p) * LDK];
                if (BF) {
                    *reinterpret_cast<uint2*>(reinterpret_cast<char*>(brow) + skq * 8) =
                        make_uint2(cg_bf2(rbw[p][0], rbw[p][1]), cg_bf2(rbw[p][2], rbw[p][3]));
                } else if (H3) {
                    const u32x4 w = __builtin_bit_cast(u32x4, rbw[p]);
                    *reinterpret_cast<uint2*>(reinterpret_cast<char*>(brow) + skq * 8) = make_uint2(w[0], w[1]);
                    *reinterpret_cast<uint2*>(reinterpret_cast<char*>(brow) + 32 + skq * 8) = make_uint2(w[2], w[3]);
                } else {
                    *reinterpret_cast<f32x4*>(brow + skq * 4) = rbw[p];
                }
            }
        };
        // ne = column blocks that take part (phase-pair form: a tap that feeds phase 0 only skips the phase-1 block)
        auto pcompute_n = [&](int cur, int tap, auto ne_c) {
            constexpr int NE = decltype(ne_c)::value;
            const int toff = (sm.dt[tap] * Fp + sm.ioff[tap]) * LDK;  // per-tap shift inside the patch (wave-uniform)
            if (BF) {
                bf16x8 ab[MI], bb[NE];
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) ab[mi] = *reinterpret_cast<const bf16x8*>(&sm.a[fa[mi] + toff]);
#pragma unroll
                for (int ni = 0; ni < NE; ++ni) bb[ni] = *reinterpret_cast<const bf16x8*>(&sm.b[cur][b_base + ni * 32 * LDK]);
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int ni = 0; ni < NE; ++ni)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab[mi], bb[ni], acc[mi][ni], 0, 0, 0);
            } else if (H3) {
                h16x8 ah[MI], al[MI], bh[NE], bl[NE];
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) {
                    const char* pa = reinterpret_cast<const char*>(&sm.a[fa[mi] + toff]);
                    ah[mi] = *reinterpret_cast<const h16x8*>(pa);
                    al[mi] = *reinterpret_cast<const h16x8*>(pa + 32);
                }
#pragma unroll
                for (int ni = 0; ni < NE; ++ni) {
                    const char* pb = reinterpret_cast<const char*>(&sm.b[cur][b_base + ni * 32 * LDK]);
                    bh[ni] = *reinterpret_cast<const h16x8*>(pb);
                    bl[ni] = *reinterpret_cast<const h16x8*>(pb + 32);
                }
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int ni = 0; ni < NE; ++ni) {
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[mi], bh[ni], acc[mi][ni], 0, 0, 0);
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mi], bl[ni], acc[mi][ni], 0, 0, 0);
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mi], bh[ni], acc[mi][ni], 0, 0, 0);
                    }
            } else {
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    f32x4 af[MI], bf[NE];
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi) af[mi] = *reinterpret_cast<const f32x4*>(&sm.a[fa[mi] + toff + g * 8]);
#pragma unroll
                    for (int ni = 0; ni < NE; ++ni)
                        bf[ni] = *reinterpret_cast<const f32x4*>(&sm.b[cur][b_base + ni * 32 * LDK + g * 8]);
#pragma unroll
                    for (int k = 0; k < 4; ++k)
#pragma unroll
                        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                            for (int ni = 0; ni < NE; ++ni)
                                acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[mi][k], bf[ni][k], acc[mi][ni], 0, 0, 0);
                }
            }
        };
        const unsigned p2mask = PH2 ? (unsigned)d.p2_mask1 : ~0u;
        auto pcompute = [&](int cur, int tap) {
            if (PH2 && !((p2mask >> tap) & 1u)) pcompute_n(cur, tap, std::integral_constant<int, 1>{});   // (wave-uniform)
            else pcompute_n(cur, tap, std::integral_constant<int, NI>{});
        };

        patch_fetch(0);
        b_fetch(0, rbw0);
        patch_stash();
        b_stash(0, rbw0);
        b_fetch(1, rbw0);                                            // stage u+1 -> rbw0, stage u+2 -> rbw1
        b_fetch(2, rbw1);
        lds_barrier();
        const int pre = d.ntaps >= 3 ? d.ntaps - 3 : 0;             // request the next patch three taps early
        int chunk = 0, tap = 0;
        auto stage = [&](int u, int cur, f32x4 (&rnext)[NI]) {
            // rnext holds the weights of stage u+1 (loaded two stages ago); it is refilled with u+3
            if (tap == pre && chunk + 1 < UPT) patch_fetch(chunk + 1);
            pcompute(cur, tap);
            if (u + 1 < NU) b_stash(cur ^ 1, rnext);
            b_fetch(u + 3, rnext);
            if (tap == d.ntaps - 1 && chunk + 1 < UPT) {
                lds_barrier();                                       // every wave is done with this chunk's patch
                patch_stash();
            }
            lds_barrier();
            if (++tap == d.ntaps) { tap = 0; ++chunk; }
        };
        for (int u = 0; u < NU; u += 2) {
            stage(u, 0, rbw0);
            if (u + 1 < NU) stage(u + 1, 1, rbw1);
        }
        __syncthreads();                                             // drain the look-ahead loads before LDS is reused
    } else {
    // Zero padding acts on the NORMALISED tensor in the reference (ConstantPad2d /
    // the transposed conv's implicit zeros come after norm+PReLU), so out-of-range
    // taps must stay exactly 0 through the fused transform: st_ok remembers which
    // staged rows are real.
    struct Stage {                  // one pipeline stage in flight in registers
        f32x4 ra[KU][MI], rb[KU][NI];
        f32x4 rb2[BF ? NI : 1];     // wide stages (bf16-stored source): the second four weights of the thread's eight
        int r_tc[KU];               // (table << 8 | first channel) of the staged float4
        bool st_ok[KU][MI];
    };
    Stage sa;                       // stage s+1 in flight while stage s is multiplied

    // Row state of the CURRENT tap, refreshed when the unit sequence enters a new tap (units run tap-major: UPT units per
    // tap): validity of the tap's source position and its byte offset in either source.  Per unit that leaves one select and
    // one add per row instead of six compares, two integer multiplies (quarter rate) and the scalar unit/tap division.
    bool tap_ok[MI];
    unsigned tap_b0[MI], tap_b1[MI];
#pragma unroll
    for (int p = 0; p < MI; ++p) {
        tap_ok[p] = false;
        tap_b0[p] = tap_b1[p] = 0u;
    }
    int g_tap = -1, g_c0 = 0, g_w = 16;                    // tap, first channel and width of the unit fetched last
    // LEAN (one unit per stage, float4 gathers, one transform): the per-stage address work is moved to where it changes.
    // Everything that depends on the unit alone -- which source, the channel offset inside it, the weight column -- is
    // workgroup-uniform and goes into the scalar offset operand of the buffer loads (soffset is not part of the range check,
    // so an out-of-range row stays out of range); the per-row byte offset of a tap's source position, with the thread's own
    // 16-byte channel group folded in, is computed once per tap.  Per stage that leaves one compare and one select per row
    // (rounds 1-3: ~30 vector instructions per stage, more than the MFMAs of a 64-column tile cost).  Same loads, same values.
    constexpr bool LEAN = KU == 1 && VEC && !DUAL;
    const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.w), 0, (unsigned)d.N * d.Kpad * 4u, 0x00020000);
    unsigned w_off[NI];
#pragma unroll
    for (int p = 0; p < NI; ++p) w_off[p] = (unsigned)((n_blk + srow + 64 * p) * d.Kpad + skq * 4) * 4u;
    auto fetch = [&](int stage, Stage& rg) {
        if constexpr (LEAN) {
            // (callers never ask for a stage past the last one; g_w = channels of the previous unit: 16, or 32 for a bf16 source)
            if (stage == 0) {
                g_tap = 0;
                g_c0 = 0;
            } else {
                g_c0 += g_w;
                if (g_c0 >= (UPT << 4)) {
                    g_c0 = 0;
                    ++g_tap;
                }
            }
            if (g_c0 == 0) {                                    // (workgroup-uniform) first unit of a tap
                const int dt = sm.dt[g_tap], io = sm.ioff[g_tap];
#pragma unroll
                for (int p = 0; p < MI; ++p) {
                    const int tt = a_t[p] + dt, fi = a_f0[p] + io;
                    tap_ok[p] = a_ok[p] && tt >= 0 && tt < d.T && fi >= 0 && fi < d.Fin;
                    const unsigned pos = (unsigned)(a_tf[p] + dt * d.Fin + io);
                    tap_b0[p] = (pos * (unsigned)d.C0 + (unsigned)(skq * (w0 ? 8 : 4))) * esz0;
                    tap_b1[p] = (pos * (unsigned)d.C1 + (unsigned)(skq * (w1 ? 8 : 4))) * esz1;
                }
            }
            const bool second = (d.C1 > 0) && (g_c0 >= d.C0);  // workgroup-uniform
            const int Cs = second ? d.C1 : d.C0;
            const int cu = second ? g_c0 - d.C0 : g_c0;         // first channel of the unit inside its source (uniform)
            constexpr bool half = WIDE;                         // the sources are stored as bf16: 32-channel units
            g_w = half ? 32 : 16;
            const bool cok = cu + skq * (half ? 8 : 4) < Cs;    // (false only in the padded tail of a source with C % 16 != 0)
            const int k0 = g_tap * (UPT << 4) + g_c0;           // first weight column of the unit
#pragma unroll
            for (int p = 0; p < MI; ++p) {
                const bool ok = tap_ok[p] && cok;
                rg.st_ok[0][p] = ok;
                const unsigned off = ok ? (second ? tap_b1[p] : tap_b0[p]) : CG_OOB;
                // (bf16 source: eight bf16 of the row = the LDS image of this thread's part of the unit; soffset in bytes)
                const u32x4 v = second ? __builtin_amdgcn_raw_buffer_load_b128(rs1, off, cu * (half ? 2 : 4), 0)
                                       : __builtin_amdgcn_raw_buffer_load_b128(rs0, off, cu * (half ? 2 : 4), 0);
                rg.ra[0][p] = __builtin_bit_cast(f32x4, v);
            }
            rg.r_tc[0] = ((half ? 1 : 0) << 9) | ((second ? 1 : 0) << 8) | cu;   // uniform: (stored as bf16, table, first channel)
#pragma unroll
            for (int p = 0; p < NI; ++p) {
                if constexpr (WIDE) {                           // eight weights per thread: columns k0 + 8 skq .. + 7
                    rg.rb[0][p] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsw, w_off[p] + skq * 16, k0 * 4, 0));
                    if constexpr (BF)
                        rg.rb2[p] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsw, w_off[p] + skq * 16, k0 * 4 + 16, 0));
                } else {
                    rg.rb[0][p] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsw, w_off[p], k0 * 4, 0));
                }
            }
            return;
        }
#pragma unroll
        for (int ku = 0; ku < KU; ++ku) {
            const int u = stage * KU + ku;
            const bool live = u < NU;                           // K tail of the last stage
            // units are fetched in order 0, 1, 2, ...: advance (tap, channel) without dividing
            if (u == 0) {
                g_tap = 0;
                g_c0 = 0;
            } else if (live) {
                g_c0 += 16;
                if (g_c0 >= (UPT << 4)) {
                    g_c0 = 0;
                    ++g_tap;
                }
            }
            const int uu = live ? u : 0;
            const int c0 = live ? g_c0 : 0;
            if (live && c0 == 0) {                              // (workgroup-uniform) first unit of a tap
                const int dt = sm.dt[g_tap], io = sm.ioff[g_tap];
#pragma unroll
                for (int p = 0; p < MI; ++p) {
                    const int tt = a_t[p] + dt, fi = a_f0[p] + io;
                    tap_ok[p] = a_ok[p] && tt >= 0 && tt < d.T && fi >= 0 && fi < d.Fin;
                    const unsigned pos = (unsigned)(a_tf[p] + dt * d.Fin + io);
                    tap_b0[p] = pos * (unsigned)(d.C0 * 4);
                    tap_b1[p] = pos * (unsigned)(d.C1 * 4);
                }
            }
            const bool second = (d.C1 > 0) && (c0 >= d.C0);     // wave-uniform
            const int Cs = second ? d.C1 : d.C0;
            const int c = (second ? c0 - d.C0 : c0) + skq * 4;
            const bool cok = live && (c < Cs);
#pragma unroll
            for (int p = 0; p < MI; ++p) {
                const bool ok = tap_ok[p] && cok;
                rg.st_ok[ku][p] = ok;
                const unsigned off = ok ? (second ? tap_b1[p] : tap_b0[p]) + (unsigned)(c * 4) : CG_OOB;
                if (VEC) {
                    const u32x4 v = second ? __builtin_amdgcn_raw_buffer_load_b128(rs1, off, 0, 0)
                                           : __builtin_amdgcn_raw_buffer_load_b128(rs0, off, 0, 0);
                    rg.ra[ku][p] = __builtin_bit_cast(f32x4, v);
                } else {
                    // channel counts that are not multiples of 4 (odd microphone counts): dword gathers
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const unsigned oj = (ok && c + j < Cs) ? off + 4u * j : CG_OOB;
                        const unsigned v = second ? __builtin_amdgcn_raw_buffer_load_b32(rs1, oj, 0, 0)
                                                  : __builtin_amdgcn_raw_buffer_load_b32(rs0, oj, 0, 0);
                        rg.ra[ku][p][j] = __builtin_bit_cast(float, v);
                    }
                }
            }
            rg.r_tc[ku] = ((second ? 1 : 0) << 8) | (cok ? c : 0);
#pragma unroll
            for (int p = 0; p < NI; ++p) {
                rg.rb[ku][p] = *reinterpret_cast<const f32x4*>(wrow[p] + (size_t)uu * 16);
                if (!live) rg.rb[ku][p] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
    };

    auto stash = [&](int buf, const Stage& rg) {
#pragma unroll
        for (int ku = 0; ku < KU; ++ku) {
            f32x4 sh01[2], sh23[2], sl[2];
            bool all_ok = false;
            if (XF != EAB_XF_NONE) {
                if constexpr (LEAN) {
                    // r_tc is workgroup-uniform (table, first channel of the unit): the table addresses are the thread's own
                    // channel group (fixed) plus a scalar -- two vector adds instead of ten.  (A channel past the end of a
                    // source with C % 16 != 0 indexes an initialised table entry and its row is zeroed by st_ok.)
                    const int tb = (rg.r_tc[ku] >> 8) & 1, cu = rg.r_tc[ku] & 0xFF;
                    const char* px = reinterpret_cast<const char*>(&sm.xft[0][skq * 4][0]) + (tb * CG_XFC + cu) * 8;
                    const char* ps = reinterpret_cast<const char*>(&sm.xsl[0][skq * 4]) + (tb * CG_XFC + cu) * 4;
                    sh01[0] = *reinterpret_cast<const f32x4*>(px);
                    sh23[0] = *reinterpret_cast<const f32x4*>(px + 16);
                    sl[0] = *reinterpret_cast<const f32x4*>(ps);
                    // every row of every lane real (all but the tiles at the rim of the tensor): no zero-selects at all
                    bool ok_all = true;
#pragma unroll
                    for (int p = 0; p < MI; ++p) ok_all = ok_all && rg.st_ok[ku][p];
                    all_ok = __builtin_amdgcn_ballot_w64(ok_all) == __builtin_amdgcn_ballot_w64(true);   // (wave-uniform)
                } else {
                const int cc = rg.r_tc[ku] & 0xFF;
#pragma unroll
                for (int k = 0; k < (DUAL ? 2 : 1); ++k) {
                    const int tb = DUAL ? k : (rg.r_tc[ku] >> 8);
                    sh01[k] = *reinterpret_cast<const f32x4*>(&sm.xft[tb][cc][0]);
                    sh23[k] = *reinterpret_cast<const f32x4*>(&sm.xft[tb][cc + 2][0]);
                    sl[k] = *reinterpret_cast<const f32x4*>(&sm.xsl[tb][cc]);
                }
                }
            }
#pragma unroll
            for (int p = 0; p < MI; ++p) {
#pragma unroll
                for (int k = 0; k < (DUAL ? 2 : 1); ++k) {
                    f32x4 v = rg.ra[ku][p];
                    if (XF != EAB_XF_NONE) {
                        v = cg_xform<XF>(v, sh01[k], sh23[k], sl[k]);
                        if (!all_ok) {                           // scalar branch (the asm keeps it one: four selects otherwise)
                            asm volatile("; rim" ::: "memory");
                            v = rg.st_ok[ku][p] ? v : f32x4{0.f, 0.f, 0.f, 0.f};
                        }
                    }
                    float* arow = &sm.a[(k * 2 + buf) * Smem::ATILE + (srow + 64 * p) * LDK + ku * 16];
                    if (BF) {
                        constexpr bool half = WIDE;                              // bf16 in memory: stored as fetched
                        if (half)                                                // eight channels of a 32-channel unit
                            *reinterpret_cast<u32x4*>(reinterpret_cast<char*>(arow) + skq * 16) = __builtin_bit_cast(u32x4, v);
                        else
                            *reinterpret_cast<uint2*>(reinterpret_cast<char*>(arow) + skq * 8) = make_uint2(cg_bf2(v[0], v[1]), cg_bf2(v[2], v[3]));
                    } else if (H3) {
                        // unit layout in LDS (64 B): 16 fp16 hi | 16 fp16 lo; this thread owns channels 4*skq..+3
                        unsigned h01, l01, h23, l23;
                        cg_split2(v[0], v[1], h01, l01);
                        cg_split2(v[2], v[3], h23, l23);
                        *reinterpret_cast<uint2*>(reinterpret_cast<char*>(arow) + skq * 8) = make_uint2(h01, h23);
                        *reinterpret_cast<uint2*>(reinterpret_cast<char*>(arow) + 32 + skq * 8) = make_uint2(l01, l23);
                    } else {
                        *reinterpret_cast<f32x4*>(arow + skq * 4) = v;
                    }
                }
            }
#pragma unroll
            for (int p = 0; p < NI; ++p) {
                float* brow = &sm.b[buf][(srow + 64 * p) * LDK + ku * 16];
                if (BF) {
                    if constexpr (WIDE) {                        // wide stage: eight weights of the thread -> 16 bytes
                        const f32x4 r0 = rg.rb[ku][p], r1 = rg.rb2[BF ? p : 0];
                        *reinterpret_cast<u32x4*>(reinterpret_cast<char*>(brow) + skq * 16) =
                            u32x4{cg_bf2(r0[0], r0[1]), cg_bf2(r0[2], r0[3]), cg_bf2(r1[0], r1[1]), cg_bf2(r1[2], r1[3])};
                    } else {
                        *reinterpret_cast<uint2*>(reinterpret_cast<char*>(brow) + skq * 8) =
                            make_uint2(cg_bf2(rg.rb[ku][p][0], rg.rb[ku][p][1]), cg_bf2(rg.rb[ku][p][2], rg.rb[ku][p][3]));
                    }
                } else if (H3) {   // global: [4 hi | 4 lo] per 4-channel group  ->  LDS: [16 hi | 16 lo] per unit
                    const u32x4 w = __builtin_bit_cast(u32x4, rg.rb[ku][p]);
                    *reinterpret_cast<uint2*>(reinterpret_cast<char*>(brow) + skq * 8) = make_uint2(w[0], w[1]);
                    *reinterpret_cast<uint2*>(reinterpret_cast<char*>(brow) + 32 + skq * 8) = make_uint2(w[2], w[3]);
                } else {
                    *reinterpret_cast<f32x4*>(brow + skq * 4) = rg.rb[ku][p];
                }
            }
        }
    };

    fetch(0, sa);
    stash(0, sa);
    lds_barrier();

    const int a_base = (wm * MI * 32 + li) * LDK + 4 * lh;
    const int b_base = (wn * NI * 32 + li) * LDK + 4 * lh;

    auto compute = [&](int cur) {
        if (BF) {
#pragma unroll
            for (int ku = 0; ku < KU; ++ku) {
                bf16x8 ab[Smem::NA][MI], bb[NI];
#pragma unroll
                for (int k = 0; k < Smem::NA; ++k)
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi)
                        ab[k][mi] = *reinterpret_cast<const bf16x8*>(&sm.a[(k * 2 + cur) * Smem::ATILE + a_base + mi * 32 * LDK + ku * 16]);
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) bb[ni] = *reinterpret_cast<const bf16x8*>(&sm.b[cur][b_base + ni * 32 * LDK + ku * 16]);
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab[DUAL ? ni : 0][mi], bb[ni], acc[mi][ni], 0, 0, 0);
            }
            if constexpr (WIDE) {                               // channels 16..31 of a 32-channel unit: bytes 32..63 of a row
                bf16x8 ab[MI], bb[NI];
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) ab[mi] = *reinterpret_cast<const bf16x8*>(&sm.a[cur * Smem::ATILE + a_base + mi * 32 * LDK + 8]);
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) bb[ni] = *reinterpret_cast<const bf16x8*>(&sm.b[cur][b_base + ni * 32 * LDK + 8]);
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab[mi], bb[ni], acc[mi][ni], 0, 0, 0);
            }
        } else if (H3) {
            // one v_mfma_f32_32x32x16_f16 spans a whole 16-channel unit: lane (i, h) holds k = 8h..8h+7
#pragma unroll
            for (int ku = 0; ku < KU; ++ku) {
                h16x8 ah[Smem::NA][MI], al[Smem::NA][MI], bh[NI], bl[NI];
#pragma unroll
                for (int k = 0; k < Smem::NA; ++k)
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi) {
                        const char* pa = reinterpret_cast<const char*>(&sm.a[(k * 2 + cur) * Smem::ATILE + a_base + mi * 32 * LDK + ku * 16]);
                        ah[k][mi] = *reinterpret_cast<const h16x8*>(pa);
                        al[k][mi] = *reinterpret_cast<const h16x8*>(pa + 32);
                    }
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) {
                    const char* pb = reinterpret_cast<const char*>(&sm.b[cur][b_base + ni * 32 * LDK + ku * 16]);
                    bh[ni] = *reinterpret_cast<const h16x8*>(pb);
                    bl[ni] = *reinterpret_cast<const h16x8*>(pb + 32);
                }
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni) {
                        const int ka = DUAL ? ni : 0;
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[ka][mi], bh[ni], acc[mi][ni], 0, 0, 0);
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[ka][mi], bl[ni], acc[mi][ni], 0, 0, 0);
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[ka][mi], bh[ni], acc[mi][ni], 0, 0, 0);
                    }
            }
        } else {
#pragma unroll
        for (int g = 0; g < 2 * KU; ++g) {
            f32x4 af[Smem::NA][MI], bf[NI];
#pragma unroll
            for (int k = 0; k < Smem::NA; ++k)
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
                    af[k][mi] = *reinterpret_cast<const f32x4*>(&sm.a[(k * 2 + cur) * Smem::ATILE + a_base + mi * 32 * LDK + g * 8]);
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
                bf[ni] = *reinterpret_cast<const f32x4*>(&sm.b[cur][b_base + ni * 32 * LDK + g * 8]);
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[DUAL ? ni : 0][mi][k], bf[ni][k],
                                                                           acc[mi][ni], 0, 0, 0);
        }
        }
    };
    // (a second register set for a two-stage prefetch costs the 128x128 tiles a wave of occupancy -- 195 vs 154
    // registers -- and measured slower there in both precisions)
    constexpr bool PF2 = MI * NI <= 2 && KU == 1 && MODE == CG_PLAIN;
    if constexpr (PF2) {
        // The 64-column unit convolutions (and the dgrads of the training programs): 16 MFMAs per stage are ~1,000 cycles
        // against a gather round trip of 2,000+, so the loads of stage s+2 are issued before stage s is multiplied
        // (two register sets, still four waves per SIMD).  Same stage order, same arithmetic.
        Stage sb;
        if (NS > 1) fetch(1, sb);
        for (int s = 0; s < NS; s += 2) {
            // even stage s: multiply LDS[0]; sb holds stage s+1 (in flight since the previous iteration); sa <- stage s+2
            if (s + 2 < NS) fetch(s + 2, sa);
            compute(0);
            if (s + 1 < NS) stash(1, sb);
            lds_barrier();
            if (s + 1 < NS) {
                // odd stage s+1: multiply LDS[1]; sa holds stage s+2; sb <- stage s+3
                if (s + 3 < NS) fetch(s + 3, sb);
                compute(1);
                if (s + 2 < NS) stash(0, sa);
                lds_barrier();
            }
        }
    } else {
    for (int s = 0; s < NS; ++s) {
        const int cur = s & 1;
        if (s + 1 < NS) fetch(s + 1, sa);
        compute(cur);
        if (s + 1 < NS) stash(cur ^ 1, sa);
        lds_barrier();
    }
    }

    }

    // ---- epilogue ---------------------------------------------------------------
    // C/D map of the 32x32 MFMA: column = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).
    constexpr int NC = GLU ? 1 : NI;            // output-channel groups of 32 held by this lane
    const int Cout = d.Cout;
    int ch[NC];
    float bias_v[NI];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
        const int n = n_blk + (wn * NI + ni) * 32 + li;
        bias_v[ni] = d.bias ? d.bias[n] : 0.0f;
    }
    if (GLU) {
        ch[0] = (n_blk >> 1) + wn * 32 + li;
    } else {
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) ch[ni] = n_blk + (wn * NI + ni) * 32 + li;
    }
    float st_slope[2][NC];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const float* sp = s == 0 ? d.stat_slope0 : d.stat_slope1;
            st_slope[s][c] = (d.stats && s < d.nsets && sp) ? sp[ch[c]] : 1.0f;
        }
    // InstanceNorm partials in Welford form: per lane a shifted single pass (shift = the lane's first
    // valid value, so a nearly constant channel loses nothing to cancellation), then Chan merges
    // lane <-> lane^32 <-> the two wm waves.  (sum, sum of squares) partials in fp32 fail the
    // reference's two-pass variance when var << mean^2, e.g. on two-frame utterances.
    float skk[2][NC], ssum[2][NC], ssq[2][NC];
    float scount = 0.0f;                        // valid rows seen by this lane (same for every column)
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int c = 0; c < NC; ++c) skk[s][c] = ssum[s][c] = ssq[s][c] = 0.0f;
    const bool two_sets = d.nsets == 2;
    // Output-side tensors through bounds-checked descriptors spanning this batch element:
    // rows past the tile end get an out-of-range offset (loads give 0, stores are dropped), and
    // the aux / running-sum operands of a 32-row block are all in flight before the first use.
    const unsigned out_bytes = (unsigned)d.T * d.Fout * Cout * 4u;
    const size_t out_b = (size_t)b * d.T * d.Fout * Cout;
    const bool need_aux = d.epi == EAB_EPI_MULSIG || d.epi == EAB_EPI_ADD;
    const __amdgpu_buffer_rsrc_t r_dst = __builtin_amdgcn_make_buffer_rsrc(d.dst + out_b, 0, out_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_aux = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(need_aux ? d.aux + out_b : d.dst + out_b), 0, need_aux ? out_bytes : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_acc = __builtin_amdgcn_make_buffer_rsrc(
        d.dst_acc ? d.dst_acc + out_b : d.dst + out_b, 0, d.dst_acc ? out_bytes : 0u, 0x00020000);

    const unsigned row_bytes = (unsigned)(d.Fout * Cout) * 4u, step_bytes = (unsigned)(d.ostride * Cout) * 4u;
    const unsigned phase_bytes = (unsigned)(d.ophase * Cout) * 4u;
    if constexpr (PH2) {
        // ---- phase-pair epilogue (EAB_EPI_PHASE2): row q = (t, o) of the tile owns out[t][2o] (accumulator block 0) and
        // out[t][2o+1] (block 1; it exists while 2o+1 < Fout, i.e. not for the last o of a frame when Fout is odd).  Both
        // belong to channel ch[0], so the InstanceNorm partial of the channel takes both (shifted single pass as below; the
        // shift is the lane's first phase-0 value -- a row past the tile's end repeats a valid row there, any finite shift
        // gives the same (mean, M2)).  Straight-line code; FULL = no row of the tile lies past the end.
        const float a = st_slope[0][0];
        const float k0 = eab_prelu(acc[0][0][0] + bias_v[0], a);
        const unsigned chan_bytes = (unsigned)Cout * 4u;
        float su = 0.0f, sq = 0.0f, cnt = 0.0f;
        auto ph2_loop = [&](auto full_c) {
            constexpr bool FULL = decltype(full_c)::value;
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    const int qg = q0 + (wm * MI + mi) * 32 + 8 * r4 + 4 * lh;     // rows 4*r4 + j, j = 0..3: consecutive q
                    const int t = cg_div((FULL || qg < Q) ? qg : 0, d.No, inv_no);
                    int o = ((FULL || qg < Q) ? qg : 0) - t * d.No;
                    unsigned row_start = (unsigned)t * row_bytes;
                    unsigned cur = row_start + (unsigned)o * step_bytes;            // byte offset of out[t][2o][0]
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int r = 4 * r4 + j;
                        const bool ok0 = FULL || qg + j < Q;
                        const bool ok1 = ok0 && 2 * o + 1 < d.Fout;
                        const float v0 = acc[mi][0][r] + bias_v[0], v1 = acc[mi][1][r] + bias_v[1];
                        const unsigned o0 = cur + 4u * ch[0];
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v0), r_dst, ok0 ? o0 : CG_OOB, 0, 0);
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v1), r_dst, ok1 ? o0 + chan_bytes : CG_OOB, 0, 0);
                        const float e0 = eab_prelu(v0, a) - k0, e1 = eab_prelu(v1, a) - k0;
                        const float m0 = FULL ? e0 : (ok0 ? e0 : 0.0f), m1 = ok1 ? e1 : 0.0f;
                        su += m0;
                        sq = fmaf(m0, m0, sq);
                        su += m1;
                        sq = fmaf(m1, m1, sq);
                        cnt += (FULL ? 1.0f : (ok0 ? 1.0f : 0.0f)) + (ok1 ? 1.0f : 0.0f);
                        cur += step_bytes;
                        if (++o == d.No) {
                            o = 0;
                            row_start += row_bytes;
                            cur = row_start;
                        }
                    }
                }
            }
        };
        if (q0 + BM <= Q) ph2_loop(std::true_type{});
        else ph2_loop(std::false_type{});
        skk[0][0] = k0;
        ssum[0][0] = su;
        ssq[0][0] = sq;
        scount = cnt;
    } else {
    // Fast path (workgroup-uniform): a full tile with the plain / gated epilogue and at most one statistics set -- every
    // 2-D convolution of the inference program except the last tile of a batch element.  Straight-line code: no per-row
    // masks, no per-element dispatch on the epilogue kind.  Values are computed by the same expressions as below (a row's
    // result does not depend on the path its tile took); the statistics use the same shifted single pass.
    const bool fast_epi = !DUAL && q0 + BM <= Q && d.dst_acc == nullptr && !two_sets &&
                          (GLU ? true : (d.epi == EAB_EPI_LINEAR || d.epi == EAB_EPI_ADD));
    if (fast_epi) {
        // the two training-side extras as compile-time variants of the same loop: the gated epilogue's factor dump (forward
        // of the training program) and the accumulate-into operand (dgrad adding to an existing gradient, residual adds)
        auto fast_loop = [&](auto dump_c, auto add_c) {
            constexpr bool DUMPV = decltype(dump_c)::value, ADDV = decltype(add_c)::value;
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                unsigned off[16];
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    const int qg = q0 + (wm * MI + mi) * 32 + 8 * r4 + 4 * lh;
                    const int t = cg_div(qg, d.No, inv_no);
                    int o = qg - t * d.No;
                    unsigned row_start = (unsigned)t * row_bytes + phase_bytes;
                    unsigned cur = row_start + (unsigned)o * step_bytes;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        off[4 * r4 + j] = cur;
                        cur += step_bytes;
                        if (++o == d.No) {
                            o = 0;
                            row_start += row_bytes;
                            cur = row_start;
                        }
                    }
                }
                float auxv[16][NC];
                if (ADDV) {
#pragma unroll
                    for (int r = 0; r < 16; ++r)
#pragma unroll
                        for (int c = 0; c < NC; ++c)
                            auxv[r][c] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r_aux, off[r] + 4u * ch[c], 0, 0));
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
#pragma unroll
                    for (int c = 0; c < NC; ++c) {
                        float v;
                        if constexpr (GLU) {
                            const float av = acc[mi][0][r] + bias_v[0], sv = cg_sigmoid(acc[mi][1][r] + bias_v[1]);
                            v = av * sv;
                            if (DUMPV) {
                                float* dp = d.glu_dump + 2 * (out_b + (off[r] >> 2)) + n_blk + wn * 64 + li;
                                dp[0] = av;
                                dp[32] = sv;
                            }
                        } else {
                            v = acc[mi][c][r] + bias_v[c];
                        }
                        if (ADDV) v = v + auxv[r][c];
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r_dst, off[r] + 4u * ch[c], 0, 0);
                        acc[mi][c][r] = v;
                    }
                }
            }
        };
        const bool dump = GLU && d.glu_dump != nullptr, add = !GLU && d.epi == EAB_EPI_ADD;
        if (dump) fast_loop(std::true_type{}, std::false_type{});
        else if (add) fast_loop(std::false_type{}, std::true_type{});
        else fast_loop(std::false_type{}, std::false_type{});
        if (d.stats) {
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const float a = st_slope[0][c];
                const float k0 = eab_prelu(acc[0][c][0], a);
                float su = 0.0f, sq = 0.0f;
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float e = eab_prelu(acc[mi][c][r], a) - k0;
                        su += e;
                        sq = fmaf(e, e, sq);
                    }
                skk[0][c] = k0;
                ssum[0][c] = su;
                ssq[0][c] = sq;
            }
            scount = (float)(16 * MI);
        }
    } else
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        unsigned off[16];
        bool rowok[16];
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
            // rows r = 4*r4 + j, j = 0..3 are consecutive q: one division per group of four
            const int qg = q0 + (wm * MI + mi) * 32 + 8 * r4 + 4 * lh;
            const int t = cg_div(qg < Q ? qg : 0, d.No, inv_no);
            int o = (qg < Q ? qg : 0) - t * d.No;
            // byte offset of (t, o) stepped incrementally: two integer multiplies per group of four rows
            // instead of three per row (v_mul_lo_u32 is quarter rate and shares the pipe with the fp32 MFMA)
            unsigned row_start = (unsigned)t * row_bytes + phase_bytes;
            unsigned cur = row_start + (unsigned)o * step_bytes;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int r = 4 * r4 + j;
                rowok[r] = qg + j < Q;
                off[r] = rowok[r] ? cur : CG_OOB;
                cur += step_bytes;
                if (++o == d.No) {
                    o = 0;
                    row_start += row_bytes;
                    cur = row_start;
                }
            }
        }
        float auxv[16][NC], accv[16][NC];
        if (need_aux) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
#pragma unroll
                for (int c = 0; c < NC; ++c)
                    auxv[r][c] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                                               r_aux, rowok[r] ? off[r] + 4u * ch[c] : CG_OOB, 0, 0));
        }
        if (d.dst_acc) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
#pragma unroll
                for (int c = 0; c < NC; ++c)
                    accv[r][c] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                                               r_acc, rowok[r] ? off[r] + 4u * ch[c] : CG_OOB, 0, 0));
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                float v;
                if (GLU) {
                    const float av = acc[mi][0][r] + bias_v[0], sv = cg_sigmoid(acc[mi][1][r] + bias_v[1]);
                    v = av * sv;
                    if (d.glu_dump && rowok[r]) {
                        // training: value and sigmoid(gate) in the packed column order (what eab_glu_bwd_f32 reads)
                        float* dp = d.glu_dump + 2 * (out_b + (off[r] >> 2)) + n_blk + wn * 64 + li;
                        dp[0] = av;
                        dp[32] = sv;
                    }
                } else {
                    v = acc[mi][c][r] + bias_v[c];
                }
                if (d.epi == EAB_EPI_RELU) v = fmaxf(v, 0.0f);
                else if (d.epi == EAB_EPI_MULSIG) v = auxv[r][c] * cg_sigmoid(v);
                else if (d.epi == EAB_EPI_ADD) v = v + auxv[r][c];
                const unsigned o4 = rowok[r] ? off[r] + 4u * ch[c] : CG_OOB;
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r_dst, o4, 0, 0);
                if (d.dst_acc)
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v + accv[r][c]), r_acc, o4, 0, 0);
                if (rowok[r]) {
                    const bool first = scount == 0.0f;
                    const float g0 = eab_prelu(v, st_slope[0][c]);
                    if (first) skk[0][c] = g0;
                    const float e0 = g0 - skk[0][c];
                    ssum[0][c] += e0;
                    ssq[0][c] = fmaf(e0, e0, ssq[0][c]);
                    if (two_sets) {
                        const float g1 = eab_prelu(v, st_slope[1][c]);
                        if (first) skk[1][c] = g1;
                        const float e1 = g1 - skk[1][c];
                        ssum[1][c] += e1;
                        ssq[1][c] = fmaf(e1, e1, ssq[1][c]);
                    }
                }
            }
            if (rowok[r]) scount += 1.0f;
        }
    }

    }
    if (d.stats) {
        // per lane: (n, mean, M2) from the shifted sums; then lanes l and l^32 (same columns), then
        // the two wm waves through LDS; fixed order everywhere => bit-reproducible partials.
        float* red = &sm.a[0];               // staging LDS is free after the last barrier
        float mean[2][NC], m2[2][NC];
        float cnt = scount;
        const float inv_n = scount > 0.0f ? 1.0f / scount : 0.0f;
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                mean[s][c] = fmaf(ssum[s][c], inv_n, skk[s][c]);
                m2[s][c] = fmaxf(ssq[s][c] - ssum[s][c] * ssum[s][c] * inv_n, 0.0f);
            }
        const float cnt_o = __shfl_xor(cnt, 32);
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                float n = cnt;
                const float mo = __shfl_xor(mean[s][c], 32), qo = __shfl_xor(m2[s][c], 32);
                // both lanes of a pair must apply the merge in the same order: lower lane first
                if (lh == 0) cg_merge(n, mean[s][c], m2[s][c], cnt_o, mo, qo);
            }
        cnt += cnt_o;
        // red[wm][wn][s][c][li][3]
        if (lh == 0) {
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    const int i = ((((wm * 2 + wn) * 2 + s) * NC + c) * 32 + li) * 3;
                    red[i] = cnt;
                    red[i + 1] = mean[s][c];
                    red[i + 2] = m2[s][c];
                }
        }
        __syncthreads();
        if (wm == 0 && lh == 0) {
            const size_t tbase = ((size_t)b * d.stat_tiles + d.stat_tile0 + tile) * d.nsets;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                if (s >= d.nsets) break;
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    const int i0 = ((((0 * 2 + wn) * 2 + s) * NC + c) * 32 + li) * 3;
                    const int i1 = ((((1 * 2 + wn) * 2 + s) * NC + c) * 32 + li) * 3;
                    float n = red[i0], mu = red[i0 + 1], q = red[i0 + 2];
                    cg_merge(n, mu, q, red[i1], red[i1 + 1], red[i1 + 2]);
                    *reinterpret_cast<f32x4*>(&d.stats[((tbase + s) * Cout + ch[c]) * 4]) = f32x4{n, mu, q, 0.0f};
                }
            }
        }
        if (d.fz_counter) {
            // ---- fused finalisation: the last-arriving tile of batch element b merges all partials.
            // release our partial (agent scope), count the arrival, and if we are last acquire everyone else's.
            __threadfence();
            __syncthreads();
            if (tid == 0) sm.last = atomicAdd(&d.fz_counter[b], 1) == d.stat_tiles - 1;
            __syncthreads();
            if (sm.last) {
                __threadfence();
                // thread -> (set s, channel c, tile slice): partial sums over tiles slice, slice+NSL, .. in fp64,
                // combined through LDS in slice order: the same fixed order whichever workgroup does it.
                const int nch = d.nsets * Cout;                       // <= 256
                const int NSL = CG_THREADS / nch;                     // tile slices (host: 1 <= nch <= 256)
                const int e = tid % nch, slice = tid / nch;
                const int s = e / Cout, c = e - s * Cout;
                double* dred = reinterpret_cast<double*>(&sm.a[0]);   // [3][NSL][nch] doubles (<= 6 KB)
                double sn = 0.0, sm_ = 0.0, sq = 0.0;
                if (slice < NSL) {
                    const float* p = d.stats + (((size_t)b * d.stat_tiles) * d.nsets + s) * Cout * 4 + (size_t)c * 4;
                    const size_t stride = (size_t)d.nsets * Cout * 4;
                    for (int t = slice; t < d.stat_tiles; t += NSL) {
                        const f32x4 v = *reinterpret_cast<const f32x4*>(p + (size_t)t * stride);
                        const double n = (double)v[0], mu = (double)v[1];
                        sn += n;
                        sm_ = fma(n, mu, sm_);
                        sq += fma(n * mu, mu, (double)v[2]);
                    }
                    dred[(0 * NSL + slice) * nch + e] = sn;
                    dred[(1 * NSL + slice) * nch + e] = sm_;
                    dred[(2 * NSL + slice) * nch + e] = sq;
                }
                __syncthreads();
                if (slice == 0) {
                    sn = sm_ = sq = 0.0;
                    for (int k = 0; k < NSL; ++k) {
                        sn += dred[(0 * NSL + k) * nch + e];
                        sm_ += dred[(1 * NSL + k) * nch + e];
                        sq += dred[(2 * NSL + k) * nch + e];
                    }
                    const double mean = sn > 0.0 ? sm_ / sn : 0.0;
                    double var = sn > 0.0 ? sq / sn - mean * mean : 0.0;
                    if (var < 0.0) var = 0.0;
                    const float* gm = s == 0 ? d.fz_gamma0 : d.fz_gamma1;
                    const float* bt = s == 0 ? d.fz_beta0 : d.fz_beta1;
                    float* xf = s == 0 ? d.fz_xf0 : d.fz_xf1;
                    const double scale = (double)gm[c] / sqrt(var + (double)d.fz_eps);
                    *reinterpret_cast<float2*>(&xf[((size_t)b * Cout + c) * 2]) = make_float2((float)scale, (float)((double)bt[c] - mean * scale));
                }
                if (tid == 0) d.fz_counter[b] = 0;                    // re-armed for the next replay of the program
            }
        }
    }
}

extern "C" int eab_conv_tiles(int T, int No, int bm) {
    if (T <= 0 || No <= 0 || (bm != 16 && bm != 32 && bm != 64 && bm != 128)) return -1;
    long long q = (long long)T * No;
    return (int)((q + bm - 1) / bm);
}

int eab_conv_st(const eab_conv_desc* d, hipStream_t s);      // conv_st.hip

template <int MI, int NI, int KU, int MODE, int XF, bool VEC>
static int cg_launch(const eab_conv_desc* d, hipStream_t s) {
    constexpr int BM = 64 * MI, BN = 64 * NI;
    const int tiles = eab_conv_tiles(d->win.pos ? d->win.count : d->T, d->No, BM);
    dim3 grid((unsigned)(d->B * tiles), (unsigned)(d->N / BN));
    constexpr bool can_patch = KU == 1 && MODE != CG_DUAL && VEC && XF != EAB_XF_PRELU_NORM;
    if (d->korder == EAB_KORDER_CHUNK) {
        if constexpr (can_patch) {
            if (d->precision == EAB_PREC_F16X3)
                hipLaunchKernelGGL((conv_gemm_kernel<MI, NI, 1, MODE, XF, true, EAB_PREC_F16X3, true>), grid,
                                   dim3(CG_THREADS), 0, s, *d);
            else if (d->precision == EAB_PREC_BF16)
                hipLaunchKernelGGL((conv_gemm_kernel<MI, NI, 1, MODE, XF, true, EAB_PREC_BF16, true>), grid,
                                   dim3(CG_THREADS), 0, s, *d);
            else
                hipLaunchKernelGGL((conv_gemm_kernel<MI, NI, 1, MODE, XF, true, EAB_PREC_F32, true>), grid,
                                   dim3(CG_THREADS), 0, s, *d);
            EAB_RETURN_LAUNCH_STATUS();
        } else {
            return EAB_EUNSUPPORTED;
        }
    }
    if constexpr (MODE == CG_PH2) {
        return EAB_EUNSUPPORTED;                 // the phase-pair form exists in the patch pipeline only
    } else {
    if (d->precision == EAB_PREC_F16X3) {
        if constexpr (VEC)
            hipLaunchKernelGGL((conv_gemm_kernel<MI, NI, KU, MODE, XF, VEC, EAB_PREC_F16X3, false>), grid,
                               dim3(CG_THREADS), 0, s, *d);
        else
            return EAB_EUNSUPPORTED;
    } else if (d->precision == EAB_PREC_BF16) {
        if constexpr (VEC) {
            if (d->src_bf16) {                   // every source stored as bf16 (host check): the 32-channel walk
                if constexpr (KU == 1 && XF == EAB_XF_NONE && MODE != CG_DUAL)
                    hipLaunchKernelGGL((conv_gemm_kernel<MI, NI, KU, MODE, XF, VEC, EAB_PREC_BF16, false, true>), grid,
                                       dim3(CG_THREADS), 0, s, *d);
                else
                    return EAB_EUNSUPPORTED;
            } else {
                hipLaunchKernelGGL((conv_gemm_kernel<MI, NI, KU, MODE, XF, VEC, EAB_PREC_BF16, false>), grid,
                                   dim3(CG_THREADS), 0, s, *d);
            }
        } else {
            return EAB_EUNSUPPORTED;
        }
    } else {
        hipLaunchKernelGGL((conv_gemm_kernel<MI, NI, KU, MODE, XF, VEC, EAB_PREC_F32, false>), grid, dim3(CG_THREADS), 0,
                           s, *d);
    }
    EAB_RETURN_LAUNCH_STATUS();
    }
}

// worst-case number of input positions a BM-row tile touches (the kernel's P)
static long long cg_patch_positions(const eab_conv_desc* d) {
    int dt_min = 0, io_min = 0, io_max = 0;
    for (int j = 0; j < d->ntaps; ++j) {
        dt_min = d->dt[j] < dt_min ? d->dt[j] : dt_min;
        io_min = d->ioff[j] < io_min ? d->ioff[j] : io_min;
        io_max = d->ioff[j] > io_max ? d->ioff[j] : io_max;
    }
    const int hi_need = (d->No - 1) * d->istride + io_max - (d->Fin - 1);
    const long long Fp = d->Fin - io_min + (hi_need > 0 ? hi_need : 0);
    const long long rows = (d->bm - 1) / d->No + 2 - dt_min;
    return rows * Fp;
}

// K units per pipeline stage: big tiles keep three workgroups per CU resident
// (41 KB of LDS each) with KU = 1; the small-M kernels of the S-TCN are latency
// bound (one workgroup per CU, K = 256..320) and take four units per barrier.
template <int MI, int NI, int MODE, int XF, bool VEC>
static int cg_pick_ku(const eab_conv_desc* d, hipStream_t s, int ku) {
    if (ku == 4) {
        if constexpr (MI == 1 && MODE != CG_GLU && VEC) return cg_launch<MI, NI, 4, MODE, XF, VEC>(d, s);
        return EAB_EUNSUPPORTED;
    }
    if (ku == 2) {
        if constexpr (VEC && MODE != CG_DUAL) return cg_launch<MI, NI, 2, MODE, XF, VEC>(d, s);
        return EAB_EUNSUPPORTED;
    }
    return cg_launch<MI, NI, 1, MODE, XF, VEC>(d, s);
}

extern "C" int eab_conv_f32(const eab_conv_desc* d, eab_stream_t stream) {
    EAB_CHECK_ARG(d && d->src0 && d->w && d->dst);
    EAB_CHECK_ARG(d->B > 0 && d->T > 0 && d->Fin > 0 && d->Fout > 0 && d->No > 0);
    EAB_CHECK_ARG(d->C0 > 0 && d->C1 >= 0 && (d->C1 == 0) == (d->src1 == nullptr));
    EAB_CHECK_ARG(d->C1 == 0 || (d->C0 % 16) == 0);
    EAB_CHECK_ARG(d->ntaps > 0 && d->ntaps <= EAB_MAX_TAPS);
    EAB_CHECK_ARG(d->ostride >= 1 && d->istride >= 1 && d->ophase >= 0 && d->ophase < d->ostride);
    EAB_CHECK_ARG((d->No - 1) * d->ostride + d->ophase < d->Fout);
    const int upt = (d->C0 + d->C1 + 15) / 16;
    EAB_CHECK_ARG(d->Kpad == d->ntaps * upt * 16);
    // taps may look ahead in time (non-causal S-TCMs): rows past the utterance read as zero through
    // the per-utterance buffer bound; the patch pipeline is laid out for causal taps only
    for (int j = 0; j < d->ntaps; ++j) {
        EAB_CHECK_ARG(d->dt[j] < (1 << 20) && d->dt[j] > -(1 << 20));
        EAB_CHECK_ARG(d->dt[j] <= 0 || d->korder != EAB_KORDER_CHUNK);
    }
    EAB_CHECK_ARG(d->epi >= EAB_EPI_LINEAR && d->epi <= EAB_EPI_PHASE2);
    EAB_CHECK_ARG(d->precision == EAB_PREC_F32 || d->precision == EAB_PREC_F16X3 || d->precision == EAB_PREC_BF16);
    EAB_CHECK_ARG(d->korder == EAB_KORDER_TAP || d->korder == EAB_KORDER_CHUNK || d->korder == EAB_KORDER_FRAG);
    EAB_CHECK_ARG(d->korder == EAB_KORDER_FRAG || d->ph1_No == 0);
    if (d->korder == EAB_KORDER_CHUNK) {   // patch pipeline: the tile's input patch must fit its LDS area
        EAB_CHECK_ARG(cg_patch_positions(d) <= CG_PMAX && d->epi != EAB_EPI_DUALGATE);
        EAB_CHECK_ARG(d->C0 % 4 == 0 && d->C1 % 4 == 0 && d->xf_mode != EAB_XF_PRELU_NORM);
    }
    if (d->win.pos) {    // streaming window: no data-dependent statistics, causal taps only
        EAB_CHECK_ARG(d->win.count > 0 && d->stats == nullptr && d->fin_stats == nullptr);
        for (int j = 0; j < d->ntaps; ++j) EAB_CHECK_ARG(d->dt[j] <= 0);
    }
    const bool dual = d->epi == EAB_EPI_DUALGATE;
    const bool glu = d->epi == EAB_EPI_GLU;
    const bool ph2 = d->epi == EAB_EPI_PHASE2;
    EAB_CHECK_ARG(d->Cout == ((glu || dual || ph2) ? d->N / 2 : d->N));
    if (ph2) {
        // phase pair of a stride-2 transposed convolution: patch pipeline only, rows (t, o) with out[t][2o], out[t][2o+1];
        // one InstanceNorm set at most, no aux / running-sum operand, no streaming window (the small-tile kernel serves those)
        EAB_CHECK_ARG(d->korder == EAB_KORDER_CHUNK && d->N == 128 && d->ostride == 2 && d->ophase == 0 && d->istride == 1);
        EAB_CHECK_ARG(2 * d->No - 1 <= d->Fout && d->Fout <= 2 * d->No && d->nsets <= 1 && !d->dst_acc && !d->aux && !d->win.pos);
        EAB_CHECK_ARG((d->p2_mask1 & 1) == 1 && (unsigned)d->p2_mask1 < (1u << d->ntaps) && !d->fz_counter && !d->glu_dump);
    } else {
        EAB_CHECK_ARG(d->p2_mask1 == 0);
    }
    // one batch element of a source / of the output must be addressable with a 31-bit byte
    // offset, and rows per batch element must stay exact in the fp32-reciprocal division
    const long long per_b = (long long)d->T * d->Fin * (d->C0 > d->C1 ? d->C0 : d->C1) * 4;
    EAB_CHECK_ARG(per_b < (1ll << 31) && (long long)d->T * d->Fout * d->Cout * 4 < (1ll << 31));
    EAB_CHECK_ARG((long long)d->T * d->No < (1ll << 22));
    EAB_CHECK_ARG(d->xf_mode >= EAB_XF_NONE && d->xf_mode <= EAB_XF_PRELU_NORM);
    if (d->src_bf16) {
        // sources stored as bf16: bf16 products, the tap-ordered gather pipeline with one unit per stage (2-D layers), no fused
        // transform (the training programs' materialised activations and gradients), channel counts in whole units
        EAB_CHECK_ARG((d->src_bf16 & ~3) == 0 && d->precision == EAB_PREC_BF16 && d->korder == EAB_KORDER_TAP);
        EAB_CHECK_ARG(d->xf_mode == EAB_XF_NONE && d->fin_stats == nullptr && d->Fin > 1 && d->epi != EAB_EPI_DUALGATE);
        EAB_CHECK_ARG(d->C0 % 16 == 0 && d->C1 % 16 == 0 && (!(d->src_bf16 & 2) || d->src1));
        EAB_CHECK_ARG((!(d->src_bf16 & 1) || d->C0 % 32 == 0) && (!(d->src_bf16 & 2) || d->C1 % 32 == 0));   // 32-channel units
        EAB_CHECK_ARG(d->src_bf16 == (d->src1 ? 3 : 1));                  // all sources of a launch alike
    }
    const bool fin = d->fin_stats != nullptr;
    if (fin) {
        EAB_CHECK_ARG(d->xf_mode != EAB_XF_NONE && !d->xf0 && !d->xf1 && d->C1 == 0);
        EAB_CHECK_ARG(d->fin_tiles > 0 && d->fin_tiles <= 64 && d->fin_count > 0);
        EAB_CHECK_ARG(d->fin_nsets == (dual ? 2 : 1) || (!dual && d->fin_nsets == 2));
        EAB_CHECK_ARG(d->fin_gamma0 && d->fin_beta0 && d->slope0);
        EAB_CHECK_ARG(!dual || (d->fin_gamma1 && d->fin_beta1 && d->slope1));
    }
    if (d->xf_mode != EAB_XF_NONE) {
        EAB_CHECK_ARG(fin || (d->xf0 == nullptr) == (d->slope0 == nullptr));
        EAB_CHECK_ARG(fin || dual || (d->xf1 == nullptr) == (d->slope1 == nullptr));
    }
    if (dual) {
        EAB_CHECK_ARG(d->C1 == 0 && d->xf_mode != EAB_XF_NONE && d->slope0 && d->slope1);
        EAB_CHECK_ARG(fin || (d->xf0 && d->xf1));
    }
    EAB_CHECK_ARG((d->epi != EAB_EPI_MULSIG && d->epi != EAB_EPI_ADD) || d->aux);
    EAB_CHECK_ARG(d->nsets >= 0 && d->nsets <= 2 && (d->nsets == 0) == (d->stats == nullptr));
    EAB_CHECK_ARG(d->korder == EAB_KORDER_FRAG ? (d->bm == 16 || d->bm == 32 || d->bm == 64) : (d->bm == 64 || d->bm == 128));
    if (d->fz_counter) {
        EAB_CHECK_ARG(d->stats && d->nsets >= 1 && d->nsets * d->Cout <= CG_THREADS && d->win.pos == nullptr);
        EAB_CHECK_ARG(d->fz_gamma0 && d->fz_beta0 && d->fz_xf0 && (d->nsets == 1 || (d->fz_gamma1 && d->fz_beta1 && d->fz_xf1)));
    }
    if (d->stats) {
        const int tiles = eab_conv_tiles(d->T, d->No, d->bm) + (d->ph1_No > 0 ? eab_conv_tiles(d->T, d->ph1_No, d->bm) : 0);
        EAB_CHECK_ARG(d->stat_tile0 >= 0 && d->stat_tile0 + tiles <= d->stat_tiles);
    }
    EAB_CHECK_ARG((long long)d->B * (eab_conv_tiles(d->T, d->No, d->bm) + (d->ph1_No > 0 ? eab_conv_tiles(d->T, d->ph1_No, d->bm) : 0)) < (1ll << 31));
    hipStream_t s = eab_stream(stream);
    if (d->korder == EAB_KORDER_FRAG) return eab_conv_st(d, s);       // small-tile kernel (conv_st.hip)
    const int mi = d->bm / 64;
    const bool vec = (d->C0 % 4 == 0) && (d->C1 % 4 == 0);
    const bool has_xf = d->xf_mode != EAB_XF_NONE && (d->xf0 || d->xf1 || fin);
    const int xf = has_xf ? d->xf_mode : EAB_XF_NONE;
    if (xf != EAB_XF_NONE)   // transform tables live in LDS: CG_XFC channels per source
        EAB_CHECK_ARG(d->C0 <= CG_XFC && d->C1 <= CG_XFC && vec);
    int ku = (mi == 1 && !glu && vec && d->Fin == 1) ? 4 : 1;
    if (d->korder == EAB_KORDER_CHUNK) ku = 1;
    if (ku == 4 && !(mi == 1 && !glu && vec)) ku = 1;       // KU = 4 exists for the 64-row plain / dual tiles only
    if (ku == 2 && (!vec || dual)) ku = 1;
    if (ph2) {
        if (!vec || xf == EAB_XF_PRELU_NORM) return EAB_EUNSUPPORTED;
        if (xf == EAB_XF_NORM_PRELU)
            return mi == 2 ? cg_launch<2, 2, 1, CG_PH2, EAB_XF_NORM_PRELU, true>(d, s) : cg_launch<1, 2, 1, CG_PH2, EAB_XF_NORM_PRELU, true>(d, s);
        return mi == 2 ? cg_launch<2, 2, 1, CG_PH2, EAB_XF_NONE, true>(d, s) : cg_launch<1, 2, 1, CG_PH2, EAB_XF_NONE, true>(d, s);
    }
    if (dual) {
        if (d->N != 128 || mi != 1 || xf != EAB_XF_PRELU_NORM) return EAB_EUNSUPPORTED;
        return cg_pick_ku<1, 2, CG_DUAL, EAB_XF_PRELU_NORM, true>(d, s, ku == 2 ? 1 : ku);
    }
    if (glu) {
        if (d->N % 128 != 0 || xf == EAB_XF_PRELU_NORM) return EAB_EUNSUPPORTED;   // N > 128: GaGNet's gated in-convs
        if (!vec) return mi == 2 ? cg_pick_ku<2, 2, CG_GLU, 0, false>(d, s, 1) : cg_pick_ku<1, 2, CG_GLU, 0, false>(d, s, 1);
        if (xf == EAB_XF_NORM_PRELU)    // gated conv on raw producers (plain U-Net, is_u2 = False)
            return mi == 2 ? cg_pick_ku<2, 2, CG_GLU, EAB_XF_NORM_PRELU, true>(d, s, ku)
                           : cg_pick_ku<1, 2, CG_GLU, EAB_XF_NORM_PRELU, true>(d, s, ku);
        return mi == 2 ? cg_pick_ku<2, 2, CG_GLU, 0, true>(d, s, ku) : cg_pick_ku<1, 2, CG_GLU, 0, true>(d, s, ku);
    }
    if (!vec) return EAB_EUNSUPPORTED;          // only the first (gated) conv can see 2M % 4 != 0 channels
#define CG_DISPATCH_XF(MI_, NI_)                                                           \
    (xf == EAB_XF_NONE        ? cg_pick_ku<MI_, NI_, CG_PLAIN, EAB_XF_NONE, true>(d, s, ku) \
     : xf == EAB_XF_NORM_PRELU ? cg_pick_ku<MI_, NI_, CG_PLAIN, EAB_XF_NORM_PRELU, true>(d, s, ku) \
                               : cg_pick_ku<MI_, NI_, CG_PLAIN, EAB_XF_PRELU_NORM, true>(d, s, ku))
    if (d->N % 128 == 0) return mi == 2 ? CG_DISPATCH_XF(2, 2) : CG_DISPATCH_XF(1, 2);
    if (d->N % 64 == 0) return mi == 2 ? CG_DISPATCH_XF(2, 1) : CG_DISPATCH_XF(1, 1);
#undef CG_DISPATCH_XF
    return EAB_EUNSUPPORTED;
}

extern "C" int eab_conv_bf16(const eab_conv_desc* d, eab_stream_t stream) {
    EAB_CHECK_ARG(d);
    eab_conv_desc dd = *d;
    dd.precision = EAB_PREC_BF16;
    return eab_conv_f32(&dd, stream);
}
