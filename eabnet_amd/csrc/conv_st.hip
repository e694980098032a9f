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

template <int RB, int NCB, int MODE, int XF, bool BF>
__global__ __launch_bounds__(ST_THREADS, MODE == ST_DUAL ? 1 : 2) void conv_st_kernel(const eab_conv_desc d) {
    // BF = EAB_PREC_BF16: tensors and weights stay fp32 in memory (the same fragment-order `w`); A is rounded to bf16 on
    // its way into LDS, B in registers, products on v_mfma_f32_16x16x32_bf16 with fp32 accumulation
    constexpr bool DUAL = MODE == ST_DUAL;
    constexpr bool GATED = MODE != ST_PLAIN;             // a wave owns value and gate of the same 16 channels (NCB = 2)
    constexpr int BM = 16 * RB;
    constexpr int NA = DUAL ? 2 : 1;
    // ALL of this wave's B operands are fetched up front and stay in registers (K/16 x NCB b128 loads in flight at once: one
    // L2 round trip, overlapped with the A burst).  These launches run one or two waves per SIMD, so registers are free;
    // U bounds the K extent a variant accepts (host check): 320 for N = 64 and 128 (a five-tap S-TCM branch), 64 for N = 256.
    constexpr int U = NCB <= 2 ? 20 : 4;
    static_assert(!GATED || NCB == 2, "gated forms: one value and one gate block per wave");
    extern __shared__ __attribute__((aligned(16))) float st_lds[];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, kq = lane >> 4;

    // ---- which tile: (batch element, phase, tile) -----------------------------------------------------------------
    const int t_lo = d.win.pos ? *d.win.pos : 0;
    const int t_hi = d.win.pos ? (t_lo + d.win.count < d.T ? t_lo + d.win.count : d.T) : d.T;
    const int Tw = d.win.pos ? d.win.count : d.T;
    // two output-column phases of a transposed convolution in ONE launch: tiles [0, ph_tiles0) of a batch element
    // belong to phase 0 (d.No, d.ophase, d.w, taps d.dt/d.ioff), the rest to phase 1 (d.ph1_*)
    const int tiles0 = (Tw * d.No + BM - 1) / BM;
    const int tiles1 = d.ph1_No > 0 ? (Tw * d.ph1_No + BM - 1) / BM : 0;
    const int tiles_per_b = tiles0 + tiles1;
    unsigned vblk = blockIdx.x;
    if (gridDim.x >= 64) {                              // XCD-aware order: each XCD walks a contiguous eighth
        const unsigned G = gridDim.x, G8 = G >> 3, rem = G & 7, xcd = vblk & 7, idx = vblk >> 3;
        vblk = xcd * G8 + (xcd < rem ? xcd : rem) + idx;
    }
    // (readfirstlane: the quotient comes out of the vector unit; everything derived from it -- buffer descriptors above
    // all -- must be PROVABLY wave-uniform or every buffer access is wrapped in a waterfall loop)
    const int b = __builtin_amdgcn_readfirstlane((int)(vblk / (unsigned)tiles_per_b));
    int tile = (int)vblk - b * tiles_per_b;
    const int stat_tile = tile;                         // phase-1 partials follow the phase-0 ones
    const bool ph1 = tile >= tiles0;
    if (ph1) tile -= tiles0;
    const int No = ph1 ? d.ph1_No : d.No;
    const int ophase = ph1 ? d.ph1_ophase : d.ophase;
    const int ntaps = ph1 ? d.ph1_ntaps : d.ntaps;
    const int Kpad = ph1 ? d.ph1_Kpad : d.Kpad;
    const float* wfrag = ph1 ? d.ph1_w : d.w;
    const int Q = t_hi * No;
    const int q0 = t_lo * No + tile * BM;
    const float inv_no = 1.0f / (float)No;
    // row stride of the A tiles in ELEMENTS (fp32, or bf16 in the BF form): 16 bytes of padding per row
    const int LD = BF ? d.Kpad + 8 : d.Kpad + 4;        // (phase 0 has the larger K)
    const int M2 = Kpad >> 4;

    float* const sa = st_lds;
    StTables& tb = *reinterpret_cast<StTables*>(st_lds + (BF ? NA * BM * LD / 2 : NA * BM * LD));
    unsigned short* const sh = reinterpret_cast<unsigned short*>(st_lds);     // the A tiles as bf16 (BF)
    // diagnostic only (tools/diag_st_stamps.py): d.glu_dump, which this kernel has no other use for, may point to
    // 8 x 64-bit cycle stamps per workgroup; no output value depends on them and production descriptors leave it NULL
    unsigned long long* const stamps = reinterpret_cast<unsigned long long*>(d.glu_dump);
    auto stamp = [&](int i) {
        if (stamps && tid == 0) stamps[(size_t)blockIdx.x * 8 + i] = __builtin_amdgcn_s_memtime();
    };
    stamp(0);

    // ---- B prefetch: fragments of this wave's column blocks, K steps 0 .. U-1 ---------------------------------------
    // packed [N/16][M2][64 lanes][4]: element j of lane (i, kq) = W[16 nb + i][16 m2 + 8 (j>>1) + 2 kq + (j&1)]
    const float* wb[NCB];
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) wb[cb] = wfrag + ((size_t)(wave * NCB + cb) * M2 * 64 + lane) * 4;
    f32x4 bq[U][NCB];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int m2 = u < M2 ? u : 0;
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) bq[u][cb] = *reinterpret_cast<const f32x4*>(wb[cb] + (size_t)m2 * 256);
    }

    // fused second convolution (eab_conv_desc.f2_*; N = 256 launches only): its 64 x 256 weights, 16 columns per wave
    constexpr bool F2OK = NCB == 4 && !GATED && !BF;
    const bool f2 = F2OK && d.f2_w != nullptr;
    f32x4 bq2[F2OK ? 16 : 1];
    if (f2) {
#pragma unroll
        for (int u = 0; u < (F2OK ? 16 : 1); ++u)
            bq2[u] = *reinterpret_cast<const f32x4*>(d.f2_w + ((size_t)(wave * 16 + u) * 64 + lane) * 4);
    }

    // ---- tap tables -> LDS (needed for the gather addresses) --------------------------------------------------------------
    {
        // (constant indices keep the descriptor in the kernarg segment: scalar loads, then a select chain per lane)
        int tdt = 0, tio = 0;
#pragma unroll
        for (int j = 0; j < EAB_MAX_TAPS; ++j) {
            const int dj = ph1 ? d.ph1_dt[j] : d.dt[j], ij = ph1 ? d.ph1_ioff[j] : d.ioff[j];
            tdt = tid == j ? dj : tdt;
            tio = tid == j ? ij : tio;
        }
        if (tid < EAB_MAX_TAPS) tb.tap[tid] = make_int2(tdt, tio);
    }
    __syncthreads();

    // ---- A burst: every (row, tap, 4-channel group) of the tile, per source ---------------------------------------------
    const unsigned bytes0 = (unsigned)d.T * d.Fin * d.C0 * 4u;
    const unsigned bytes1 = (unsigned)d.T * d.Fin * d.C1 * 4u;
    const __amdgpu_buffer_rsrc_t rs0 = st_rsrc(d.src0 + (size_t)b * d.T * d.Fin * d.C0, bytes0);
    const __amdgpu_buffer_rsrc_t rs1 = st_rsrc(d.src1 ? d.src1 + (size_t)b * d.T * d.Fin * d.C1 : d.src0, d.src1 ? bytes1 : 0u);
    const int Ctot = d.C0 + d.C1;
    const int Cpad = (Ctot + 15) & ~15;                 // K extent of one tap
    const int Qm1 = Q > 0 ? Q - 1 : 0;
    // A source with Cs = 4 << SH4 channels: 1 << SH4 lanes fetch one (row, tap) position, the 256 threads cover RPS rows
    // per step, so a thread works on NS fixed rows (its (t, o) are computed once) and a fixed 4-channel group; the taps are
    // walked TB at a time with TB * NS >= 8 gathers in flight.  Branch-free addressing: a position outside the tensor or a
    // row past the tile's end gets an out-of-range offset (the load returns 0).
    constexpr int NSMAX = BM / 4;                        // row slots per thread for the widest source (256 channels)
    struct Rows {
        int t[NSMAX], f[NSMAX];
        bool ok[NSMAX];
    };
    constexpr int BSZ = NSMAX > 8 ? NSMAX : 8;          // gathers in flight per thread
    struct Batch {
        f32x4 v[BSZ];
        bool ok[BSZ];
    };
    auto rows_of = [&](auto sh4c, Rows& rw) {
        constexpr int SH4 = decltype(sh4c)::value, RPS = ST_THREADS >> SH4, NS = BM / RPS;
#pragma unroll
        for (int j = 0; j < NS; ++j) {
            const int q = q0 + (tid >> SH4) + j * RPS;
            rw.ok[j] = q < Q;
            const int qc = rw.ok[j] ? q : Qm1;
            const int t = st_div(qc, No, inv_no);
            rw.t[j] = t;
            rw.f[j] = (qc - t * No) * d.istride;
        }
    };
    auto issue = [&](auto sh4c, const Rows& rw, Batch& bt, const __amdgpu_buffer_rsrc_t rs, int tap0) {
        constexpr int SH4 = decltype(sh4c)::value, RPS = ST_THREADS >> SH4, NS = BM / RPS;
        constexpr int TB = NS >= 8 ? 1 : 8 / NS;
        const int Cs = 4 << SH4, c = (tid & ((1 << SH4) - 1)) << 2;
#pragma unroll
        for (int tt = 0; tt < TB; ++tt) {
            const bool live = tap0 + tt < ntaps;                              // workgroup-uniform
            const int2 tp = tb.tap[live ? tap0 + tt : 0];
#pragma unroll
            for (int j = 0; j < NS; ++j) {
                const int ti = rw.t[j] + tp.x, fi = rw.f[j] + tp.y;
                const bool ok = live & rw.ok[j] & (ti >= 0) & (ti < d.T) & (fi >= 0) & (fi < d.Fin);
                const unsigned off = ok ? (unsigned)(((ti * d.Fin + fi) * Cs + c) * 4) : ST_OOB;
                bt.ok[tt * NS + j] = ok;
                bt.v[tt * NS + j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0));
            }
        }
    };
    auto finish = [&](auto sh4c, const Batch& bt, int tap0, int coff, int table) {
        constexpr int SH4 = decltype(sh4c)::value, RPS = ST_THREADS >> SH4, NS = BM / RPS;
        constexpr int TB = NS >= 8 ? 1 : 8 / NS;
        const int c = (tid & ((1 << SH4) - 1)) << 2;
        f32x4 sh01[NA], sh23[NA], sl[NA];
        if (XF != EAB_XF_NONE) {
#pragma unroll
            for (int a = 0; a < NA; ++a) {
                const int tbl = DUAL ? a : table;
                sh01[a] = *reinterpret_cast<const f32x4*>(&tb.xft[tbl][c][0]);
                sh23[a] = *reinterpret_cast<const f32x4*>(&tb.xft[tbl][c + 2][0]);
                sl[a] = *reinterpret_cast<const f32x4*>(&tb.xsl[tbl][c]);
            }
        }
#pragma unroll
        for (int tt = 0; tt < TB; ++tt) {
            if (tap0 + tt < ntaps) {                                           // workgroup-uniform
#pragma unroll
                for (int j = 0; j < NS; ++j) {
                    float* dst = sa + ((tid >> SH4) + j * RPS) * LD + (tap0 + tt) * Cpad + coff + c;
#pragma unroll
                    for (int a = 0; a < NA; ++a) {
                        f32x4 x = bt.v[tt * NS + j];
                        if (XF != EAB_XF_NONE) {
                            x = st_xform<XF>(x, sh01[a], sh23[a], sl[a]);
                            x = bt.ok[tt * NS + j] ? x : f32x4{0.f, 0.f, 0.f, 0.f};   // zero padding acts on the NORMALISED tensor
                        }
                        if constexpr (BF) {
                            // bf16 image with the K order of the 16x16x32 MFMA's lanes: inside a 32-deep block, k = 16a + 8b +
                            // 2kq + e sits at 8kq + 4a + 2b + e, so lane kq reads its 8 operands of a block as one b128 and
                            // they pair with the fp32 weight fragments of two consecutive 16-deep steps
                            const int k0 = (tap0 + tt) * Cpad + coff + c;          // multiple of 4
                            const int blk = k0 & ~31, r32 = k0 & 31;
                            const int pos = blk + 8 * ((r32 & 7) >> 1) + 4 * (r32 >> 4) + 2 * ((r32 >> 3) & 1);
                            unsigned short* hrow = sh + (a * BM + (tid >> SH4) + j * RPS) * LD;
                            *reinterpret_cast<unsigned*>(hrow + pos) = st_bf2(x[0], x[1]);
                            *reinterpret_cast<unsigned*>(hrow + pos + 8) = st_bf2(x[2], x[3]);
                        } else {
                            *reinterpret_cast<f32x4*>(dst + a * BM * LD) = x;
                        }
                    }
                }
            }
        }
    };
    // the taps of a source from tap_start on, TB at a time
    auto stage_rest = [&](auto sh4c, const __amdgpu_buffer_rsrc_t rs, int coff, int table, const Rows& rw, bool skip_first) {
        constexpr int SH4 = decltype(sh4c)::value, NS = BM / (ST_THREADS >> SH4), TB = NS >= 8 ? 1 : 8 / NS;
        for (int tap0 = skip_first ? TB : 0; tap0 < ntaps; tap0 += TB) {
            Batch bt;
            issue(sh4c, rw, bt, rs, tap0);
            finish(sh4c, bt, tap0, coff, table);
        }
    };
    using SH16 = std::integral_constant<int, 4>;         // 64 channels
    using SH32 = std::integral_constant<int, 5>;         // 128
    using SH64 = std::integral_constant<int, 6>;         // 256
    const int sh4_0 = 31 - __builtin_clz((unsigned)(d.C0 >> 2));
    Rows rw0;
    Batch b0;
    // first batch of source 0: in flight while the transform tables are made
    if (sh4_0 == 4) { rows_of(SH16{}, rw0); issue(SH16{}, rw0, b0, rs0, 0); }
    else if (sh4_0 == 5) { rows_of(SH32{}, rw0); issue(SH32{}, rw0, b0, rs0, 0); }
    else { rows_of(SH64{}, rw0); issue(SH64{}, rw0, b0, rs0, 0); }
    stamp(1);

    // ---- transform tables -> LDS ----------------------------------------------------------------------------------------------
    if (XF != EAB_XF_NONE) {
        if (d.fin_stats) {
            // The producer left few partial tiles: merge them here.  (set, channel) pairs x tile slices over the 256 threads
            // (slice s takes tiles s, s + S, ..), fp64 partial sums joined through LDS in slice order -- the same bits in
            // every workgroup of the launch.
            const int P = d.fin_nsets * d.C0;            // <= 256 (host check)
            const int S = ST_THREADS / P;
            const int pair = tid % P, slice = tid / P;
            const int k = pair / d.C0, c = pair - k * d.C0;
            double sn = 0.0, sm = 0.0, sq = 0.0;
            if (slice < S) {
                const float* fp = d.fin_stats + ((((size_t)b * d.fin_tiles) * d.fin_nsets + k) * d.C0 + c) * 4;
                const size_t fstride = (size_t)d.fin_nsets * d.C0 * 4;
                constexpr int FC = 8;                    // partials fetched together: one round trip per 8 tiles of a slice
                for (int t0 = slice; t0 < d.fin_tiles; t0 += FC * S) {
                    f32x4 pv[FC];
#pragma unroll
                    for (int j = 0; j < FC; ++j) {
                        const int t = t0 + j * S < d.fin_tiles ? t0 + j * S : d.fin_tiles - 1;
                        pv[j] = *reinterpret_cast<const f32x4*>(fp + (size_t)t * fstride);
                    }
#pragma unroll
                    for (int j = 0; j < FC; ++j) {
                        const bool live = t0 + j * S < d.fin_tiles;
                        const double n = live ? (double)pv[j][0] : 0.0, mu = (double)pv[j][1];
                        sn += n;
                        sm = fma(n, mu, sm);
                        sq += live ? fma(n * mu, mu, (double)pv[j][2]) : 0.0;
                    }
                }
                tb.fin[0][slice * P + pair] = sn;
                tb.fin[1][slice * P + pair] = sm;
                tb.fin[2][slice * P + pair] = sq;
            }
            __syncthreads();
            if (slice == 0) {
                for (int j = 1; j < S; ++j) {
                    sn += tb.fin[0][j * P + pair];
                    sm += tb.fin[1][j * P + pair];
                    sq += tb.fin[2][j * P + pair];
                }
                const float* gm = k == 0 ? d.fin_gamma0 : d.fin_gamma1;
                const float* bt_ = k == 0 ? d.fin_beta0 : d.fin_beta1;
                const float* slk = k == 0 ? d.slope0 : d.slope1;
                const double mean = sn > 0.0 ? sm / sn : 0.0;
                double var = sn > 0.0 ? sq / sn - mean * mean : 0.0;
                if (var < 0.0) var = 0.0;
                const double scale = (double)gm[c] / sqrt(var + (double)d.fin_eps);
                tb.xft[k][c][0] = (float)scale;
                tb.xft[k][c][1] = (float)((double)bt_[c] - mean * scale);
                tb.xsl[k][c] = slk[c];
            }
        } else {
            const int k = tid >> 7, c = tid & (ST_XFC - 1);
            const int Ck = (k == 0 || DUAL) ? d.C0 : d.C1;
            const float* xfk = k == 0 ? d.xf0 : d.xf1;
            const float* slk = k == 0 ? d.slope0 : d.slope1;
            float sc = 1.0f, sh = 0.0f, sl = 1.0f;
            if (c < Ck && xfk) {
                const float2 v = *reinterpret_cast<const float2*>(&xfk[((size_t)b * Ck + c) * 2]);
                sc = v.x;
                sh = v.y;
                sl = slk[c];
            }
            tb.xft[k][c][0] = sc;
            tb.xft[k][c][1] = sh;
            tb.xsl[k][c] = sl;
        }
        __syncthreads();
    }
    if (sh4_0 == 4) { finish(SH16{}, b0, 0, 0, 0); stage_rest(SH16{}, rs0, 0, 0, rw0, true); }
    else if (sh4_0 == 5) { finish(SH32{}, b0, 0, 0, 0); stage_rest(SH32{}, rs0, 0, 0, rw0, true); }
    else { finish(SH64{}, b0, 0, 0, 0); stage_rest(SH64{}, rs0, 0, 0, rw0, true); }
    if (d.C1 > 0) {
        const int sh4_1 = 31 - __builtin_clz((unsigned)(d.C1 >> 2));
        Rows rw1;
        if (sh4_1 == 4) { rows_of(SH16{}, rw1); stage_rest(SH16{}, rs1, d.C0, 1, rw1, false); }
        else if (sh4_1 == 5) { rows_of(SH32{}, rw1); stage_rest(SH32{}, rs1, d.C0, 1, rw1, false); }
        else { rows_of(SH64{}, rw1); stage_rest(SH64{}, rs1, d.C0, 1, rw1, false); }
    }
    if (!BF && Cpad != Ctot) {                           // channel padding of a tap (C0 + C1 not a multiple of 16): zeros
        const int padc = Cpad - Ctot;
        for (int e = tid; e < BM * ntaps * padc; e += ST_THREADS) {
            const int c = e % padc, rt = e / padc;
            const int tap = rt / BM, row = rt & (BM - 1);
#pragma unroll
            for (int a = 0; a < NA; ++a) sa[a * BM * LD + row * LD + tap * Cpad + Ctot + c] = 0.0f;
        }
    }
    stamp(2);
    __syncthreads();
    stamp(3);

    // ---- main loop ----------------------------------------------------------------------------------------------------------
    // A fragments of UA K steps are read together (the LDS latency is paid once per UA steps, not once per MFMA pair);
    // every 16x16 block runs two accumulation chains (e = 0 / 1: the 16x16x4 MFMA's dependent latency is 40 cycles against
    // an issue interval of 32), summed at the end in a fixed order
    constexpr int NACC = 2;       // (the same chain structure for every tile shape: a row's bits must not depend on the tile it is in)
    constexpr int UA0 = RB * NA >= 8 ? 1 : RB * NA == 4 ? 2 : RB * NA == 2 ? 4 : 8;
    constexpr int UA = UA0 < U ? UA0 : U;
    f32x4 acc[NACC][RB][NCB];
#pragma unroll
    for (int n = 0; n < NACC; ++n)
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) acc[n][rb][cb] = f32x4{0.f, 0.f, 0.f, 0.f};
    if constexpr (BF) {
        constexpr int UB = U / 2;                        // 32-deep steps
        constexpr int UAB = UA >= 2 ? UA / 2 : 1;
        const unsigned short* h_lane = sh + li * LD + 8 * kq;
        const int M32 = M2 >> 1;
#pragma unroll
        for (int ua = 0; ua < UB; ua += UAB) {
            if (ua < M32) {                              // workgroup-uniform
                st_bf16x8 ah[UAB][NA][RB];
#pragma unroll
                for (int sI = 0; sI < UAB; ++sI) {
                    const int uc = ua + sI < M32 ? ua + sI : M32 - 1;
#pragma unroll
                    for (int a = 0; a < NA; ++a)
#pragma unroll
                        for (int rb = 0; rb < RB; ++rb)
                            ah[sI][a][rb] = *reinterpret_cast<const st_bf16x8*>(h_lane + (a * BM + rb * 16) * LD + 32 * uc);
                }
#pragma unroll
                for (int sI = 0; sI < UAB; ++sI) {
                    const int u = ua + sI;
                    if (u < UB && u < M32) {
#pragma unroll
                        for (int cb = 0; cb < NCB; ++cb) {
                            const f32x4 w0 = bq[2 * u < U ? 2 * u : 0][cb], w1 = bq[2 * u + 1 < U ? 2 * u + 1 : 0][cb];
                            const u32x4 bw = {st_bf2(w0[0], w0[1]), st_bf2(w0[2], w0[3]), st_bf2(w1[0], w1[1]), st_bf2(w1[2], w1[3])};
                            const st_bf16x8 bh = __builtin_bit_cast(st_bf16x8, bw);
#pragma unroll
                            for (int rb = 0; rb < RB; ++rb)
                                acc[u & 1][rb][cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[sI][DUAL ? cb : 0][rb], bh,
                                                                                               acc[u & 1][rb][cb], 0, 0, 0);
                        }
                    }
                }
            }
        }
    } else {
    const float* a_lane = sa + li * LD + 2 * kq;
    // (K extents beyond U steps -- the gated convolutions of the decoder, up to 768 deep -- take further passes: the weights of
    // the next U steps are fetched when a pass ends; single-pass launches never enter the reload)
    constexpr bool MP = MODE == ST_GLU && !BF;
    for (int mp = 0; mp < (MP ? M2 : 1); mp += U) {
    if (MP && mp > 0) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int m2 = mp + u < M2 ? mp + u : mp;
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) bq[u][cb] = *reinterpret_cast<const f32x4*>(wb[cb] + (size_t)m2 * 256);
        }
    }
#pragma unroll
    for (int ua = 0; ua < U; ua += UA) {
        if (mp + ua < M2) {                              // workgroup-uniform
            f32x2 af[UA][2][NA][RB];
#pragma unroll
            for (int sI = 0; sI < UA; ++sI) {
                const int m2c = mp + ua + sI < M2 ? mp + ua + sI : M2 - 1;
#pragma unroll
                for (int ms = 0; ms < 2; ++ms)
#pragma unroll
                    for (int a = 0; a < NA; ++a)
#pragma unroll
                        for (int rb = 0; rb < RB; ++rb)
                            af[sI][ms][a][rb] =
                                *reinterpret_cast<const f32x2*>(a_lane + a * BM * LD + rb * 16 * LD + 16 * m2c + 8 * ms);
            }
#pragma unroll
            for (int sI = 0; sI < UA; ++sI) {
                const int u = ua + sI;
                if (u < U && mp + u < M2) {
#pragma unroll
                    for (int ms = 0; ms < 2; ++ms)
#pragma unroll
                        for (int e = 0; e < 2; ++e)
#pragma unroll
                            for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                                for (int cb = 0; cb < NCB; ++cb)
                                    acc[NACC == 2 ? e : 0][rb][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(
                                        af[sI][ms][DUAL ? cb : 0][rb][e], bq[u < U ? u : 0][cb][2 * ms + e], acc[NACC == 2 ? e : 0][rb][cb], 0, 0, 0);
                }
            }
        }
    }
    }
    }
    if (NACC == 2) {
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) acc[0][rb][cb] += acc[NACC - 1][rb][cb];
    }
    stamp(4);
    // ---- epilogue -------------------------------------------------------------------------------------------------------------
    // lane holds, per (rb, cb): column (wave*NCB + cb)*16 + li (DUAL: channel wave*16 + li), rows rb*16 + 4*kq + r
    constexpr int NC = GATED ? 1 : NCB;                  // output-channel blocks held by this lane
    const int Cout = d.Cout;
    int ch[NC];
    float bias_v[NCB];
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb)      // gated: bias in the convolution's own order (value rows, then gate rows)
        bias_v[cb] = d.bias ? d.bias[GATED ? cb * d.Cout + wave * 16 + li : (wave * NCB + cb) * 16 + li] : 0.0f;
    if (GATED) {
        ch[0] = wave * 16 + li;
    } else {
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) ch[cb] = (wave * NCB + cb) * 16 + li;
    }
    const unsigned out_bytes = (unsigned)d.T * d.Fout * Cout * 4u;
    const size_t out_b = (size_t)b * d.T * d.Fout * Cout;
    const bool need_aux = d.epi == EAB_EPI_ADD;
    const __amdgpu_buffer_rsrc_t r_dst = st_rsrc(d.dst + out_b, out_bytes);
    const __amdgpu_buffer_rsrc_t r_aux = st_rsrc(need_aux ? d.aux + out_b : d.dst + out_b, need_aux ? out_bytes : 0u);
    const __amdgpu_buffer_rsrc_t r_acc = st_rsrc(d.dst_acc ? d.dst_acc + out_b : d.dst + out_b, d.dst_acc ? out_bytes : 0u);
    const unsigned row_bytes = (unsigned)(d.Fout * Cout) * 4u, step_bytes = (unsigned)(d.ostride * Cout) * 4u;
    const unsigned phase_bytes = (unsigned)(ophase * Cout) * 4u;

    unsigned off[RB][4];
    bool rowok[RB][4];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
        const int qg = q0 + rb * 16 + 4 * kq;            // four consecutive rows
        const int t = st_div(qg < Q ? qg : 0, No, inv_no);
        int o = (qg < Q ? qg : 0) - t * No;
        unsigned row_start = (unsigned)t * row_bytes + phase_bytes;
        unsigned cur = row_start + (unsigned)o * step_bytes;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            rowok[rb][r] = qg + r < Q;
            off[rb][r] = rowok[rb][r] ? cur : ST_OOB;
            cur += step_bytes;
            if (++o == No) {
                o = 0;
                row_start += row_bytes;
                cur = row_start;
            }
        }
    }
    float auxv[RB][4][NC], accv[RB][4][NC];
    if (need_aux) {
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c < NC; ++c)
                    auxv[rb][r][c] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                                                   r_aux, rowok[rb][r] ? off[rb][r] + 4u * ch[c] : ST_OOB, 0, 0));
    }
    if (d.dst_acc) {
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c < NC; ++c)
                    accv[rb][r][c] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                                                   r_acc, rowok[rb][r] ? off[rb][r] + 4u * ch[c] : ST_OOB, 0, 0));
    }
    float st_slope[2][NC];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const float* sp = s == 0 ? d.stat_slope0 : d.stat_slope1;
            st_slope[s][c] = (d.stats && s < d.nsets && sp) ? sp[ch[c]] : 1.0f;
        }
    // InstanceNorm partials in Welford form: per lane a shifted single pass over its <= 4*RB rows (shift = first valid
    // value), then Chan merges over the four lanes (kq) that hold the same column -- fixed order, bit-reproducible
    float skk[2][NC], ssum[2][NC], ssq[2][NC];
    float scount = 0.0f;
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int c = 0; c < NC; ++c) skk[s][c] = ssum[s][c] = ssq[s][c] = 0.0f;
    const bool two_sets = d.nsets == 2;
    constexpr int LD2 = 256 + 4;                         // fused second convolution: its A tile [BM][256 + 4] behind the tables
    float* const t2 = reinterpret_cast<float*>(&tb + 1);
    auto epi_loop = [&](auto with_stats) {
        constexpr bool STATS = decltype(with_stats)::value;
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                float v;
                if constexpr (GATED) {
                    v = (acc[0][rb][0][r] + bias_v[0]) * st_sigmoid(acc[0][rb][1][r] + bias_v[1]);
                } else {
                    v = acc[0][rb][c][r] + bias_v[c];
                }
                if (d.epi == EAB_EPI_RELU) v = fmaxf(v, 0.0f);
                else if (d.epi == EAB_EPI_ADD) v = v + auxv[rb][r][c];
                const unsigned o4 = rowok[rb][r] ? off[rb][r] + 4u * ch[c] : ST_OOB;
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r_dst, o4, 0, 0);
                if (F2OK && f2) t2[(rb * 16 + 4 * kq + r) * LD2 + ch[c]] = v;      // A operand of the fused second convolution
                if (d.dst_acc)
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v + accv[rb][r][c]), r_acc, o4, 0, 0);
                if constexpr (STATS) {   // statistics (selects, no branches: the shift is the lane's first valid value)
                    const bool ok = rowok[rb][r], first = ok && scount == 0.0f;
                    const float g0 = eab_prelu(v, st_slope[0][c]);
                    skk[0][c] = first ? g0 : skk[0][c];
                    const float e0 = ok ? g0 - skk[0][c] : 0.0f;
                    ssum[0][c] += e0;
                    ssq[0][c] = fmaf(e0, e0, ssq[0][c]);
                    const float g1 = eab_prelu(v, st_slope[1][c]);
                    skk[1][c] = first ? g1 : skk[1][c];
                    const float e1 = ok ? g1 - skk[1][c] : 0.0f;
                    ssum[1][c] += e1;
                    ssq[1][c] = fmaf(e1, e1, ssq[1][c]);
                }
            }
            scount += rowok[rb][r] ? 1.0f : 0.0f;
        }
    };
    if (d.stats) epi_loop(std::true_type{});
    else epi_loop(std::false_type{});
    if (d.stats) {
        const float inv_n = scount > 0.0f ? __builtin_amdgcn_rcpf(scount) : 0.0f;
        const size_t tbase = ((size_t)b * d.stat_tiles + d.stat_tile0 + stat_tile) * d.nsets;
        // (n, mean, M2) of every (set, column block), then the two merge steps over the lanes kq that hold the same column:
        // pairs (0,1) and (2,3), then (0,2); all exchanges of a step are issued together, every lane merges (only kq = 0 is
        // kept) -- the same order everywhere
        float sn[2][NC], smu[2][NC], sm2[2][NC];
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                sn[s][c] = scount;
                smu[s][c] = fmaf(ssum[s][c], inv_n, skk[s][c]);
                sm2[s][c] = fmaxf(ssq[s][c] - ssum[s][c] * ssum[s][c] * inv_n, 0.0f);
            }
#pragma unroll
        for (int step = 16; step <= 32; step <<= 1) {
            float on[2][NC], omu[2][NC], om2[2][NC];
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    on[s][c] = __shfl_xor(sn[s][c], step);
                    omu[s][c] = __shfl_xor(smu[s][c], step);
                    om2[s][c] = __shfl_xor(sm2[s][c], step);
                }
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int c = 0; c < NC; ++c) st_merge(sn[s][c], smu[s][c], sm2[s][c], on[s][c], omu[s][c], om2[s][c]);
        }
        if (kq == 0) {
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                if (s >= d.nsets) break;
#pragma unroll
                for (int c = 0; c < NC; ++c)
                    *reinterpret_cast<f32x4*>(&d.stats[((tbase + s) * Cout + ch[c]) * 4]) = f32x4{sn[s][c], smu[s][c], sm2[s][c], 0.0f};
            }
        }
    }
    if constexpr (F2OK) {
        if (f2) {
            // ---- fused second convolution: y2[row][n2] = sum_c W2[n2][c] * out[row][c] on the tile this workgroup just wrote
            __syncthreads();                             // every wave's 64 output columns are in t2
            f32x4 acc2[2][RB];
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) acc2[n][rb] = f32x4{0.f, 0.f, 0.f, 0.f};
            const float* a2_lane = t2 + li * LD2 + 2 * kq;
#pragma unroll
            for (int ua = 0; ua < 16; ua += 4) {
                f32x2 af2[4][2][RB];
#pragma unroll
                for (int sI = 0; sI < 4; ++sI)
#pragma unroll
                    for (int ms = 0; ms < 2; ++ms)
#pragma unroll
                        for (int rb = 0; rb < RB; ++rb)
                            af2[sI][ms][rb] = *reinterpret_cast<const f32x2*>(a2_lane + rb * 16 * LD2 + 16 * (ua + sI) + 8 * ms);
#pragma unroll
                for (int sI = 0; sI < 4; ++sI)
#pragma unroll
                    for (int ms = 0; ms < 2; ++ms)
#pragma unroll
                        for (int e = 0; e < 2; ++e)
#pragma unroll
                            for (int rb = 0; rb < RB; ++rb)
                                acc2[e][rb] = __builtin_amdgcn_mfma_f32_16x16x4f32(af2[sI][ms][rb][e], bq2[ua + sI][2 * ms + e],
                                                                                    acc2[e][rb], 0, 0, 0);
            }
            const int N2 = d.f2_N, ch2 = wave * 16 + li;
            const __amdgpu_buffer_rsrc_t r_d2 = st_rsrc(d.f2_dst + (size_t)b * d.T * N2, (unsigned)d.T * N2 * 4u);
            float sl2[2];
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const float* sp = s2 == 0 ? d.f2_stat_slope0 : d.f2_stat_slope1;
                sl2[s2] = (d.f2_stats && s2 < d.f2_nsets && sp) ? sp[ch2] : 1.0f;
            }
            float kk2[2] = {0.f, 0.f}, su2[2] = {0.f, 0.f}, sq2[2] = {0.f, 0.f}, cnt2 = 0.0f;
#pragma unroll
            for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int q = q0 + rb * 16 + 4 * kq + r;                       // No == 1: row = frame
                    const bool okr = q < Q;
                    const float v2 = acc2[0][rb][r] + acc2[1][rb][r];
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v2), r_d2,
                                                          okr ? (unsigned)((q * N2 + ch2) * 4) : ST_OOB, 0, 0);
                    if (d.f2_stats && okr) {
                        const bool first = cnt2 == 0.0f;
#pragma unroll
                        for (int s2 = 0; s2 < 2; ++s2) {
                            const float g = eab_prelu(v2, sl2[s2]);
                            if (first) kk2[s2] = g;
                            const float e2 = g - kk2[s2];
                            su2[s2] += e2;
                            sq2[s2] = fmaf(e2, e2, sq2[s2]);
                        }
                    }
                    if (okr) cnt2 += 1.0f;
                }
            if (d.f2_stats) {
                const float inv_n = cnt2 > 0.0f ? 1.0f / cnt2 : 0.0f;
                const size_t tbase = ((size_t)b * d.f2_stat_tiles + stat_tile) * d.f2_nsets;
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    if (s2 >= d.f2_nsets) break;
                    float n = cnt2;
                    float mean = fmaf(su2[s2], inv_n, kk2[s2]);
                    float m2 = fmaxf(sq2[s2] - su2[s2] * su2[s2] * inv_n, 0.0f);
                    float no = __shfl_xor(n, 16), mo = __shfl_xor(mean, 16), qo = __shfl_xor(m2, 16);
                    if ((kq & 1) == 0) st_merge(n, mean, m2, no, mo, qo);
                    no = __shfl_xor(n, 32);
                    mo = __shfl_xor(mean, 32);
                    qo = __shfl_xor(m2, 32);
                    if (kq == 0) {
                        st_merge(n, mean, m2, no, mo, qo);
                        *reinterpret_cast<f32x4*>(&d.f2_stats[((tbase + s2) * N2 + ch2) * 4]) = f32x4{n, mean, m2, 0.0f};
                    }
                }
            }
        }
    }
    if (stamps) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        stamp(5);
    }
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
    static bool attr_set = false;                        // per instantiation: allow more than the default 64 KB
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_st_kernel<RB, NCB, MODE, XF, BF>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return eab_hip_status(e);
        attr_set = true;
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
    EAB_CHECK_ARG(d->fz_counter == nullptr);          // (glu_dump: diagnostic stamp buffer or NULL)
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
