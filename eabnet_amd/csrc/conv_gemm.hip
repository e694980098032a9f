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
// Workgroup = 4 waves (2x2), wave tile (32*MI) x (32*NI), BK = 16.
// LDS tiles are [rows][16+4] floats: the +4 (one ds_read_b128 width) makes the
// 16-byte-slot stride odd (5), so the 64 lanes' fragment reads are conflict free
// (MI355X_MICROARCH §LDS).  Lane (i = l&31, h = l>>5) reads floats
// [8g+4h, 8g+4h+4) of its row as ONE ds_read_b128 and feeds them to four
// successive MFMA k-steps; A and B use the same k permutation, so the sum is
// unchanged.
//
// Pipeline: unit u+1 is fetched global->registers while unit u is multiplied;
// the fused producer-side InstanceNorm affine + PReLU is applied on the way
// registers->LDS; one barrier per unit, two LDS buffers.
//
// fp32-in MFMA is exact fp32 (fmaf chain), 64 FLOP/clk/SIMD: one MFMA occupies
// its SIMD for 64 cycles while the wave needs one A and one B VGPR for it, so
// LDS and staging traffic are far below their limits and the kernel is bound by
// the matrix pipe (roofline: "mfma", fp32 dense 157.3 TFLOP/s).
#include "common.h"

#define CG_THREADS 256
#define CG_BK 16
#define CG_LDK 20   // padded LDS row (floats)

template <int MI, int NI>
struct CgSmem {
    static constexpr int BM = 64 * MI, BN = 64 * NI;
    float a[2][BM * CG_LDK];
    float b[2][BN * CG_LDK];
    int dt[EAB_MAX_TAPS];
    int ioff[EAB_MAX_TAPS];
};

__device__ __forceinline__ f32x4 cg_xform(f32x4 v, f32x4 sh01, f32x4 sh23, f32x4 sl, int mode) {
    // sh01 = (scale0, shift0, scale1, shift1), sh23 likewise for channels 2,3
    const float sc[4] = {sh01[0], sh01[2], sh23[0], sh23[2]};
    const float sf[4] = {sh01[1], sh01[3], sh23[1], sh23[3]};
    f32x4 r;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (mode == EAB_XF_NORM_PRELU)
            r[j] = eab_prelu(fmaf(v[j], sc[j], sf[j]), sl[j]);
        else
            r[j] = fmaf(eab_prelu(v[j], sl[j]), sc[j], sf[j]);
    }
    return r;
}

template <int MI, int NI, bool GLU>
__global__ __launch_bounds__(CG_THREADS) void conv_gemm_kernel(const eab_conv_desc d) {
    constexpr int BM = 64 * MI, BN = 64 * NI;
    __shared__ __attribute__((aligned(16))) CgSmem<MI, NI> sm;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 31, lh = lane >> 5;

    const int Q = d.T * d.No;                       // rows per batch element
    const int tiles_per_b = (Q + BM - 1) / BM;
    const int b = blockIdx.x / tiles_per_b;
    const int tile = blockIdx.x - b * tiles_per_b;
    const int q0 = tile * BM;
    const int n_blk = blockIdx.y * BN;

    // constant indices keep the descriptor in the kernarg segment (a lane-indexed
    // read would force a scratch copy of the struct)
#pragma unroll
    for (int j = 0; j < EAB_MAX_TAPS; ++j)
        if (tid == j) {
            sm.dt[j] = d.dt[j];
            sm.ioff[j] = d.ioff[j];
        }

    const int Ctot = d.C0 + d.C1;
    const int UPT = (Ctot + 15) >> 4;
    const int NU = d.ntaps * UPT;
    const bool vec = ((d.C0 & 3) == 0) && ((d.C1 & 3) == 0);

    // ---- per-thread staging coordinates --------------------------------------
    const int srow = tid >> 2;      // 0..63
    const int skq = tid & 3;        // which float4 of the 16-wide unit
    int a_t[MI], a_f0[MI];
    long long a_pos[MI];
    bool a_ok[MI];
#pragma unroll
    for (int p = 0; p < MI; ++p) {
        int q = q0 + srow + 64 * p;
        a_ok[p] = q < Q;
        int t = a_ok[p] ? q / d.No : 0;
        int o = a_ok[p] ? q - t * d.No : 0;
        a_t[p] = t;
        a_f0[p] = o * d.istride;
        a_pos[p] = ((long long)b * d.T + t) * d.Fin + o * d.istride;
    }
    const float* wrow[NI];
#pragma unroll
    for (int p = 0; p < NI; ++p) wrow[p] = d.w + (size_t)(n_blk + srow + 64 * p) * d.Kpad + skq * 4;

    // Zero padding acts on the NORMALISED tensor in the reference (ConstantPad2d /
    // the transposed conv's implicit zeros come after norm+PReLU), so out-of-range
    // taps must stay exactly 0 through the fused transform: st_ok remembers which
    // staged rows are real.
    f32x4 ra[MI], rb[NI], r_sh01, r_sh23, r_sl;
    bool st_ok[MI];
    bool r_xf = false;

    __syncthreads();   // tap tables visible

    auto fetch = [&](int u) {
        const int tap = u / UPT;
        const int c0 = (u - tap * UPT) << 4;
        const int dt = sm.dt[tap], io = sm.ioff[tap];
        const bool second = (d.C1 > 0) && (c0 >= d.C0);
        const float* src = second ? d.src1 : d.src0;
        const float* xf = second ? d.xf1 : d.xf0;
        const float* sl = second ? d.slope1 : d.slope0;
        const int Cs = second ? d.C1 : d.C0;
        const int c = (second ? c0 - d.C0 : c0) + skq * 4;
        r_xf = (xf != nullptr) && (d.xf_mode != EAB_XF_NONE);
        if (vec) {
            const bool cok = c < Cs;        // Cs % 4 == 0: a float4 is all-in or all-out
#pragma unroll
            for (int p = 0; p < MI; ++p) {
                const int tt = a_t[p] + dt, fi = a_f0[p] + io;
                const bool ok = a_ok[p] && cok && tt >= 0 && fi >= 0 && fi < d.Fin;
                ra[p] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (ok) ra[p] = *reinterpret_cast<const f32x4*>(src + (a_pos[p] + (long long)dt * d.Fin + io) * Cs + c);
                st_ok[p] = ok;
            }
            if (r_xf && cok) {
                const float* xp = xf + ((size_t)b * Cs + c) * 2;
                r_sh01 = *reinterpret_cast<const f32x4*>(xp);
                r_sh23 = *reinterpret_cast<const f32x4*>(xp + 4);
                r_sl = *reinterpret_cast<const f32x4*>(sl + c);
            } else {
                r_sh01 = r_sh23 = f32x4{1.f, 0.f, 1.f, 0.f};
                r_sl = f32x4{1.f, 1.f, 1.f, 1.f};
            }
        } else {
            // channel counts that are not multiples of 4 (odd microphone counts): scalar gathers
#pragma unroll
            for (int p = 0; p < MI; ++p) {
                const int tt = a_t[p] + dt, fi = a_f0[p] + io;
                const bool ok = a_ok[p] && tt >= 0 && fi >= 0 && fi < d.Fin;
                st_ok[p] = ok;
                const float* ptr = src + (a_pos[p] + (long long)dt * d.Fin + io) * Cs;
#pragma unroll
                for (int j = 0; j < 4; ++j) ra[p][j] = (ok && c + j < Cs) ? ptr[c + j] : 0.0f;
            }
            float sc[4], sf[4], sv[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bool cok = r_xf && (c + j < Cs);
                sc[j] = cok ? xf[((size_t)b * Cs + c + j) * 2] : 1.0f;
                sf[j] = cok ? xf[((size_t)b * Cs + c + j) * 2 + 1] : 0.0f;
                sv[j] = cok ? sl[c + j] : 1.0f;
            }
            r_sh01 = f32x4{sc[0], sf[0], sc[1], sf[1]};
            r_sh23 = f32x4{sc[2], sf[2], sc[3], sf[3]};
            r_sl = f32x4{sv[0], sv[1], sv[2], sv[3]};
        }
#pragma unroll
        for (int p = 0; p < NI; ++p) rb[p] = *reinterpret_cast<const f32x4*>(wrow[p] + (size_t)u * CG_BK);
    };

    auto stash = [&](int buf) {
#pragma unroll
        for (int p = 0; p < MI; ++p) {
            f32x4 v = ra[p];
            if (r_xf) {
                f32x4 x = cg_xform(v, r_sh01, r_sh23, r_sl, d.xf_mode);
                v = st_ok[p] ? x : f32x4{0.f, 0.f, 0.f, 0.f};
            }
            *reinterpret_cast<f32x4*>(&sm.a[buf][(srow + 64 * p) * CG_LDK + skq * 4]) = v;
        }
#pragma unroll
        for (int p = 0; p < NI; ++p)
            *reinterpret_cast<f32x4*>(&sm.b[buf][(srow + 64 * p) * CG_LDK + skq * 4]) = rb[p];
    };

    f32x16 acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.0f;

    fetch(0);
    stash(0);
    __syncthreads();

    const int a_base = (wm * MI * 32 + li) * CG_LDK + 4 * lh;
    const int b_base = (wn * NI * 32 + li) * CG_LDK + 4 * lh;

    for (int u = 0; u < NU; ++u) {
        const int cur = u & 1;
        if (u + 1 < NU) fetch(u + 1);
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            f32x4 af[MI], bf[NI];
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
                af[mi] = *reinterpret_cast<const f32x4*>(&sm.a[cur][a_base + mi * 32 * CG_LDK + g * 8]);
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
                bf[ni] = *reinterpret_cast<const f32x4*>(&sm.b[cur][b_base + ni * 32 * CG_LDK + g * 8]);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[mi][s], bf[ni][s], acc[mi][ni], 0, 0, 0);
        }
        if (u + 1 < NU) stash(cur ^ 1);
        __syncthreads();
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
    float ssum[2][NC], ssq[2][NC];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int c = 0; c < NC; ++c) ssum[s][c] = ssq[s][c] = 0.0f;

#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = (wm * MI + mi) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            const int q = q0 + m;
            if (q >= Q) continue;
            const int t = q / d.No, o = q - t * d.No;
            const long long pos = ((long long)b * d.T + t) * d.Fout + o * d.ostride + d.ophase;
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                float v;
                if (GLU) {
                    v = (acc[mi][0][r] + bias_v[0]) * eab_sigmoid(acc[mi][1][r] + bias_v[1]);
                } else {
                    v = acc[mi][c][r] + bias_v[c];
                }
                const long long idx = pos * Cout + ch[c];
                if (d.epi == EAB_EPI_RELU) v = fmaxf(v, 0.0f);
                else if (d.epi == EAB_EPI_MULSIG) v = d.aux[idx] * eab_sigmoid(v);
                else if (d.epi == EAB_EPI_ADD) v = v + d.aux[idx];
                d.dst[idx] = v;
                if (d.dst_acc) d.dst_acc[idx] += v;
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const float g = eab_prelu(v, st_slope[s][c]);
                    ssum[s][c] += g;
                    ssq[s][c] = fmaf(g, g, ssq[s][c]);
                }
            }
        }
    }

    if (d.stats) {
        // lanes l and l^32 hold the same columns; then the two wm waves; fixed
        // order everywhere => bit-reproducible partials.
        float* red = &sm.a[0][0];               // staging LDS is free after the last barrier
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                ssum[s][c] += __shfl_xor(ssum[s][c], 32);
                ssq[s][c] += __shfl_xor(ssq[s][c], 32);
            }
        // red[wm][wn][s][c][li][2]
        if (lh == 0) {
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    const int i = ((((wm * 2 + wn) * 2 + s) * NC + c) * 32 + li) * 2;
                    red[i] = ssum[s][c];
                    red[i + 1] = ssq[s][c];
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
                    const int i0 = ((((0 * 2 + wn) * 2 + s) * NC + c) * 32 + li) * 2;
                    const int i1 = ((((1 * 2 + wn) * 2 + s) * NC + c) * 32 + li) * 2;
                    float2 o2 = make_float2(red[i0] + red[i1], red[i0 + 1] + red[i1 + 1]);
                    *reinterpret_cast<float2*>(&d.stats[((tbase + s) * Cout + ch[c]) * 2]) = o2;
                }
            }
        }
    }
}

extern "C" int eab_conv_tiles(int T, int No, int bm) {
    if (T <= 0 || No <= 0 || (bm != 64 && bm != 128)) return -1;
    long long q = (long long)T * No;
    return (int)((q + bm - 1) / bm);
}

template <int MI, int NI, bool GLU>
static int cg_launch(const eab_conv_desc* d, hipStream_t s) {
    constexpr int BM = 64 * MI, BN = 64 * NI;
    const int tiles = eab_conv_tiles(d->T, d->No, BM);
    dim3 grid((unsigned)(d->B * tiles), (unsigned)(d->N / BN));
    hipLaunchKernelGGL((conv_gemm_kernel<MI, NI, GLU>), grid, dim3(CG_THREADS), 0, s, *d);
    EAB_RETURN_LAUNCH_STATUS();
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
    for (int j = 0; j < d->ntaps; ++j) EAB_CHECK_ARG(d->dt[j] <= 0 && d->dt[j] > -(1 << 20));
    EAB_CHECK_ARG(d->xf_mode >= EAB_XF_NONE && d->xf_mode <= EAB_XF_PRELU_NORM);
    if (d->xf_mode != EAB_XF_NONE) {
        EAB_CHECK_ARG((d->xf0 == nullptr) == (d->slope0 == nullptr));
        EAB_CHECK_ARG((d->xf1 == nullptr) == (d->slope1 == nullptr));
    }
    EAB_CHECK_ARG(d->epi >= EAB_EPI_LINEAR && d->epi <= EAB_EPI_ADD);
    const bool glu = d->epi == EAB_EPI_GLU;
    EAB_CHECK_ARG(d->Cout == (glu ? d->N / 2 : d->N));
    EAB_CHECK_ARG((d->epi != EAB_EPI_MULSIG && d->epi != EAB_EPI_ADD) || d->aux);
    EAB_CHECK_ARG(d->nsets >= 0 && d->nsets <= 2 && (d->nsets == 0) == (d->stats == nullptr));
    EAB_CHECK_ARG(d->bm == 64 || d->bm == 128);
    if (d->stats) {
        const int tiles = eab_conv_tiles(d->T, d->No, d->bm);
        EAB_CHECK_ARG(d->stat_tile0 >= 0 && d->stat_tile0 + tiles <= d->stat_tiles);
    }
    EAB_CHECK_ARG((long long)d->B * eab_conv_tiles(d->T, d->No, d->bm) < (1ll << 31));
    hipStream_t s = eab_stream(stream);
    const int mi = d->bm / 64;
    if (glu) {
        if (d->N != 128) return EAB_EUNSUPPORTED;
        return mi == 2 ? cg_launch<2, 2, true>(d, s) : cg_launch<1, 2, true>(d, s);
    }
    if (d->N % 128 == 0) return mi == 2 ? cg_launch<2, 2, false>(d, s) : cg_launch<1, 2, false>(d, s);
    if (d->N % 64 == 0) return mi == 2 ? cg_launch<2, 1, false>(d, s) : cg_launch<1, 1, false>(d, s);
    return EAB_EUNSUPPORTED;
}
