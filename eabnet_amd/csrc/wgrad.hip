// Weight gradient of every convolution / linear map of the path, on fp32 MFMA (v_mfma_f32_32x32x2_f32):
//     dW[n][tap][c] += sum over rows (b, t, o) of  dz[b][t][o*ostride+ophase][n] * x[b][t+dt_tap][o*istride+ioff_tap][c]
// i.e. the forward gather geometry of eab_conv_f32 with the roles of "output channel" and "row" exchanged: the
// reduction runs over the rows.  (What autograd computes for Conv2d / ConvTranspose2d / Conv1d / Linear / the LSTM
// input and recurrent matrices in train_distributed.py:228; reference layers as listed for eab_conv_f32.)
//
// GEMM view: M = N output channels (tile TN = 64 or 128), N = Kpad weight columns (tile TC = 64 or 128 = 4 or 8 units of 16
// channels), K = rows.  dz rows and gathered x rows are staged [16 rows][cols] in LDS by coalesced 16-byte loads
// (a unit of one row is 64 contiguous bytes, zero where the tap leaves the tensor), and both MFMA operands are
// read K-major straight from there: lane (i, h) takes  As[2s+h][n0+i]  and  Bs[2s+h][c0+i]  -- 32 consecutive
// floats per half wave, row stride = 32 mod 64 floats, so the two halves use disjoint banks.
// Split-K over the rows: grid.x row groups accumulate into dW with fp32 atomics (dW zeroed by the caller).  The
// column sums of dz (the bias gradient) ride along in the workgroups of column block 0 (dbias, optional).
// Roofline "mfma" (fp32 dense, 157.3 TFLOP/s); algorithmic bytes = rows * (N + ntaps*C) * 4 per launch.
#include "common.h"
#include <math.h>

#define WG_THREADS 256
#define WG_ROWS 16           // rows per pipeline stage

template <int TN, int TC>
struct WgSmem {
    static constexpr int LDA = TN + 32, LDB = TC + 32;
    float a[2][WG_ROWS * LDA];
    float b[2][WG_ROWS * LDB];
};

// (b, t, o) of a flattened row index, advanced by a constant number of rows per stage without divisions
struct WgRow {
    int b, t, o;
    __device__ __forceinline__ void init(long long r, int T, int No) {
        const long long rows_b = (long long)T * No;
        b = (int)(r / rows_b);
        const int q = (int)(r - (long long)b * rows_b);
        t = q / No;
        o = q - t * No;
    }
    // advance by n = a * No + rem rows (a, rem workgroup-uniform, computed once): one conditional wrap of o instead of a
    // per-lane loop (for the deep layers No is 4..10, so a 16-row stage wraps o several times)
    __device__ __forceinline__ void advance(int a, int rem, int T, int No) {
        o += rem;
        t += a;
        if (o >= No) {
            o -= No;
            ++t;
        }
        while (t >= T) {                                     // (only at a batch-element boundary)
            t -= T;
            ++b;
        }
    }
    // the same step, telling whether o wrapped: byte offsets of a row are then stepped without multiplies --
    // offset(row) is affine in (b*T + t, o), and b*T + t just keeps counting across batch elements
    __device__ __forceinline__ bool step(int a, int rem, int T, int No) {
        o += rem;
        t += a;
        const bool wrap = o >= No;
        if (wrap) {
            o -= No;
            ++t;
        }
        while (t >= T) {
            t -= T;
            ++b;
        }
        return wrap;
    }
};

// One launch serves up to WG_MAXB weight gradients of IDENTICAL geometry (the 18 S-TCMs' in/left/right/out convolutions, the
// repeated U-Net levels): the shared descriptor plus per-member pointers travel in the kernel arguments, blockIdx.x =
// member * groups + row group.
#define WG_MAXB 24
struct WgBatch {
    eab_wgrad_desc d;
    int n, groups;
    const float* dz[WG_MAXB];
    const float* src0[WG_MAXB];
    const float* src1[WG_MAXB];
    float* dw[WG_MAXB];
    float* dbias[WG_MAXB];
};

// VEC (compile time): channel counts are multiples of 4 -> 16-byte gathers; false only for the network input of an odd
// number of microphones.  A run-time flag around the operand loads breaks their burst (see wgrad_bf_kernel).
template <int TN, int TC, bool VEC>
__global__ __launch_bounds__(WG_THREADS) void wgrad_kernel(const WgBatch bt) {
    using Smem = WgSmem<TN, TC>;
    const eab_wgrad_desc& d = bt.d;
    const int member = bt.n > 1 ? (int)(blockIdx.x / (unsigned)bt.groups) : 0;
    const int row_group = (int)blockIdx.x - member * bt.groups;
    const float* const m_dz = bt.dz[member];
    const float* const m_src0 = bt.src0[member];
    const float* const m_src1 = bt.src1[member];
    float* const m_dw = bt.dw[member];
    float* const m_dbias = bt.dbias[member];
    constexpr int LDA = Smem::LDA, LDB = Smem::LDB, MI = TN / 64, NJ = TC / 64;
    constexpr int AP = TN / 64;                                  // float4 loads of dz per thread per stage
    constexpr int BP = TC / 64;                                  // float4 loads of x per thread per stage
    __shared__ __attribute__((aligned(16))) Smem sm;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, li = lane & 31, lh = lane >> 5;

    const long long R = (long long)d.T * d.No * d.B;
    const int adv_a = WG_ROWS / d.No, adv_r = WG_ROWS - adv_a * d.No;         // a stage advances every row walker by WG_ROWS rows
    const long long r_begin = (long long)row_group * d.rows_per_wg;
    const long long r_end = r_begin + d.rows_per_wg < R ? r_begin + d.rows_per_wg : R;
    const int col0 = blockIdx.y * TC, n0 = blockIdx.z * TN;
    const int Ctot = d.C0 + d.C1, UPT = (Ctot + 15) >> 4, NU = d.ntaps * UPT;

    // ---- staging roles --------------------------------------------------------------------------------
    // A (dz): thread -> row (tid>>5) + 8p (TN = 128) / tid>>4 (TN = 64), float4 column
    const int a_row = TN == 128 ? tid >> 5 : tid >> 4, a_c4 = TN == 128 ? tid & 31 : tid & 15;
    // B (gathered x): thread -> row tid>>4, float4 (tid&15) + 16j of the TC-wide block: unit = (tid&15)/4 + 4j
    const int b_row = tid >> 4, b_q = tid & 3;
    int tdt[BP], tio[BP], b_cc[BP], b_Cs[BP];
    const float* b_src[BP];
    bool b_ok[BP];
#pragma unroll
    for (int j = 0; j < BP; ++j) {
        const int u = (col0 >> 4) + ((tid & 15) >> 2) + 4 * j;
        const bool u_ok = u < NU;
        const int tap = u_ok ? u / UPT : 0;
        const int c0 = (u_ok ? u - tap * UPT : 0) << 4;
        const bool second = d.C1 > 0 && c0 >= d.C0;
        b_Cs[j] = second ? d.C1 : d.C0;
        b_cc[j] = (second ? c0 - d.C0 : c0) + b_q * 4;
        b_ok[j] = u_ok && b_cc[j] < b_Cs[j];
        b_src[j] = second ? m_src1 : m_src0;
        tdt[j] = tio[j] = 0;
#pragma unroll
        for (int k = 0; k < EAB_MAX_TAPS; ++k)
            if (k == tap) {
                tdt[j] = d.dt[k];
                tio[j] = d.ioff[k];
            }
    }
    // Operands through bounds-checked buffer descriptors with 32-bit byte offsets (host: every tensor < 2 GiB), stepped
    // incrementally: a stage moves every row walker by WG_ROWS rows = (adv_a, adv_r) in (t, o) plus one conditional wrap, and
    // a row's offset is affine in (b*T + t, o) -- no multiplies, no 64-bit address arithmetic, no branches around the loads
    // (an invalid row or tap gets an out-of-range offset and reads 0).
    const __amdgpu_buffer_rsrc_t rs_dz = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(m_dz), 0, (unsigned)((size_t)d.B * d.T * d.Fz * d.N * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_s0 = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(m_src0), 0, (unsigned)((size_t)d.B * d.T * d.Fin * d.C0 * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_s1 = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(m_src1 ? m_src1 : m_src0), 0, m_src1 ? (unsigned)((size_t)d.B * d.T * d.Fin * d.C1 * 4) : 0u, 0x00020000);
    constexpr unsigned WG_OOB = 0x80000000u;
    WgRow ra_[AP], rb_;
    unsigned offA[AP];
    int offB[BP];                                                // (may be negative while the tap is out of range: unused then)
#pragma unroll
    for (int p = 0; p < AP; ++p) {
        ra_[p].init(r_begin + a_row + 8 * p, d.T, d.No);
        offA[p] = (unsigned)(((((size_t)ra_[p].b * d.T + ra_[p].t) * d.Fz + (size_t)ra_[p].o * d.ostride + d.ophase) * d.N + n0 + a_c4 * 4) * 4);
    }
    rb_.init(r_begin + b_row, d.T, d.No);
    const unsigned dA0 = (unsigned)((adv_a * d.Fz + adv_r * d.ostride) * d.N * 4), dA1 = (unsigned)((d.Fz - d.No * d.ostride) * d.N * 4);
    int dB0[BP], dB1[BP];
    bool b_second[BP];
#pragma unroll
    for (int j = 0; j < BP; ++j) {
        b_second[j] = d.C1 > 0 && b_src[j] == m_src1;
        const long long bt = (long long)rb_.b * d.T + rb_.t + tdt[j];
        offB[j] = (int)(((bt * d.Fin + (long long)rb_.o * d.istride + tio[j]) * b_Cs[j] + b_cc[j]) * 4);
        dB0[j] = (adv_a * d.Fin + adv_r * d.istride) * b_Cs[j] * 4;
        dB1[j] = (d.Fin - d.No * d.istride) * b_Cs[j] * 4;
    }

    f32x4 ra[AP], rb[BP];
    f32x4 bsum[AP];                                              // bias gradient: column sums of dz (column block 0 only)
#pragma unroll
    for (int p = 0; p < AP; ++p) bsum[p] = f32x4{0.f, 0.f, 0.f, 0.f};
    const bool do_bias = m_dbias != nullptr && blockIdx.y == 0;
    constexpr bool vec_ok = VEC;                                   // (two sources require C0 % 16 == 0 and C1 % 4 == 0)
    auto fetch = [&](long long r0) {
#pragma unroll
        for (int p = 0; p < AP; ++p) {
            const bool ok = r0 + a_row + 8 * p < r_end;
            ra[p] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_dz, ok ? offA[p] : WG_OOB, 0, 0));
            offA[p] += dA0 + (ra_[p].step(adv_a, adv_r, d.T, d.No) ? dA1 : 0u);
        }
        const bool rok = r0 + b_row < r_end;
#pragma unroll
        for (int j = 0; j < BP; ++j) {
            const int tt = rb_.t + tdt[j], fi = rb_.o * d.istride + tio[j];
            const bool ok = rok && b_ok[j] && tt >= 0 && tt < d.T && fi >= 0 && fi < d.Fin;
            const unsigned off = ok ? (unsigned)offB[j] : WG_OOB;
            if (vec_ok) {
                rb[j] = __builtin_bit_cast(f32x4, b_second[j] ? __builtin_amdgcn_raw_buffer_load_b128(rs_s1, off, 0, 0)
                                                              : __builtin_amdgcn_raw_buffer_load_b128(rs_s0, off, 0, 0));
            } else {                // channel count not a multiple of 4 (2M = 18 network inputs): dword loads, row tail guarded
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    rb[j][e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                                             rs_s0, (ok && b_cc[j] + e < b_Cs[j]) ? off + 4u * e : WG_OOB, 0, 0));
            }
        }
        const bool wrap = rb_.step(adv_a, adv_r, d.T, d.No);
#pragma unroll
        for (int j = 0; j < BP; ++j) offB[j] += dB0[j] + (wrap ? dB1[j] : 0);
    };
    auto stash = [&](int buf) {
#pragma unroll
        for (int p = 0; p < AP; ++p) {
            *reinterpret_cast<f32x4*>(&sm.a[buf][(a_row + 8 * p) * LDA + a_c4 * 4]) = ra[p];
            if (do_bias) bsum[p] += ra[p];
        }
#pragma unroll
        for (int j = 0; j < BP; ++j) *reinterpret_cast<f32x4*>(&sm.b[buf][b_row * LDB + (tid & 15) * 4 + 64 * j]) = rb[j];
    };

    f32x16 acc[MI][NJ];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int nj = 0; nj < NJ; ++nj)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][nj][r] = 0.0f;

    auto lds_barrier = [] { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
    const int a_off = wm * (TN / 2) + li, b_off = wn * (TC / 2) + li;

    if (r_begin < r_end) {
        fetch(r_begin);
        stash(0);
        lds_barrier();
        int cur = 0;
        for (long long r0 = r_begin; r0 < r_end; r0 += WG_ROWS) {
            const bool more = r0 + WG_ROWS < r_end;
            if (more) fetch(r0 + WG_ROWS);
#pragma unroll
            for (int s = 0; s < WG_ROWS / 2; ++s) {
                float av[MI], bv[NJ];
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) av[mi] = sm.a[cur][(2 * s + lh) * LDA + a_off + mi * 32];
#pragma unroll
                for (int nj = 0; nj < NJ; ++nj) bv[nj] = sm.b[cur][(2 * s + lh) * LDB + b_off + nj * 32];
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int nj = 0; nj < NJ; ++nj)
                        acc[mi][nj] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[mi], bv[nj], acc[mi][nj], 0, 0, 0);
            }
            if (more) stash(cur ^ 1);
            lds_barrier();
            cur ^= 1;
        }
    }

    // ---- accumulate the tile into dW[n][col]: C/D map of the 32x32 MFMA: column = lane&31, row = (r&3)+8*(r>>2)+4*(lane>>5)
#pragma unroll
    for (int nj = 0; nj < NJ; ++nj) {
        const int col = col0 + wn * (TC / 2) + nj * 32 + li;
        if (col < d.Kpad) {
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int n = n0 + wm * (TN / 2) + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    atomicAdd(&m_dw[(size_t)n * d.Kpad + col], acc[mi][nj][r]);
                }
        }
    }
    if (m_dbias != nullptr && blockIdx.y == 0) {     // (workgroup-uniform)
        // the 16 row lanes of a column are summed through LDS first: ONE atomic per column and workgroup (per-thread
        // atomics on the same 128 addresses from every row group serialised in L2: 9 -> 34 ms per step when tried)
        __syncthreads();
#pragma unroll
        for (int p = 0; p < AP; ++p) *reinterpret_cast<f32x4*>(&sm.a[0][(a_row + 8 * p) * LDA + a_c4 * 4]) = bsum[p];
        __syncthreads();
        if (tid < TN) {
            float t = 0.0f;
#pragma unroll
            for (int r = 0; r < WG_ROWS; ++r) t += sm.a[0][r * LDA + tid];
            atomicAdd(&m_dbias[n0 + tid], t);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// bf16 products (EAB_PREC_BF16; the training dtype of BASELINE configs[3]): the same split-K tiling, but the two operands are
// rounded to bf16 on their way into LDS and multiplied on v_mfma_f32_32x32x16_bf16 (one instruction per 16-row stage instead
// of eight fp32 ones); fp32 accumulation, fp32 atomics into dW, bias gradient from the unrounded dz.
// LDS holds the operands TRANSPOSED, [column][row pair] with two bf16 (rows 2p, 2p+1) per dword and 9 dwords per column (odd
// stride: the dword writes of a float4's four columns and the 4-dword fragment reads both spread over the banks): lane
// (i, g) of the MFMA reads the dwords 4g..4g+3 of column i = its eight k values.  Thread -> row pair tid/32, float4 tid%32.
// ---------------------------------------------------------------------------------------------------------------
typedef __bf16 wg_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 wg_bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int wg_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int wg_u32x2 __attribute__((ext_vector_type(2)));
#define WGB_S 9

__device__ __forceinline__ unsigned wg_bf2(float x0, float x1) {
    const wg_bf16x2 v = {(__bf16)x0, (__bf16)x1};
    return __builtin_bit_cast(unsigned, v);
}

// HALF (compile time, so that the operand loads of a stage stay one straight-line burst): bit 0 = dz, bit 1 = x stored as bf16
template <int TN, int TC, int HALF, bool VEC>
__global__ __launch_bounds__(WG_THREADS) void wgrad_bf_kernel(const WgBatch bt) {
    constexpr int MI = TN / 64, NJ = TC / 64;
    __shared__ unsigned a_t[2][TN * WGB_S];
    __shared__ unsigned b_t[2][TC * WGB_S];
    __shared__ float bias_red[8][TN];
    const eab_wgrad_desc& d = bt.d;
    const int member = bt.n > 1 ? (int)(blockIdx.x / (unsigned)bt.groups) : 0;
    const int row_group = (int)blockIdx.x - member * bt.groups;
    const float* const m_dz = bt.dz[member];
    const float* const m_src0 = bt.src0[member];
    const float* const m_src1 = bt.src1[member];
    float* const m_dw = bt.dw[member];
    float* const m_dbias = bt.dbias[member];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, li = lane & 31, lh = lane >> 5;

    const long long R = (long long)d.T * d.No * d.B;
    const int adv_a = WG_ROWS / d.No, adv_r = WG_ROWS - adv_a * d.No;         // a stage advances every row walker by WG_ROWS rows
    const long long r_begin = (long long)row_group * d.rows_per_wg;
    const long long r_end = r_begin + d.rows_per_wg < R ? r_begin + d.rows_per_wg : R;
    const int col0 = blockIdx.y * TC, n0 = blockIdx.z * TN;
    const int Ctot = d.C0 + d.C1, UPT = (Ctot + 15) >> 4, NU = d.ntaps * UPT;

    const int rp = tid >> 5, c4 = tid & 31;                        // row pair of the stage, float4 column
    const bool a_live = c4 * 4 < TN, b_live = c4 * 4 < TC;
    // gathered-x column block of this thread: unit u = one tap x 16 channels
    const int u = (col0 >> 4) + (c4 >> 2);
    const bool u_ok = b_live && u < NU;
    const int tap = u_ok ? u / UPT : 0;
    const int cu = (u_ok ? u - tap * UPT : 0) << 4;
    const bool second = d.C1 > 0 && cu >= d.C0;
    const int b_Cs = second ? d.C1 : d.C0;
    const int b_cc = (second ? cu - d.C0 : cu) + (c4 & 3) * 4;
    const bool b_ok = u_ok && b_cc < b_Cs;
    const float* const b_src = second ? m_src1 : m_src0;
    int tdt = 0, tio = 0;
#pragma unroll
    for (int k = 0; k < EAB_MAX_TAPS; ++k)
        if (k == tap) {
            tdt = d.dt[k];
            tio = d.ioff[k];
        }
    constexpr bool vec_ok = VEC;
    // operands STORED as bf16 (eab_wgrad_desc.bf16_mask: the bf16 training programs' convolution-output gradients and
    // normalised activations): four bf16 of a row per load, and the LDS image of a row pair is a byte permute of the two loads
    // instead of four conversions -- the same operand bits as rounding the fp32 tensor here, half the bytes
    // (the two sources of a concatenation are either both stored as bf16 or both fp32 -- host check -- so the flag is uniform)
    constexpr bool a_half = (HALF & 1) != 0, b_half = (HALF & 2) != 0;
    const int eszA = a_half ? 2 : 4, eszB = b_half ? 2 : 4;
    // (buffer descriptors and incremental 32-bit offsets as in wgrad_kernel; two row walkers per thread: rows 2 rp, 2 rp + 1)
    const __amdgpu_buffer_rsrc_t rs_dz = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(m_dz), 0, (unsigned)((size_t)d.B * d.T * d.Fz * d.N * eszA), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_src = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(b_src), 0, (unsigned)((size_t)d.B * d.T * d.Fin * b_Cs * eszB), 0x00020000);
    constexpr unsigned WG_OOB = 0x80000000u;
    WgRow rw[2];
    unsigned offA[2];
    int offB[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        rw[e].init(r_begin + 2 * rp + e, d.T, d.No);
        const long long bt = (long long)rw[e].b * d.T + rw[e].t;
        offA[e] = (unsigned)((((bt * d.Fz) + (long long)rw[e].o * d.ostride + d.ophase) * d.N + n0 + c4 * 4) * eszA);
        offB[e] = (int)((((bt + tdt) * d.Fin + (long long)rw[e].o * d.istride + tio) * b_Cs + b_cc) * eszB);
    }
    const unsigned dA0 = (unsigned)((adv_a * d.Fz + adv_r * d.ostride) * d.N * eszA), dA1 = (unsigned)((d.Fz - d.No * d.ostride) * d.N * eszA);
    const int dB0 = (adv_a * d.Fin + adv_r * d.istride) * b_Cs * eszB, dB1 = (d.Fin - d.No * d.istride) * b_Cs * eszB;

    f32x4 ra[2], rb[2];
    f32x4 bsum = {0.f, 0.f, 0.f, 0.f};
    const bool do_bias = m_dbias != nullptr && blockIdx.y == 0;
    auto fetch = [&](long long r0) {
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const bool rok = r0 + 2 * rp + e < r_end;
            if (a_half) {                                         // (workgroup-uniform)
                const wg_u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rs_dz, (rok && a_live) ? offA[e] : WG_OOB, 0, 0);
                ra[e] = __builtin_bit_cast(f32x4, wg_u32x4{v[0], v[1], 0u, 0u});
            } else {
                ra[e] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_dz, (rok && a_live) ? offA[e] : WG_OOB, 0, 0));
            }
            const int tt = rw[e].t + tdt, fi = rw[e].o * d.istride + tio;
            const bool ok = rok && b_ok && tt >= 0 && tt < d.T && fi >= 0 && fi < d.Fin;
            const unsigned off = ok ? (unsigned)offB[e] : WG_OOB;
            if (b_half) {                                         // (uniform per thread's column block; C % 16 == 0: host check)
                const wg_u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rs_src, off, 0, 0);
                rb[e] = __builtin_bit_cast(f32x4, wg_u32x4{v[0], v[1], 0u, 0u});
            } else if (vec_ok) {
                rb[e] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_src, off, 0, 0));
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    rb[e][q] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                                             rs_src, (ok && b_cc + q < b_Cs) ? off + 4u * q : WG_OOB, 0, 0));
            }
            const bool wrap = rw[e].step(adv_a, adv_r, d.T, d.No);
            offA[e] += dA0 + (wrap ? dA1 : 0u);
            offB[e] += dB0 + (wrap ? dB1 : 0);
        }
    };
    auto stash = [&](int buf) {
        // row pair -> one dword per column: (row 2rp, row 2rp+1) as (low, high) bf16
        auto pair4 = [](const f32x4 (&r)[2], bool half, unsigned (&o)[4]) {
            if (half) {                                           // r[e] = two dwords holding the row's four bf16
                const wg_u32x4 r0 = __builtin_bit_cast(wg_u32x4, r[0]), r1 = __builtin_bit_cast(wg_u32x4, r[1]);
                o[0] = __builtin_amdgcn_perm(r1[0], r0[0], 0x05040100u);
                o[1] = __builtin_amdgcn_perm(r1[0], r0[0], 0x07060302u);
                o[2] = __builtin_amdgcn_perm(r1[1], r0[1], 0x05040100u);
                o[3] = __builtin_amdgcn_perm(r1[1], r0[1], 0x07060302u);
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = wg_bf2(r[0][j], r[1][j]);
            }
        };
        if (a_live) {
            unsigned o[4];
            pair4(ra, a_half, o);
#pragma unroll
            for (int j = 0; j < 4; ++j) a_t[buf][(c4 * 4 + j) * WGB_S + rp] = o[j];
            if (do_bias) {
                if (a_half) {                                     // bias gradient from the stored (bf16) gradient
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        bsum[j] += __builtin_bit_cast(float, o[j] << 16) + __builtin_bit_cast(float, o[j] & 0xFFFF0000u);
                } else {
                    bsum += ra[0] + ra[1];
                }
            }
        }
        if (b_live) {
            unsigned o[4];
            pair4(rb, b_half, o);
#pragma unroll
            for (int j = 0; j < 4; ++j) b_t[buf][(c4 * 4 + j) * WGB_S + rp] = o[j];
        }
    };

    f32x16 acc[MI][NJ];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int nj = 0; nj < NJ; ++nj)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][nj][r] = 0.0f;

    auto lds_barrier = [] { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
    const int a_off = (wm * (TN / 2) + li) * WGB_S + 4 * lh, b_off = (wn * (TC / 2) + li) * WGB_S + 4 * lh;
    auto frag = [](const unsigned* p) {
        const wg_u32x4 v = {p[0], p[1], p[2], p[3]};
        return __builtin_bit_cast(wg_bf16x8, v);
    };

    if (r_begin < r_end) {
        fetch(r_begin);
        stash(0);
        lds_barrier();
        int cur = 0;
        for (long long r0 = r_begin; r0 < r_end; r0 += WG_ROWS) {
            const bool more = r0 + WG_ROWS < r_end;
            if (more) fetch(r0 + WG_ROWS);
            wg_bf16x8 av[MI], bv[NJ];
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) av[mi] = frag(&a_t[cur][a_off + mi * 32 * WGB_S]);
#pragma unroll
            for (int nj = 0; nj < NJ; ++nj) bv[nj] = frag(&b_t[cur][b_off + nj * 32 * WGB_S]);
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int nj = 0; nj < NJ; ++nj)
                    acc[mi][nj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[mi], bv[nj], acc[mi][nj], 0, 0, 0);
            if (more) stash(cur ^ 1);
            lds_barrier();
            cur ^= 1;
        }
    }

#pragma unroll
    for (int nj = 0; nj < NJ; ++nj) {
        const int col = col0 + wn * (TC / 2) + nj * 32 + li;
        if (col < d.Kpad) {
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int n = n0 + wm * (TN / 2) + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    atomicAdd(&m_dw[(size_t)n * d.Kpad + col], acc[mi][nj][r]);
                }
        }
    }
    if (m_dbias != nullptr && blockIdx.y == 0) {     // (workgroup-uniform): the 8 row-pair lanes of a column meet in LDS
        __syncthreads();
        if (a_live) *reinterpret_cast<f32x4*>(&bias_red[rp][c4 * 4]) = bsum;
        __syncthreads();
        if (tid < TN) {
            float t = 0.0f;
#pragma unroll
            for (int r = 0; r < 8; ++r) t += bias_red[r][tid];
            atomicAdd(&m_dbias[n0 + tid], t);
        }
    }
}

static bool wg_same_geometry(const eab_wgrad_desc* a, const eab_wgrad_desc* b) {
    if (a->N != b->N || a->C0 != b->C0 || a->C1 != b->C1 || a->Kpad != b->Kpad || a->B != b->B || a->T != b->T || a->Fin != b->Fin ||
        a->Fz != b->Fz || a->No != b->No || a->ostride != b->ostride || a->ophase != b->ophase || a->istride != b->istride ||
        a->ntaps != b->ntaps || a->precision != b->precision || a->bf16_mask != b->bf16_mask ||
        (a->src1 == nullptr) != (b->src1 == nullptr) ||
        (a->dbias == nullptr) != (b->dbias == nullptr))
        return false;
    for (int k = 0; k < a->ntaps; ++k)
        if (a->dt[k] != b->dt[k] || a->ioff[k] != b->ioff[k]) return false;
    return true;
}

// How many of descs[0..n) (at most WG_MAXB) can share one launch with descs[0]
extern "C" int eab_wgrad_batchable(const eab_wgrad_desc* descs, int n, int stride_bytes) {
    if (!descs || n <= 0) return 0;
    int k = 1;
    while (k < n && k < WG_MAXB &&
           wg_same_geometry(descs, reinterpret_cast<const eab_wgrad_desc*>(reinterpret_cast<const char*>(descs) + (size_t)k * stride_bytes)))
        ++k;
    return k;
}

extern "C" int eab_wgrad_batch_f32(const eab_wgrad_desc* descs, int n, int stride_bytes, eab_stream_t stream) {
    EAB_CHECK_ARG(descs && n >= 1 && n <= WG_MAXB && stride_bytes >= (int)sizeof(eab_wgrad_desc));
    const eab_wgrad_desc* d = descs;
    auto at = [&](int k) { return reinterpret_cast<const eab_wgrad_desc*>(reinterpret_cast<const char*>(descs) + (size_t)k * stride_bytes); };
    for (int k = 0; k < n; ++k) {
        EAB_CHECK_ARG(at(k)->dz && at(k)->src0 && at(k)->dw);
        EAB_CHECK_ARG(k == 0 || wg_same_geometry(d, at(k)));
    }
    EAB_CHECK_ARG(d->B > 0 && d->T > 0 && d->Fin > 0 && d->Fz > 0 && d->No > 0 && d->N > 0 && (d->N % 64) == 0);
    EAB_CHECK_ARG(d->precision == EAB_PREC_F32 || d->precision == EAB_PREC_BF16);
    if (d->bf16_mask) {        // operands stored as bf16: bf16 products, whole 16-channel units, flagged sources must exist
        EAB_CHECK_ARG((d->bf16_mask & ~7) == 0 && d->precision == EAB_PREC_BF16 && (d->N % 4) == 0);
        EAB_CHECK_ARG(!(d->bf16_mask & 2) || (d->C0 % 16) == 0);
        EAB_CHECK_ARG(!(d->bf16_mask & 4) || (d->src1 && (d->C1 % 16) == 0 && (d->C0 % 16) == 0));
        EAB_CHECK_ARG(d->src1 == nullptr || ((d->bf16_mask >> 1) & 1) == ((d->bf16_mask >> 2) & 1));   // both sources alike
    }
    // 32-bit byte offsets inside the kernels
    EAB_CHECK_ARG((unsigned long long)d->B * d->T * d->Fz * d->N * 4 < (1ull << 31));
    EAB_CHECK_ARG((unsigned long long)d->B * d->T * d->Fin * (d->C0 > d->C1 ? d->C0 : d->C1) * 4 < (1ull << 31));
    EAB_CHECK_ARG(d->C0 > 0 && d->C1 >= 0 && (d->C1 == 0) == (d->src1 == nullptr));
    EAB_CHECK_ARG((d->C1 % 4) == 0 && (d->C1 == 0 || (d->C0 % 16) == 0));      // a single source may have any channel count
    EAB_CHECK_ARG(d->ntaps > 0 && d->ntaps <= EAB_MAX_TAPS && d->ostride >= 1 && d->istride >= 1);
    EAB_CHECK_ARG(d->ophase >= 0 && d->ophase < d->ostride && (d->No - 1) * d->ostride + d->ophase < d->Fz);
    const int upt = (d->C0 + d->C1 + 15) / 16;
    EAB_CHECK_ARG(d->Kpad == d->ntaps * upt * 16);
    const long long R = (long long)d->B * d->T * d->No;
    const int tn = (d->N % 128 == 0) ? 128 : 64;
    const int tc = (d->Kpad % 128 == 0) ? 128 : 64;
    const int cb = (d->Kpad + tc - 1) / tc, nb = d->N / tn;
    // Row groups G (split-K).  Two costs pull in opposite directions, both measured on MI355X:
    //   * every group adds its whole tile to dW with device-scope atomics, and those retire at ~1.2e11 per second chip-wide
    //     (en.3.in_conv: 338 groups x 128 x 384 atomics = 127 us, whatever the matrix work) -> t_atomic = G * cb*nb*tn*tc / 1.2e11;
    //   * a group is a serial chain of 16-row stages (global load -> LDS -> barrier -> MFMA, ~0.5-1 us each with
    //     neighbours hiding the latency) -> t_chain = R / (16 G) * t_stage.
    // G* = sqrt(R * t_stage / 16 / (atomics per group / 1.2e11)) minimises the sum; never more than ~four workgroups per CU
    // over the grid, never fewer than fill the chip once (when R allows it at >= 64 rows per group).
    const double t_stage = (tn == 128 && tc == 128) ? 1.0e-6 : (tn == 128 || tc == 128) ? 0.7e-6 : 0.5e-6;
    const double per_group = (double)cb * nb * tn * tc / 1.2e11;
    long long groups = (long long)(sqrt((double)R * t_stage / 16.0 / per_group) + 0.5);
    const long long per_rg = (long long)cb * nb * n;              // workgroups per row group over the whole launch
    const long long g_max = (1024 + per_rg - 1) / per_rg;
    const long long g_min = (256 + per_rg - 1) / per_rg;
    if (groups > g_max) groups = g_max;
    if (groups < g_min) groups = g_min;
    long long rpw = (R + groups - 1) / groups;
    if (rpw < 64) rpw = 64;
    rpw = (rpw + WG_ROWS - 1) / WG_ROWS * WG_ROWS;
    WgBatch bt;
    bt.d = *d;
    bt.d.rows_per_wg = (int)rpw;
    bt.n = n;
    bt.groups = (int)((R + rpw - 1) / rpw);
    for (int k = 0; k < WG_MAXB; ++k) {
        const eab_wgrad_desc* m = at(k < n ? k : 0);
        bt.dz[k] = m->dz;
        bt.src0[k] = m->src0;
        bt.src1[k] = m->src1;
        bt.dw[k] = m->dw;
        bt.dbias[k] = m->dbias;
    }
    EAB_CHECK_ARG((long long)bt.groups * n < (1ll << 31));
    dim3 grid((unsigned)(bt.groups * n), (unsigned)cb, (unsigned)nb);
    hipStream_t s = eab_stream(stream);
    const bool vec = (d->C0 & 3) == 0;               // 16-byte gathers (compile-time property of the kernel instance)
    if (d->precision == EAB_PREC_BF16) {
        const int half = d->bf16_mask & 3;       // (both sources of a concatenation alike: checked above)
#define WG_BF_LAUNCH(TN_, TC_)                                                                                              \
    do {                                                                                                                    \
        if (!vec) {       /* odd channel counts: the network input only, never stored as bf16 (dz may be) */               \
            if (half & 2) return EAB_EUNSUPPORTED;                                                                          \
            if (half == 0) hipLaunchKernelGGL((wgrad_bf_kernel<TN_, TC_, 0, false>), grid, dim3(WG_THREADS), 0, s, bt);    \
            else hipLaunchKernelGGL((wgrad_bf_kernel<TN_, TC_, 1, false>), grid, dim3(WG_THREADS), 0, s, bt);              \
        } else if (half == 0) hipLaunchKernelGGL((wgrad_bf_kernel<TN_, TC_, 0, true>), grid, dim3(WG_THREADS), 0, s, bt);  \
        else if (half == 1) hipLaunchKernelGGL((wgrad_bf_kernel<TN_, TC_, 1, true>), grid, dim3(WG_THREADS), 0, s, bt);    \
        else if (half == 2) hipLaunchKernelGGL((wgrad_bf_kernel<TN_, TC_, 2, true>), grid, dim3(WG_THREADS), 0, s, bt);    \
        else hipLaunchKernelGGL((wgrad_bf_kernel<TN_, TC_, 3, true>), grid, dim3(WG_THREADS), 0, s, bt);                   \
    } while (0)
        if (tn == 128 && tc == 128) WG_BF_LAUNCH(128, 128);
        else if (tn == 128) WG_BF_LAUNCH(128, 64);
        else if (tc == 128) WG_BF_LAUNCH(64, 128);
        else WG_BF_LAUNCH(64, 64);
#undef WG_BF_LAUNCH
        EAB_RETURN_LAUNCH_STATUS();
    }
#define WG_F32_LAUNCH(TN_, TC_)                                                                                             \
    do {                                                                                                                    \
        if (vec) hipLaunchKernelGGL((wgrad_kernel<TN_, TC_, true>), grid, dim3(WG_THREADS), 0, s, bt);                     \
        else hipLaunchKernelGGL((wgrad_kernel<TN_, TC_, false>), grid, dim3(WG_THREADS), 0, s, bt);                        \
    } while (0)
    if (tn == 128 && tc == 128) WG_F32_LAUNCH(128, 128);
    else if (tn == 128) WG_F32_LAUNCH(128, 64);
    else if (tc == 128) WG_F32_LAUNCH(64, 128);
    else WG_F32_LAUNCH(64, 64);
#undef WG_F32_LAUNCH
    EAB_RETURN_LAUNCH_STATUS();
}

extern "C" int eab_wgrad_f32(const eab_wgrad_desc* d, eab_stream_t stream) {
    return eab_wgrad_batch_f32(d, 1, (int)sizeof(eab_wgrad_desc), stream);
}
