// Program runner + ABI housekeeping.  The op list is the whole EaBNet.forward (reference
// EaBNet.py:88-117) or GaGNet.forward (GaGNet.py:76-90) lowered by eabnet_amd/program.py.
#include "common.h"

// Zero fill as a KERNEL: a hipMemsetAsync captured into a hipGraph was observed (ROCm 7.2, 6.5 MB) not
// to be ordered against the neighbouring kernel nodes when the graph is replayed on an idle stream --
// the S-TCN running sum then accumulated across replays.  A kernel node has no such ambiguity.
__global__ __launch_bounds__(256) void zero_fill_kernel(float* __restrict__ p, size_t n4, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t k = i; k < n4; k += stride) reinterpret_cast<f32x4*>(p)[k] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (size_t k = 4 * n4 + i; k < n; k += stride) p[k] = 0.0f;
}

// rows [*t_pos, *t_pos + count) of a [B][T][row] tensor (row % 4 == 0): the streaming form of the fill
__global__ __launch_bounds__(256) void zero_rows_kernel(float* __restrict__ p, int T, int row4, const int* __restrict__ t_pos,
                                                        int count, long long total4) {
    const int t_lo = *t_pos;
    const long long per_b = (long long)count * row4;
    for (long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x; j < total4; j += (long long)gridDim.x * blockDim.x) {
        const long long b = j / per_b, rem = j - b * per_b;
        if (t_lo + rem / row4 >= T) continue;
        reinterpret_cast<f32x4*>(p)[(b * T + t_lo) * row4 + rem] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
}

extern "C" int eab_zero_rows_f32(float* ptr, int B, int T, int row_floats, eab_time_window win, eab_stream_t stream) {
    EAB_CHECK_ARG(ptr && B > 0 && T > 0 && row_floats > 0 && (row_floats % 4) == 0 && ((uintptr_t)ptr & 15) == 0);
    EAB_CHECK_ARG(win.pos == nullptr || win.count > 0);
    if (!win.pos) {
        const size_t n = (size_t)B * T * row_floats, n4 = n / 4;
        size_t g = (n4 + 255) / 256;
        if (g > 2048) g = 2048;
        hipLaunchKernelGGL(zero_fill_kernel, dim3((unsigned)g), dim3(256), 0, eab_stream(stream), ptr, n4, n);
        EAB_RETURN_LAUNCH_STATUS();
    }
    const long long total4 = (long long)B * win.count * (row_floats / 4);
    long long g = (total4 + 255) / 256;
    if (g > 2048) g = 2048;
    hipLaunchKernelGGL(zero_rows_kernel, dim3((unsigned)g), dim3(256), 0, eab_stream(stream), ptr, T, row_floats / 4, win.pos,
                       win.count, total4);
    EAB_RETURN_LAUNCH_STATUS();
}

extern "C" int eab_abi_version(void) { return EAB_ABI_VERSION; }

extern "C" const char* eab_error_string(int code) {
    if (code == EAB_OK) return "ok";
    if (code == EAB_EINVAL) return "invalid argument (shape, limit or null pointer) -- nothing was launched";
    if (code == EAB_EUNSUPPORTED) return "configuration not built into libeabnet_hip";
    if (code >= EAB_EHIP_BASE) return hipGetErrorString((hipError_t)(code - EAB_EHIP_BASE));
    return "unknown eabnet_hip error";
}

extern "C" int eab_run_program(const eab_op* ops, int n_ops, eab_stream_t stream) {
    EAB_CHECK_ARG(ops && n_ops >= 0);
    for (int k = 0; k < n_ops; ++k) {
        const eab_op& o = ops[k];
        int rc;
        switch (o.kind) {
            case EAB_OP_CONV:
                rc = eab_conv_f32(&o.conv, stream);
                break;
            case EAB_OP_CONV_CHAIN:
                rc = eab_conv_st_chain_run((const eab_conv_desc*)o.p[0], (const int*)o.p[1], o.i[0], o.i[1], o.i[2], o.i[3], stream);
                break;
            case EAB_OP_IN_FINALIZE:
                rc = eab_in_finalize_mr_f32((const float*)o.p[0], o.i[0], o.i[1], o.i[2], o.i[3], o.i[4], o.f[0],
                                            (const float*)o.p[1], (const float*)o.p[2], (float*)o.p[3],
                                            (const float*)o.p[4], (const float*)o.p[5], (float*)o.p[6], (float*)o.p[7],
                                            (float*)o.p[8], stream);
                break;
            case EAB_OP_NORM_ACT:
                if (o.win.pos)
                    rc = o.i[3] > 0 && o.i[1] % o.i[3] == 0
                             ? eab_norm_act_win_f32((const float*)o.p[0], (const float*)o.p[1], (const float*)o.p[2],
                                                    (const float*)o.p[3], (const float*)o.p[4], (const float*)o.p[5],
                                                    (float*)o.p[6], o.i[0], o.i[3], o.i[1] / o.i[3], o.i[2], o.win, stream)
                             : EAB_EINVAL;
                else
                    rc = eab_norm_act_f32((const float*)o.p[0], (const float*)o.p[1], (const float*)o.p[2],
                                          (const float*)o.p[3], (const float*)o.p[4], (const float*)o.p[5],
                                          (float*)o.p[6], o.i[0], o.i[1], o.i[2], stream);
                break;
            case EAB_OP_LSTM64:
                rc = eab_lstm64_stream_f32((const float*)o.p[0], (const float*)o.p[1], (const float*)o.p[2], o.f[0],
                                           (const float*)o.p[3], (const float*)o.p[4], (float*)o.p[5], (float*)o.p[6],
                                           o.i[0], o.i[1], o.i[2], o.i[3], o.win, stream);
                break;
            case EAB_OP_BFW_FS:
                rc = eab_mlp_bfw_filter_sum_f32((const float*)o.p[0], (const float*)o.p[6], (const float*)o.p[7],
                                                (const float*)o.p[1], (const float*)o.p[2], (const float*)o.p[3],
                                                (float*)o.p[4], (float*)o.p[5], o.i[0], o.i[1], o.i[2], o.i[3], o.win,
                                                stream);
                break;
            case EAB_OP_MEMSET0: {
                if (o.win.pos) {
                    rc = eab_zero_rows_f32((float*)const_cast<void*>(o.p[0]), o.i[2], o.i[3], o.i[4], o.win, stream);
                    break;
                }
                const size_t bytes = ((size_t)(uint32_t)o.i[1] << 32) | (uint32_t)o.i[0];
                if (!o.p[0] || (bytes & 3) || ((uintptr_t)o.p[0] & 15)) {
                    rc = EAB_EINVAL;
                    break;
                }
                const size_t n = bytes / 4, n4 = n / 4;
                size_t g = (n4 + 255) / 256;
                if (g > 2048) g = 2048;
                if (g == 0) g = 1;
                hipLaunchKernelGGL(zero_fill_kernel, dim3((unsigned)g), dim3(256), 0, eab_stream(stream),
                                   (float*)const_cast<void*>(o.p[0]), n4, n);
                rc = eab_hip_status(hipGetLastError());
                break;
            }
            case EAB_OP_GAG_PACK:
                rc = eab_gag_pack_f32((const float*)o.p[0], (const float*)o.p[1], (float*)o.p[2], (float*)o.p[3], o.i[0],
                                      o.i[1], o.i[2], o.i[3], o.win, stream);
                break;
            case EAB_OP_GAG_CRM:
                rc = eab_gag_crm_f32((const float*)o.p[0], (const float*)o.p[1], (const float*)o.p[2],
                                     (const float*)o.p[3], (float*)o.p[4], (float*)o.p[5], o.i[0], o.i[1], o.i[2], o.i[3],
                                     o.i[4], o.i[5], o.win, stream);
                break;
#define EAB_P(k) ((const float*)o.p[k])
#define EAB_W(k) ((float*)const_cast<void*>(o.p[k]))
#define EAB_N64(k) ((long long)(((unsigned long long)(uint32_t)o.i[(k) + 1] << 32) | (uint32_t)o.i[k]))
            case EAB_OP_GATHER:
                rc = eab_gather_f32(EAB_P(0), (const int32_t*)o.p[1], (const int32_t*)o.p[2], EAB_W(3), EAB_N64(0), stream);
                break;
            case EAB_OP_IN_STATS:
                rc = o.p[6] && o.i[3] > 0
                         ? eab_train_in1d_multi_f32(EAB_P(0), EAB_P(1), o.i[0], o.i[1], o.i[2], o.i[3], o.f[0], EAB_P(2), EAB_P(3),
                                                    EAB_W(4), EAB_W(5), EAB_W(6), stream)
                     : o.p[6] ? eab_train_in1d_f32(EAB_P(0), EAB_P(1), o.i[0], o.i[1], o.i[2], o.f[0], EAB_P(2), EAB_P(3), EAB_W(4),
                                                 EAB_W(5), EAB_W(6), stream)
                            : eab_train_in_stats_f32(EAB_P(0), EAB_P(1), o.i[0], o.i[1], o.i[2], o.f[0], EAB_P(2), EAB_P(3),
                                                     EAB_W(4), EAB_W(5), stream);
                break;
            case EAB_OP_TR_NORM_ACT:
                rc = eab_train_norm_act_f32(EAB_P(0), EAB_P(1), EAB_P(2), EAB_P(3), EAB_W(4), o.i[0], o.i[1], o.i[2], o.i[3],
                                            stream);
                break;
            case EAB_OP_NORM_BWD:
                rc = o.i[4] > 0 ? eab_train_norm_bwd_multi_f32(EAB_P(0), EAB_P(4), EAB_P(1), EAB_P(2), EAB_P(3), EAB_P(5), EAB_W(6),
                                                               EAB_P(7), EAB_W(8), EAB_W(9), EAB_W(10), EAB_W(11), o.i[0], o.i[1],
                                                               o.i[4], stream)
                                : eab_train_norm_bwd_f32(EAB_P(0), EAB_P(1), EAB_P(2), EAB_P(3), EAB_P(4), EAB_P(5), EAB_W(6), EAB_P(7),
                                                         EAB_W(8), EAB_W(9), EAB_W(10), EAB_W(11), o.i[0], o.i[1], o.i[2], o.i[3],
                                                         stream);
                break;
            case EAB_OP_GLU_BWD:
                rc = eab_glu_bwd_ex_f32(EAB_P(0), EAB_P(1), EAB_W(2), EAB_N64(0), o.i[2], o.i[3], stream);
                break;
            case EAB_OP_GATE_FWD:
                rc = eab_gate_fwd_f32(EAB_P(0), EAB_P(1), EAB_W(2), EAB_N64(0), stream);
                break;
            case EAB_OP_GATE_BWD:
                rc = eab_gate_bwd_f32(EAB_P(0), EAB_P(1), EAB_P(2), EAB_W(3), EAB_W(4), EAB_N64(0), stream);
                break;
            case EAB_OP_ADD:
                rc = eab_add_f32(EAB_P(0), EAB_P(1), EAB_W(2), EAB_N64(0), stream);
                break;
            case EAB_OP_RELU_BWD:
                rc = eab_relu_bwd_f32(EAB_P(0), EAB_P(1), EAB_W(2), EAB_N64(0), stream);
                break;
            case EAB_OP_COLSUM:
                rc = eab_colsum_f32(EAB_P(0), EAB_W(1), EAB_N64(0), o.i[2], stream);
                break;
            case EAB_OP_FILTER_SUM:
                rc = eab_filter_sum_ld_f32(EAB_P(0), EAB_P(1), EAB_W(2), o.i[0], o.i[1], o.i[2], o.i[3], o.i[4], stream);
                break;
            case EAB_OP_FS_BWD:
                rc = eab_filter_sum_bwd_f32(EAB_P(0), EAB_P(1), EAB_W(2), o.i[0], o.i[1], o.i[2], o.i[3], o.i[4], stream);
                break;
            case EAB_OP_LN_FWD:
                rc = eab_layernorm64_fwd_f32(EAB_P(0), EAB_P(1), EAB_P(2), o.f[0], EAB_W(3), EAB_W(4), EAB_N64(0), stream);
                break;
            case EAB_OP_LN_BWD:
                rc = eab_layernorm64_bwd_f32(EAB_P(0), EAB_P(1), EAB_P(2), EAB_P(3), EAB_W(4), EAB_W(5), EAB_W(6), EAB_N64(0),
                                             stream);
                break;
            case EAB_OP_LSTM_TRAIN:
                rc = eab_lstm64_train_fwd_prec_f32(EAB_P(0), EAB_P(1), EAB_P(2), EAB_W(3), EAB_W(4), o.i[0], o.i[1], o.i[2], o.i[3], stream);
                break;
            case EAB_OP_LSTM_BWD:
                rc = eab_lstm64_bwd_prec_f32(EAB_P(0), EAB_P(1), EAB_P(2), EAB_W(3), o.i[0], o.i[1], o.i[2], o.i[3], stream);
                break;
            case EAB_OP_GAG_CRM_BWD:
                rc = eab_gag_crm_bwd_f32(EAB_P(0), EAB_P(1), EAB_P(2), EAB_P(3), EAB_P(4), EAB_W(5), EAB_W(6), EAB_W(7), EAB_W(8), o.i[0],
                                         o.i[1], o.i[2], o.i[3], o.i[4], o.i[5], stream);
                break;
            case EAB_OP_WGRAD: {
                // consecutive weight gradients of identical geometry share one launch (the lowering sorts them so)
                int run = 1;
                while (k + run < n_ops && ops[k + run].kind == EAB_OP_WGRAD) ++run;
                const int nb = eab_wgrad_batchable(&o.wgrad, run, (int)sizeof(eab_op));
                rc = eab_wgrad_batch_f32(&o.wgrad, nb, (int)sizeof(eab_op), stream);
                k += nb - 1;
                break;
            }
            case EAB_OP_CLN_STATS:
                rc = eab_cln_stats_f32(EAB_P(0), EAB_P(1), o.i[0], o.i[1], o.i[2], o.i[3], o.f[0], (double*)const_cast<void*>(o.p[2]),
                                       (double*)const_cast<void*>(o.p[3]), EAB_W(4), o.win, stream);
                break;
            case EAB_OP_CLN_APPLY:
                rc = eab_cln_apply_f32(EAB_P(0), EAB_P(1), EAB_P(2), EAB_P(3), EAB_P(4), EAB_P(5), EAB_W(6), o.i[0], o.i[1], o.i[2],
                                       o.i[3], o.i[4], o.win, stream);
                break;
            case EAB_OP_CLN_STEP:
                rc = eab_cln_step_f32(EAB_P(0), EAB_P(1), (double*)const_cast<void*>(o.p[2]), (double*)const_cast<void*>(o.p[3]),
                                      EAB_W(4), EAB_P(5), EAB_P(6), EAB_P(7), EAB_P(8), EAB_W(9), o.i[0], o.i[1], o.i[2], o.i[3],
                                      o.i[4], o.f[0], o.win, stream);
                break;
            case EAB_OP_CLN_BWD:
                rc = eab_train_cln_bwd_f32(EAB_P(0), EAB_P(1), EAB_P(2), EAB_P(3), EAB_P(4), EAB_P(5), (double*)const_cast<void*>(o.p[6]),
                                           EAB_W(7), EAB_W(8), EAB_P(9), EAB_W(10), o.i[0], o.i[1], o.i[2], o.i[3], o.i[4], stream);
                break;
            case EAB_OP_GATE_ROWS:
                rc = eab_gate_rows_f32(EAB_P(0), EAB_P(1), EAB_W(2), o.i[0], o.i[1], o.i[2], o.win, stream);
                break;
#undef EAB_P
#undef EAB_W
#undef EAB_N64
            default:
                rc = EAB_EINVAL;
        }
        if (rc != EAB_OK) return rc;
    }
    return EAB_OK;
}

// struct-layout handshake for the ctypes mirror (eabnet_amd/_lib.py)
extern "C" int eab_sizeof_conv_desc(void) { return (int)sizeof(eab_conv_desc); }
extern "C" int eab_sizeof_op(void) { return (int)sizeof(eab_op); }
extern "C" int eab_sizeof_wgrad_desc(void) { return (int)sizeof(eab_wgrad_desc); }
