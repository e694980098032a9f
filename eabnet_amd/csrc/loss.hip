// Training losses fused (first piece): com_mag_mse_loss (reference EaBNet.py:627-640) and,
// stage by stage, stagewise_com_mag_mse_loss (GaGNet.py:601-619) -- forward value and the gradient w.r.t.
// the estimate in ONE pass over the two spectra:
//   mask[b,t,f] = t < frames[b];   n = sum(mask)
//   loss = 0.5 * ( sum mask (|e| - |l|)^2 / n  +  sum mask ((e_r-l_r)^2 + (e_i-l_i)^2) / (2n) )
//   d loss / d e_c = mask * ( (|e| - |l|) e_c / |e|  +  (e_c - l_c) / 2 ) / n          (0 where |e| = 0)
// The reference builds the masks on the CPU with pad_sequence, moves them to the device and makes ~10
// full-tensor passes; here both sums come out of one read of esti and label.  Deterministic: per-block
// partial sums (fixed tree in LDS), then one block adds the partials in index order.
// Layout: esti, label, grad [B][2][T][F] fp32.  Bound: HBM (2 reads [+1 write] of B*2*T*F floats).
#include "common.h"

#define LOSS_THREADS 256
#define LOSS_MAXB 64

struct LossFrames {
    int n[LOSS_MAXB];
};

__global__ __launch_bounds__(LOSS_THREADS) void loss_partial_kernel(const float* __restrict__ esti, const float* __restrict__ label,
                                                                    const LossFrames frames, int T, int F, long long bins,
                                                                    float inv_n, float* __restrict__ partial,
                                                                    float* __restrict__ grad) {
    __shared__ float red[2][LOSS_THREADS];
    const long long plane = (long long)T * F;
    float s_mag = 0.0f, s_com = 0.0f;
    for (long long i = (long long)blockIdx.x * LOSS_THREADS + threadIdx.x; i < bins; i += (long long)gridDim.x * LOSS_THREADS) {
        const long long b = i / plane, rem = i - b * plane;
        const int t = (int)(rem / F);
        const long long o = b * 2 * plane + rem;
        const bool live = t < frames.n[b];
        float gr = 0.0f, gi = 0.0f;
        if (live) {
            const float er = esti[o], ei = esti[o + plane], lr = label[o], li = label[o + plane];
            const float me = sqrtf(er * er + ei * ei), ml = sqrtf(lr * lr + li * li);
            const float dm = me - ml, dr = er - lr, di = ei - li;
            s_mag = fmaf(dm, dm, s_mag);
            s_com += dr * dr + di * di;
            if (grad) {
                const float k = me > 0.0f ? dm / me : 0.0f;
                gr = (k * er + 0.5f * dr) * inv_n;
                gi = (k * ei + 0.5f * di) * inv_n;
            }
        }
        if (grad) {
            grad[o] = gr;
            grad[o + plane] = gi;
        }
    }
    red[0][threadIdx.x] = s_mag;
    red[1][threadIdx.x] = s_com;
    __syncthreads();
    for (int s = LOSS_THREADS / 2; s > 0; s >>= 1) {
        if (threadIdx.x < s) {
            red[0][threadIdx.x] += red[0][threadIdx.x + s];
            red[1][threadIdx.x] += red[1][threadIdx.x + s];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        partial[2 * blockIdx.x] = red[0][0];
        partial[2 * blockIdx.x + 1] = red[1][0];
    }
}

__global__ void loss_final_kernel(const float* __restrict__ partial, int nblocks, float inv_n, float* __restrict__ out) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        double a = 0.0, c = 0.0;
        for (int k = 0; k < nblocks; ++k) {
            a += (double)partial[2 * k];
            c += (double)partial[2 * k + 1];
        }
        out[0] = (float)(0.5 * (a + 0.5 * c) * (double)inv_n);
    }
}

extern "C" int eab_com_mag_mse_loss_f32(const float* esti, const float* label, const int32_t* frames, int B, int T, int F,
                                        float* partial, int partial_blocks, float* loss, float* grad,
                                        eab_stream_t stream) {
    EAB_CHECK_ARG(esti && label && frames && partial && loss && B > 0 && B <= LOSS_MAXB && T > 0 && F > 0);
    EAB_CHECK_ARG(partial_blocks >= 1 && partial_blocks <= 4096);
    LossFrames fr;
    long long n = 0;
    for (int b = 0; b < LOSS_MAXB; ++b) {
        fr.n[b] = b < B ? frames[b] : 0;                 // HOST array: the frame counts come from the data loader
        if (b < B) {
            EAB_CHECK_ARG(frames[b] >= 0 && frames[b] <= T);
            n += (long long)frames[b] * F;
        }
    }
    EAB_CHECK_ARG(n > 0);
    const long long bins = (long long)B * T * F;
    long long g = (bins + LOSS_THREADS - 1) / LOSS_THREADS;
    if (g > partial_blocks) g = partial_blocks;
    const float inv_n = (float)(1.0 / (double)n);
    hipLaunchKernelGGL(loss_partial_kernel, dim3((unsigned)g), dim3(LOSS_THREADS), 0, eab_stream(stream), esti, label, fr, T, F,
                       bins, inv_n, partial, grad);
    hipLaunchKernelGGL(loss_final_kernel, dim3(1), dim3(64), 0, eab_stream(stream), partial, (int)g, inv_n, loss);
    EAB_RETURN_LAUNCH_STATUS();
}
