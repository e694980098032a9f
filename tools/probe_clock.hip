// Probe: what the fp32 matrix pipe of one MI355X delivers under a sustained load, and at which clock
// (MI355X_MICROARCH.md, DVFS give-back item 6: in-kernel clock = d s_memtime / d s_memrealtime x 100 MHz).
//   hipcc -O3 --offload-arch=gfx950 tools/probe_clock.hip -o tools/build/probe_clock && tools/build/probe_clock
// Loops (256 threads = one wave per SIMD, GRID workgroups per CU x 256 CUs, random operands):
//   bare32   v_mfma_f32_32x32x2_f32, 4 independent accumulators, operands in registers
//   bare16   v_mfma_f32_16x16x4_f32, 16 independent accumulators, operands in registers
//   lds32    the 128x128 conv tile's inner loop: per 16-deep K stage 4 ds_read_b128 (A) + 4 (B) per wave, 32 MFMAs, one barrier
//   lds32v   the same + NV VALU instructions per MFMA (the kernel's transform / address work)
//   lds16    the stage on v_mfma_f32_16x16x4_f32 (same wave tile 64x64 as 16 blocks of 16x16)
// Each loop runs back to back for >= 1.5 s before the measured launches.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

struct Stamp { unsigned long long c0, r0, c1, r1; };

__device__ __forceinline__ float frand(unsigned s) {
    s = s * 1664525u + 1013904223u; s ^= s >> 15; s *= 2246822519u; s ^= s >> 13;
    return (float)(s & 0xFFFF) / 32768.0f - 1.0f;
}

template <int SHAPE>   // 0: 32x32x2, 1: 16x16x4
__global__ __launch_bounds__(256) void bare(float* out, Stamp* st, int iters) {
    const int l = threadIdx.x, g = blockIdx.x;
    float a[8], b[8];
    for (int j = 0; j < 8; ++j) { a[j] = frand(g * 7919 + l * 31 + j); b[j] = frand(g * 104729 + l * 17 + j + 100); }
    float s = 0.f;
    unsigned long long c0 = 0, r0 = 0;
    if constexpr (SHAPE == 0) {
        f32x16 acc[4];
        for (int q = 0; q < 4; ++q) for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
        c0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime();
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(j + q) & 7], b[j], acc[q], 0, 0, 0);
            }
        }
        for (int q = 0; q < 4; ++q) s += acc[q][q];
    } else {
        f32x4 acc[16];
        for (int q = 0; q < 16; ++q) acc[q] = f32x4{0.f, 0.f, 0.f, 0.f};
        c0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime();
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
#pragma unroll
                for (int q = 0; q < 16; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[(j + q) & 7], b[(j + (q >> 2)) & 7], acc[q], 0, 0, 0);
            }
        }
        for (int q = 0; q < 16; ++q) s += acc[q][q & 3];
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[g * 256 + l] = s;
    if (l == 0) st[g] = Stamp{c0, r0, c1, r1};
}

// the 128x128x16 stage of conv_gemm_kernel<2,2,1,...>: LDS rows of 20 floats, wave tile 64x64
template <int SHAPE, int NV>
__global__ __launch_bounds__(256, 3) void ldsloop(float* out, Stamp* st, int iters) {
    constexpr int LDK = 20;
    __shared__ __attribute__((aligned(16))) float sa[2][128 * LDK], sb[2][128 * LDK];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1, g = blockIdx.x;
    for (int i = tid; i < 2 * 128 * LDK; i += 256) { (&sa[0][0])[i] = frand(g * 7919 + i); (&sb[0][0])[i] = frand(g * 104729 + i + 77); }
    __syncthreads();
    float s = 0.f;
    float v[8];
    for (int j = 0; j < 8; ++j) v[j] = frand(tid + j);
    unsigned long long c0, r0;
    if constexpr (SHAPE == 0) {
        const int li = lane & 31, lh = lane >> 5;
        const int a_base = (wm * 64 + li) * LDK + 4 * lh, b_base = (wn * 64 + li) * LDK + 4 * lh;
        f32x16 acc[2][2];
        for (int mi = 0; mi < 2; ++mi) for (int ni = 0; ni < 2; ++ni) for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
        c0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime();
        for (int i = 0; i < iters; ++i) {
            const int cur = i & 1;
#pragma unroll
            for (int gg = 0; gg < 2; ++gg) {
                f32x4 af[2], bf[2];
#pragma unroll
                for (int mi = 0; mi < 2; ++mi) af[mi] = *reinterpret_cast<const f32x4*>(&sa[cur][a_base + mi * 32 * LDK + gg * 8]);
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) bf[ni] = *reinterpret_cast<const f32x4*>(&sb[cur][b_base + ni * 32 * LDK + gg * 8]);
#pragma unroll
                for (int k = 0; k < 4; ++k)
#pragma unroll
                    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                        for (int ni = 0; ni < 2; ++ni) {
                            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[mi][k], bf[ni][k], acc[mi][ni], 0, 0, 0);
#pragma unroll
                            for (int q = 0; q < NV; ++q) v[q & 7] = fmaf(v[q & 7], 1.0001f, v[(q + 1) & 7]);
                        }
            }
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
        for (int mi = 0; mi < 2; ++mi) for (int ni = 0; ni < 2; ++ni) s += acc[mi][ni][mi * 2 + ni];
    } else {
        const int li = lane & 15, lq = lane >> 4;
        const int a_base = (wm * 64 + li) * LDK + 4 * lq, b_base = (wn * 64 + li) * LDK + 4 * lq;
        f32x4 acc[4][4];
        for (int mi = 0; mi < 4; ++mi) for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
        c0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime();
        for (int i = 0; i < iters; ++i) {
            const int cur = i & 1;
            f32x4 af[4], bf[4];
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) af[mi] = *reinterpret_cast<const f32x4*>(&sa[cur][a_base + mi * 16 * LDK]);
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) bf[ni] = *reinterpret_cast<const f32x4*>(&sb[cur][b_base + ni * 16 * LDK]);
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                    for (int ni = 0; ni < 4; ++ni) {
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[mi][k], bf[ni][k], acc[mi][ni], 0, 0, 0);
                        if ((ni & 1) == 0) {
#pragma unroll
                            for (int q = 0; q < NV; ++q) v[q & 7] = fmaf(v[q & 7], 1.0001f, v[(q + 1) & 7]);
                        }
                    }
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
        for (int mi = 0; mi < 4; ++mi) for (int ni = 0; ni < 4; ++ni) s += acc[mi][ni][(mi + ni) & 3];
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    for (int j = 0; j < 8; ++j) s += v[j];
    out[g * 256 + tid] = s;
    if (tid == 0) st[g] = Stamp{c0, r0, c1, r1};
}

template <typename F>
static void run(const char* name, F launch, int grid, double flop_per_wg_iter, int iters) {
    float* out; Stamp* st;
    hipMalloc(&out, (size_t)grid * 256 * 4); hipMalloc(&st, (size_t)grid * sizeof(Stamp));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    // soak: >= 1.5 s of back-to-back launches, then 5 measured ones
    launch(out, st, grid, iters); hipDeviceSynchronize();
    hipEventRecord(e0); launch(out, st, grid, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms1; hipEventElapsedTime(&ms1, e0, e1);
    const int soak = (int)(1500.0f / ms1) + 1;
    for (int i = 0; i < soak; ++i) launch(out, st, grid, iters);
    hipEventRecord(e0);
    const int reps = 5;
    for (int i = 0; i < reps; ++i) launch(out, st, grid, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
    std::vector<Stamp> h(grid);
    hipMemcpy(h.data(), st, grid * sizeof(Stamp), hipMemcpyDeviceToHost);
    std::vector<double> clk, cyc;
    for (auto& s : h) { clk.push_back((double)(s.c1 - s.c0) / (double)(s.r1 - s.r0) * 0.1); cyc.push_back((double)(s.c1 - s.c0)); }
    std::sort(clk.begin(), clk.end()); std::sort(cyc.begin(), cyc.end());
    const double tf = flop_per_wg_iter * iters * grid / (ms * 1e-3) / 1e12;
    const double cyc_per_iter = cyc[grid / 2] / iters;
    printf("%-10s grid %5d  %8.3f ms  %7.2f TFLOP/s  in-kernel clock median %.3f GHz (min %.3f max %.3f)  %.1f cycles/iter  -> pipe use %.3f\n",
           name, grid, ms, tf, clk[grid / 2], clk.front(), clk.back(), cyc_per_iter, flop_per_wg_iter / 256.0 / cyc_per_iter);
    fflush(stdout);
    hipFree(out); hipFree(st);
}

int main(int argc, char** argv) {
    const int wgs_per_cu = argc > 1 ? atoi(argv[1]) : 1;
    const int grid = 256 * wgs_per_cu;
    // FLOP per workgroup per iteration: bare32: 4 waves x 32 MFMA x 4096; bare16: 4 waves x 64 MFMA x 2048
    run("bare32", [](float* o, Stamp* s, int g, int it) { bare<0><<<g, 256>>>(o, s, it); }, grid, 4.0 * 32 * 4096, 4000);
    run("bare16", [](float* o, Stamp* s, int g, int it) { bare<1><<<g, 256>>>(o, s, it); }, grid, 4.0 * 64 * 2048, 4000);
    run("lds32", [](float* o, Stamp* s, int g, int it) { ldsloop<0, 0><<<g, 256>>>(o, s, it); }, grid, 4.0 * 32 * 4096, 4000);
    run("lds32v1", [](float* o, Stamp* s, int g, int it) { ldsloop<0, 1><<<g, 256>>>(o, s, it); }, grid, 4.0 * 32 * 4096, 4000);
    run("lds32v2", [](float* o, Stamp* s, int g, int it) { ldsloop<0, 2><<<g, 256>>>(o, s, it); }, grid, 4.0 * 32 * 4096, 4000);
    run("lds32v4", [](float* o, Stamp* s, int g, int it) { ldsloop<0, 4><<<g, 256>>>(o, s, it); }, grid, 4.0 * 32 * 4096, 4000);
    run("lds16", [](float* o, Stamp* s, int g, int it) { ldsloop<1, 0><<<g, 256>>>(o, s, it); }, grid, 4.0 * 64 * 2048, 4000);
    run("lds16v2", [](float* o, Stamp* s, int g, int it) { ldsloop<1, 2><<<g, 256>>>(o, s, it); }, grid, 4.0 * 64 * 2048, 4000);
    run("lds16v4", [](float* o, Stamp* s, int g, int it) { ldsloop<1, 4><<<g, 256>>>(o, s, it); }, grid, 4.0 * 64 * 2048, 4000);
    return 0;
}
