// Probe: operand / result layout and timing of v_mfma_f32_4x4x1_16b_f32 on gfx950.
//   hipcc --offload-arch=gfx950 tools/probe_mfma_4x4.hip -o /tmp/probe4 && /tmp/probe4
// Claim to check: lane l = 4*block + i supplies A_block[i][0] and B_block[0][i]; result VGPR r of lane
// l = 4*block + j holds D_block[r][j] = sum_k A_block[r][k] * B_block[k][j].
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void k(const float* a, const float* b, float* out) {
    const int l = threadIdx.x;
    f4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l], b[l], c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) out[l * 4 + r] = c[r];
}
template <int NC>
__global__ void timing(float* out, int iters) {
    const int l = threadIdx.x;
    f4 c[NC];
    for (int q = 0; q < NC; ++q) c[q] = f4{0.f, 0.f, 0.f, 0.f};
    float a = 1.0f + l, b = 0.5f;
    long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int q = 0; q < NC; ++q) c[q] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c[q], 0, 0, 0);
    }
    long long t1 = clock64();
    float s = 0.f;
    for (int q = 0; q < NC; ++q) s += c[q][q & 3];
    out[l] = s;
    if (l == 0) out[64] = (float)(t1 - t0) / ((float)NC * iters);
}
typedef float f16v __attribute__((ext_vector_type(4)));
template <int NC>
__global__ void timing16(float* out, int iters) {      // v_mfma_f32_16x16x4_f32 for comparison
    const int l = threadIdx.x;
    f4 c[NC];
    for (int q = 0; q < NC; ++q) c[q] = f4{0.f, 0.f, 0.f, 0.f};
    float a = 1.0f + l, b = 0.5f;
    long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int q = 0; q < NC; ++q) c[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c[q], 0, 0, 0);
    }
    long long t1 = clock64();
    float s = 0.f;
    for (int q = 0; q < NC; ++q) s += c[q][q & 3];
    out[l] = s;
    if (l == 0) out[64] = (float)(t1 - t0) / ((float)NC * iters);
}
int main() {
    float ha[64], hb[64], ho[256], *da, *db, *dout;
    hipMalloc(&da, 256); hipMalloc(&db, 256); hipMalloc(&dout, 2048);
    for (int l = 0; l < 64; ++l) { ha[l] = 1.0f + (l & 3) + 10.0f * (l >> 2); hb[l] = 100.0f * ((l & 3) + 1) + 1000.0f * (l >> 2); }
    hipMemcpy(da, ha, 256, hipMemcpyHostToDevice); hipMemcpy(db, hb, 256, hipMemcpyHostToDevice);
    k<<<1, 64>>>(da, db, dout); hipMemcpy(ho, dout, 1024, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l)
        for (int r = 0; r < 4; ++r) {
            const int blk = l >> 2, j = l & 3;
            const float want = ha[4 * blk + r] * hb[4 * blk + j];
            if (ho[l * 4 + r] != want) { if (bad < 5) printf("lane %d r %d: got %g want %g\n", l, r, ho[l * 4 + r], want); ++bad; }
        }
    printf("layout claim %s (%d mismatches)\n", bad ? "WRONG" : "confirmed", bad);
#define RUN(K, NC) K<NC><<<1, 64>>>(dout, 10000); hipMemcpy(ho, dout, 65 * 4, hipMemcpyDeviceToHost); \
    printf(#K " with %d independent chains: %.2f clock64 ticks per MFMA\n", NC, ho[64]);
    RUN(timing, 1) RUN(timing, 3) RUN(timing, 6) RUN(timing, 12)
    RUN(timing16, 1) RUN(timing16, 4) RUN(timing16, 8)
    return 0;
}
