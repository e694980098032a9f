// Probe: does v_mfma_f32_32x32x16_f16 flush fp16 subnormal inputs?  And is the
// 3-term split product accurate?  (hipcc --offload-arch=gfx950 tools/probe_mfma_f16.hip -o /tmp/probe && /tmp/probe)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
__global__ void k(const float* av, const float* bv, float* out) {
    const int l = threadIdx.x, r = l & 31, h = l >> 5;
    h8 a, b;
    for (int j = 0; j < 8; ++j) {   // A[r][k] = av[k] (all rows equal), B[k][c] = bv[k] (all cols equal), k = 8h + j
        a[j] = (_Float16)av[8 * h + j];
        b[j] = (_Float16)bv[8 * h + j];
    }
    f16v c = {0};
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    if (l == 0) out[0] = c[0];
}
int main() {
    float ha[16], hb[16], *da, *db, *dout, res;
    hipMalloc(&da, 64); hipMalloc(&db, 64); hipMalloc(&dout, 4);
    // case 1: one subnormal-f16 a (3e-6 is subnormal: min normal 6.1e-5) times b = 1024
    for (int i = 0; i < 16; ++i) ha[i] = hb[i] = 0.f;
    ha[0] = 3.0e-6f; hb[0] = 1024.f;
    hipMemcpy(da, ha, 64, hipMemcpyHostToDevice); hipMemcpy(db, hb, 64, hipMemcpyHostToDevice);
    k<<<1, 64>>>(da, db, dout); hipMemcpy(&res, dout, 4, hipMemcpyDeviceToHost);
    printf("subnormal a=3e-6 (f16 %g) * 1024 -> %g  (expect %g if denormals honoured, 0 if flushed)\n",
           (float)(_Float16)3.0e-6f, res, (float)(_Float16)3.0e-6f * 1024.f);
    // case 2: subnormal on the B side
    ha[0] = 1024.f; hb[0] = 3.0e-6f;
    hipMemcpy(da, ha, 64, hipMemcpyHostToDevice); hipMemcpy(db, hb, 64, hipMemcpyHostToDevice);
    k<<<1, 64>>>(da, db, dout); hipMemcpy(&res, dout, 4, hipMemcpyDeviceToHost);
    printf("subnormal b -> %g\n", res);
    return 0;
}
