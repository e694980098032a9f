// Probe: do packed-fp32 VALU instructions (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32) return wrong results while ANOTHER
// kernel streams matrix instructions on the same SIMDs?  (Round 2 found wrong imaginary parts in bfw_filter_sum_kernel --
// high half of a v_pk_add_f32, lanes 48-63 -- only next to kernels that stream f16 MFMAs, and removed packed fp32 from the
// build; this is the isolated form of that experiment: no LDS, no MFMA and no memory traffic in the victim's loop.)
//   hipcc -O3 --offload-arch=gfx950 tools/probe_pk_f32.hip -o tools/build/probe_pk_f32 && tools/build/probe_pk_f32
// Victim (stream 0): every lane iterates exact small-integer arithmetic in packed form (inline asm) and in scalar form and
// compares with the closed-form result.  Aggressor (stream 1): back-to-back launches of an MFMA loop of one type
// (none / f32 32x32x2 / f16 32x32x8 / bf16 32x32x16 / f16 16x16x16), sized so that both kernels share every CU.
// Output: per (aggressor, victim form) the number of wrong lanes-results out of all, and which lanes / halves.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

// ---- aggressors ------------------------------------------------------------------------------------------------------------
template <int KIND>   // 1: f32 32x32x2, 2: f16 32x32x8, 3: bf16 32x32x16, 4: f16 16x16x16
__global__ __launch_bounds__(256) void aggressor(float* out, int iters) {
    const int l = threadIdx.x;
    f32x16 acc[2];
    f32x4 acc4[4];
    for (int q = 0; q < 2; ++q) for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
    for (int q = 0; q < 4; ++q) acc4[q] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float fa = 1.0f + (l & 7) * 0.125f, fb = 0.5f + (l & 3) * 0.25f;
    f16x4 ha = {(_Float16)fa, (_Float16)fb, (_Float16)fa, (_Float16)fb};
    f16x8 ha8 = {(_Float16)fa, (_Float16)fb, (_Float16)fa, (_Float16)fb, (_Float16)fa, (_Float16)fb, (_Float16)fa, (_Float16)fb};
    bf16x8 ba8;
    for (int j = 0; j < 8; ++j) ba8[j] = (__bf16)(j & 1 ? fb : fa);
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if constexpr (KIND == 1) {
                acc[j & 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, acc[j & 1], 0, 0, 0);
            } else if constexpr (KIND == 2) {
                acc[j & 1] = __builtin_amdgcn_mfma_f32_32x32x8f16(ha, ha, acc[j & 1], 0, 0, 0);
            } else if constexpr (KIND == 3) {
                acc[j & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ba8, ba8, acc[j & 1], 0, 0, 0);
            } else {
                acc4[j & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha8, ha8, acc4[j & 3], 0, 0, 0);
            }
        }
    }
    float s = 0.f;
    for (int q = 0; q < 2; ++q) s += acc[q][q];
    for (int q = 0; q < 4; ++q) s += acc4[q][q & 3];
    if (s == 12345.678f) out[blockIdx.x * 256 + l] = s;      // keep the loop alive
}

// ---- victims ---------------------------------------------------------------------------------------------------------------
// acc = (0, 0); repeat n times: acc = acc + (1, 2); acc2 = acc2 * (1, 1) + ... exact in fp32 while n < 2^22
template <bool PACKED>
__global__ __launch_bounds__(256) void victim(f32x2* out, int n) {
    f32x2 acc = {0.f, 0.f}, inc = {1.f, 2.f}, acc2 = {(float)(threadIdx.x & 63), 1.0f}, one = {1.f, 1.f}, k = {3.f, 5.f};
    for (int i = 0; i < n; ++i) {
        if constexpr (PACKED) {
            asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(acc) : "v"(inc));
            asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(acc2) : "v"(one));
            asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc2) : "v"(one), "v"(k));
        } else {
            asm volatile("v_add_f32 %0, %0, %1" : "+v"(acc.x) : "v"(inc.x));
            asm volatile("v_add_f32 %0, %0, %1" : "+v"(acc.y) : "v"(inc.y));
            asm volatile("v_mul_f32 %0, %0, %1" : "+v"(acc2.x) : "v"(one.x));
            asm volatile("v_mul_f32 %0, %0, %1" : "+v"(acc2.y) : "v"(one.y));
            asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc2.x) : "v"(one.x), "v"(k.x));
            asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc2.y) : "v"(one.y), "v"(k.y));
        }
    }
    const size_t e = ((size_t)blockIdx.x * 256 + threadIdx.x) * 2;
    out[e] = acc;
    out[e + 1] = acc2;
}

// the form the production kernel had: packed adds whose operands are the VGPR-form results of this wave's own fp32 MFMAs
// (a = b = 1: every element of a 16x16x4 product is exactly 4)
template <bool PACKED>
__global__ __launch_bounds__(256) void victim_mfma(f32x2* out, int n) {
    f32x2 s01 = {0.f, 0.f}, s23 = {0.f, 0.f};
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    float one = 1.0f;
    asm volatile("" : "+v"(one));
    for (int i = 0; i < n; ++i) {
        f32x4 r = __builtin_amdgcn_mfma_f32_16x16x4f32(one, one, zero, 0, 0, 0);
        f32x2 lo = {r[0], r[1]}, hi = {r[2], r[3]};
        if constexpr (PACKED) {
            asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(s01) : "v"(lo));
            asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(s23) : "v"(hi));
        } else {
            asm volatile("v_add_f32 %0, %0, %1" : "+v"(s01.x) : "v"(lo.x));
            asm volatile("v_add_f32 %0, %0, %1" : "+v"(s01.y) : "v"(lo.y));
            asm volatile("v_add_f32 %0, %0, %1" : "+v"(s23.x) : "v"(hi.x));
            asm volatile("v_add_f32 %0, %0, %1" : "+v"(s23.y) : "v"(hi.y));
        }
    }
    const size_t e = ((size_t)blockIdx.x * 256 + threadIdx.x) * 2;
    out[e] = s01;
    out[e + 1] = s23;
}

template <int KIND>
static void launch_aggr(float* dbuf, int grid, int iters, hipStream_t s) {
    hipLaunchKernelGGL(aggressor<KIND>, dim3(grid), dim3(256), 0, s, dbuf, iters);
}

int main() {
    const int cus = 256, vgrid = cus * 4, agrid = cus * 4, n = 200000;
    hipStream_t s0, s1;
    CK(hipStreamCreate(&s0));
    CK(hipStreamCreate(&s1));
    f32x2* dout;
    float* dagg;
    CK(hipMalloc(&dout, (size_t)vgrid * 256 * 2 * sizeof(f32x2)));
    CK(hipMalloc(&dagg, (size_t)agrid * 256 * sizeof(float)));
    std::vector<f32x2> h((size_t)vgrid * 256 * 2);
    const char* names[5] = {"none", "f32 32x32x2", "f16 32x32x8", "bf16 32x32x16", "f16 16x16x32"};
    for (int form = 0; form < 2; ++form)
    for (int kind = 0; kind < 5; ++kind) {
        for (int packed = 1; packed >= 0; --packed) {
            long long bad = 0, bad_hi = 0, bad_lane48 = 0, total = 0;
            int first_lane = -1, first_half = -1;
            for (int rep = 0; rep < 6; ++rep) {
                // keep the aggressor running for the whole life of the victim: several launches queued on stream 1
                for (int q = 0; q < 6 && kind > 0; ++q) {
                    switch (kind) {
                        case 1: launch_aggr<1>(dagg, agrid, 60000, s1); break;
                        case 2: launch_aggr<2>(dagg, agrid, 60000, s1); break;
                        case 3: launch_aggr<3>(dagg, agrid, 60000, s1); break;
                        default: launch_aggr<4>(dagg, agrid, 60000, s1); break;
                    }
                }
                if (form == 0) {
                    if (packed) hipLaunchKernelGGL(victim<true>, dim3(vgrid), dim3(256), 0, s0, dout, n);
                    else hipLaunchKernelGGL(victim<false>, dim3(vgrid), dim3(256), 0, s0, dout, n);
                } else {
                    if (packed) hipLaunchKernelGGL(victim_mfma<true>, dim3(vgrid), dim3(256), 0, s0, dout, n);
                    else hipLaunchKernelGGL(victim_mfma<false>, dim3(vgrid), dim3(256), 0, s0, dout, n);
                }
                CK(hipStreamSynchronize(s0));
                CK(hipMemcpy(h.data(), dout, h.size() * sizeof(f32x2), hipMemcpyDeviceToHost));
                CK(hipStreamSynchronize(s1));
                for (size_t t = 0; t < (size_t)vgrid * 256; ++t) {
                    const int lane = (int)(t & 63);
                    const float w4 = 4.0f * n;
                    const float want[4] = {form ? w4 : (float)n, form ? w4 : 2.0f * n, form ? w4 : (float)lane + 3.0f * n,
                                           form ? w4 : 1.0f + 5.0f * n};
                    const float got[4] = {h[t * 2][0], h[t * 2][1], h[t * 2 + 1][0], h[t * 2 + 1][1]};
                    for (int c = 0; c < 4; ++c) {
                        ++total;
                        if (got[c] != want[c]) {
                            ++bad;
                            if (c & 1) ++bad_hi;
                            if (lane >= 48) ++bad_lane48;
                            if (first_lane < 0) { first_lane = lane; first_half = c & 1; }
                        }
                    }
                }
            }
            printf("%s | aggressor %-14s victim %-6s: %lld wrong of %lld results", form ? "adds of own fp32-MFMA results" : "register-only loop", names[kind], packed ? "packed" : "scalar", bad, total);
            if (bad) printf("  (high half: %lld, lanes 48-63: %lld, first: lane %d half %d)", bad_hi, bad_lane48, first_lane, first_half);
            printf("\n");
            fflush(stdout);
        }
    }
    return 0;
}
