// Shared host/device helpers for libeabnet_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/eabnet_hip.h"

#define EAB_CHECK_ARG(cond)            \
    do {                               \
        if (!(cond)) return EAB_EINVAL; \
    } while (0)

static inline int eab_hip_status(hipError_t e) { return e == hipSuccess ? EAB_OK : EAB_EHIP_BASE + (int)e; }

// Launch status without a device sync: hipGetLastError picks up invalid launch
// configurations; execution faults surface at the caller's next sync.
#define EAB_RETURN_LAUNCH_STATUS() return eab_hip_status(hipGetLastError())

static inline hipStream_t eab_stream(eab_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float eab_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }
// tanh through the same exp: |abs err| ~1e-7, well inside the 1e-4 parity bar.
__device__ __forceinline__ float eab_tanh(float x) {
    float ax = fabsf(x);
    float e = expf(-2.0f * ax);
    float t = (1.0f - e) / (1.0f + e);
    return copysignf(t, x);
}
__device__ __forceinline__ float eab_prelu(float x, float a) { return x > 0.0f ? x : a * x; }
