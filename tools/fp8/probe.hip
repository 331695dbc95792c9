// Diagnostic (not part of libovhip): one v_mfma_scale_f32_16x16x128_f8f6f4 per wave on caller-provided per-lane operand bytes,
// to establish the A/B lane maps of the MX-scaled MFMA with exact integer data, and a bare issue-rate loop.
#include <hip/hip_runtime.h>
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void probe_kernel(const unsigned char* a, const unsigned char* b, float* out, int scale_a, int scale_b) {
    const int lane = threadIdx.x;
    i32x8 va, vb;
    for (int i = 0; i < 8; ++i) {
        va[i] = ((const int*)(a + lane * 32))[i];
        vb[i] = ((const int*)(b + lane * 32))[i];
    }
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(va, vb, c, 0, 0, 0, scale_a, 0, scale_b);
    for (int i = 0; i < 4; ++i) out[lane * 4 + i] = c[i];
}

// 4 independent accumulators per wave, `iters` x 4 MFMAs, operands in registers: matrix-pipe rate of the scaled fp8 form
__global__ __launch_bounds__(256) void rate_kernel(float* out, int iters, int seed) {
    i32x8 va, vb;
    for (int i = 0; i < 8; ++i) { va[i] = 0x38404438 ^ (threadIdx.x * 0x01010101 & 0x07070707) ^ seed; vb[i] = 0x3c383038 + i; }
    f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    for (int it = 0; it < iters; ++it) {
        c0 = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(va, vb, c0, 0, 0, 0, 127, 0, 127);
        c1 = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(vb, va, c1, 0, 0, 0, 127, 0, 127);
        c2 = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(va, va, c2, 0, 0, 0, 127, 0, 127);
        c3 = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(vb, vb, c3, 0, 0, 0, 127, 0, 127);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
}

extern "C" int fp8_probe(const unsigned char* a, const unsigned char* b, float* out, int scale_a, int scale_b, void* stream) {
    hipLaunchKernelGGL(probe_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, a, b, out, scale_a, scale_b);
    return (int)hipGetLastError();
}
extern "C" int fp8_rate(float* out, int blocks, int iters, void* stream) {
    hipLaunchKernelGGL(rate_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, out, iters, 0);
    return (int)hipGetLastError();
}
