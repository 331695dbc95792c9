// quant.hip — activation quantisation to OCP fp8 e4m3 for the fp8 GEMM path (BASELINE.json config #5), gfx950.
//
//   ov_quant_rows_fp8       q[r, :] = e4m3(x[r, :] / s_r),  s_r = max|x[r, :]| / 448           (x bf16)
//   ov_layernorm_quant_fp8  y = LayerNorm(x[r, :]) * gamma + beta (fp32, eps, biased variance: open_clip/transformer.py:15-30),
//                           then the same row quantisation of y -- the LN output never exists in bf16
// One 64-lane wave per row, the row held in registers (read once), two-pass statistics as in layernorm.hip, row maximum by
// wave reduction, v_cvt_pk_fp8_f32 for the conversion (round to nearest even; |y / s| <= 448 by construction, so no overflow).
// HBM-bound: 2 bytes read + 1 byte written per element (+ 4 bytes of scale per row).
#include "common.h"

namespace {

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

template <int NCH, bool LN>
__global__ __launch_bounds__(256) void quant_rows(const ov_bf16* __restrict__ x, int64_t ldx, const float* __restrict__ gamma,
                                                 const float* __restrict__ beta, unsigned char* __restrict__ q, int64_t ldq,
                                                 float* __restrict__ scale, int64_t rows, int D, float eps, float* __restrict__ amax_acc) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int nchunk = D >> 3;
    const float invD = 1.0f / (float)D;
    for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < rows; row += (int64_t)gridDim.x * 4) {
        float v[NCH][8];
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int ch = lane + c * 64;
            if (ch < nchunk) {
                const u32x4_t w = *(const u32x4_t*)(x + row * ldx + ch * 8);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[c][2 * e] = bf16lo_to_f32(w[e]);
                    v[c][2 * e + 1] = bf16hi_to_f32(w[e]);
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) s += v[c][e];
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[c][e] = 0.f;
            }
        }
        if (LN) {
            const float mean = wave_sum(s) * invD;
            float qq = 0.f;
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                if (lane + c * 64 < nchunk) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float d = v[c][e] - mean;
                        qq += d * d;
                    }
                }
            }
            const float rstd = rsqrtf(wave_sum(qq) * invD + eps);
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                const int ch = lane + c * 64;
                if (ch < nchunk) {
                    const float4 g0 = *(const float4*)(gamma + ch * 8), g1 = *(const float4*)(gamma + ch * 8 + 4);
                    const float4 b0 = *(const float4*)(beta + ch * 8), b1 = *(const float4*)(beta + ch * 8 + 4);
                    const float gg[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
                    const float bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[c][e] = (v[c][e] - mean) * rstd * gg[e] + bb[e];
                }
            }
        }
        float amax = 0.f;
#pragma unroll
        for (int c = 0; c < NCH; ++c)
#pragma unroll
            for (int e = 0; e < 8; ++e) amax = fmaxf(amax, fabsf(v[c][e]));
        amax = fmaxf(wave_max(amax), 1e-12f);
        const float sc = amax * (1.0f / 448.0f);
        const float inv = 448.0f / amax;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int ch = lane + c * 64;
            if (ch < nchunk) {
                int lo = 0, hi = 0;
                lo = __builtin_amdgcn_cvt_pk_fp8_f32(v[c][0] * inv, v[c][1] * inv, lo, false);
                lo = __builtin_amdgcn_cvt_pk_fp8_f32(v[c][2] * inv, v[c][3] * inv, lo, true);
                hi = __builtin_amdgcn_cvt_pk_fp8_f32(v[c][4] * inv, v[c][5] * inv, hi, false);
                hi = __builtin_amdgcn_cvt_pk_fp8_f32(v[c][6] * inv, v[c][7] * inv, hi, true);
                *(u32x2_t*)(q + row * ldq + ch * 8) = u32x2_t{(unsigned)lo, (unsigned)hi};
            }
        }
        if (lane == 0) {
            scale[row] = sc;
            // calibration: running maximum over all rows (values >= 0: their bit patterns order like unsigned integers); the plain
            // read skips the atomic once the maximum has settled
            if (amax_acc != nullptr && amax > *(volatile float*)amax_acc) atomicMax((unsigned*)amax_acc, __float_as_uint(amax));
        }
    }
}

__global__ void amax_roll_kernel(float* cur, const float* next, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) cur[i] = fmaxf(cur[i], next[i]);
}

template <bool LN>
int launch_quant(const ov_bf16* x, int64_t ldx, const float* g, const float* b, unsigned char* q, int64_t ldq, float* scale,
                 int64_t rows, int D, float eps, float* amax_acc, hipStream_t st) {
    int64_t blocks = (rows + 3) / 4;
    if (blocks > 16384) blocks = 16384;
    const dim3 grid((unsigned)blocks), blk(256);
    const int nch = (D / 8 + 63) / 64;
#define OV_Q(N) hipLaunchKernelGGL((quant_rows<N, LN>), grid, blk, 0, st, x, ldx, g, b, q, ldq, scale, rows, D, eps, amax_acc)
    if (nch <= 1) OV_Q(1);
    else if (nch <= 2) OV_Q(2);
    else if (nch <= 3) OV_Q(3);
    else if (nch <= 4) OV_Q(4);
    else if (nch <= 8) OV_Q(8);
    else OV_Q(16);
#undef OV_Q
    OV_LAUNCH_CHECK();
    return OV_OK;
}

}  // namespace

extern "C" int ov_quant_rows_fp8(const ov_bf16* x, int64_t ldx, unsigned char* q, int64_t ldq, float* rowscale, int64_t rows, int D,
                                 float* amax_acc, ov_stream_t stream) {
    if (!x || !q || !rowscale || rows <= 0 || D <= 0) return OV_ERR_INVALID;
    if (D % 8 || D > 8192 || ldx % 8 || ldx < D || ldq % 8 || ldq < D) return OV_ERR_UNSUPPORTED;
    if (((uintptr_t)x & 15) || ((uintptr_t)q & 7)) return OV_ERR_INVALID;
    return launch_quant<false>(x, ldx, nullptr, nullptr, q, ldq, rowscale, rows, D, 0.f, amax_acc, (hipStream_t)stream);
}

extern "C" int ov_layernorm_quant_fp8(const ov_bf16* x, int64_t ldx, const float* gamma, const float* beta, unsigned char* q,
                                      int64_t ldq, float* rowscale, int64_t rows, int D, float eps, ov_stream_t stream) {
    if (!x || !q || !rowscale || !gamma || !beta || rows <= 0 || D <= 0) return OV_ERR_INVALID;
    if (D % 8 || D > 8192 || ldx % 8 || ldx < D || ldq % 8 || ldq < D) return OV_ERR_UNSUPPORTED;
    if (((uintptr_t)x & 15) || ((uintptr_t)q & 7) || (((uintptr_t)gamma | (uintptr_t)beta) & 15)) return OV_ERR_INVALID;
    return launch_quant<true>(x, ldx, gamma, beta, q, ldq, rowscale, rows, D, eps, nullptr, (hipStream_t)stream);
}

// Delayed scaling: cur[i] = max(cur[i], next[i]) -- the maxima recorded by the producers of the previous forward become the scales of
// this one (launched once at the top of a forward, so every kernel of a forward sees one consistent value).
extern "C" int ov_amax_roll(float* cur, const float* next, int n, ov_stream_t stream) {
    if (!cur || !next || n <= 0) return OV_ERR_INVALID;
    hipLaunchKernelGGL(amax_roll_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, cur, next, n);
    OV_LAUNCH_CHECK();
    return OV_OK;
}
