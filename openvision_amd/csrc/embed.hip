// embed.hip — the memory-bound glue kernels either side of the block stack (gfx950).
//
//   ov_im2col_patches  operand gather for conv1 (k = s = P, no bias)      open_clip/transformer.py:469,610-612
//   ov_cls_rows        class_embedding concat + pos-emb row 0             transformer.py:615-617
//   ov_mean_pool       _global_pool 'avg' (cls excluded) / 'tok'          transformer.py:599-603
//   ov_text_embed      token_embedding(text) + positional_embedding      model.py:272-274
//   ov_gather_rows     text_global_pool 'last' / 'first'                  transformer.py:655-658
//   ov_convert         dtype casts (.to(cast_dtype))
//   ov_l2norm          F.normalize(x, dim=-1)                             model.py:267,284
// All are 16-byte-per-lane streaming kernels; none re-reads its input.
#include "common.h"

namespace {

template <bool IMG_F32>
__global__ __launch_bounds__(256) void im2col_kernel(const void* __restrict__ img, ov_bf16* __restrict__ out,
                                                     int B, int S, int P, int g, int Kpad, int64_t total) {
    // one thread per 8 output columns
    const int cpr = Kpad >> 3;
    const int PP = P * P, K = 3 * PP;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = i / cpr;
        const int ch = (int)(i - row * cpr);
        const int b = (int)(row / (g * g));
        const int pr = (int)(row - (int64_t)b * g * g);
        const int py = pr / g, px = pr - py * g;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int k = ch * 8 + e;
            float x = 0.f;
            if (k < K) {
                const int c = k / PP, rem = k - c * PP;
                const int ii = rem / P, jj = rem - ii * P;
                const int64_t src = (((int64_t)b * 3 + c) * S + (py * P + ii)) * S + (px * P + jj);
                x = IMG_F32 ? ((const float*)img)[src] : bf16_bits_to_f32(((const ov_bf16*)img)[src]);
            }
            v[e] = x;
        }
        u32x4_t w = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7])};
        *(u32x4_t*)(out + row * Kpad + ch * 8) = w;
    }
}

__global__ __launch_bounds__(256) void cls_rows_kernel(ov_bf16* __restrict__ x, int64_t ldx, const float* __restrict__ cls,
                                                       const float* __restrict__ pos0, int B, int L, int D) {
    const int total = B * D;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int b = i / D, d = i - b * D;
        const float v = round_bf16(cls[d]) + round_bf16(pos0[d]);
        x[(int64_t)b * L * ldx + d] = f32_to_bf16_bits(v);
    }
}

// grid (B, ceil(D/8/64)), block 256: wave w sums tokens first+w, first+w+4, ...; lanes own 16-B column chunks
__global__ __launch_bounds__(256) void mean_pool_kernel(const ov_bf16* __restrict__ x, int64_t ldx, float* __restrict__ out,
                                                        int L, int D, int first) {
    __shared__ float red[4][64][8];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.x;
    const int ch = blockIdx.y * 64 + lane;
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (ch * 8 < D) {
        const ov_bf16* p = x + (int64_t)b * L * ldx + ch * 8;
        for (int t = first + wave; t < L; t += 4) {
            const u32x4_t w = *(const u32x4_t*)(p + (int64_t)t * ldx);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                acc[2 * e] += bf16lo_to_f32(w[e]);
                acc[2 * e + 1] += bf16hi_to_f32(w[e]);
            }
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) red[wave][lane][e] = acc[e];
    __syncthreads();
    if (wave == 0 && ch * 8 < D) {
        const float inv = 1.0f / (float)(L - first);
        float o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (red[0][lane][e] + red[1][lane][e] + red[2][lane][e] + red[3][lane][e]) * inv;
        float4* q = (float4*)(out + (int64_t)b * D + ch * 8);
        q[0] = make_float4(o[0], o[1], o[2], o[3]);
        q[1] = make_float4(o[4], o[5], o[6], o[7]);
    }
}

__global__ __launch_bounds__(256) void text_embed_kernel(const int64_t* __restrict__ tokens, const ov_bf16* __restrict__ table,
                                                         const ov_bf16* __restrict__ pos, ov_bf16* __restrict__ x, int64_t ldx,
                                                         int T, int D, int V, int* __restrict__ err, int64_t total) {
    const int cpr = D >> 3;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = i / cpr;
        const int ch = (int)(i - row * cpr);
        const int t = (int)(row % T);
        int64_t id = tokens[row];
        if (id < 0 || id >= V) {
            if (err) *err = 1;
            id = id < 0 ? 0 : V - 1;
        }
        const u32x4_t a = *(const u32x4_t*)(table + id * D + ch * 8);
        const u32x4_t p = *(const u32x4_t*)(pos + (int64_t)t * D + ch * 8);
        u32x4_t o;
#pragma unroll
        for (int e = 0; e < 4; ++e)
            o[e] = pack_bf16x2(bf16lo_to_f32(a[e]) + bf16lo_to_f32(p[e]), bf16hi_to_f32(a[e]) + bf16hi_to_f32(p[e]));
        *(u32x4_t*)(x + row * ldx + ch * 8) = o;
    }
}

__global__ __launch_bounds__(256) void gather_rows_kernel(const ov_bf16* __restrict__ x, int64_t ldx, ov_bf16* __restrict__ out,
                                                          int64_t ldo, int B, int L, int t, int D) {
    const int cpr = D >> 3;
    const int total = B * cpr;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int b = i / cpr, ch = i - b * cpr;
        *(u32x4_t*)(out + (int64_t)b * ldo + ch * 8) = *(const u32x4_t*)(x + ((int64_t)b * L + t) * ldx + ch * 8);
    }
}

template <bool SRC_F32, bool DST_F32>
__global__ __launch_bounds__(256) void convert_kernel(const void* __restrict__ src, int64_t lds, void* __restrict__ dst,
                                                      int64_t ldd, int64_t rows, int cols) {
    const int64_t total = rows * cols;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / cols;
        const int c = (int)(i - r * cols);
        const float v = SRC_F32 ? ((const float*)src)[r * lds + c] : bf16_bits_to_f32(((const ov_bf16*)src)[r * lds + c]);
        if (DST_F32) ((float*)dst)[r * ldd + c] = v;
        else ((ov_bf16*)dst)[r * ldd + c] = f32_to_bf16_bits(v);
    }
}

// one wave per row
template <bool X_F32>
__global__ __launch_bounds__(256) void l2norm_kernel(const void* __restrict__ x, int64_t ldx, float* __restrict__ y, int64_t ldy,
                                                     int64_t rows, int E) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < rows; row += (int64_t)gridDim.x * 4) {
        float ss = 0.f;
        for (int c = lane; c < E; c += 64) {
            const float v = X_F32 ? ((const float*)x)[row * ldx + c] : bf16_bits_to_f32(((const ov_bf16*)x)[row * ldx + c]);
            ss += v * v;
        }
        const float nrm = sqrtf(wave_sum(ss));
        const float inv = 1.0f / fmaxf(nrm, 1e-12f);
        for (int c = lane; c < E; c += 64) {
            const float v = X_F32 ? ((const float*)x)[row * ldx + c] : bf16_bits_to_f32(((const ov_bf16*)x)[row * ldx + c]);
            y[row * ldy + c] = v * inv;
        }
    }
}

inline unsigned grid_for(int64_t total, int block) {
    int64_t g = (total + block - 1) / block;
    if (g > 8192) g = 8192;
    if (g < 1) g = 1;
    return (unsigned)g;
}

}  // namespace

extern "C" int ov_im2col_patches(const void* image, int img_dtype, ov_bf16* out, int B, int S, int P, int Kpad,
                                 ov_stream_t stream) {
    if (!image || !out || B <= 0 || S <= 0 || P <= 0) return OV_ERR_INVALID;
    if (S % P || Kpad % 8 || Kpad < 3 * P * P || ((uintptr_t)out & 15)) return OV_ERR_INVALID;
    const int g = S / P;
    const int64_t total = (int64_t)B * g * g * (Kpad / 8);
    hipStream_t st = (hipStream_t)stream;
    if (img_dtype == OV_F32)
        hipLaunchKernelGGL(im2col_kernel<true>, dim3(grid_for(total, 256)), dim3(256), 0, st, image, out, B, S, P, g, Kpad, total);
    else if (img_dtype == OV_BF16)
        hipLaunchKernelGGL(im2col_kernel<false>, dim3(grid_for(total, 256)), dim3(256), 0, st, image, out, B, S, P, g, Kpad, total);
    else
        return OV_ERR_INVALID;
    OV_LAUNCH_CHECK();
    return OV_OK;
}

extern "C" int ov_cls_rows(ov_bf16* x, int64_t ldx, const float* cls, const float* pos0, int B, int L, int D,
                           ov_stream_t stream) {
    if (!x || !cls || !pos0 || B <= 0 || L <= 0 || D <= 0 || ldx < D) return OV_ERR_INVALID;
    hipLaunchKernelGGL(cls_rows_kernel, dim3(grid_for((int64_t)B * D, 256)), dim3(256), 0, (hipStream_t)stream, x, ldx, cls,
                       pos0, B, L, D);
    OV_LAUNCH_CHECK();
    return OV_OK;
}

extern "C" int ov_mean_pool(const ov_bf16* x, int64_t ldx, float* out, int B, int L, int D, int first,
                            ov_stream_t stream) {
    if (!x || !out || B <= 0 || L <= 0 || D <= 0 || first < 0 || first >= L) return OV_ERR_INVALID;
    if (D % 8 || ldx % 8 || ldx < D || (((uintptr_t)x | (uintptr_t)out) & 15)) return OV_ERR_INVALID;
    hipLaunchKernelGGL(mean_pool_kernel, dim3((unsigned)B, (unsigned)((D / 8 + 63) / 64)), dim3(256), 0, (hipStream_t)stream, x,
                       ldx, out, L, D, first);
    OV_LAUNCH_CHECK();
    return OV_OK;
}

extern "C" int ov_text_embed(const int64_t* tokens, const ov_bf16* table, const ov_bf16* pos, ov_bf16* x, int64_t ldx,
                             int B, int T, int D, int V, int* err_flag, ov_stream_t stream) {
    if (!tokens || !table || !pos || !x || B <= 0 || T <= 0 || D <= 0 || V <= 0) return OV_ERR_INVALID;
    if (D % 8 || ldx % 8 || ldx < D || (((uintptr_t)table | (uintptr_t)pos | (uintptr_t)x) & 15)) return OV_ERR_INVALID;
    const int64_t total = (int64_t)B * T * (D / 8);
    hipLaunchKernelGGL(text_embed_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, tokens, table, pos, x,
                       ldx, T, D, V, err_flag, total);
    OV_LAUNCH_CHECK();
    return OV_OK;
}

extern "C" int ov_gather_rows(const ov_bf16* x, int64_t ldx, ov_bf16* out, int64_t ldo, int B, int L, int t, int D,
                              ov_stream_t stream) {
    if (!x || !out || B <= 0 || L <= 0 || t < 0 || t >= L || D <= 0) return OV_ERR_INVALID;
    if (D % 8 || ldx % 8 || ldo % 8 || ldx < D || ldo < D || (((uintptr_t)x | (uintptr_t)out) & 15)) return OV_ERR_INVALID;
    hipLaunchKernelGGL(gather_rows_kernel, dim3(grid_for((int64_t)B * (D / 8), 256)), dim3(256), 0, (hipStream_t)stream, x, ldx,
                       out, ldo, B, L, t, D);
    OV_LAUNCH_CHECK();
    return OV_OK;
}

extern "C" int ov_convert(const void* src, int src_dtype, int64_t lds, void* dst, int dst_dtype, int64_t ldd, int64_t rows,
                          int cols, ov_stream_t stream) {
    if (!src || !dst || rows <= 0 || cols <= 0 || lds < cols || ldd < cols) return OV_ERR_INVALID;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid(grid_for(rows * cols, 256)), blk(256);
    if (src_dtype == OV_F32 && dst_dtype == OV_BF16)
        hipLaunchKernelGGL((convert_kernel<true, false>), grid, blk, 0, st, src, lds, dst, ldd, rows, cols);
    else if (src_dtype == OV_BF16 && dst_dtype == OV_F32)
        hipLaunchKernelGGL((convert_kernel<false, true>), grid, blk, 0, st, src, lds, dst, ldd, rows, cols);
    else if (src_dtype == OV_F32 && dst_dtype == OV_F32)
        hipLaunchKernelGGL((convert_kernel<true, true>), grid, blk, 0, st, src, lds, dst, ldd, rows, cols);
    else if (src_dtype == OV_BF16 && dst_dtype == OV_BF16)
        hipLaunchKernelGGL((convert_kernel<false, false>), grid, blk, 0, st, src, lds, dst, ldd, rows, cols);
    else
        return OV_ERR_INVALID;
    OV_LAUNCH_CHECK();
    return OV_OK;
}

extern "C" int ov_l2norm(const void* x, int x_dtype, int64_t ldx, float* y, int64_t ldy, int64_t rows, int E,
                         ov_stream_t stream) {
    if (!x || !y || rows <= 0 || E <= 0 || ldx < E || ldy < E) return OV_ERR_INVALID;
    int64_t blocks = (rows + 3) / 4;
    if (blocks > 8192) blocks = 8192;
    hipStream_t st = (hipStream_t)stream;
    if (x_dtype == OV_F32)
        hipLaunchKernelGGL(l2norm_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, st, x, ldx, y, ldy, rows, E);
    else if (x_dtype == OV_BF16)
        hipLaunchKernelGGL(l2norm_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, st, x, ldx, y, ldy, rows, E);
    else
        return OV_ERR_INVALID;
    OV_LAUNCH_CHECK();
    return OV_OK;
}
