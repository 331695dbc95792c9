// eval.hip — consumers of the encode + logits path: zero-shot classifier weights and top-k ranking.
//
//   ov_class_mean_normalize  class_embeddings.reshape(C, T, E).mean(1) / norm   open_clip/zero_shot_classifier.py:54-57
//   ov_topk                  argmax / argsort()[:k] of a logit or distance row   src/evaluators/proj/image_text/
//                            discriminative_classifier.py:308 (argmax), image_text_retrieval.py:44-50,78-83 (ranks[:k])
// Index work is bit-exact: ties are broken towards the smaller index (what a stable argsort of the negated row gives).
#include "common.h"

namespace {

// one wave per class: out[c,:] = normalize(mean_t emb[c*T + t, :])
__global__ __launch_bounds__(256) void class_mean_norm_kernel(const float* __restrict__ emb, float* __restrict__ out, int C, int T,
                                                              int E) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float invT = 1.0f / (float)T;
    for (int c = blockIdx.x * 4 + wave; c < C; c += gridDim.x * 4) {
        float ss = 0.f;
        for (int e = lane; e < E; e += 64) {
            float s = 0.f;
            for (int t = 0; t < T; ++t) s += emb[((int64_t)c * T + t) * E + e];
            s *= invT;
            out[(int64_t)c * E + e] = s;
            ss += s * s;
        }
        const float inv = 1.0f / sqrtf(wave_sum(ss));            // the reference divides by the plain norm (no eps)
        for (int e = lane; e < E; e += 64) out[(int64_t)c * E + e] *= inv;
    }
}

// (value, index) ordering: larger value first (smaller when !LARGEST), then smaller index
template <bool LARGEST>
__device__ __forceinline__ bool better(float v, int i, float bv, int bi) {
    return LARGEST ? (v > bv || (v == bv && i < bi)) : (v < bv || (v == bv && i < bi));
}

// one wave per row; k selection rounds, each a lexicographic arg-best over the elements ranked after the previous winner
template <bool LARGEST>
__global__ __launch_bounds__(256) void topk_kernel(const float* __restrict__ x, int64_t ldx, int rows, int cols, int k,
                                                   int64_t* __restrict__ idx_out, float* __restrict__ val_out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float worst = LARGEST ? -INFINITY : INFINITY;
    for (int row = blockIdx.x * 4 + wave; row < rows; row += gridDim.x * 4) {
        const float* xr = x + (int64_t)row * ldx;
        float pv = LARGEST ? INFINITY : -INFINITY;
        int pi = -1;
        for (int r = 0; r < k; ++r) {
            float bv = worst;
            int bi = 0x7fffffff;
            for (int c = lane; c < cols; c += 64) {
                const float v = xr[c];
                const bool after_prev = (pi < 0) || better<LARGEST>(pv, pi, v, c);     // strictly after the previous winner
                if (after_prev && (bi == 0x7fffffff || better<LARGEST>(v, c, bv, bi))) { bv = v; bi = c; }
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const float ov = __shfl_xor(bv, o, 64);
                const int oi = __shfl_xor(bi, o, 64);
                if (oi != 0x7fffffff && (bi == 0x7fffffff || better<LARGEST>(ov, oi, bv, bi))) { bv = ov; bi = oi; }
            }
            if (lane == 0) {
                idx_out[(int64_t)row * k + r] = bi == 0x7fffffff ? -1 : bi;
                if (val_out) val_out[(int64_t)row * k + r] = bv;
            }
            pv = bv; pi = bi;
        }
    }
}

}  // namespace

extern "C" int ov_class_mean_normalize(const float* emb, float* out, int C, int T, int E, ov_stream_t stream) {
    if (!emb || !out || C <= 0 || T <= 0 || E <= 0) return OV_ERR_INVALID;
    int blocks = (C + 3) / 4;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(class_mean_norm_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, emb, out, C, T, E);
    OV_LAUNCH_CHECK();
    return OV_OK;
}

extern "C" int ov_topk(const float* x, int64_t ldx, int rows, int cols, int k, int largest, int64_t* idx_out, float* val_out,
                       ov_stream_t stream) {
    if (!x || !idx_out || rows <= 0 || cols <= 0 || k <= 0 || k > cols || ldx < cols) return OV_ERR_INVALID;
    int blocks = (rows + 3) / 4;
    if (blocks > 8192) blocks = 8192;
    if (largest)
        hipLaunchKernelGGL(topk_kernel<true>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, ldx, rows, cols, k, idx_out, val_out);
    else
        hipLaunchKernelGGL(topk_kernel<false>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, ldx, rows, cols, k, idx_out, val_out);
    OV_LAUNCH_CHECK();
    return OV_OK;
}
