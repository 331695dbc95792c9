// optim.hip — the parameter update of the training step (gfx950), SURVEY §8f row 4's "training step".
//
// The reference's trainer builds its update as an optax chain (src/optim/build_optax.py:272-278, used at src/main_clip.py:480-483):
//     clip_by_global_norm -> scale_by_adam(b1, b2, mu_dtype=bfloat16) -> add_decayed_weights(wd, kernels only) -> scale(lr) ->
//     scale_by_schedule -> scale(-1)
// i.e. per element, at step t (1-based), with s = clip / max(||g||, clip) (1 without clipping):
//     g'  = s g                                          (times grad_scale = 1 / world_size when the all-reduce summed)
//     mu  = bf16( b1 mu + (1 - b1) g' )                  first moment, STORED in bfloat16 (config.optax: mu_dtype='bfloat16')
//     nu  = b2 nu + (1 - b2) g'^2                        second moment, fp32
//     u   = (mu / (1 - b1^t)) / (sqrt(nu / (1 - b2^t)) + eps)
//     p  -= lr_t (u + wd p)                              decoupled decay, scaled by the scheduled learning rate; wd = 0 off the mask
// optax itself is not installed here: the formulas are restated from its published scale_by_adam / add_decayed_weights /
// clip_by_global_norm (optax 0.2.x, optax/_src/transform.py, clipping.py); oracle/optim_ref.py is the numpy restatement the tests
// compare with -- parity unpinned against a run of optax.
//
// Both kernels are one pass over flat fp32 buffers, HBM-bound: 14 bytes read + 10 written per element (p, g, nu fp32; mu bf16).
// The squared gradient norm is a two-stage fixed-order reduction (deterministic) that stays on the device: the update kernel reads
// it through a pointer, so clipping adds no host synchronisation.
#include "common.h"
#include <math.h>

namespace {

constexpr int SUMSQ_BLOCKS = 1024;

__global__ __launch_bounds__(256) void sumsq_partial(const float* __restrict__ g, int64_t n, float* __restrict__ part) {
    __shared__ float red[4];
    float acc = 0.f;
    const int64_t n4 = n >> 2;
    const f32x4_t* g4 = (const f32x4_t*)g;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const f32x4_t v = g4[i];
        acc = fmaf(v[0], v[0], acc); acc = fmaf(v[1], v[1], acc); acc = fmaf(v[2], v[2], acc); acc = fmaf(v[3], v[3], acc);
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) { const float v = g[(n4 << 2) + threadIdx.x]; acc = fmaf(v, v, acc); }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// out[0] = (accumulate ? out[0] : 0) + sum(part[0..nparts))   -- one block, fixed order
__global__ __launch_bounds__(256) void sumsq_final(const float* __restrict__ part, int nparts, float* __restrict__ out, int accumulate) {
    __shared__ float red[4];
    float acc = 0.f;
    for (int i = threadIdx.x; i < nparts; i += 256) acc += part[i];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = (accumulate ? out[0] : 0.f) + ((red[0] + red[1]) + (red[2] + red[3]));
}

struct AdamArgs {
    float* p; const float* g; unsigned short* mu; float* nu;
    int64_t n;
    float lr, b1, b2, eps, wd, bc1, bc2;     // bc = 1 - b^t
    const float* gnorm_sq;                   // device scalar: sum of squares of the UNSCALED gradients (NULL = no clipping)
    float clip, gscale;                      // gscale multiplies every gradient first (1 / world_size after a sum all-reduce)
};

__device__ __forceinline__ void adam_one(float& p, float g, unsigned short& mu, float& nu, const AdamArgs& a, float s) {
    // every product and sum rounded on its own (no FMA contraction): the arithmetic of the restated optax formulas, bit for bit
    g = __fmul_rn(g, s);
    // optax.scale_by_adam: the update is formed from the fp32 moment; only the STORED copy is cast to mu_dtype afterwards
    const float m = __fadd_rn(__fmul_rn(a.b1, bf16_bits_to_f32(mu)), __fmul_rn(1.0f - a.b1, g));
    const float v = __fadd_rn(__fmul_rn(a.b2, nu), __fmul_rn(1.0f - a.b2, __fmul_rn(g, g)));
    mu = f32_to_bf16_bits(m);
    nu = v;
    const float u = __fdiv_rn(__fdiv_rn(m, a.bc1), __fadd_rn(__fsqrt_rn(__fdiv_rn(v, a.bc2)), a.eps));
    p = __fsub_rn(p, __fmul_rn(a.lr, __fadd_rn(u, __fmul_rn(a.wd, p))));
}

__global__ __launch_bounds__(256) void adamw_kernel(const AdamArgs a) {
    float s = a.gscale;
    if (a.gnorm_sq != nullptr) {
        const float norm = sqrtf(*a.gnorm_sq) * a.gscale;
        s *= a.clip / fmaxf(norm, a.clip);         // optax.clip_by_global_norm: g * clip / max(norm, clip); one factor for scale and clip
    }
    const int64_t n4 = a.n >> 2;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const f32x4_t pv = ((f32x4_t*)a.p)[i];
        const f32x4_t g = ((const f32x4_t*)a.g)[i];
        const f32x4_t nv = ((f32x4_t*)a.nu)[i];
        const u32x2_t mw = ((u32x2_t*)a.mu)[i];
        float p[4] = {pv[0], pv[1], pv[2], pv[3]}, nu[4] = {nv[0], nv[1], nv[2], nv[3]};
        unsigned short m[4] = {(unsigned short)(mw[0] & 0xffffu), (unsigned short)(mw[0] >> 16), (unsigned short)(mw[1] & 0xffffu),
                               (unsigned short)(mw[1] >> 16)};
#pragma unroll
        for (int e = 0; e < 4; ++e) adam_one(p[e], g[e], m[e], nu[e], a, s);
        ((f32x4_t*)a.p)[i] = f32x4_t{p[0], p[1], p[2], p[3]};
        ((f32x4_t*)a.nu)[i] = f32x4_t{nu[0], nu[1], nu[2], nu[3]};
        ((u32x2_t*)a.mu)[i] = u32x2_t{(unsigned)m[0] | ((unsigned)m[1] << 16), (unsigned)m[2] | ((unsigned)m[3] << 16)};
    }
    if (blockIdx.x == 0 && threadIdx.x < (a.n & 3)) {
        const int64_t i = (n4 << 2) + threadIdx.x;
        adam_one(a.p[i], a.g[i], a.mu[i], a.nu[i], a, s);
    }
}

}  // namespace

extern "C" size_t ov_sumsq_workspace_bytes(void) { return SUMSQ_BLOCKS * sizeof(float); }

extern "C" int ov_sumsq(const float* g, int64_t n, float* out, int accumulate, void* workspace, size_t workspace_bytes,
                        ov_stream_t stream) {
    if (!g || !out || !workspace || n <= 0) return OV_ERR_INVALID;
    if (((uintptr_t)g | (uintptr_t)workspace) & 15) return OV_ERR_INVALID;
    if (workspace_bytes < ov_sumsq_workspace_bytes()) return OV_ERR_WORKSPACE;
    int64_t want = (n / 4 + 255) / 256;
    const int blocks = (int)(want < 1 ? 1 : (want > SUMSQ_BLOCKS ? SUMSQ_BLOCKS : want));
    hipLaunchKernelGGL(sumsq_partial, dim3(blocks), dim3(256), 0, (hipStream_t)stream, g, n, (float*)workspace);
    OV_LAUNCH_CHECK();
    hipLaunchKernelGGL(sumsq_final, dim3(1), dim3(256), 0, (hipStream_t)stream, (const float*)workspace, blocks, out, accumulate);
    OV_LAUNCH_CHECK();
    return OV_OK;
}

extern "C" int ov_adamw_step(float* p, const float* g, ov_bf16* mu, float* nu, int64_t n, float lr, float b1, float b2, float eps,
                             float wd, int step, float grad_scale, const float* gnorm_sq, float clip_norm, ov_stream_t stream) {
    if (!p || !g || !mu || !nu || n <= 0 || step < 1) return OV_ERR_INVALID;
    if (!(b1 >= 0.f && b1 < 1.f) || !(b2 >= 0.f && b2 < 1.f) || !(eps >= 0.f)) return OV_ERR_INVALID;
    if (gnorm_sq != nullptr && !(clip_norm > 0.f)) return OV_ERR_INVALID;
    if (((uintptr_t)p | (uintptr_t)g | (uintptr_t)nu) & 15 || ((uintptr_t)mu & 7)) return OV_ERR_INVALID;
    AdamArgs a;
    a.p = p; a.g = g; a.mu = mu; a.nu = nu; a.n = n;
    a.lr = lr; a.b1 = b1; a.b2 = b2; a.eps = eps; a.wd = wd;
    a.bc1 = (float)(1.0 - pow((double)b1, (double)step));
    a.bc2 = (float)(1.0 - pow((double)b2, (double)step));
    a.gnorm_sq = gnorm_sq; a.clip = clip_norm; a.gscale = grad_scale;
    int64_t want = (n / 4 + 255) / 256;
    const int blocks = (int)(want < 1 ? 1 : (want > 8192 ? 8192 : want));
    hipLaunchKernelGGL(adamw_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
    OV_LAUNCH_CHECK();
    return OV_OK;
}
