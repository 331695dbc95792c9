// backward.hip — operator-level backward kernels of the block (gfx950): the second piece of SURVEY §8f row 4.
//
// The reference has no backward code of its own: it is torch autograd through nn.Linear / nn.LayerNorm / nn.GELU
// (open_clip/transformer.py:15-30, 232-236).  What autograd computes for those modules is restated here:
//   Linear     y = x W^T + b     ->  dx = dy W,  dW = dy^T x (split over row ranges, bf16 partials, fp32 sum),  db = sum_rows dy
//   LayerNorm  y = xhat g + b    ->  dx = rstd (q - mean(q) - xhat mean(q xhat)), q = dy g;  dg = sum dy xhat;  db = sum dy
//   GELU       h = gelu(a)       ->  da = dh gelu'(a)     (exact erf for the vision tower, tanh form for the text tower)
// The two Linear products run on the tuned forward GEMM (ov_gemm: C = A W^T with K contiguous on both operands), fed by
// bf16 transposes staged through LDS (HBM-bound); everything else is one HBM pass with fp32 arithmetic.  Reductions over
// rows are two-stage with a fixed order (no atomics): results are deterministic.
#include "common.h"

extern "C" int ov_gemm(const ov_bf16* A, int64_t lda, const ov_bf16* W, int64_t ldw, const float* bias, ov_bf16* C, int64_t ldc,
                       int64_t M, int N, int K, int epilogue, const ov_bf16* R, int64_t ldr, int out_group, int resid_mod,
                       int resid_off, ov_stream_t stream);
extern "C" int ov_layernorm(const void* x, int x_dtype, int64_t ldx, const float* gamma, const float* beta, void* y, int y_dtype,
                            int64_t ldy, int64_t rows, int D, float eps, ov_stream_t stream);
extern "C" int ov_attention(const ov_bf16* qkv, int64_t ld_qkv, ov_bf16* out, int64_t ld_out, int B, int L, int H, int hd, float scale,
                            ov_stream_t stream);
extern "C" int ov_attention_backward(const ov_bf16* qkv, int64_t ld_qkv, const ov_bf16* out, int64_t ld_out, const ov_bf16* dout,
                                     int64_t ld_dout, ov_bf16* dqkv, int64_t ld_dqkv, int B, int L, int H, int hd, float scale,
                                     void* workspace, size_t workspace_bytes, ov_stream_t stream);
extern "C" int ov_attention_backward_saved(const ov_bf16* qkv, int64_t ld_qkv, const ov_bf16* out, int64_t ld_out, const ov_bf16* dout,
                                           int64_t ld_dout, ov_bf16* dqkv, int64_t ld_dqkv, const float* lse, int B, int L, int H, int hd,
                                           float scale, void* workspace, size_t workspace_bytes, ov_stream_t stream);
extern "C" size_t ov_attention_backward_workspace_bytes(int B, int L, int H, int hd);
extern "C" int ov_gemm_batched(const ov_bf16* A, int64_t lda, int64_t stride_a, const ov_bf16* W, int64_t ldw, int64_t stride_w,
                               ov_bf16* C, int64_t ldc, int64_t stride_c, int64_t M, int N, int K, int batch, ov_stream_t stream);
extern "C" int ov_gemm_tn_batched(const ov_bf16* P, int64_t ldp, const ov_bf16* Q, int64_t ldq, ov_bf16* C, int64_t ldc, int64_t stride_c,
                                  int64_t Mc, int NI, int NJ, int64_t chunk, int batch, float* psum, ov_stream_t stream);

namespace {

// out[c, r] = in[r, c] for r < R, 0 for R <= r < Rpad (Rpad = R rounded up to 64: the GEMM's K granule).  64 x 64 tiles.
// SUMS: also writes the fp32 column sums of each 64-row tile to colpart[row tile][C] (the bias gradient rides on the pass that
// transposes dY for the weight gradient).
template <bool SUMS>
__global__ __launch_bounds__(256) void transpose_bf16(const unsigned short* __restrict__ in, int64_t ld_in, int64_t R, int C,
                                                      unsigned short* __restrict__ out, int64_t ld_out, float* __restrict__ colpart) {
    __shared__ unsigned short tile[64][66];                       // tile[c][r], 33-dword rows: conflict-free both ways
    const int64_t r0 = (int64_t)blockIdx.x * 64;
    const int c0 = blockIdx.y * 64;
    const int t = threadIdx.x;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int r = (t >> 3) + 32 * h, c8 = (t & 7) * 8;
        u32x4_t w = {0u, 0u, 0u, 0u};
        if (r0 + r < R && c0 + c8 < C) w = *(const u32x4_t*)(in + (r0 + r) * ld_in + c0 + c8);   // C % 8 == 0
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            tile[c8 + 2 * e][r] = (unsigned short)(w[e] & 0xffffu);
            tile[c8 + 2 * e + 1][r] = (unsigned short)(w[e] >> 16);
        }
    }
    __syncthreads();
    if (SUMS) {       // thread (c, quarter) sums 16 rows of column c as eight packed pairs; the quarters meet in LDS
        __shared__ float red[4][64];
        const int c = t & 63, qd = t >> 6;
        const unsigned int* rp = (const unsigned int*)&tile[c][qd * 16];       // 132-byte rows: 4-byte aligned
        float acc = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) { const unsigned int w = rp[i]; acc += bf16lo_to_f32(w); acc += bf16hi_to_f32(w); }
        red[qd][c] = acc;
        __syncthreads();
        if (t < 64 && c0 + t < C) colpart[(int64_t)blockIdx.x * C + c0 + t] = ((red[0][t] + red[1][t]) + red[2][t]) + red[3][t];
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int c = (t >> 3) + 32 * h, r8 = (t & 7) * 8;
        if (c0 + c < C) {
            u32x4_t w;
#pragma unroll
            for (int e = 0; e < 4; ++e) w[e] = (unsigned)tile[c][r8 + 2 * e] | ((unsigned)tile[c][r8 + 2 * e + 1] << 16);
            *(u32x4_t*)(out + (int64_t)(c0 + c) * ld_out + r0 + r8) = w;       // r0 + r8 < Rpad by construction
        }
    }
}

// Column sums of a bf16 [M, N] matrix in fp32: stage 1 sums 256-row chunks (a thread owns two adjacent columns).
constexpr int CS_ROWS = 256;
__global__ __launch_bounds__(256) void colsum_partial(const unsigned int* __restrict__ x, int64_t ld2, int64_t M, int N2,
                                                      float* __restrict__ part) {
    const int c2 = blockIdx.x * 256 + threadIdx.x;                 // column pair
    if (c2 >= N2) return;
    const int64_t m0 = (int64_t)blockIdx.y * CS_ROWS;
    const int64_t m1 = m0 + CS_ROWS < M ? m0 + CS_ROWS : M;
    float s0 = 0.f, s1 = 0.f;
    int64_t m = m0;
    for (; m + 8 <= m1; m += 8) {
        unsigned int w[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) w[u] = x[(m + u) * ld2 + c2];
#pragma unroll
        for (int u = 0; u < 8; ++u) { s0 += bf16lo_to_f32(w[u]); s1 += bf16hi_to_f32(w[u]); }
    }
    for (; m < m1; ++m) {
        const unsigned int w = x[m * ld2 + c2];
        s0 += bf16lo_to_f32(w);
        s1 += bf16hi_to_f32(w);
    }
    float2* p = (float2*)(part + ((int64_t)blockIdx.y * N2 + c2) * 2);
    *p = make_float2(s0, s1);
}
// out[split][n] = sum of partial rows [split * rows_per, (split + 1) * rows_per) (fixed order: wave w takes rows = w mod 4, eight
// loads in flight, then the four waves are combined through LDS).  Block = 64 columns x 4 waves; grid (ceil(N / 64), nsplit).
constexpr int RS_SPLIT = 32;
// out_hi != NULL: single-split launches only; columns >= n_lo go to out_hi[n - n_lo] (two vectors out of one pass: dgamma | dbeta)
__global__ __launch_bounds__(256) void rows_sum(const float* __restrict__ part, int64_t nrow, int N, int64_t stride, int64_t rows_per,
                                                float* __restrict__ out, float* __restrict__ out_hi = nullptr, int n_lo = 0) {
    __shared__ float red[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = blockIdx.x * 64 + lane;
    const int64_t r0 = (int64_t)blockIdx.y * rows_per;
    const int64_t r1 = r0 + rows_per < nrow ? r0 + rows_per : nrow;
    float s = 0.f;
    if (n < N) {
        int64_t r = r0 + wave;
        for (; r + 28 < r1; r += 32) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = part[(r + 4 * u) * stride + n];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; r < r1; r += 4) s += part[r * stride + n];
    }
    red[wave][lane] = s;
    __syncthreads();
    if (wave == 0 && n < N) {
        const float v = ((red[0][lane] + red[1][lane]) + red[2][lane]) + red[3][lane];
        if (out_hi != nullptr && n >= n_lo) out_hi[n - n_lo] = v;
        else out[(int64_t)blockIdx.y * N + n] = v;
    }
}

// column sums of `nrow` partial rows into out[N]; scratch = RS_SPLIT * N floats.  out_hi: the columns from n_lo on go there instead
// (one pair of launches for two vectors that sit side by side in the partial rows)
int launch_rows_sum(const float* part, int64_t nrow, int N, int64_t stride, float* scratch, float* out, hipStream_t st,
                    float* out_hi = nullptr, int n_lo = 0) {
    const unsigned gx = (unsigned)((N + 63) / 64);
    if (nrow <= 4 * RS_SPLIT) {
        hipLaunchKernelGGL(rows_sum, dim3(gx, 1), dim3(256), 0, st, part, nrow, N, stride, nrow, out, out_hi, n_lo);
        OV_LAUNCH_CHECK();
        return OV_OK;
    }
    const int64_t per = (nrow + RS_SPLIT - 1) / RS_SPLIT;
    const int64_t nsplit = (nrow + per - 1) / per;
    hipLaunchKernelGGL(rows_sum, dim3(gx, (unsigned)nsplit), dim3(256), 0, st, part, nrow, N, stride, per, scratch, (float*)nullptr, 0);
    OV_LAUNCH_CHECK();
    hipLaunchKernelGGL(rows_sum, dim3(gx, 1), dim3(256), 0, st, (const float*)scratch, nsplit, N, (int64_t)N, nsplit, out, out_hi, n_lo);
    OV_LAUNCH_CHECK();
    return OV_OK;
}

// out = bf16( sum_z fp32(part[z]) ) over `nz` split-K partials of `total8` 8-element groups each (fixed order)
__global__ __launch_bounds__(256) void splitk_sum(const ov_bf16* __restrict__ part, int64_t stride, int nz, int64_t total8, int K8,
                                                  ov_bf16* __restrict__ out, int64_t ldo) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total8; i += (int64_t)gridDim.x * 256) {
        float acc[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = 0.f;
        for (int z = 0; z < nz; ++z) {
            const u32x4_t w = *(const u32x4_t*)(part + z * stride + i * 8);
#pragma unroll
            for (int e = 0; e < 4; ++e) { acc[2 * e] += bf16lo_to_f32(w[e]); acc[2 * e + 1] += bf16hi_to_f32(w[e]); }
        }
        const int64_t r = i / K8;
        const int c = (int)(i - r * K8) * 8;
        const u32x4_t o = {pack_bf16x2(acc[0], acc[1]), pack_bf16x2(acc[2], acc[3]), pack_bf16x2(acc[4], acc[5]), pack_bf16x2(acc[6], acc[7])};
        *(u32x4_t*)(out + r * ldo + c) = o;
    }
}

// LayerNorm backward, wave per row (row in registers, like the forward).  dgamma / dbeta partial sums stay in the wave's
// registers across the rows it owns and are written once: part[wave_global][2][D].
template <int NCH>
__global__ __launch_bounds__(256) void layernorm_bwd_rows(const ov_bf16* __restrict__ x, int64_t ldx, const float* __restrict__ gamma,
                                                          const ov_bf16* __restrict__ dy, int64_t lddy, const ov_bf16* __restrict__ dres,
                                                          int64_t lddres, ov_bf16* __restrict__ dx, int64_t lddx, int64_t rows, int D,
                                                          float eps, float* __restrict__ part) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nchunk = D >> 3;
    const float invD = 1.0f / (float)D;
    float ag[NCH][8], ab[NCH][8];
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int e = 0; e < 8; ++e) { ag[c][e] = 0.f; ab[c][e] = 0.f; }
    for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < rows; row += (int64_t)gridDim.x * 4) {
        float v[NCH][8], q[NCH][8];
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int ch = lane + c * 64;
            if (ch < nchunk) {
                const u32x4_t w = *(const u32x4_t*)(x + row * ldx + ch * 8);
                const u32x4_t d = *(const u32x4_t*)(dy + row * lddy + ch * 8);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[c][2 * e] = bf16lo_to_f32(w[e]); v[c][2 * e + 1] = bf16hi_to_f32(w[e]);
                    q[c][2 * e] = bf16lo_to_f32(d[e]); q[c][2 * e + 1] = bf16hi_to_f32(d[e]);
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) s += v[c][e];
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) { v[c][e] = 0.f; q[c][e] = 0.f; }
            }
        }
        const float mean = wave_sum(s) * invD;
        float ss = 0.f;
#pragma unroll
        for (int c = 0; c < NCH; ++c)
            if (lane + c * 64 < nchunk)
#pragma unroll
                for (int e = 0; e < 8; ++e) { const float d = v[c][e] - mean; ss += d * d; }
        const float rstd = rsqrtf(wave_sum(ss) * invD + eps);
        float sq = 0.f, sqx = 0.f;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int ch = lane + c * 64;
            if (ch < nchunk) {
                const float4 g0 = *(const float4*)(gamma + ch * 8), g1 = *(const float4*)(gamma + ch * 8 + 4);
                const float g[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float xh = (v[c][e] - mean) * rstd;
                    ag[c][e] = fmaf(q[c][e], xh, ag[c][e]);          // dgamma += dy * xhat
                    ab[c][e] += q[c][e];                             // dbeta  += dy
                    v[c][e] = xh;
                    q[c][e] *= g[e];                                 // q = dy * gamma
                    sq += q[c][e];
                    sqx = fmaf(q[c][e], xh, sqx);
                }
            }
        }
        const float mq = wave_sum(sq) * invD, mqx = wave_sum(sqx) * invD;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int ch = lane + c * 64;
            if (ch < nchunk) {
                float o[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = rstd * (q[c][e] - mq - v[c][e] * mqx);
                if (dres != nullptr) {                               // gradient of the residual branch around the LayerNorm (x + f(LN(x)))
                    const u32x4_t rw = *(const u32x4_t*)(dres + row * lddres + ch * 8);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { o[2 * e] += bf16lo_to_f32(rw[e]); o[2 * e + 1] += bf16hi_to_f32(rw[e]); }
                }
                const u32x4_t w = {pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3]), pack_bf16x2(o[4], o[5]), pack_bf16x2(o[6], o[7])};
                *(u32x4_t*)(dx + row * lddx + ch * 8) = w;
            }
        }
    }
    float* pg = part + ((int64_t)blockIdx.x * 4 + wave) * 2 * D;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int ch = lane + c * 64;
        if (ch < nchunk) {
            *(float4*)(pg + ch * 8) = make_float4(ag[c][0], ag[c][1], ag[c][2], ag[c][3]);
            *(float4*)(pg + ch * 8 + 4) = make_float4(ag[c][4], ag[c][5], ag[c][6], ag[c][7]);
            *(float4*)(pg + D + ch * 8) = make_float4(ab[c][0], ab[c][1], ab[c][2], ab[c][3]);
            *(float4*)(pg + D + ch * 8 + 4) = make_float4(ab[c][4], ab[c][5], ab[c][6], ab[c][7]);
        }
    }
}

// d gelu / d a.  erf form: Phi(a) + a phi(a), Phi through the same A&S 7.1.26 erfc as the forward epilogue's reference form;
// tanh form: 0.5 (1 + t) + 0.5 a (1 - t^2) u'(a), t = tanh(u), u = sqrt(2/pi) (a + 0.044715 a^3).
// d/da gelu(a) and gelu(a) from ONE evaluation of the shared pieces (one v_rcp + one v_exp per element: the kernel is VALU-bound on
// the quarter-rate transcendentals, and the backward needs both the gradient factor and, when the activation is not kept, h itself):
//   erf form : Phi(a) from erfc(|a| / sqrt 2) = poly(t) exp(-a^2 / 2) (A&S 7.1.26),  gelu' = Phi + a phi(a),  gelu = a Phi
//   tanh form: sg = sigmoid(2u), u = sqrt(2/pi) (a + 0.044715 a^3),  gelu' = sg + 2 a sg (1 - sg) u',  gelu = a sg
__device__ __forceinline__ void gelu_erf_both(float a, float& grad, float& h) {
    const float z = fabsf(a) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
    float p = fmaf(t, 1.061405429f, -1.453152027f);
    p = fmaf(t, p, 1.421413741f);
    p = fmaf(t, p, -0.284496736f);
    p = fmaf(t, p, 0.254829592f);
    p *= t;
    const float ex = __builtin_amdgcn_exp2f(-1.4426950408889634f * z * z);      // exp(-a^2 / 2)
    const float half_erfc = 0.5f * p * ex;
    const float Phi = a >= 0.f ? 1.0f - half_erfc : half_erfc;
    grad = fmaf(a * 0.3989422804014327f, ex, Phi);
    h = a * Phi;
}
__device__ __forceinline__ void gelu_tanh_both(float a, float& grad, float& h) {
    const float a2 = a * a;
    const float u = 0.7978845608028654f * a * fmaf(0.044715f, a2, 1.0f);
    const float sg = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-2.8853900817779268f * u));   // sigmoid(2u) = (1 + tanh u) / 2
    const float du = 0.7978845608028654f * fmaf(3.0f * 0.044715f, a2, 1.0f);
    grad = fmaf(2.0f * a * sg * (1.0f - sg), du, sg);              // 0.5 (1 - t^2) = 2 sg (1 - sg)
    h = a * sg;
}

template <bool TANH>
__global__ __launch_bounds__(256) void gelu_bwd(const ov_bf16* a, int64_t lda, const ov_bf16* dh, int64_t lddh, ov_bf16* da, int64_t ldda,
                                                ov_bf16* h_out, int64_t ldh, int64_t rows, int nchunk) {   // da may alias dh, h_out may alias a
    const int64_t total = rows * nchunk;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / nchunk;
        const int c = (int)(i - r * nchunk) * 8;
        const u32x4_t av = *(const u32x4_t*)(a + r * lda + c);
        const u32x4_t dv = *(const u32x4_t*)(dh + r * lddh + c);
        u32x4_t o, ho;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float a0 = bf16lo_to_f32(av[e]), a1 = bf16hi_to_f32(av[e]);
            float g0, g1, h0, h1;
            if (TANH) { gelu_tanh_both(a0, g0, h0); gelu_tanh_both(a1, g1, h1); }
            else { gelu_erf_both(a0, g0, h0); gelu_erf_both(a1, g1, h1); }
            o[e] = pack_bf16x2(bf16lo_to_f32(dv[e]) * g0, bf16hi_to_f32(dv[e]) * g1);
            ho[e] = pack_bf16x2(h0, h1);
        }
        *(u32x4_t*)(da + r * ldda + c) = o;
        if (h_out != nullptr) *(u32x4_t*)(h_out + r * ldh + c) = ho;
    }
}

inline int64_t pad64(int64_t v) { return (v + 63) / 64 * 64; }
inline size_t align256(size_t v) { return (v + 255) / 256 * 256; }

// split-K plan of dW = dY^T X: the output has only (N / 256) (K / 256) tiles, so the M contraction is cut into `nz` ranges of
// `chunk` rows (a multiple of 64) to put ~2 workgroups on every CU; rows past M are zeros written by the transposes
struct SplitK { int nz; int64_t chunk, mp; };
inline SplitK plan_splitk(int64_t M, int N, int K) {
    // The dW kernels run one 256 x 256 tile per workgroup and one workgroup per CU at a time: the launch takes
    // ceil(tiles * nz / CUs) rounds of (rows per range) / 64 K-tiles.  Pick the nz in [1, 32] with the shortest launch (ties: fewer
    // partials, i.e. less fp32 summing): e.g. QKV (48 tiles): nz = 16 -> 768 workgroups = 3 full rounds of 256 CUs, where the former
    // "about 512 workgroups" rule gave 11 ranges = 528 workgroups = two full rounds and a third one for 16 of them.
    const int64_t tiles = (int64_t)((N + 255) / 256) * ((K + 255) / 256);
    const int ncu = ov_num_cus();
    const int64_t max_nz = (M + 511) / 512;                       // at least 8 K-tiles per range
    int64_t best_nz = 1, best_cost = -1;
    for (int64_t nz = 1; nz <= 32 && nz <= (max_nz < 1 ? 1 : max_nz); ++nz) {
        const int64_t chunk = pad64((M + nz - 1) / nz);
        const int64_t real_nz = (M + chunk - 1) / chunk;
        const int64_t rounds = (tiles * real_nz + ncu - 1) / ncu;
        const int64_t cost = rounds * (chunk / 64 + 6) * 16 + real_nz;      // + 6: tile prologue / epilogue in K-tile units
        if (best_cost < 0 || cost < best_cost) { best_cost = cost; best_nz = real_nz; }
    }
    SplitK p;
    p.chunk = pad64((M + best_nz - 1) / best_nz);
    p.nz = (int)((M + p.chunk - 1) / p.chunk);
    p.mp = p.chunk * p.nz;
    return p;
}

// Rpad: multiple of 64 >= R; tiles past R are written as zeros
int launch_transpose(const ov_bf16* in, int64_t ld_in, int64_t R, int64_t Rpad, int C, ov_bf16* out, int64_t ld_out, hipStream_t st,
                     float* colpart = nullptr) {
    const dim3 grid((unsigned)(Rpad / 64), (unsigned)((C + 63) / 64));
    if (colpart)
        hipLaunchKernelGGL(transpose_bf16<true>, grid, dim3(256), 0, st, (const unsigned short*)in, ld_in, R, C, (unsigned short*)out, ld_out, colpart);
    else
        hipLaunchKernelGGL(transpose_bf16<false>, grid, dim3(256), 0, st, (const unsigned short*)in, ld_in, R, C, (unsigned short*)out, ld_out,
                           (float*)nullptr);
    OV_LAUNCH_CHECK();
    return OV_OK;
}

}  // namespace

extern "C" int ov_transpose_bf16(const ov_bf16* in, int64_t ld_in, int64_t rows, int cols, ov_bf16* out, int64_t ld_out,
                                 ov_stream_t stream) {
    if (!in || !out || rows <= 0 || cols <= 0) return OV_ERR_INVALID;
    if (cols % 8 || ld_in % 8 || ld_out % 8 || ld_in < cols || ld_out < pad64(rows)) return OV_ERR_UNSUPPORTED;
    if (((uintptr_t)in | (uintptr_t)out) & 15) return OV_ERR_INVALID;
    if (pad64(rows) / 64 > 0x7fffffffLL) return OV_ERR_UNSUPPORTED;
    return launch_transpose(in, ld_in, rows, pad64(rows), cols, out, ld_out, (hipStream_t)stream);
}

extern "C" size_t ov_linear_backward_workspace_bytes(int64_t M, int N, int K) {
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    const SplitK sp = plan_splitk(M, N, K);
    return align256((size_t)K * N * 2) + align256((size_t)N * sp.mp * 2) + align256((size_t)K * sp.mp * 2) +
           align256((size_t)sp.nz * N * K * 2) + align256((size_t)(sp.mp / 64) * N * 4) +     // >= the M / 256 chunk partials
           align256((size_t)RS_SPLIT * N * 4);
}

// dx_epi / dx_r: the epilogue of the dX product and its second operand [M, K] (ov_block_backward: the GELU derivative at the c_fc
// pre-activation multiplies dy Wproj right there, OV_EPI_GELU_GRAD_*); OV_EPI_BIAS / NULL = plain dX
static int linear_backward(const ov_bf16* dY, int64_t lddy, const ov_bf16* X, int64_t ldx, const ov_bf16* W, int64_t ldw,
                           int64_t M, int N, int K, ov_bf16* dX, int64_t lddx, ov_bf16* dW, int64_t lddw, float* db,
                           void* workspace, size_t workspace_bytes, ov_stream_t stream, int dx_epi, const ov_bf16* dx_r, int64_t lddxr) {
    if (!dY || !W || !workspace || M <= 0 || N <= 0 || K <= 0) return OV_ERR_INVALID;
    if ((!dX && !dW && !db) || (dW && !X)) return OV_ERR_INVALID;
    if (N % 64 || K % 64 || lddy % 8 || ldw % 8 || lddy < N || ldw < K || (X && (ldx % 8 || ldx < K))) return OV_ERR_UNSUPPORTED;
    if ((dX && (lddx % 8 || lddx < K)) || (dW && (lddw % 8 || lddw < K))) return OV_ERR_UNSUPPORTED;
    if (((uintptr_t)dY | (uintptr_t)X | (uintptr_t)W | (uintptr_t)dX | (uintptr_t)dW | (uintptr_t)db | (uintptr_t)workspace) & 15)
        return OV_ERR_INVALID;
    if (workspace_bytes < ov_linear_backward_workspace_bytes(M, N, K)) return OV_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const SplitK sp = plan_splitk(M, N, K);
    const int64_t mp = sp.mp;
    char* ws = (char*)workspace;
    ov_bf16* Wt = (ov_bf16*)ws;  ws += align256((size_t)K * N * 2);          // [K, N]
    ov_bf16* dYt = (ov_bf16*)ws; ws += align256((size_t)N * mp * 2);         // [N, Mpad]
    ov_bf16* Xt = (ov_bf16*)ws;  ws += align256((size_t)K * mp * 2);         // [K, Mpad]
    ov_bf16* dWp = (ov_bf16*)ws; ws += align256((size_t)sp.nz * N * K * 2); // [nz][N, K] split-K partials
    float* part = (float*)ws;
    int rc;
    if (dX) {       // dX[M, K] = dY[M, N] . W[N, K]  =  ov_gemm(A = dY, "W" = W^T [K, N]) contracting over N
        if ((rc = launch_transpose(W, ldw, N, N, K, Wt, N, st)) != OV_OK) return rc;
        if ((rc = ov_gemm(dY, lddy, Wt, N, nullptr, dX, lddx, M, K, N, dx_epi, dx_r, lddxr, 0, 0, 0, stream)) != OV_OK) return rc;
    }
    // dW = dY^T X straight from the row-major operands (transposing LDS reads) when the row count is a multiple of the 64-row
    // K-tile (B * L of every tower is: 257 * 256, 80 * 256, ...); OVHIP_DW_TRANSPOSE=1 forces the explicit-transpose route.  Both
    // routes accumulate the same products in the same order: bitwise the same dW.
    static int force_tr = -1;
    if (force_tr < 0) { const char* e = getenv("OVHIP_DW_TRANSPOSE"); force_tr = (e && e[0] == '1') ? 1 : 0; }
    const bool tn = dW && (M % 64 == 0) && !force_tr;
    if (dW && tn) {
        if (sp.nz == 1) {
            if ((rc = ov_gemm_tn_batched(dY, lddy, X, ldx, dW, lddw, 0, M, N, K, sp.chunk, 1, db, stream)) != OV_OK) return rc;
        } else {
            if ((rc = ov_gemm_tn_batched(dY, lddy, X, ldx, dWp, K, (int64_t)N * K, M, N, K, sp.chunk, sp.nz, db ? part : nullptr, stream)) != OV_OK)
                return rc;
            const int64_t total8 = (int64_t)N * K / 8;
            int64_t blocks = (total8 + 255) / 256;
            if (blocks > 4096) blocks = 4096;
            hipLaunchKernelGGL(splitk_sum, dim3((unsigned)blocks), dim3(256), 0, st, (const ov_bf16*)dWp, (int64_t)N * K, sp.nz, total8, K / 8,
                               dW, lddw);
            OV_LAUNCH_CHECK();
        }
    } else if (dW) {       // dW[N, K] = dY^T[N, M] . X[M, K]  =  ov_gemm(A = dY^T [N, Mpad], "W" = X^T [K, Mpad]) contracting over Mpad (zeros past M)
        if (sp.chunk > 0x7fffffffLL) return OV_ERR_UNSUPPORTED;
        if ((rc = launch_transpose(dY, lddy, M, mp, N, dYt, mp, st, db ? part : nullptr)) != OV_OK) return rc;   // + per-tile column sums
        if ((rc = launch_transpose(X, ldx, M, mp, K, Xt, mp, st)) != OV_OK) return rc;
        if (sp.nz == 1) {
            if ((rc = ov_gemm(dYt, mp, Xt, mp, nullptr, dW, lddw, N, K, (int)mp, OV_EPI_BIAS, nullptr, 0, 0, 0, 0, stream)) != OV_OK) return rc;
        } else {     // range z contracts columns [z chunk, (z + 1) chunk) of both transposes into partial z; then one fp32 sum
            if ((rc = ov_gemm_batched(dYt, mp, sp.chunk, Xt, mp, sp.chunk, dWp, K, (int64_t)N * K, N, K, (int)sp.chunk, sp.nz, stream)) != OV_OK)
                return rc;
            const int64_t total8 = (int64_t)N * K / 8;
            int64_t blocks = (total8 + 255) / 256;
            if (blocks > 4096) blocks = 4096;
            hipLaunchKernelGGL(splitk_sum, dim3((unsigned)blocks), dim3(256), 0, st, (const ov_bf16*)dWp, (int64_t)N * K, sp.nz, total8, K / 8,
                               dW, lddw);
            OV_LAUNCH_CHECK();
        }
    }
    if (db && tn) {         // the column sums of dY came out of the TN kernel, one partial row per split (nz == 1: straight into db)
        if (sp.nz > 1) {
            float* scratch = (float*)((char*)part + align256((size_t)(mp / 64) * N * 4));
            if ((rc = launch_rows_sum(part, sp.nz, N, (int64_t)N, scratch, db, st)) != OV_OK) return rc;
        }
    } else if (db && dW) {  // the column sums of dY came with its transpose: one partial row per 64-row tile
        const int64_t ntile = mp / 64;
        float* scratch = (float*)((char*)part + align256((size_t)ntile * N * 4));
        if ((rc = launch_rows_sum(part, ntile, N, (int64_t)N, scratch, db, st)) != OV_OK) return rc;
    } else if (db) {
        const int64_t nchunk = (M + CS_ROWS - 1) / CS_ROWS;
        if (nchunk > 65535) return OV_ERR_UNSUPPORTED;
        hipLaunchKernelGGL(colsum_partial, dim3((unsigned)((N / 2 + 255) / 256), (unsigned)nchunk), dim3(256), 0, st,
                           (const unsigned int*)dY, lddy / 2, M, N / 2, part);
        OV_LAUNCH_CHECK();
        float* scratch = (float*)((char*)part + align256((size_t)(mp / 64) * N * 4));
        if ((rc = launch_rows_sum(part, nchunk, N, (int64_t)N, scratch, db, st)) != OV_OK) return rc;
    }
    return OV_OK;
}

extern "C" int ov_linear_backward(const ov_bf16* dY, int64_t lddy, const ov_bf16* X, int64_t ldx, const ov_bf16* W, int64_t ldw,
                                  int64_t M, int N, int K, ov_bf16* dX, int64_t lddx, ov_bf16* dW, int64_t lddw, float* db,
                                  void* workspace, size_t workspace_bytes, ov_stream_t stream) {
    return linear_backward(dY, lddy, X, ldx, W, ldw, M, N, K, dX, lddx, dW, lddw, db, workspace, workspace_bytes, stream, OV_EPI_BIAS, nullptr, 0);
}

namespace {
constexpr int LNB_BLOCKS = 1024;
inline int64_t lnb_blocks(int64_t rows) { const int64_t b = (rows + 3) / 4; return b < LNB_BLOCKS ? b : LNB_BLOCKS; }
}

extern "C" size_t ov_layernorm_backward_workspace_bytes(int64_t rows, int D) {
    if (rows <= 0 || D <= 0) return 0;
    return (size_t)lnb_blocks(rows) * 4 * 2 * D * sizeof(float) + (size_t)RS_SPLIT * 2 * D * sizeof(float);
}

extern "C" int ov_layernorm_backward(const ov_bf16* x, int64_t ldx, const float* gamma, const ov_bf16* dy, int64_t lddy,
                                     const ov_bf16* dres, int64_t lddres, ov_bf16* dx, int64_t lddx, float* dgamma, float* dbeta,
                                     int64_t rows, int D, float eps, void* workspace, size_t workspace_bytes, ov_stream_t stream) {
    if (!x || !gamma || !dy || !dx || !dgamma || !dbeta || !workspace || rows <= 0 || D <= 0) return OV_ERR_INVALID;
    if (D % 8 || D > 4096 || ldx % 8 || lddy % 8 || lddx % 8 || ldx < D || lddy < D || lddx < D) return OV_ERR_UNSUPPORTED;
    if (dres && (lddres % 8 || lddres < D)) return OV_ERR_UNSUPPORTED;
    if (((uintptr_t)x | (uintptr_t)gamma | (uintptr_t)dy | (uintptr_t)dres | (uintptr_t)dx | (uintptr_t)workspace) & 15) return OV_ERR_INVALID;
    if (workspace_bytes < ov_layernorm_backward_workspace_bytes(rows, D)) return OV_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const int64_t blocks = lnb_blocks(rows);
    const dim3 grid((unsigned)blocks), blk(256);
    float* part = (float*)workspace;
    const int nch = (D / 8 + 63) / 64;
    if (nch <= 1) hipLaunchKernelGGL(layernorm_bwd_rows<1>, grid, blk, 0, st, x, ldx, gamma, dy, lddy, dres, lddres, dx, lddx, rows, D, eps, part);
    else if (nch <= 2) hipLaunchKernelGGL(layernorm_bwd_rows<2>, grid, blk, 0, st, x, ldx, gamma, dy, lddy, dres, lddres, dx, lddx, rows, D, eps, part);
    else if (nch <= 3) hipLaunchKernelGGL(layernorm_bwd_rows<3>, grid, blk, 0, st, x, ldx, gamma, dy, lddy, dres, lddres, dx, lddx, rows, D, eps, part);
    else if (nch <= 4) hipLaunchKernelGGL(layernorm_bwd_rows<4>, grid, blk, 0, st, x, ldx, gamma, dy, lddy, dres, lddres, dx, lddx, rows, D, eps, part);
    else hipLaunchKernelGGL(layernorm_bwd_rows<8>, grid, blk, 0, st, x, ldx, gamma, dy, lddy, dres, lddres, dx, lddx, rows, D, eps, part);
    OV_LAUNCH_CHECK();
    // part[w][0][:] = dgamma partial, part[w][1][:] = dbeta partial of wave w: two strided column sums
    float* scratch = part + (size_t)blocks * 4 * 2 * D;
    // part[w] = [dgamma partial | dbeta partial]: both column sums from one pair of launches (the same additions in the same order)
    return launch_rows_sum(part, blocks * 4, 2 * D, (int64_t)2 * D, scratch, dgamma, st, dbeta, D);
}

extern "C" int ov_gelu_backward(const ov_bf16* a, int64_t lda, const ov_bf16* dh, int64_t lddh, ov_bf16* da, int64_t ldda, ov_bf16* h_out,
                                int64_t ldh, int64_t rows, int N, int tanh_form, ov_stream_t stream) {
    if (!a || !dh || !da || rows <= 0 || N <= 0) return OV_ERR_INVALID;
    if (N % 8 || lda % 8 || lddh % 8 || ldda % 8 || lda < N || lddh < N || ldda < N) return OV_ERR_UNSUPPORTED;
    if (h_out && (ldh % 8 || ldh < N)) return OV_ERR_UNSUPPORTED;
    if (((uintptr_t)a | (uintptr_t)dh | (uintptr_t)da | (uintptr_t)h_out) & 15) return OV_ERR_INVALID;
    const int64_t total = rows * (N / 8);
    int64_t blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipStream_t st = (hipStream_t)stream;
    if (tanh_form) hipLaunchKernelGGL(gelu_bwd<true>, dim3((unsigned)blocks), dim3(256), 0, st, a, lda, dh, lddh, da, ldda, h_out, ldh, rows, N / 8);
    else hipLaunchKernelGGL(gelu_bwd<false>, dim3((unsigned)blocks), dim3(256), 0, st, a, lda, dh, lddh, da, ldda, h_out, ldh, rows, N / 8);
    OV_LAUNCH_CHECK();
    return OV_OK;
}

// ---- one ResidualAttentionBlock (transformer.py:254-265: x1 = x + attn(ln_1(x)); y = x1 + mlp(ln_2(x1))) ----------------------
// Activation recomputation: only the block input x is kept by the caller; ln_1, qkv, attention, x1, ln_2 and the c_fc
// pre-activation are recomputed here with the forward's own kernels, then the chain rule runs back through the operators above.
namespace {
struct BlockBufs { ov_bf16 *n1, *qkv, *o, *x1, *n2, *a, *dh, *t1, *dx1, *dqkv; char* lin; char* ln; char* att; size_t lin_bytes, ln_bytes, att_bytes, total; };
inline BlockBufs plan_block(const ov_tower_cfg* c, int B, int L, char* base) {
    const int64_t M = (int64_t)B * L;
    const int D = c->width, F = c->mlp_pad;
    BlockBufs b;
    size_t off = 0;
    auto take = [&](size_t bytes) { char* p = base ? base + off : nullptr; off += align256(bytes); return p; };
    b.n1 = (ov_bf16*)take((size_t)M * D * 2);   b.qkv = (ov_bf16*)take((size_t)M * 3 * D * 2);  b.o = (ov_bf16*)take((size_t)M * D * 2);
    b.x1 = (ov_bf16*)take((size_t)M * D * 2);   b.n2 = (ov_bf16*)take((size_t)M * D * 2);       b.a = (ov_bf16*)take((size_t)M * F * 2);
    b.dh = (ov_bf16*)take((size_t)M * F * 2);   b.t1 = (ov_bf16*)take((size_t)M * D * 2);       b.dx1 = (ov_bf16*)take((size_t)M * D * 2);
    b.dqkv = (ov_bf16*)take((size_t)M * 3 * D * 2);
    size_t lb = ov_linear_backward_workspace_bytes(M, 3 * D, D);
    const size_t l2 = ov_linear_backward_workspace_bytes(M, D, D), l3 = ov_linear_backward_workspace_bytes(M, F, D),
                 l4 = ov_linear_backward_workspace_bytes(M, D, F);
    lb = lb > l2 ? lb : l2; lb = lb > l3 ? lb : l3; lb = lb > l4 ? lb : l4;
    b.lin_bytes = lb; b.lin = take(lb);
    b.ln_bytes = ov_layernorm_backward_workspace_bytes(M, D); b.ln = take(b.ln_bytes);
    b.att_bytes = ov_attention_backward_workspace_bytes(B, L, c->heads, c->width / c->heads); b.att = take(b.att_bytes + 256);
    b.total = off;
    return b;
}
inline bool block_cfg_ok(const ov_tower_cfg* c) {
    if (!c || c->width <= 0 || c->heads <= 0 || c->width % 64 || c->width % c->heads || c->width > 4096) return false;
    const int hd = c->width / c->heads;
    return hd % 8 == 0 && hd <= 96 && c->mlp > 0 && c->mlp <= c->mlp_pad && c->mlp_pad % 64 == 0;
}
}  // namespace

extern "C" size_t ov_block_backward_workspace_bytes(const ov_tower_cfg* cfg, int B, int L) {
    if (!block_cfg_ok(cfg) || B <= 0 || L <= 0) return 0;
    return plan_block(cfg, B, L, nullptr).total;
}

extern "C" int ov_block_backward(const ov_tower_cfg* cfg, const ov_block_weights* w, const ov_bf16* x, const ov_block_saved* saved,
                                 const ov_bf16* dy, ov_bf16* dx, const ov_block_grads* g, int B, int L, void* workspace,
                                 size_t workspace_bytes, ov_stream_t stream) {
    if (!cfg || !w || !x || !dy || !dx || !g || !workspace || B <= 0 || L <= 0) return OV_ERR_INVALID;
    if (saved && (!saved->qkv || !saved->attn_out || !saved->x1)) return OV_ERR_INVALID;
    if (!block_cfg_ok(cfg)) return OV_ERR_UNSUPPORTED;                        // head_dim % 8 == 0 and <= 96, width % 64 == 0
    if (w->qkv_colsum || w->fc_colsum) return OV_ERR_UNSUPPORTED;             // needs the module's own (unfolded) weights
    if (!w->ln1_w || !w->ln1_b || !w->qkv_w || !w->qkv_b || !w->out_w || !w->out_b || !w->ln2_w || !w->ln2_b || !w->fc_w || !w->fc_b ||
        !w->proj_w || !w->proj_b)
        return OV_ERR_INVALID;
    if (!g->ln1_w || !g->ln1_b || !g->qkv_w || !g->qkv_b || !g->out_w || !g->out_b || !g->ln2_w || !g->ln2_b || !g->fc_w || !g->fc_b ||
        !g->proj_w || !g->proj_b)
        return OV_ERR_INVALID;
    const int D = cfg->width, F = cfg->mlp_pad, H = cfg->heads;
    const int64_t M = (int64_t)B * L;
    if (workspace_bytes < ov_block_backward_workspace_bytes(cfg, B, L)) return OV_ERR_WORKSPACE;
    if (((uintptr_t)workspace | (uintptr_t)x | (uintptr_t)dy | (uintptr_t)dx) & 15) return OV_ERR_INVALID;
    BlockBufs b = plan_block(cfg, B, L, (char*)workspace);
    const int hd = D / H;
    const float eps = cfg->ln_eps, scale = 1.0f / sqrtf((float)hd);
    int rc;
#define OV_TRY(call) do { if ((rc = (call)) != OV_OK) return rc; } while (0)
    // ---- the forward's intermediates: kept by ov_tower_forward_saving or recomputed here
    if (saved && saved->ln1_out) b.n1 = const_cast<ov_bf16*>(saved->ln1_out);
    else OV_TRY(ov_layernorm(x, OV_BF16, D, w->ln1_w, w->ln1_b, b.n1, OV_BF16, D, M, D, eps, stream));
    if (saved) {
        b.qkv = const_cast<ov_bf16*>(saved->qkv); b.o = const_cast<ov_bf16*>(saved->attn_out); b.x1 = const_cast<ov_bf16*>(saved->x1);
    } else {
        OV_TRY(ov_gemm(b.n1, D, w->qkv_w, D, w->qkv_b, b.qkv, 3 * D, M, 3 * D, D, OV_EPI_BIAS, nullptr, 0, 0, 0, 0, stream));
        OV_TRY(ov_attention(b.qkv, 3 * D, b.o, D, B, L, H, hd, scale, stream));
        OV_TRY(ov_gemm(b.o, D, w->out_w, D, w->out_b, b.x1, D, M, D, D, OV_EPI_BIAS_RESIDUAL, x, D, 0, 0, 0, stream));
    }
    if (saved && saved->ln2_out) b.n2 = const_cast<ov_bf16*>(saved->ln2_out);
    else OV_TRY(ov_layernorm(b.x1, OV_BF16, D, w->ln2_w, w->ln2_b, b.n2, OV_BF16, D, M, D, eps, stream));
    const ov_bf16* pre = (saved && saved->fc_pre) ? saved->fc_pre : b.a;
    if (pre == b.a) OV_TRY(ov_gemm(b.n2, D, w->fc_w, D, w->fc_b, b.a, F, M, F, D, OV_EPI_BIAS, nullptr, 0, 0, 0, 0, stream));
    // ---- MLP branch: y = x1 + c_proj(gelu(a))
    const ov_bf16* act = b.a;
    if (saved && saved->fc_act && saved->fc_pre) {      // da = (dy Wproj) * gelu'(a) in the product's own epilogue; gelu(a) was kept
        OV_TRY(linear_backward(dy, D, nullptr, 0, w->proj_w, F, M, D, F, b.dh, F, nullptr, 0, nullptr, b.lin, b.lin_bytes, stream,
                               cfg->gelu_tanh ? OV_EPI_GELU_GRAD_TANH : OV_EPI_GELU_GRAD_ERF, pre, F));
        act = saved->fc_act;
    } else {
        OV_TRY(ov_linear_backward(dy, D, nullptr, 0, w->proj_w, F, M, D, F, b.dh, F, nullptr, 0, nullptr, b.lin, b.lin_bytes, stream));  // dh = dy Wproj
        OV_TRY(ov_gelu_backward(pre, F, b.dh, F, b.dh, F, b.a, F, M, F, cfg->gelu_tanh, stream));                                      // dh -> da (in place), b.a = gelu(a)
    }
    OV_TRY(ov_linear_backward(dy, D, act, F, w->proj_w, F, M, D, F, nullptr, 0, g->proj_w, F, g->proj_b, b.lin, b.lin_bytes, stream));
    OV_TRY(ov_linear_backward(b.dh, F, b.n2, D, w->fc_w, D, M, F, D, b.t1, D, g->fc_w, D, g->fc_b, b.lin, b.lin_bytes, stream));        // t1 = d ln_2 out
    OV_TRY(ov_layernorm_backward(b.x1, D, w->ln2_w, b.t1, D, dy, D, b.dx1, D, g->ln2_w, g->ln2_b, M, D, eps, b.ln, b.ln_bytes, stream)); // dx1 = dy + ...
    // ---- attention branch: x1 = x + out_proj(attn(qkv))
    OV_TRY(ov_linear_backward(b.dx1, D, b.o, D, w->out_w, D, M, D, D, b.t1, D, g->out_w, D, g->out_b, b.lin, b.lin_bytes, stream));     // t1 = d attention out
    OV_TRY(ov_attention_backward_saved(b.qkv, 3 * D, b.o, D, b.t1, D, b.dqkv, 3 * D, saved ? saved->attn_lse : nullptr, B, L, H, hd, scale, b.att,
                                       b.att_bytes, stream));
    OV_TRY(ov_linear_backward(b.dqkv, 3 * D, b.n1, D, w->qkv_w, D, M, 3 * D, D, b.t1, D, g->qkv_w, D, g->qkv_b, b.lin, b.lin_bytes, stream));  // t1 = d ln_1 out
    OV_TRY(ov_layernorm_backward(x, D, w->ln1_w, b.t1, D, b.dx1, D, dx, D, g->ln1_w, g->ln1_b, M, D, eps, b.ln, b.ln_bytes, stream));
#undef OV_TRY
    return OV_OK;
}
