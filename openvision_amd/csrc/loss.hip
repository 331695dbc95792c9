// loss.hip — fused CLIP InfoNCE on local logit strips (gfx950).
//
// Replaces ClipLoss.get_logits + F.cross_entropy both ways (reference open_clip/loss.py:102-131, local_loss
// branch :108-110, labels arange(b) + b*rank :93-94; JAX twin src/losses/common.py:120-189):
//     loss = ( CE(s * img @ all_txt^T, i + off) + CE(s * txt @ all_img^T, i + off) ) / 2
// The [b, N] logit strips are never written: each 32x32 logit tile lives in one f32x16 MFMA accumulator
// (exact-fp32 v_mfma_f32_32x32x2_f32, k-ordered fmaf chain), and only a running (max, sum-exp) pair per
// local row plus the diagonal logit leave the kernel.  The gathered side is the MFMA A operand, so a lane
// holds 16 of a tile's 32 gathered entries for ONE local row: the row reductions are lane-local.
// Two launches, no atomics, deterministic: partials per (direction, column split, row) -> finalize.
#include "common.h"

namespace {

struct LossArgs {
    const float* x[2];      // local rows  [b, E]   (dir 0: img, dir 1: txt)
    const float* y[2];      // gathered    [N, E]   (dir 0: all_txt, dir 1: all_img)
    float* part;            // [2][nsplit][bpad][2]  (max, sumexp) in natural-log units
    float* diag;            // [2][bpad]
    int b, N, E, bpad, nsplit, tiles_per_split, ntiles, label_offset;
    const float* scale;     // device scalar: the logit multiplier exp(logit_scale) (ABI 2: never crosses the host)
};

__global__ __launch_bounds__(256) void clip_logits_partial(const LossArgs a) {
    __shared__ float red[4][32][2];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 31, half = lane >> 5;
    const int split = blockIdx.x, rt = blockIdx.y, dir = blockIdx.z;
    const float* __restrict__ X = a.x[dir];
    const float* __restrict__ Y = a.y[dir];
    const int row = rt * 32 + j;
    const int rowc = row < a.b ? row : a.b - 1;
    const float* xp = X + (int64_t)rowc * a.E + 4 * half;
    const int label = row + a.label_offset;
    const float scale = *a.scale;

    float m = -INFINITY, s = 0.f;
    const int t0 = split * a.tiles_per_split;
    int t1 = t0 + a.tiles_per_split;
    if (t1 > a.ntiles) t1 = a.ntiles;
    for (int t = t0 + wave; t < t1; t += 4) {
        int gi = t * 32 + j;
        gi = gi < a.N ? gi : a.N - 1;
        const float* yp = Y + (int64_t)gi * a.E + 4 * half;
        f32x16_t acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll 4
        for (int k0 = 0; k0 < a.E; k0 += 8) {
            const float4 av = *(const float4*)(yp + k0);
            const float4 bv = *(const float4*)(xp + k0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bv.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bv.w, acc, 0, 0, 0);
        }
        // acc[i] = <Y[t*32 + (i&3) + 8*(i>>2) + 4*half], X[row]>
        float mx = -INFINITY;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int g = t * 32 + (i & 3) + 8 * (i >> 2) + 4 * half;
            float v = acc[i] * scale;
            if (g == label && row < a.b) a.diag[dir * a.bpad + row] = v;
            if (g >= a.N) v = -INFINITY;
            acc[i] = v;
            mx = fmaxf(mx, v);
        }
        if (mx > -INFINITY) {
            const float mn = fmaxf(m, mx);
            float ps = 0.f;
#pragma unroll
            for (int i = 0; i < 16; ++i) ps += __expf(acc[i] - mn);
            s = s * __expf(m - mn) + ps;
            m = mn;
        }
    }
    // combine the two lane halves of each row, then the four waves
    {
        const float mo = __shfl_xor(m, 32, 64), so = __shfl_xor(s, 32, 64);
        const float mn = fmaxf(m, mo);
        if (mn > -INFINITY) s = s * __expf(m - mn) + so * __expf(mo - mn);
        m = mn;
    }
    if (half == 0) { red[wave][j][0] = m; red[wave][j][1] = s; }
    __syncthreads();
    if (wave == 0 && half == 0 && row < a.b) {
        float M = red[0][j][0], S = red[0][j][1];
#pragma unroll
        for (int w = 1; w < 4; ++w) {
            const float mw = red[w][j][0], sw = red[w][j][1];
            const float mn = fmaxf(M, mw);
            if (mn > -INFINITY) S = S * __expf(M - mn) + sw * __expf(mw - mn);
            M = mn;
        }
        float* p = a.part + (((int64_t)dir * a.nsplit + split) * a.bpad + row) * 2;
        p[0] = M; p[1] = S;
    }
}

__global__ __launch_bounds__(256) void clip_loss_finalize(const float* __restrict__ part, const float* __restrict__ diag,
                                                          int b, int bpad, int nsplit, float* __restrict__ loss_out,
                                                          float* __restrict__ lse_out) {
    __shared__ float red[4];
    float local = 0.f;
    for (int i = threadIdx.x; i < 2 * b; i += blockDim.x) {
        const int dir = i / b, row = i - dir * b;
        float M = -INFINITY, S = 0.f;
        for (int sp = 0; sp < nsplit; ++sp) {
            const float* p = part + (((int64_t)dir * nsplit + sp) * bpad + row) * 2;
            const float mw = p[0], sw = p[1];
            const float mn = fmaxf(M, mw);
            if (mn > -INFINITY) S = S * __expf(M - mn) + sw * __expf(mw - mn);
            M = mn;
        }
        const float lse = M + logf(S);
        const float d = diag[dir * bpad + row];
        if (lse_out) {
            lse_out[(2 * dir) * b + row] = lse;
            lse_out[(2 * dir + 1) * b + row] = d;
        }
        local += lse - d;
    }
    local = wave_sum(local);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = local;
    __syncthreads();
    if (threadIdx.x == 0) loss_out[0] = (red[0] + red[1] + red[2] + red[3]) / (2.0f * (float)b);
}

// out[i, j] = scale * <X[i, :], Y[j, :]>  (CLIP.get_logits, model.py:286-293; fp32-exact MFMA). One wave per 32x32 tile.
__global__ __launch_bounds__(64) void logits_kernel(const float* __restrict__ X, const float* __restrict__ Y,
                                                    float* __restrict__ out, int64_t ldo, int n1, int n2, int E,
                                                    float scale, const float* __restrict__ scale_dev) {
    if (scale_dev) scale *= *scale_dev;
    const int lane = threadIdx.x & 63;
    const int j = lane & 31, half = lane >> 5;
    const int i0 = blockIdx.y * 32, j0 = blockIdx.x * 32;
    const int xi = (i0 + j) < n1 ? (i0 + j) : n1 - 1;
    const int yj = (j0 + j) < n2 ? (j0 + j) : n2 - 1;
    const float* xp = X + (int64_t)xi * E + 4 * half;
    const float* yp = Y + (int64_t)yj * E + 4 * half;
    f32x16_t acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    for (int k0 = 0; k0 < E; k0 += 8) {
        const float4 av = *(const float4*)(yp + k0);
        const float4 bv = *(const float4*)(xp + k0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bv.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bv.w, acc, 0, 0, 0);
    }
    const int row = i0 + j;
    if (row < n1) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int col = j0 + (i & 3) + 8 * (i >> 2) + 4 * half;
            if (col < n2) out[(int64_t)row * ldo + col] = acc[i] * scale;
        }
    }
}


// ---- backward of the InfoNCE strips (first piece of the training path: loss.py:102-131 differentiated) -----------------------
// d loss / d logits = (softmax - onehot) / (2 b) per direction; the logit tiles are recomputed with the same exact-fp32 MFMA
// (never materialised), P = exp(s * S - lse[row]) - [label] is formed in registers from the forward's per-row LSE, and the
// product P . X_in accumulates into [32 rows x E] fp32 MFMA accumulators split over the four waves of a workgroup by 32-column
// e-tile.  The S tile's K reduction is split the same way (each wave contracts its own e-tiles) and summed through LDS.
//   MODE_B = false: out rows = LOCAL rows (lse by out row):     d x_local[r]  = c * sum_g P[r, g] * y_all[g]
//   MODE_B = true : out rows = GATHERED rows (lse by in row):   d y_all[g]    = c * sum_r P[r, g] * x_local[r]
// One workgroup per (32-row tile, direction); the in-side loop is not split, so the result is deterministic (no atomics).
constexpr int BWD_MAXT = 9;          // e-tiles per wave: E <= 4 * 9 * 32 = 1152

struct LossBwdArgs {
    const float* xo[2];     // out-side rows [no, E]
    const float* xi[2];     // in-side rows  [ni, E]
    const float* lse[2];    // per LOCAL row
    float* out[2];          // [no, E]
    float* dsc_part;        // [2][nrt]  (MODE_B = false only)
    int no, ni, E, label_offset, nrt;
    const float* scale;     // device scalars (ABI 2): logit multiplier, upstream gradient of the loss (NULL = 1)
    const float* grad;
    float inv2b;
};

template <bool MODE_B>
__global__ __launch_bounds__(256) void clip_loss_bwd(const LossBwdArgs a) {
    __shared__ float part[4][16][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 31, half = lane >> 5;
    const int rt = blockIdx.x, dir = blockIdx.y;
    const float* __restrict__ XO = a.xo[dir];
    const float* __restrict__ XI = a.xi[dir];
    const float* __restrict__ LSE = a.lse[dir];
    float* __restrict__ OUT = a.out[dir];
    if (OUT == nullptr) return;                                   // direction not requested (workgroup-uniform)
    const int E = a.E, net = E >> 5;
    const int nown = (net - wave + 3) >> 2;                       // e-tiles wave, wave + 4, ...
    const int o = rt * 32 + j;
    const int oc = o < a.no ? o : a.no - 1;
    const float* xop = XO + (int64_t)oc * E + 4 * half;
    const float lse_o = MODE_B ? 0.f : LSE[oc];
    const float scale = *a.scale;
    const float coef = (a.grad ? *a.grad : 1.f) * a.inv2b * scale;

    f32x16_t acc_o[BWD_MAXT];
#pragma unroll
    for (int n = 0; n < BWD_MAXT; ++n)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc_o[n][i] = 0.f;
    float dsc = 0.f;

    const int ntiles = (a.ni + 31) >> 5;
    for (int t = 0; t < ntiles; ++t) {
        int gi = t * 32 + j;
        gi = gi < a.ni ? gi : a.ni - 1;
        const float* yip = XI + (int64_t)gi * E + 4 * half;
        f32x16_t acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
        for (int n = 0; n < nown; ++n) {
            const int e0 = (wave + 4 * n) * 32;
#pragma unroll
            for (int k0 = 0; k0 < 32; k0 += 8) {
                const float4 av = *(const float4*)(yip + e0 + k0);
                const float4 bv = *(const float4*)(xop + e0 + k0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv.x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv.y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bv.z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bv.w, acc, 0, 0, 0);
            }
        }
        __syncthreads();                                          // the previous tile's partials have been consumed
#pragma unroll
        for (int i = 0; i < 16; ++i) part[wave][i][lane] = acc[i];
        __syncthreads();
        // acc[i] = <XI[t*32 + (i&3) + 8*(i>>2) + 4*half], XO[o]>, summed over the waves in a fixed order
        f32x16_t p;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const float sdot = ((part[0][i][lane] + part[1][i][lane]) + part[2][i][lane]) + part[3][i][lane];
            const int g = t * 32 + (i & 3) + 8 * (i >> 2) + 4 * half;
            const bool valid = o < a.no && g < a.ni;
            const float lse_v = MODE_B ? LSE[g < a.ni ? g : a.ni - 1] : lse_o;
            const bool hit = MODE_B ? (o == g + a.label_offset) : (g == o + a.label_offset);
            const float pv = valid ? __expf(sdot * scale - lse_v) - (hit ? 1.f : 0.f) : 0.f;
            p[i] = pv;
            dsc = fmaf(pv, sdot, dsc);
        }
        // out[o, e] += sum_g P[o, g] * XI[g, e]: contraction step s pairs g0(s) = (s&3) + 8*(s>>2) (k = 0, held by the lower lane
        // half as register s) with g0(s) + 4 (k = 1, upper half): the A operand is this lane's own p[s]
#pragma unroll
        for (int n = 0; n < BWD_MAXT; ++n) {
            if (n < nown) {
                const int e = (wave + 4 * n) * 32 + j;
#pragma unroll
                for (int s2 = 0; s2 < 16; ++s2) {
                    int g = t * 32 + (s2 & 3) + 8 * (s2 >> 2) + 4 * half;
                    g = g < a.ni ? g : a.ni - 1;
                    const float yv = XI[(int64_t)g * E + e];
                    acc_o[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(p[s2], yv, acc_o[n], 0, 0, 0);
                }
            }
        }
    }
#pragma unroll
    for (int n = 0; n < BWD_MAXT; ++n) {
        if (n < nown) {
            const int e = (wave + 4 * n) * 32 + j;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int row = rt * 32 + (i & 3) + 8 * (i >> 2) + 4 * half;
                if (row < a.no) OUT[(int64_t)row * E + e] = acc_o[n][i] * coef;
            }
        }
    }
    if (!MODE_B) {                                                // d loss / d scale: every wave holds the same P; wave 0 reports
        dsc = wave_sum(dsc);
        if (wave == 0 && lane == 0) a.dsc_part[dir * a.nrt + rt] = dsc;
    }
}

__global__ __launch_bounds__(64) void clip_loss_bwd_scale(const float* __restrict__ part, int n, float inv2b,
                                                          const float* __restrict__ grad, float* __restrict__ d_scale) {
    float v = 0.f;
    for (int i = threadIdx.x; i < n; i += 64) v += part[i];
    v = wave_sum(v);
    if (threadIdx.x == 0) d_scale[0] = v * inv2b * (grad ? *grad : 1.f);
}

struct Plan { int bpad, nrt, ntiles, nsplit, tps; };

inline Plan make_plan(int b, int N) {
    Plan p;
    p.nrt = (b + 31) / 32;
    p.bpad = p.nrt * 32;
    p.ntiles = (N + 31) / 32;
    int want = 1024 / (2 * p.nrt);
    if (want < 1) want = 1;
    int maxsplit = (p.ntiles + 3) / 4;
    if (maxsplit < 1) maxsplit = 1;
    p.nsplit = want < maxsplit ? want : maxsplit;
    p.tps = (p.ntiles + p.nsplit - 1) / p.nsplit;
    p.nsplit = (p.ntiles + p.tps - 1) / p.tps;
    return p;
}

}  // namespace

extern "C" size_t ov_clip_loss_workspace_bytes(int b, int N) {
    if (b <= 0 || N <= 0) return 0;
    const Plan p = make_plan(b, N);
    return ((size_t)2 * p.nsplit * p.bpad * 2 + (size_t)2 * p.bpad) * sizeof(float);
}

extern "C" int ov_clip_loss(const float* img, const float* txt, const float* all_img, const float* all_txt, int b,
                            int N, int E, const float* logit_scale, int label_offset, float* loss_out, float* lse_out,
                            void* workspace, size_t workspace_bytes, ov_stream_t stream) {
    if (!img || !txt || !all_img || !all_txt || !loss_out || !workspace || !logit_scale) return OV_ERR_INVALID;
    if (b <= 0 || N < b || E <= 0 || label_offset < 0 || label_offset + b > N) return OV_ERR_INVALID;
    if (E % 8) return OV_ERR_UNSUPPORTED;
    if (((uintptr_t)img | (uintptr_t)txt | (uintptr_t)all_img | (uintptr_t)all_txt | (uintptr_t)workspace) & 15)
        return OV_ERR_INVALID;
    if (workspace_bytes < ov_clip_loss_workspace_bytes(b, N)) return OV_ERR_WORKSPACE;
    const Plan p = make_plan(b, N);
    LossArgs a;
    a.x[0] = img; a.y[0] = all_txt;
    a.x[1] = txt; a.y[1] = all_img;
    a.part = (float*)workspace;
    a.diag = a.part + (size_t)2 * p.nsplit * p.bpad * 2;
    a.b = b; a.N = N; a.E = E; a.bpad = p.bpad; a.nsplit = p.nsplit; a.tiles_per_split = p.tps; a.ntiles = p.ntiles;
    a.label_offset = label_offset; a.scale = logit_scale;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(clip_logits_partial, dim3((unsigned)p.nsplit, (unsigned)p.nrt, 2), dim3(256), 0, st, a);
    OV_LAUNCH_CHECK();
    hipLaunchKernelGGL(clip_loss_finalize, dim3(1), dim3(256), 0, st, a.part, a.diag, b, p.bpad, p.nsplit, loss_out, lse_out);
    OV_LAUNCH_CHECK();
    return OV_OK;
}


extern "C" size_t ov_clip_loss_backward_workspace_bytes(int b, int N) {
    if (b <= 0 || N <= 0) return 0;
    return (size_t)2 * ((b + 31) / 32) * sizeof(float) + 64;
}

extern "C" int ov_clip_loss_backward(const float* img, const float* txt, const float* all_img, const float* all_txt, int b, int N,
                                     int E, const float* logit_scale, int label_offset, const float* lse_terms,
                                     const float* grad_loss, float* d_img, float* d_txt, float* d_all_img, float* d_all_txt,
                                     float* d_scale, void* workspace, size_t workspace_bytes, ov_stream_t stream) {
    if (!img || !txt || !all_img || !all_txt || !lse_terms || !d_img || !d_txt || !workspace || !logit_scale) return OV_ERR_INVALID;
    if (b <= 0 || N < b || E <= 0 || label_offset < 0 || label_offset + b > N) return OV_ERR_INVALID;
    if (E % 32 || E > 4 * BWD_MAXT * 32) return OV_ERR_UNSUPPORTED;
    if (((uintptr_t)img | (uintptr_t)txt | (uintptr_t)all_img | (uintptr_t)all_txt | (uintptr_t)workspace) & 15) return OV_ERR_INVALID;
    if (workspace_bytes < ov_clip_loss_backward_workspace_bytes(b, N)) return OV_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const float inv2b = 1.0f / (2.0f * (float)b);
    LossBwdArgs a;
    a.E = E; a.label_offset = label_offset; a.scale = logit_scale; a.grad = grad_loss; a.inv2b = inv2b;
    a.lse[0] = lse_terms; a.lse[1] = lse_terms + (size_t)2 * b;
    a.dsc_part = (float*)workspace;
    // local side: d img = c * P_i . all_txt, d txt = c * P_t . all_img
    a.xo[0] = img; a.xi[0] = all_txt; a.out[0] = d_img;
    a.xo[1] = txt; a.xi[1] = all_img; a.out[1] = d_txt;
    a.no = b; a.ni = N; a.nrt = (b + 31) / 32;
    hipLaunchKernelGGL(clip_loss_bwd<false>, dim3((unsigned)a.nrt, 2), dim3(256), 0, st, a);
    OV_LAUNCH_CHECK();
    if (d_scale) {
        hipLaunchKernelGGL(clip_loss_bwd_scale, dim3(1), dim3(64), 0, st, a.dsc_part, 2 * a.nrt, inv2b, grad_loss, d_scale);
        OV_LAUNCH_CHECK();
    }
    if (d_all_img || d_all_txt) {
        // gathered side: d all_txt = c * P_i^T . img (direction 0), d all_img = c * P_t^T . txt (direction 1)
        a.xo[0] = all_txt; a.xi[0] = img; a.out[0] = d_all_txt;
        a.xo[1] = all_img; a.xi[1] = txt; a.out[1] = d_all_img;
        a.no = N; a.ni = b; a.nrt = (N + 31) / 32;
        hipLaunchKernelGGL(clip_loss_bwd<true>, dim3((unsigned)a.nrt, 2), dim3(256), 0, st, a);
        OV_LAUNCH_CHECK();
    }
    return OV_OK;
}

extern "C" int ov_logits(const float* X, const float* Y, float* out, int64_t ldo, int n1, int n2, int E, float scale,
                         const float* scale_dev, ov_stream_t stream) {
    if (!X || !Y || !out || n1 <= 0 || n2 <= 0 || E <= 0 || ldo < n2) return OV_ERR_INVALID;
    if (E % 8) return OV_ERR_UNSUPPORTED;
    if (((uintptr_t)X | (uintptr_t)Y) & 15) return OV_ERR_INVALID;
    hipLaunchKernelGGL(logits_kernel, dim3((unsigned)((n2 + 31) / 32), (unsigned)((n1 + 31) / 32)), dim3(64), 0,
                       (hipStream_t)stream, X, Y, out, ldo, n1, n2, E, scale, scale_dev);
    OV_LAUNCH_CHECK();
    return OV_OK;
}
