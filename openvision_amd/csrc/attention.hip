// attention.hip — fused non-causal multi-head self-attention, head_dim 64, gfx950 (MI355X).
//
// Replaces the softmax(q k^T / sqrt(hd)) v core of nn.MultiheadAttention (reference
// open_clip/transformer.py:225,239-252; no mask: attn_mask=None, no dropout in eval) with a flash-style
// kernel: the [L, L] score matrix never leaves registers.  The online-softmax recurrence is the one of
// the reference's blockwise attention (src/models/bpt.py:108-124: running max, rescaled numerator and
// denominator per key block).
//
// Input is the packed qkv activation [B*L, 3*H*64] written by the QKV GEMM (q | k | v column blocks), output
// is [B*L, H*64] with heads merged -- exactly the operand layout of the out-proj GEMM, so no head
// split/merge kernels exist.
//
// One workgroup = one (batch, head) x NW query tiles of 32 rows (one tile per wave).  K and V of the
// (batch, head) are staged once per key chunk into LDS and shared by all waves:
//   K image  [key][64 d]   128-B rows, 16-B chunk index ^= (key >> 1) & 7  -> conflict-free ds_read_b128
//   V image  [d half][key][32 d] 64-B rows                                  -> conflict-free ds_read_b64_tr_b16
// Per 32-key tile and wave (v_mfma_f32_32x32x16_bf16 throughout):
//   S^T = K . Q^T          (K rows as the A operand, Q fragments kept in registers as B)  -> a lane holds
//                          16 of the 32 keys of ONE query, so row max / row sum are lane-local + 1 shuffle
//   P^T = exp2(S^T - m)    packed pairwise to bf16: the accumulator tile IS the next B operand
//   O^T += V^T . P^T       (V^T fragments via the transposing LDS read)
#include "common.h"

namespace {

struct AttnArgs {
    const ov_bf16* qkv; int64_t ldq;
    ov_bf16* out; int64_t ldo;
    int B, L, H, nqt, KC;
    float scale_log2;
};

typedef __attribute__((address_space(3))) s16x4_t* lds_s16x4_ptr;

__device__ __forceinline__ bf16x8_t tr_pair(const char* p0, const char* p1) {
    const s16x4_t a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)p0);
    const s16x4_t b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)p1);
    typedef short s16x8_t __attribute__((ext_vector_type(8)));
    const s16x8_t c = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    return __builtin_bit_cast(bf16x8_t, c);
}

// <= 96 VGPRs: 5 waves per SIMD, so two 9-wave workgroups (L = 257) are co-resident per CU (LDS 2 x 72 KiB)
__global__ __launch_bounds__(576, 5) void attn_fwd_hd64(const AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, nthreads = blockDim.x;
    const int lane = tid & 63, wave = tid >> 6, nw = nthreads >> 6;
    const int r = lane & 31, h2 = lane >> 5;
    const int bh = blockIdx.x, b = bh / a.H, h = bh - b * a.H;
    const int L = a.L, KC = a.KC;
    const int HD = a.H * 64;
    char* ks = smem;                         // K image: KC * 128 B
    char* vs = smem + KC * 128;              // V image: 2 halves * KC * 64 B
    const ov_bf16* base = a.qkv + (int64_t)b * L * a.ldq + h * 64;

    const int qt = blockIdx.y * nw + wave;
    const bool active = qt < a.nqt;
    const int q0 = qt * 32;

    bf16x8_t qf[4];
    {
        int qrow = q0 + r;
        qrow = (active && qrow < L) ? qrow : L - 1;
        const ov_bf16* qp = base + (int64_t)qrow * a.ldq + 8 * h2;
#pragma unroll
        for (int s = 0; s < 4; ++s) qf[s] = *(const bf16x8_t*)(qp + 16 * s);
    }
    float m = -INFINITY, lsum = 0.f;
    f32x16_t o0, o1;
#pragma unroll
    for (int i = 0; i < 16; ++i) { o0[i] = 0.f; o1[i] = 0.f; }

    // per-lane LDS offsets
    const int k_row_off = r * 128;                                   // + tile*4096
    const int k_sw = (r >> 1) & 7;
    const int vi = lane & 15, vg = (lane >> 4) & 1;
    const int v_lane_off = (4 * h2 + (vi >> 2)) * 64 + (16 * vg + 4 * (vi & 3)) * 2;

    for (int kc0 = 0; kc0 < L; kc0 += KC) {
        if (kc0) __syncthreads();
        // ---- stage K and V rows [kc0, kc0+KC) of this (batch, head) ----
        for (int idx = tid; idx < KC * 8; idx += nthreads) {
            const int row = idx >> 3, c = idx & 7;
            u32x4_t kv = {0u, 0u, 0u, 0u}, vv = {0u, 0u, 0u, 0u};
            if (kc0 + row < L) {
                const ov_bf16* p = base + (int64_t)(kc0 + row) * a.ldq + HD + c * 8;
                kv = *(const u32x4_t*)p;
                vv = *(const u32x4_t*)(p + HD);
            }
            *(u32x4_t*)(ks + row * 128 + ((c ^ ((row >> 1) & 7)) << 4)) = kv;
            *(u32x4_t*)(vs + (c >> 2) * (KC * 64) + row * 64 + (c & 3) * 16) = vv;
        }
        __syncthreads();
        if (!active) continue;
        const int nk = (L - kc0) < KC ? (L - kc0) : KC;
        const int ntile = (nk + 31) >> 5;
        for (int kt = 0; kt < ntile; ++kt) {
            // ---- S^T = K . Q^T ----
            f32x16_t s;
#pragma unroll
            for (int i = 0; i < 16; ++i) s[i] = 0.f;
            const char* kp = ks + kt * 4096 + k_row_off;
#pragma unroll
            for (int st = 0; st < 4; ++st) {
                const bf16x8_t kf = *(const bf16x8_t*)(kp + (((2 * st + h2) ^ k_sw) << 4));
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[st], s, 0, 0, 0);
            }
            // s[i] <-> key kt*32 + (i&3) + 8*(i>>2) + 4*h2, query r   (raw q.k; the softmax scale rides in the exp2 FMA)
            if (kt * 32 + 32 > nk) {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int key = kt * 32 + (i & 3) + 8 * (i >> 2) + 4 * h2;
                    if (key >= nk) s[i] = -INFINITY;
                }
            }
            float mx = fmaxf(fmaxf(fmaxf(s[0], s[1]), fmaxf(s[2], s[3])), fmaxf(fmaxf(s[4], s[5]), fmaxf(s[6], s[7])));
            mx = fmaxf(mx, fmaxf(fmaxf(fmaxf(s[8], s[9]), fmaxf(s[10], s[11])), fmaxf(fmaxf(s[12], s[13]), fmaxf(s[14], s[15]))));
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64)) * a.scale_log2;
            // deferred rescale (bpt.py:108-124 recurrence, rescale only when the running max grows by more than 2^8):
            // P stays <= 2^8, exact in fp32 sums and scale-free in bf16; wave-uniform branch.
            if (!__all(mx - m <= 8.0f)) {
                const float mn = fmaxf(m, mx);
                const float alpha = __builtin_amdgcn_exp2f(m - mn);
                m = mn;
                lsum *= alpha;
#pragma unroll
                for (int i = 0; i < 16; ++i) { o0[i] *= alpha; o1[i] *= alpha; }
            }
            const float nm = -m;
            float ps = 0.f;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                s[i] = __builtin_amdgcn_exp2f(fmaf(s[i], a.scale_log2, nm));
                ps += s[i];
            }
            lsum += ps;
            // ---- P^T as the B operand of the two k-steps ----
            bf16x8_t pf[2];
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                u32x4_t w = {pack_bf16x2(s[8 * st + 0], s[8 * st + 1]), pack_bf16x2(s[8 * st + 2], s[8 * st + 3]),
                             pack_bf16x2(s[8 * st + 4], s[8 * st + 5]), pack_bf16x2(s[8 * st + 6], s[8 * st + 7])};
                pf[st] = __builtin_bit_cast(bf16x8_t, w);
            }
            // ---- O^T += V^T . P^T ----
            const char* vp = vs + kt * 32 * 64 + v_lane_off;
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                const char* v0 = vp + st * 16 * 64;
                const bf16x8_t vf0 = tr_pair(v0, v0 + 8 * 64);
                const bf16x8_t vf1 = tr_pair(v0 + KC * 64, v0 + KC * 64 + 8 * 64);
                o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf0, pf[st], o0, 0, 0, 0);
                o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf1, pf[st], o1, 0, 0, 0);
            }
        }
    }
    if (!active) return;
    const float l = lsum + __shfl_xor(lsum, 32, 64);
    const float inv = 1.0f / l;
    const int q = q0 + r;
    if (q < L) {
        ov_bf16* op = a.out + ((int64_t)b * L + q) * a.ldo + h * 64 + 4 * h2;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            u32x2_t w0 = {pack_bf16x2(o0[4 * g] * inv, o0[4 * g + 1] * inv),
                          pack_bf16x2(o0[4 * g + 2] * inv, o0[4 * g + 3] * inv)};
            u32x2_t w1 = {pack_bf16x2(o1[4 * g] * inv, o1[4 * g + 1] * inv),
                          pack_bf16x2(o1[4 * g + 2] * inv, o1[4 * g + 3] * inv)};
            *(u32x2_t*)(op + 8 * g) = w0;
            *(u32x2_t*)(op + 32 + 8 * g) = w1;
        }
    }
}

}  // namespace

extern "C" int ov_attention(const ov_bf16* qkv, int64_t ld_qkv, ov_bf16* out, int64_t ld_out, int B, int L,
                            int H, int hd, float scale, ov_stream_t stream) {
    if (!qkv || !out || B <= 0 || L <= 0 || H <= 0) return OV_ERR_INVALID;
    if (hd != 64) return OV_ERR_UNSUPPORTED;
    if (ld_qkv % 8 || ld_out % 8 || ld_qkv < 3 * H * hd || ld_out < H * hd) return OV_ERR_INVALID;
    if (((uintptr_t)qkv | (uintptr_t)out) & 15) return OV_ERR_INVALID;
    AttnArgs a;
    a.qkv = qkv; a.ldq = ld_qkv; a.out = out; a.ldo = ld_out;
    a.B = B; a.L = L; a.H = H;
    a.nqt = (L + 31) / 32;
    a.scale_log2 = scale * 1.4426950408889634f;
    const int lp = a.nqt * 32;
    int nw;
    if (lp <= 320) { a.KC = lp; nw = a.nqt; }          // whole K/V of a head resident: one chunk
    else { a.KC = 256; nw = 8; }
    const int gy = (a.nqt + nw - 1) / nw;
    const size_t smem = (size_t)a.KC * 256;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)attn_fwd_hd64, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           160 * 1024);
        if (e != hipSuccess) return OV_ERR_HIP - (int)e;
        attr_set = true;
    }
    hipLaunchKernelGGL(attn_fwd_hd64, dim3((unsigned)(B * H), (unsigned)gy), dim3(nw * 64), smem,
                       (hipStream_t)stream, a);
    OV_LAUNCH_CHECK();
    return OV_OK;
}
