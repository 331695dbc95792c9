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
#include <stdlib.h>

namespace {

struct AttnArgs {
    const ov_bf16* qkv; int64_t ldq;
    ov_bf16* out; int64_t ldo;
    int B, L, H, nqt, KC;
    float scale_log2;
    int mode;      // diagnostics only (OVHIP_ATTN_MODE): 0 = full, 1 = staging only, 2 = compute only
};

typedef __attribute__((address_space(3))) s16x4_t* lds_s16x4_ptr;
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

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
        // ---- stage K and V rows [kc0, kc0+KC) of this (batch, head): LDS-DMA, every piece in flight at once ----
        // LDS 16-B chunk q of the K image holds key q>>3, d-chunk (q&7) ^ ((key>>1)&7); chunk q of V half dh holds key q>>2,
        // d-chunk dh*4 + (q&3).  The DMA writes LDS lane-linearly, so the permutation sits on the per-lane SOURCE address.
        // Keys >= L are clamped to L-1: finite data (their scores are masked to -inf, their P is exactly 0).
        {
            const int nchunk = KC * 8;                                   // per image (K) and for both V halves together
            for (int q0c = wave * 64; q0c < nchunk && a.mode != 2; q0c += nthreads) {   // wave-uniform trip count (64 | KC*8)
                const int q = q0c + lane;
                {
                    const int key = q >> 3;
                    int row = kc0 + key;
                    row = row < L ? row : L - 1;
                    const int c = (q & 7) ^ ((key >> 1) & 7);
                    __builtin_amdgcn_global_load_lds((gptr_t)(base + (int64_t)row * a.ldq + HD + c * 8), (lptr_t)(ks + q0c * 16),
                                                     16, 0, 0);
                }
                {
                    const int dh = q0c >= KC * 4 ? 1 : 0;                // wave-uniform (64 | KC*4)
                    const int qq = q - dh * KC * 4;
                    int row = kc0 + (qq >> 2);
                    row = row < L ? row : L - 1;
                    __builtin_amdgcn_global_load_lds((gptr_t)(base + (int64_t)row * a.ldq + 2 * HD + (dh * 4 + (qq & 3)) * 8),
                                                     (lptr_t)(vs + q0c * 16), 16, 0, 0);
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
        if (!active) continue;
        const int nk = (L - kc0) < KC ? (L - kc0) : KC;
        const int ntile = a.mode == 1 ? 0 : (nk + 31) >> 5;
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

// =====================================================================================================
// Persistent variant (default when two heads' K/V fit in LDS, i.e. padded L <= 320): a workgroup of nqt waves (one per
// 32-row query tile) walks heads bh = blockIdx.x, blockIdx.x + gridDim.x, ...  K/V of head j live in LDS slot j & 1.
// Per head: vmcnt(0) + ONE barrier (head j has landed for everybody, and everybody has left head j-1), then the LDS-DMA
// of head j+1 into the other slot is issued and runs under the MFMAs/softmax of head j -- HBM reads overlap compute.
// The kernel is sized for <= 168 VGPRs (3 waves on the busiest SIMD at L = 257), which leaves room to fetch all K and V
// fragments of a key tile ahead of the MFMAs that use them.
// Transposing LDS reads issued from inline asm: hipcc treats the ds_read_tr builtin as possibly aliasing the LDS-DMA
// (global_load_lds) writes of the NEXT head and would drain them with vmcnt(0) before every V read, serialising the
// prefetch.  The asm forms are invisible to its memory model; completion is enforced by tr_wait(), which names every
// destination "+v" (cdna guide 5.7 form ii) so no consumer can be scheduled above the lgkmcnt wait.
__device__ __forceinline__ u32x2_t tr_read_asm(const char* p) {
    u32x2_t v;
    const unsigned addr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const char*)p;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(addr));
    return v;
}
__device__ __forceinline__ void tr_wait(u32x2_t (&t)[8]) {
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(t[0]), "+v"(t[1]), "+v"(t[2]), "+v"(t[3]), "+v"(t[4]), "+v"(t[5]), "+v"(t[6]), "+v"(t[7]));
}
__device__ __forceinline__ bf16x8_t tr_join(u32x2_t a, u32x2_t b) {
    const u32x4_t w = {a[0], a[1], b[0], b[1]};
    return __builtin_bit_cast(bf16x8_t, w);
}

// 4 floats -> 4 e4m3 bytes (saturating)
__device__ __forceinline__ unsigned pack_fp8x4(float a, float b, float c, float d) {
    int w = 0;
    w = __builtin_amdgcn_cvt_pk_fp8_f32(__builtin_amdgcn_fmed3f(a, -448.f, 448.f), __builtin_amdgcn_fmed3f(b, -448.f, 448.f), w, false);
    w = __builtin_amdgcn_cvt_pk_fp8_f32(__builtin_amdgcn_fmed3f(c, -448.f, 448.f), __builtin_amdgcn_fmed3f(d, -448.f, 448.f), w, true);
    return (unsigned)w;
}

struct AttnPArgs {
    const ov_bf16* qkv; int64_t ldq;
    ov_bf16* out; int64_t ldo;
    int L, H, nqt, KC, nheads;
    float scale_log2;
    const float* out_amax;        // OUT8: the output is e4m3 bytes (ldo in bytes) with the static scale 2 * (*out_amax) / 448
    float* amax_next;             // OUT8, optional: running maximum of |output| (the next call's scale)
    int lone_valu;                // 1: a single-key last tile is folded in on the VALU (OVHIP_ATTN_LONEKEY=0 keeps the tile step)
    float* lse;                   // optional (training forward, bf16 output): lse[head][KC] = row log-sum-exp of scale * q.k in log2 units
};

__device__ __forceinline__ void stage_head(const AttnPArgs& a, int bh, char* slot, int wave, int lane, int nthreads) {
    const int b = bh / a.H, h = bh - b * a.H;
    const int HD = a.H * 64, KC = a.KC, L = a.L;
    // wave-uniform base + 32-bit per-lane byte offsets (L * ldq * 2 < 4 GiB: checked by the launcher): the DMA issue is on every
    // head's critical path, 64-bit per-piece address arithmetic made it ~300 cycles per 1-KiB piece
    const char* base = (const char*)(a.qkv + (int64_t)b * L * a.ldq + h * 64);
    const unsigned rowb = (unsigned)(a.ldq * 2);
    const unsigned last = (unsigned)(L - 1) * rowb;
    char* ks = slot;
    char* vs = slot + KC * 128;
    const int nchunk = KC * 8;
    // K piece q = q0c + lane: key q >> 3, source d-chunk (q & 7) ^ ((key >> 1) & 7).  V piece: d-half dh = (q0c >= 4 KC), key
    // (q - dh 4 KC) >> 2, d-chunk dh * 4 + (q & 3).  q0c is a multiple of 64, so q & 7 = lane & 7 and q & 3 = lane & 3.
    for (int q0c = wave * 64; q0c < nchunk; q0c += nthreads) {   // nothing per-lane is carried between trips (register room)
        {
            const int kkey = (q0c + lane) >> 3;
            const unsigned ro = kkey < L ? (unsigned)kkey * rowb : last;
            const int c = (lane & 7) ^ ((kkey >> 1) & 7);
            __builtin_amdgcn_global_load_lds((gptr_t)(base + (ro + (unsigned)((HD + c * 8) * 2))), (lptr_t)(ks + q0c * 16), 16, 0, 0);
        }
        {
            const int dh = q0c >= KC * 4 ? 1 : 0;                 // wave-uniform
            const int vkey = (q0c - dh * KC * 4 + lane) >> 2;
            const unsigned ro = vkey < L ? (unsigned)vkey * rowb : last;
            __builtin_amdgcn_global_load_lds((gptr_t)(base + (ro + (unsigned)((2 * HD + (dh * 4 + (lane & 3)) * 8) * 2))),
                                             (lptr_t)(vs + q0c * 16), 16, 0, 0);
        }
    }
}

__device__ __forceinline__ float max3_asm(float x, float y, float z) {       // no canonicalising v_max in front
    float d;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(x), "v"(y), "v"(z));
    return d;
}
typedef float f32x2_t __attribute__((ext_vector_type(2)));
// p = exp2(s * c - m) on 16 scores, two per v_pk_fma_f32 / v_pk_add_f32 (the softmax is VALU-issue bound; v_exp has no
// packed form); acc collects the two interleaved partial row sums.
__device__ __forceinline__ void exp_rows(f32x16_t& s, float c, float nm, f32x2_t& acc) {
    const f32x2_t cc = {c, c}, mm = {nm, nm};
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        f32x2_t v = {s[2 * i], s[2 * i + 1]};
        v = __builtin_elementwise_fma(v, cc, mm);
        v[0] = __builtin_amdgcn_exp2f(v[0]);
        v[1] = __builtin_amdgcn_exp2f(v[1]);
        acc += v;
        s[2 * i] = v[0];
        s[2 * i + 1] = v[1];
    }
}
template <int OFF>
__device__ __forceinline__ u32x2_t tr_read_off(unsigned addr) {
    u32x2_t v;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}

// DEEP = workgroups of <= 8 waves (2 per SIMD, 256 VGPRs each): both K tiles of the next step are fetched behind the S MFMAs
// and all V fragments of a step are in flight from its start.
// OUT8: e4m3 output with a static scale (fp8 path: feeds the out-proj GEMM without a bf16 round trip); same 8 stores per head.
// LSE: also store every query row's log-sum-exp (a.lse; training forward).  A template parameter, not a test of the pointer: a branch
// between the stores and the vmcnt wait that names the prefetched Q registers would let hipcc copy those registers at the merge
// point BEFORE the wait (it did: nondeterministic outputs).
template <bool DEEP, bool OUT8, bool LSE = false>
__global__ __launch_bounds__(DEEP ? 512 : 640, DEEP ? 2 : 3) void attn_fwd_hd64_persist(const AttnPArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h2 = lane >> 5;
    const int L = a.L, KC = a.KC;
    const int slot_bytes = KC * 256;
    const int nk_tiles = (L + 31) >> 5;
    const bool lone_on = a.lone_valu != 0;
    const bool lone_key = lone_on && (L & 31) == 1 && (nk_tiles & 1) && nk_tiles > 1;   // an odd last key tile that holds a single key
    const int n = (a.nheads - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;     // heads of this workgroup
    if (n <= 0) return;
    // wave-uniform scalars, pinned to SGPRs (v_readfirstlane): as VGPRs they were two of the loop invariants the 9-wave variant spilled
    const float inv8 = OUT8 ? __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(448.0f / (2.0f * fmaxf(*a.out_amax, 1e-30f))))) : 1.0f;
    const float next_thr = (OUT8 && a.amax_next != nullptr)       // read once: no global load inside the head loop
                               ? __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(*a.amax_next))) : 0.f;

    // K fragment st of this lane sits at row r, 16-byte chunk (2 st + h2) ^ k_sw, k_sw = (r >> 1) & 7: byte offset
    // kb ^ (32 st) with kb = r * 128 + ((h2 ^ k_sw) << 4) (bits 5-6 of everything else are zero).  ONE register; the XOR is
    // applied to the tile's address inside load_k (4 VALU ops per tile) instead of keeping four loop-invariant offsets live.
    const int kb = r * 128 + ((h2 ^ ((r >> 1) & 7)) << 4);
    const int vi = lane & 15, vg = (lane >> 4) & 1;
    const int v_lane0 = KC * 128 + (4 * h2 + (vi >> 2)) * 64 + (16 * vg + 4 * (vi & 3)) * 2;   // d half 0
    const int v_lane1 = v_lane0 + KC * 64;                                                    // d half 1
    const int q0 = wave * 32;
    int qrow = q0 + r;
    qrow = qrow < L ? qrow : L - 1;

    auto head_base = [&](int bh) {
        const int b = bh / a.H, h = bh - b * a.H;
        return a.qkv + (int64_t)b * L * a.ldq + h * 64;
    };
    // VM-op order per head and wave: [LDS-DMA pieces of head j+1 (top of head j)] ... [4 Q loads of head j+1][NS stores of head j],
    // NS = 4 (bf16 output: dwordx4 after the half-wave swap) or 8 (e4m3 output: dword stores).
    // The wait is vmcnt(NS) at the END of head j, right behind the stores: everything but the NS youngest operations (the
    // stores) has completed, so the stores drain under the next head's MFMAs while its K/V pieces (issued a whole head
    // earlier) and Q rows are known to have landed.  The count is exact because the stores are never predicated (lanes
    // past the last query replicate query L-1 and rewrite its row with identical values) and are inline asm.
    // The Q loads are inline asm too (hipcc would drain the LDS-DMA queue at the first use of an ordinary load); the wait
    // names their registers "+v" in the same basic block, before any loop-carried copy can be made of them.
    // Per-lane parts of the Q-row and output-row addresses as 32-bit BYTE offsets from a wave-uniform (SGPR) base per head
    // (saddr form of the global instructions): two VGPRs instead of two 64-bit pointers kept live across the whole head loop.
    // The launcher guarantees L * ld * 2 < 2^31 for both pitches.
    // They are RE-derived from the lane id at each use (once per head, ~6 VALU ops) behind an opaque asm: as loop invariants hipcc
    // hoisted them out of the head loop and then spilled them in the 9-wave variant (168 VGPRs).
    auto row_offset = [&](int ld, int per_h2, unsigned bytes) {
        unsigned lf = (unsigned)lane;
        asm volatile("" : "+v"(lf));
        int qr = q0 + (int)(lf & 31u);
        qr = qr < L ? qr : L - 1;
        return (unsigned)(qr * ld + per_h2 * (int)(lf >> 5)) * bytes;
    };
    auto load_q = [&](u32x4_t (&q)[4], int bh) {
        const ov_bf16* qb = head_base(bh);                         // wave-uniform
        const unsigned qoff = row_offset((int)a.ldq, 8, 2u);
        asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(q[0]) : "v"(qoff), "s"(qb) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, %2 offset:32" : "=v"(q[1]) : "v"(qoff), "s"(qb) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, %2 offset:64" : "=v"(q[2]) : "v"(qoff), "s"(qb) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, %2 offset:96" : "=v"(q[3]) : "v"(qoff), "s"(qb) : "memory");
    };
    stage_head(a, blockIdx.x, smem, wave, lane, blockDim.x);
    u32x4_t qn[4];
    load_q(qn, blockIdx.x);
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(qn[0]), "+v"(qn[1]), "+v"(qn[2]), "+v"(qn[3]) :: "memory");

    for (int j = 0; j < n; ++j) {
        const int bh = blockIdx.x + j * gridDim.x;
        bf16x8_t qf[4];
#pragma unroll
        for (int st = 0; st < 4; ++st) qf[st] = __builtin_bit_cast(bf16x8_t, qn[st]);
        asm volatile("s_barrier" ::: "memory");   // raw barrier (a __syncthreads fence would drain vmcnt to 0, stores included);
                                                  // the memory clobber keeps every LDS read of head j below it:
                                        // everybody's pieces have landed and everybody has left head j-1
        if (j + 1 < n) stage_head(a, bh + gridDim.x, smem + ((j + 1) & 1) * slot_bytes, wave, lane, blockDim.x);
        const char* ks = smem + (j & 1) * slot_bytes;
        const unsigned ks_u = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const char*)ks;

        float m = -INFINITY, lsum = 0.f;
        f32x16_t o0, o1;
#pragma unroll
        for (int i = 0; i < 16; ++i) { o0[i] = 0.f; o1[i] = 0.f; }
        // Key tiles are consumed two at a time (64 keys: two independent S accumulators, one max / rescale decision,
        // 32 exps) with the K fragments of the NEXT step fetched right behind the S MFMAs, so LDS latency and the
        // MFMA->VALU->MFMA dependency chain of one tile are covered by the other's work; an odd last tile runs alone.
        bf16x8_t kfa[4], kfb[4];
        auto load_k = [&](bf16x8_t (&kf)[4], int kt) {
            const unsigned kp = (unsigned)(kt * 4096 + kb);          // slot bases are multiples of 256: the XOR stays inside the row
#pragma unroll
            for (int st = 0; st < 4; ++st) kf[st] = *(const bf16x8_t*)(ks + (kp ^ (unsigned)(st * 32)));
        };
        auto mask_tail = [&](f32x16_t& sc, int kt) {
            if (kt * 32 + 32 > L) {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int key = kt * 32 + (i & 3) + 8 * (i >> 2) + 4 * h2;
                    if (key >= L) sc[i] = -INFINITY;
                }
            }
        };
        auto tile_max = [&](const f32x16_t& sc) {
            float mx = max3_asm(max3_asm(sc[0], sc[1], sc[2]), max3_asm(sc[3], sc[4], sc[5]), max3_asm(sc[6], sc[7], sc[8]));
            return max3_asm(mx, max3_asm(sc[9], sc[10], sc[11]), max3_asm(max3_asm(sc[12], sc[13], sc[14]), sc[15], sc[15]));
        };
        auto pack_p = [&](const f32x16_t& sc, int st) {
            const u32x4_t w = {pack_bf16x2(sc[8 * st + 0], sc[8 * st + 1]), pack_bf16x2(sc[8 * st + 2], sc[8 * st + 3]),
                               pack_bf16x2(sc[8 * st + 4], sc[8 * st + 5]), pack_bf16x2(sc[8 * st + 6], sc[8 * st + 7])};
            return __builtin_bit_cast(bf16x8_t, w);
        };
        const int npair = nk_tiles >> 1;
        load_k(kfa, 0);
        if (DEEP && npair > 0) load_k(kfb, 1);
        for (int it = 0; it < npair; ++it) {
            const int kt = 2 * it;
            const unsigned va0 = ks_u + kt * 2048 + v_lane0, va1 = ks_u + kt * 2048 + v_lane1;
            if (!DEEP) load_k(kfb, kt + 1);                            // second tile's K: covered by the first tile's MFMAs
            u32x2_t vt[8], vu[8];
            auto read_vt = [&]() {
                vt[0] = tr_read_off<0>(va0);    vt[1] = tr_read_off<512>(va0);
                vt[2] = tr_read_off<0>(va1);    vt[3] = tr_read_off<512>(va1);
                vt[4] = tr_read_off<1024>(va0); vt[5] = tr_read_off<1536>(va0);
                vt[6] = tr_read_off<1024>(va1); vt[7] = tr_read_off<1536>(va1);
            };
            // 9+ waves (3 per SIMD: 168 VGPRs): the first tile's V fragments are fetched behind the S MFMAs instead of in front
            // of them -- still a whole softmax ahead of their use, and 16 registers fewer while both K tiles are live (the kernel
            // spilled 10 VGPRs otherwise, which the inline-asm loads and counted waits must not meet: tests/test_build_scratch.py)
            if (DEEP) read_vt();
            if (DEEP) {
                vu[0] = tr_read_off<2048>(va0); vu[1] = tr_read_off<2560>(va0);
                vu[2] = tr_read_off<2048>(va1); vu[3] = tr_read_off<2560>(va1);
                vu[4] = tr_read_off<3072>(va0); vu[5] = tr_read_off<3584>(va0);
                vu[6] = tr_read_off<3072>(va1); vu[7] = tr_read_off<3584>(va1);
            }
            f32x16_t sa, sb;
#pragma unroll
            for (int i = 0; i < 16; ++i) { sa[i] = 0.f; sb[i] = 0.f; }
#pragma unroll
            for (int st = 0; st < 4; ++st) sa = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfa[st], qf[st], sa, 0, 0, 0);
#pragma unroll
            for (int st = 0; st < 4; ++st) sb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfb[st], qf[st], sb, 0, 0, 0);
            if (!DEEP) read_vt();
            if (kt + 2 < nk_tiles) load_k(kfa, kt + 2);                // next step's first tile, fetched under the softmax
            else if (DEEP) asm volatile("s_nop 7" : "+v"(sb));         // last step of an even tile count: nothing else separates the sb
                                                                       // MFMAs from tile_max's inline-asm reads (see the single-tile step)
            if (DEEP && kt + 3 < nk_tiles) load_k(kfb, kt + 3);
            mask_tail(sb, kt + 1);
            float mx = fmaxf(tile_max(sa), tile_max(sb));
            {
                const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
                mx = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1])) * a.scale_log2;
            }
            if (!__all(mx - m <= 8.0f)) {
                const float mn = fmaxf(m, mx);
                const float alpha = __builtin_amdgcn_exp2f(m - mn);
                m = mn;
                lsum *= alpha;
#pragma unroll
                for (int i = 0; i < 16; ++i) { o0[i] *= alpha; o1[i] *= alpha; }
            }
            const float nm = -m;
            f32x2_t ps = {0.f, 0.f};
            exp_rows(sa, a.scale_log2, nm, ps);
            exp_rows(sb, a.scale_log2, nm, ps);
            lsum += ps[0] + ps[1];
            const bf16x8_t pa0 = pack_p(sa, 0), pa1 = pack_p(sa, 1), pb0 = pack_p(sb, 0), pb1 = pack_p(sb, 1);
            tr_wait(vt);
            __builtin_amdgcn_sched_barrier(0);
            if (!DEEP) {   // second tile's V fragments: their LDS latency hides under the first tile's four P.V MFMAs
                vu[0] = tr_read_off<2048>(va0); vu[1] = tr_read_off<2560>(va0);
                vu[2] = tr_read_off<2048>(va1); vu[3] = tr_read_off<2560>(va1);
                vu[4] = tr_read_off<3072>(va0); vu[5] = tr_read_off<3584>(va0);
                vu[6] = tr_read_off<3072>(va1); vu[7] = tr_read_off<3584>(va1);
            }
            o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_join(vt[0], vt[1]), pa0, o0, 0, 0, 0);
            o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_join(vt[2], vt[3]), pa0, o1, 0, 0, 0);
            o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_join(vt[4], vt[5]), pa1, o0, 0, 0, 0);
            o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_join(vt[6], vt[7]), pa1, o1, 0, 0, 0);
            tr_wait(vu);
            __builtin_amdgcn_sched_barrier(0);
            o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_join(vu[0], vu[1]), pb0, o0, 0, 0, 0);
            o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_join(vu[2], vu[3]), pb0, o1, 0, 0, 0);
            o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_join(vu[4], vu[5]), pb1, o0, 0, 0, 0);
            o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_join(vu[6], vu[7]), pb1, o1, 0, 0, 0);
        }
        if (j + 1 < n) load_q(qn, bh + gridDim.x);                   // next head's Q rows, live only across the tail step
        if (lone_key) {
            // L = 32 k + 1 (ViT: patches + cls): the odd last key tile holds ONE key.  A tile step for it is a full dependent
            // chain (LDS -> 4 MFMAs -> softmax on 32 columns -> 4 MFMAs, ~2.3 k cycles per head and wave); the same update on the
            // VALU: the staged rows past L-1 are copies of row L-1, so this lane's own K fragments of the tile ARE that key
            // (its d range), 16 v_dot2 give q.k, then one exp2 and 32 FMAs with the key's V row.
            float s0 = 0.f, s1 = 0.f;
#pragma unroll
            for (int st = 0; st < 4; ++st) {
                const u32x4_t kq = __builtin_bit_cast(u32x4_t, kfa[st]), qq = __builtin_bit_cast(u32x4_t, qf[st]);
                // inline asm: hipcc's lowering of the fdot2 builtin on dwords extracted from the fragments picked dword 0 of every
                // fragment for all four products (checked in the ISA)
                asm("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(s0) : "v"(qq[0]), "v"(kq[0]));
                asm("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(s1) : "v"(qq[1]), "v"(kq[1]));
                asm("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(s0) : "v"(qq[2]), "v"(kq[2]));
                asm("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(s1) : "v"(qq[3]), "v"(kq[3]));
            }
            asm("s_nop 2" : "+v"(s0), "+v"(s1));                      // DOT result -> ordinary VALU read: 3 wait states the assembler
                                                                      // does not insert inside inline asm
            float sd = s0 + s1;                                       // this lane's half of d; the other half sits in lane ^ 32
            {
                const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(sd), __float_as_uint(sd), false, false);
                sd = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
            }
            // the key's V row: d = 8 g + 4 h2 + e (o0), + 32 (o1): four 8-byte reads per d-half of the V image
            const char* vrow = ks + KC * 128 + (L - 1) * 64 + 8 * h2;
            u32x2_t v0[4];                                            // d-half 0 now, d-half 1 into the same registers afterwards
#pragma unroll
            for (int g = 0; g < 4; ++g) v0[g] = *(const u32x2_t*)(vrow + g * 16);
            const float mx = sd * a.scale_log2;
            if (!__all(mx - m <= 8.0f)) {
                const float mn = fmaxf(m, mx);
                const float alpha = __builtin_amdgcn_exp2f(m - mn);
                m = mn;
                lsum *= alpha;
#pragma unroll
                for (int i = 0; i < 16; ++i) { o0[i] *= alpha; o1[i] *= alpha; }
            }
            const float pk = __builtin_amdgcn_exp2f(mx - m);
            lsum += h2 ? 0.f : pk;                                    // the halves' sums are added at the end: count the key once
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                o0[4 * g + 0] = fmaf(pk, bf16lo_to_f32(v0[g][0]), o0[4 * g + 0]);
                o0[4 * g + 1] = fmaf(pk, bf16hi_to_f32(v0[g][0]), o0[4 * g + 1]);
                o0[4 * g + 2] = fmaf(pk, bf16lo_to_f32(v0[g][1]), o0[4 * g + 2]);
                o0[4 * g + 3] = fmaf(pk, bf16hi_to_f32(v0[g][1]), o0[4 * g + 3]);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int g = 0; g < 4; ++g) v0[g] = *(const u32x2_t*)(vrow + KC * 64 + g * 16);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                o1[4 * g + 0] = fmaf(pk, bf16lo_to_f32(v0[g][0]), o1[4 * g + 0]);
                o1[4 * g + 1] = fmaf(pk, bf16hi_to_f32(v0[g][0]), o1[4 * g + 1]);
                o1[4 * g + 2] = fmaf(pk, bf16lo_to_f32(v0[g][1]), o1[4 * g + 2]);
                o1[4 * g + 3] = fmaf(pk, bf16hi_to_f32(v0[g][1]), o1[4 * g + 3]);
            }
        } else if (nk_tiles & 1) {
            const int kt = nk_tiles - 1;
            // the lane's V-image offset, re-derived behind an opaque asm (see row_offset: not a second live copy of v_lane0)
            unsigned lf = (unsigned)lane;
            asm volatile("" : "+v"(lf));
            const unsigned vif = lf & 15u, vgf = (lf >> 4) & 1u, h2f = lf >> 5;
            const unsigned vl0 = (unsigned)KC * 128u + (4u * h2f + (vif >> 2)) * 64u + (16u * vgf + 4u * (vif & 3u)) * 2u;
            const unsigned va0 = ks_u + kt * 2048 + vl0, va1 = va0 + (unsigned)KC * 64u;
            u32x2_t vt[8];
            vt[0] = tr_read_off<0>(va0);    vt[1] = tr_read_off<512>(va0);
            vt[2] = tr_read_off<0>(va1);    vt[3] = tr_read_off<512>(va1);
            vt[4] = tr_read_off<1024>(va0); vt[5] = tr_read_off<1536>(va0);
            vt[6] = tr_read_off<1024>(va1); vt[7] = tr_read_off<1536>(va1);
            f32x16_t sa;
#pragma unroll
            for (int i = 0; i < 16; ++i) sa[i] = 0.f;
#pragma unroll
            for (int st = 0; st < 4; ++st) sa = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfa[st], qf[st], sa, 0, 0, 0);
            mask_tail(sa, kt);
            // tile_max reads the accumulator from inline asm (v_max3): the wait states an MFMA result needs before a VALU read are
            // not inserted in front of inline asm, and when the tile is full (mask_tail reads nothing) no other instruction separates
            // the two here.  (In the pair loop a dozen LDS reads and the other tile's maximum sit in between.)  Read too early the
            // maximum is some older value: harmless for the softmax -- any shift m gives the same result up to rounding -- but
            // the output then differs in the last bit from run to run (seen in an experiment of round 2 with this very pattern).
            asm volatile("s_nop 15" : "+v"(sa));
            float mx = tile_max(sa);
            {
                const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
                mx = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1])) * a.scale_log2;
            }
            if (!__all(mx - m <= 8.0f)) {
                const float mn = fmaxf(m, mx);
                const float alpha = __builtin_amdgcn_exp2f(m - mn);
                m = mn;
                lsum *= alpha;
#pragma unroll
                for (int i = 0; i < 16; ++i) { o0[i] *= alpha; o1[i] *= alpha; }
            }
            const float nm = -m;
            f32x2_t ps = {0.f, 0.f};
            exp_rows(sa, a.scale_log2, nm, ps);
            lsum += ps[0] + ps[1];
            const bf16x8_t pa0 = pack_p(sa, 0), pa1 = pack_p(sa, 1);
            tr_wait(vt);
            __builtin_amdgcn_sched_barrier(0);
            o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_join(vt[0], vt[1]), pa0, o0, 0, 0, 0);
            o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_join(vt[2], vt[3]), pa0, o1, 0, 0, 0);
            o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_join(vt[4], vt[5]), pa1, o0, 0, 0, 0);
            o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_join(vt[6], vt[7]), pa1, o1, 0, 0, 0);
        }
        float l;
        {
            const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(lsum), __float_as_uint(lsum), false, false);
            l = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
        }
        float inv = 1.0f / l;
        if (OUT8) {
            if (a.amax_next != nullptr) {          // delayed scaling; the (rare) atomic is issued AHEAD of the 8 stores the vmcnt(8) below counts
                float lm = 0.f;
#pragma unroll
                for (int i = 0; i < 16; ++i) lm = fmaxf(lm, fmaxf(fabsf(o0[i]), fabsf(o1[i])));
                lm *= inv;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) lm = fmaxf(lm, __shfl_xor(lm, o, 64));
                if (lane == 0 && lm > next_thr) atomicMax((unsigned*)a.amax_next, __float_as_uint(lm));
            }
            inv *= inv8;
            const unsigned char* ob = (const unsigned char*)a.out + (int64_t)(bh / a.H) * L * a.ldo + (bh % a.H) * 64;   // uniform
            const unsigned ooff = row_offset((int)a.ldo, 4, 1u);
#define OV_ST8(GQ)                                                                                                         \
            {                                                                                                              \
                const unsigned w0 = pack_fp8x4(o0[4 * GQ] * inv, o0[4 * GQ + 1] * inv, o0[4 * GQ + 2] * inv, o0[4 * GQ + 3] * inv); \
                const unsigned w1 = pack_fp8x4(o1[4 * GQ] * inv, o1[4 * GQ + 1] * inv, o1[4 * GQ + 2] * inv, o1[4 * GQ + 3] * inv); \
                asm volatile("global_store_dword %0, %1, %2 offset:%3" :: "v"(ooff), "v"(w0), "s"(ob), "n"(8 * GQ) : "memory");        \
                asm volatile("global_store_dword %0, %1, %2 offset:%3\n\ts_nop 0" :: "v"(ooff), "v"(w1), "s"(ob), "n"(32 + 8 * GQ) : "memory"); \
            }
            OV_ST8(0) OV_ST8(1) OV_ST8(2) OV_ST8(3)
#undef OV_ST8
        } else {
            // Widened store tail: after packing, a query row is split over the two half-waves (lane r: d 8g..8g+3, lane r+32:
            // d 8g+4..8g+7 of group g).  One v_permlane32_swap per dword and pair of groups (g, g+1) leaves 16 contiguous bytes in
            // every lane (lower half: d 8g..8g+7, upper half: d 8g+8..8g+15): FOUR global_store_dwordx4 per head instead of eight
            // dwordx2 -- the tail is store-issue bound (each row-per-lane store is 64 separate requests), so half the instructions
            // is half the time.
            const ov_bf16* ob = a.out + (int64_t)(bh / a.H) * L * a.ldo + (bh % a.H) * 64;                       // uniform
            const unsigned ooff = row_offset((int)a.ldo, 8, 2u);
#define OV_ST16(O, P, BYTE_OFF)                                                                                              \
            {                                                                                                               \
                const unsigned ax = pack_bf16x2(O[8 * P + 0] * inv, O[8 * P + 1] * inv), ay = pack_bf16x2(O[8 * P + 2] * inv, O[8 * P + 3] * inv); \
                const unsigned bx = pack_bf16x2(O[8 * P + 4] * inv, O[8 * P + 5] * inv), by = pack_bf16x2(O[8 * P + 6] * inv, O[8 * P + 7] * inv); \
                const auto sx = __builtin_amdgcn_permlane32_swap(ax, bx, false, false);                                       \
                const auto sy = __builtin_amdgcn_permlane32_swap(ay, by, false, false);                                       \
                const u32x4_t w = {sx[0], sy[0], sx[1], sy[1]};                                                               \
                /* s_nop 1: a store of more than 8 bytes reads its data VGPRs late -- the next VALU write of them needs 2 wait states, and */ \
                /* the assembler's hazard recogniser does not look inside inline asm */                                      \
                asm volatile("global_store_dwordx4 %0, %1, %2 offset:%3\n\ts_nop 1" :: "v"(ooff), "v"(w), "s"(ob), "n"(BYTE_OFF) : "memory"); \
            }
            OV_ST16(o0, 0, 0) OV_ST16(o0, 1, 32) OV_ST16(o1, 0, 64) OV_ST16(o1, 1, 96)
#undef OV_ST16
        }
        if (LSE && !OUT8) {
            // kept for the backward (ov_attention_backward_saved): m and l are the same in both half-waves, rows past L repeat row
            // L - 1 (clamped Q row): duplicate stores of one value.  One more store behind the four of the output.
            const float v = m + __builtin_amdgcn_logf(l);                                                      // v_log_f32 = log2
            const float* lb = a.lse + (int64_t)bh * KC;                                                         // uniform
            const unsigned loff = row_offset(1, 0, 4u);
            asm volatile("global_store_dword %0, %1, %2" :: "v"(loff), "v"(v), "s"(lb) : "memory");
        }
        // everything but this head's stores (8 dword stores with e4m3 output, 4 dwordx4 stores otherwise, + 1 with LSE) has completed
        if (OUT8) asm volatile("s_waitcnt vmcnt(8)" : "+v"(qn[0]), "+v"(qn[1]), "+v"(qn[2]), "+v"(qn[3]) :: "memory");
        else if (LSE) asm volatile("s_waitcnt vmcnt(5)" : "+v"(qn[0]), "+v"(qn[1]), "+v"(qn[2]), "+v"(qn[3]) :: "memory");
        else asm volatile("s_waitcnt vmcnt(4)" : "+v"(qn[0]), "+v"(qn[1]), "+v"(qn[2]), "+v"(qn[3]) :: "memory");
    }
}

// =====================================================================================================
// Streaming variant for long sequences (padded L > 320; S/8@384: 2305 tokens).  One workgroup = 8 waves = 256 queries of
// one (batch, head); K/V stream through a ring of four 16-KiB LDS slots in 64-key chunks (K image [64][128 B] +
// V image [2 d-halves][64][64 B], same swizzles as above), LDS-DMA'd three chunks ahead.  Per chunk: a counted
// vmcnt (the two younger chunks stay in flight), ONE raw barrier (the chunk has landed for everybody and everybody has
// left the slot about to be refilled), the DMA of chunk c+3, then the two-key-tile step of the persistent kernel.
// <= 128 VGPRs: two workgroups (2 x 64 KiB LDS) = 4 waves per SIMD share a CU, so one wave's softmax VALU work runs
// under another's MFMAs.  Workgroup ids are dealt to XCDs so that the q-blocks of a head share that XCD's L2.
struct AttnSArgs {
    const ov_bf16* qkv; int64_t ldq;
    ov_bf16* out; int64_t ldo;
    int L, H, nqt, nqb, nheads, nchunks;
    float scale_log2;
    const float* out_amax;        // OUT8 (see AttnPArgs)
    float* amax_next;
    int lone_valu;                // 1: a single-key last chunk (L = 64 k + 1) is folded in on the VALU (OVHIP_ATTN_LONEKEY=0: a chunk step)
};

template <bool OUT8>
__global__ __launch_bounds__(512, 4) void attn_fwd_hd64_stream(const AttnSArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h2 = lane >> 5;
    const int L = a.L;
    // XCD-aware work map: workgroup id -> (head, q-block); the q-blocks of a head run back to back on one XCD
    const int wid = blockIdx.x;
    const int xcd = wid & 7, slot_id = wid >> 3;
    const int bh = (slot_id / a.nqb) * 8 + xcd, qb = slot_id % a.nqb;
    if (bh >= a.nheads) return;
    const int b = bh / a.H, h = bh - b * a.H;
    const int HD = a.H * 64;
    const ov_bf16* base = a.qkv + (int64_t)b * L * a.ldq + h * 64;
    const int qt = qb * 8 + wave;
    const bool active = qt < a.nqt;
    int qrow = qt * 32 + r;
    qrow = qrow < L ? qrow : L - 1;

    // ---- staging: thread tid moves one 16-B piece of K and one of V per chunk (32-bit byte offsets from `base`) ----
    // K piece: key tid>>3, source d-chunk (tid&7) ^ ((key>>1)&7) (swizzle on the source side); V piece: d-half tid>>8, key
    // (tid&255)>>2, d-chunk within the half tid&3.
    char* const sdst = smem + wave * 1024;
    const int nc = a.nchunks;
    const char* const bbase = (const char*)base;
    unsigned kofs = (unsigned)(((tid >> 3) * a.ldq + HD + (((tid & 7) ^ ((tid >> 4) & 7)) * 8)) * 2);
    unsigned vofs = (unsigned)(((((tid & 255) >> 2)) * a.ldq + 2 * HD + ((tid >> 8) * 4 + (tid & 3)) * 8) * 2);
    const unsigned cstep = (unsigned)(64 * a.ldq * 2);
    auto stage = [&](int c) {                                     // chunks are staged in order 0, 1, 2, ...
        char* dst = sdst + (c & 3) * 16384;
        unsigned ko = kofs, vo = vofs;
        if (c == nc - 1) {                                        // last chunk: rows past the sequence re-read row L-1 (finite; masked)
            const int t = threadIdx.x;
            const int over_k = c * 64 + (t >> 3) - (L - 1), over_v = c * 64 + ((t & 255) >> 2) - (L - 1);
            if (over_k > 0) ko -= (unsigned)(over_k * a.ldq * 2);
            if (over_v > 0) vo -= (unsigned)(over_v * a.ldq * 2);
        }
        __builtin_amdgcn_global_load_lds((gptr_t)(bbase + ko), (lptr_t)dst, 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gptr_t)(bbase + vo), (lptr_t)(dst + 8192), 16, 0, 0);
        kofs += cstep;
        vofs += cstep;
    };
    // Q rows: ordinary loads, waited for by the compiler (once per workgroup).  Not the inline-asm form of the persistent
    // kernel: an asm load's destination can be spilled or copied right behind the asm statement, before the data has landed
    // (seen with a 128-VGPR ping-pong variant of this kernel: the late write then hit live address registers).
    bf16x8_t qf[4];
    {
        const ov_bf16* qp = base + (int64_t)qrow * a.ldq + 8 * h2;
#pragma unroll
        for (int st = 0; st < 4; ++st) qf[st] = *(const bf16x8_t*)(qp + 16 * st);
    }
    stage(0);
    if (nc > 1) stage(1);
    if (nc > 2) stage(2);

    const int k_lane = r * 128;
    const int k_sw = (r >> 1) & 7;
    // fragment st of a key row sits at chunk (2 st + h2) ^ k_sw = (h2 ^ k_sw) ^ 2 st: one base, st by XOR (bits 5-6)
    const int kbase = k_lane + ((h2 ^ k_sw) << 4);
    const int vi = lane & 15, vg = (lane >> 4) & 1;
    const int v_lane0 = 8192 + (4 * h2 + (vi >> 2)) * 64 + (16 * vg + 4 * (vi & 3)) * 2;    // d half 0 (half 1 = + 4096)
    const unsigned smem_u = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const char*)smem;

    float m = -INFINITY, lsum = 0.f;
    f32x16_t o0, o1;
#pragma unroll
    for (int i = 0; i < 16; ++i) { o0[i] = 0.f; o1[i] = 0.f; }
    auto tile_max = [&](const f32x16_t& sc) {
        float mx = max3_asm(max3_asm(sc[0], sc[1], sc[2]), max3_asm(sc[3], sc[4], sc[5]), max3_asm(sc[6], sc[7], sc[8]));
        return max3_asm(mx, max3_asm(sc[9], sc[10], sc[11]), max3_asm(max3_asm(sc[12], sc[13], sc[14]), sc[15], sc[15]));
    };
    auto pack_p = [&](const f32x16_t& sc, int st) {
        const u32x4_t w = {pack_bf16x2(sc[8 * st + 0], sc[8 * st + 1]), pack_bf16x2(sc[8 * st + 2], sc[8 * st + 3]),
                           pack_bf16x2(sc[8 * st + 4], sc[8 * st + 5]), pack_bf16x2(sc[8 * st + 6], sc[8 * st + 7])};
        return __builtin_bit_cast(bf16x8_t, w);
    };

    // L = 64 k + 1 (every ViT grid of a multiple of 8 plus the class token: 577, 1025, 2305): the last chunk holds ONE key.  It is staged
    // like any other chunk but folded in on the VALU behind the loop (its K / V rows are row 0 of its ring slot), as the persistent
    // kernel does for its single-key tile: a chunk step for it is a whole barrier + 16 MFMAs + 32 exps per lane (2.8 % of the launch at
    // L = 2305).
    const bool lone_key = a.lone_valu != 0 && (L & 63) == 1 && nc > 1;
    const int nc_mfma = lone_key ? nc - 1 : nc;
    for (int c = 0; c < nc_mfma; ++c) {
        if (c + 2 < nc) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else if (c + 1 < nc) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_barrier" ::: "memory");
        if (c + 3 < nc) stage(c + 3);
        if (!active) continue;                                    // idle waves of the last q-block still stage and synchronise
        const char* ks = smem + (c & 3) * 16384;
        const unsigned va0 = smem_u + (c & 3) * 16384 + v_lane0, va1 = va0 + 4096;
        // the second tile's K fragments are fetched behind the first tile's MFMAs (register room: 128 VGPRs)
        bf16x8_t kfa[4], kfb[4];
#pragma unroll
        for (int st = 0; st < 4; ++st) kfa[st] = *(const bf16x8_t*)(ks + (kbase ^ (st << 5)));
        f32x16_t sa, sb;
#pragma unroll
        for (int i = 0; i < 16; ++i) { sa[i] = 0.f; sb[i] = 0.f; }
#pragma unroll
        for (int st = 0; st < 4; ++st) sa = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfa[st], qf[st], sa, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int st = 0; st < 4; ++st) kfb[st] = *(const bf16x8_t*)(ks + 4096 + (kbase ^ (st << 5)));
#pragma unroll
        for (int st = 0; st < 4; ++st) sb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfb[st], qf[st], sb, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        u32x2_t vt[8], vu[8];
        vt[0] = tr_read_off<0>(va0);    vt[1] = tr_read_off<512>(va0);
        vt[2] = tr_read_off<0>(va1);    vt[3] = tr_read_off<512>(va1);
        vt[4] = tr_read_off<1024>(va0); vt[5] = tr_read_off<1536>(va0);
        vt[6] = tr_read_off<1024>(va1); vt[7] = tr_read_off<1536>(va1);
        if (c * 64 + 64 > L) {                                    // last chunk: keys past the sequence score -inf (P = 0)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int key = c * 64 + (i & 3) + 8 * (i >> 2) + 4 * h2;
                if (key >= L) sa[i] = -INFINITY;
                if (key + 32 >= L) sb[i] = -INFINITY;
            }
        }
        float mx = fmaxf(tile_max(sa), tile_max(sb));
        {
            const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
            mx = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1])) * a.scale_log2;
        }
        if (!__all(mx - m <= 8.0f)) {
            const float mn = fmaxf(m, mx);
            const float alpha = __builtin_amdgcn_exp2f(m - mn);
            m = mn;
            lsum *= alpha;
#pragma unroll
            for (int i = 0; i < 16; ++i) { o0[i] *= alpha; o1[i] *= alpha; }
        }
        const float nm = -m;
        f32x2_t ps = {0.f, 0.f};
        exp_rows(sa, a.scale_log2, nm, ps);
        const bf16x8_t pa0 = pack_p(sa, 0), pa1 = pack_p(sa, 1);     // packed at once: the fp32 tile's registers retire early
        exp_rows(sb, a.scale_log2, nm, ps);
        const bf16x8_t pb0 = pack_p(sb, 0), pb1 = pack_p(sb, 1);
        lsum += ps[0] + ps[1];
        tr_wait(vt);
        __builtin_amdgcn_sched_barrier(0);
        vu[0] = tr_read_off<2048>(va0); vu[1] = tr_read_off<2560>(va0);
        vu[2] = tr_read_off<2048>(va1); vu[3] = tr_read_off<2560>(va1);
        vu[4] = tr_read_off<3072>(va0); vu[5] = tr_read_off<3584>(va0);
        vu[6] = tr_read_off<3072>(va1); vu[7] = tr_read_off<3584>(va1);
        o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_join(vt[0], vt[1]), pa0, o0, 0, 0, 0);
        o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_join(vt[2], vt[3]), pa0, o1, 0, 0, 0);
        o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_join(vt[4], vt[5]), pa1, o0, 0, 0, 0);
        o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_join(vt[6], vt[7]), pa1, o1, 0, 0, 0);
        tr_wait(vu);
        __builtin_amdgcn_sched_barrier(0);
        o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_join(vu[0], vu[1]), pb0, o0, 0, 0, 0);
        o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_join(vu[2], vu[3]), pb0, o1, 0, 0, 0);
        o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_join(vu[4], vu[5]), pb1, o0, 0, 0, 0);
        o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_join(vu[6], vu[7]), pb1, o1, 0, 0, 0);
    }
    if (lone_key) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the last chunk's two pieces (issued three chunks ago)
        asm volatile("s_barrier" ::: "memory");
    }
    if (!active) return;
    if (lone_key) {
        const char* ks = smem + ((nc - 1) & 3) * 16384;           // key L-1 = row 0 of the slot: K chunk c at byte 16 c, V d-half h at 8192 + 4096 h
        float s0 = 0.f, s1 = 0.f;
#pragma unroll
        for (int st = 0; st < 4; ++st) {
            const u32x4_t kq = *(const u32x4_t*)(ks + (2 * st + h2) * 16), qq = __builtin_bit_cast(u32x4_t, qf[st]);
            asm("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(s0) : "v"(qq[0]), "v"(kq[0]));
            asm("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(s1) : "v"(qq[1]), "v"(kq[1]));
            asm("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(s0) : "v"(qq[2]), "v"(kq[2]));
            asm("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(s1) : "v"(qq[3]), "v"(kq[3]));
        }
        asm("s_nop 2" : "+v"(s0), "+v"(s1));                      // DOT result -> ordinary VALU read (not inserted inside inline asm)
        float sd = s0 + s1;                                       // this lane's half of d; the other half sits in lane ^ 32
        {
            const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(sd), __float_as_uint(sd), false, false);
            sd = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
        }
        const float mx = sd * a.scale_log2;
        if (!__all(mx - m <= 8.0f)) {
            const float mn = fmaxf(m, mx);
            const float alpha = __builtin_amdgcn_exp2f(m - mn);
            m = mn;
            lsum *= alpha;
#pragma unroll
            for (int i = 0; i < 16; ++i) { o0[i] *= alpha; o1[i] *= alpha; }
        }
        const float pk = __builtin_amdgcn_exp2f(mx - m);
        lsum += h2 ? 0.f : pk;                                    // the halves' sums are added below: count the key once
        const char* vrow = ks + 8192 + 8 * h2;                    // o0[4 g + e]: d = 8 g + 4 h2 + e; o1: + 32 (the other d-half)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const u32x2_t v0 = *(const u32x2_t*)(vrow + g * 16), v1 = *(const u32x2_t*)(vrow + 4096 + g * 16);
            o0[4 * g + 0] = fmaf(pk, bf16lo_to_f32(v0[0]), o0[4 * g + 0]);
            o0[4 * g + 1] = fmaf(pk, bf16hi_to_f32(v0[0]), o0[4 * g + 1]);
            o0[4 * g + 2] = fmaf(pk, bf16lo_to_f32(v0[1]), o0[4 * g + 2]);
            o0[4 * g + 3] = fmaf(pk, bf16hi_to_f32(v0[1]), o0[4 * g + 3]);
            o1[4 * g + 0] = fmaf(pk, bf16lo_to_f32(v1[0]), o1[4 * g + 0]);
            o1[4 * g + 1] = fmaf(pk, bf16hi_to_f32(v1[0]), o1[4 * g + 1]);
            o1[4 * g + 2] = fmaf(pk, bf16lo_to_f32(v1[1]), o1[4 * g + 2]);
            o1[4 * g + 3] = fmaf(pk, bf16hi_to_f32(v1[1]), o1[4 * g + 3]);
        }
    }
    float l;
    {
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(lsum), __float_as_uint(lsum), false, false);
        l = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
    }
    float inv = 1.0f / l;
    if (OUT8 && a.amax_next != nullptr) {                         // delayed scaling: running maximum of |output| (all lanes take part)
        float lm = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) lm = fmaxf(lm, fmaxf(fabsf(o0[i]), fabsf(o1[i])));
        lm = qt * 32 + r < L ? lm * inv : 0.f;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) lm = fmaxf(lm, __shfl_xor(lm, o, 64));
        if (lane == 0 && lm > *(volatile float*)a.amax_next) atomicMax((unsigned*)a.amax_next, __float_as_uint(lm));
    }
    if (qt * 32 + r < L) {
        if (OUT8) {
            inv *= 448.0f / (2.0f * fmaxf(*a.out_amax, 1e-30f));
            unsigned char* op = (unsigned char*)a.out + ((int64_t)b * L + qrow) * a.ldo + h * 64 + 4 * h2;
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                *(unsigned*)(op + 8 * gq) = pack_fp8x4(o0[4 * gq] * inv, o0[4 * gq + 1] * inv, o0[4 * gq + 2] * inv, o0[4 * gq + 3] * inv);
                *(unsigned*)(op + 32 + 8 * gq) = pack_fp8x4(o1[4 * gq] * inv, o1[4 * gq + 1] * inv, o1[4 * gq + 2] * inv, o1[4 * gq + 3] * inv);
            }
        } else {
            ov_bf16* op = a.out + ((int64_t)b * L + qrow) * a.ldo + h * 64 + 4 * h2;
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const u32x2_t w0 = {pack_bf16x2(o0[4 * gq] * inv, o0[4 * gq + 1] * inv),
                                    pack_bf16x2(o0[4 * gq + 2] * inv, o0[4 * gq + 3] * inv)};
                const u32x2_t w1 = {pack_bf16x2(o1[4 * gq] * inv, o1[4 * gq + 1] * inv),
                                    pack_bf16x2(o1[4 * gq + 2] * inv, o1[4 * gq + 3] * inv)};
                *(u32x2_t*)(op + 8 * gq) = w0;
                *(u32x2_t*)(op + 32 + 8 * gq) = w1;
            }
        }
    }
}

// =====================================================================================================
// Generic head_dim kernel (head_dim 72 = So400m, 80 = H/14; any multiple of 8 up to 96): same S^T / P^T / O^T scheme with
// the head padded to 96 in LDS (zero columns), 256-key chunks staged through registers, one key tile per step.  A plain,
// correctness-first variant: these model sizes are not on the benchmark configuration.
template <int HDP>
__global__ __launch_bounds__(640) void attn_fwd_generic(const AttnArgs a, int hd) {
    constexpr int KS = HDP / 16;           // k-steps of S^T = K.Q^T
    constexpr int DT = HDP / 32;           // 32-row tiles of O^T
    constexpr int CH = HDP / 8;            // 16-byte chunks per (padded) row
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, nthreads = blockDim.x;
    const int lane = tid & 63, wave = tid >> 6, nw = nthreads >> 6;
    const int r = lane & 31, h2 = lane >> 5;
    const int bh = blockIdx.x, b = bh / a.H, h = bh - b * a.H;
    const int L = a.L, KC = a.KC;
    const int HD = a.H * hd;
    char* ks = smem;                                 // K image: KC rows x HDP*2 bytes
    char* vs = smem + KC * HDP * 2;                  // V image: DT tiles x KC rows x 64 bytes
    const ov_bf16* base = a.qkv + (int64_t)b * L * a.ldq + h * hd;
    const int qt = blockIdx.y * nw + wave;
    const bool active = qt < a.nqt;
    const int q0 = qt * 32;

    bf16x8_t qf[KS];
    {
        int qrow = q0 + r;
        qrow = (active && qrow < L) ? qrow : L - 1;
        const ov_bf16* qp = base + (int64_t)qrow * a.ldq;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int d = 16 * s + 8 * h2;
            u32x4_t w = {0u, 0u, 0u, 0u};
            if (d < hd) w = *(const u32x4_t*)(qp + d);
            qf[s] = __builtin_bit_cast(bf16x8_t, w);
        }
    }
    float m = -INFINITY, lsum = 0.f;
    f32x16_t o[DT];
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) o[t][i] = 0.f;
    const int vi = lane & 15, vg = (lane >> 4) & 1;
    const int v_lane_off = (4 * h2 + (vi >> 2)) * 64 + (16 * vg + 4 * (vi & 3)) * 2;

    for (int kc0 = 0; kc0 < L; kc0 += KC) {
        if (kc0) __syncthreads();
        for (int idx = tid; idx < KC * CH; idx += nthreads) {
            const int row = idx / CH, c = idx - row * CH;
            u32x4_t kv = {0u, 0u, 0u, 0u}, vv = {0u, 0u, 0u, 0u};
            if (kc0 + row < L && c * 8 < hd) {
                const ov_bf16* p = base + (int64_t)(kc0 + row) * a.ldq + HD + c * 8;
                kv = *(const u32x4_t*)p;
                vv = *(const u32x4_t*)(p + HD);
            }
            *(u32x4_t*)(ks + row * (HDP * 2) + c * 16) = kv;
            *(u32x4_t*)(vs + (c >> 2) * (KC * 64) + row * 64 + (c & 3) * 16) = vv;
        }
        __syncthreads();
        if (!active) continue;
        const int nk = (L - kc0) < KC ? (L - kc0) : KC;
        // a last key tile that holds ONE key (L = 32 k + 1: 257) is folded in on the VALU behind the tile loop, as in the head_dim 64
        // kernels: a tile step for it is a ninth of the loop at L = 257
        const bool lone_key = (nk & 31) == 1 && nk > 32;
        const int ntile = ((nk + 31) >> 5) - (lone_key ? 1 : 0);
        for (int kt = 0; kt < ntile; ++kt) {
            f32x16_t s;
#pragma unroll
            for (int i = 0; i < 16; ++i) s[i] = 0.f;
            const char* kp = ks + (kt * 32 + r) * (HDP * 2) + h2 * 16;
#pragma unroll
            for (int st = 0; st < KS; ++st) {
                const bf16x8_t kf = *(const bf16x8_t*)(kp + st * 32);
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[st], s, 0, 0, 0);
            }
            if (kt * 32 + 32 > nk) {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int key = kt * 32 + (i & 3) + 8 * (i >> 2) + 4 * h2;
                    if (key >= nk) s[i] = -INFINITY;
                }
            }
            // (the maximum reads the accumulator from inline asm: the MFMA -> VALU wait states are not inserted in front of inline asm,
            // and a full tile leaves no other instruction in between -- see the single-tile step of attn_fwd_hd64_persist)
            asm volatile("s_nop 15" : "+v"(s));
            float mx = max3_asm(max3_asm(s[0], s[1], s[2]), max3_asm(s[3], s[4], s[5]), max3_asm(s[6], s[7], s[8]));
            mx = max3_asm(mx, max3_asm(s[9], s[10], s[11]), max3_asm(max3_asm(s[12], s[13], s[14]), s[15], s[15]));
            {
                const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
                mx = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1])) * a.scale_log2;
            }
            if (!__all(mx - m <= 8.0f)) {
                const float mn = fmaxf(m, mx);
                const float alpha = __builtin_amdgcn_exp2f(m - mn);
                m = mn;
                lsum *= alpha;
#pragma unroll
                for (int t = 0; t < DT; ++t)
#pragma unroll
                    for (int i = 0; i < 16; ++i) o[t][i] *= alpha;
            }
            const float nm = -m;
            f32x2_t ps = {0.f, 0.f};
            exp_rows(s, a.scale_log2, nm, ps);
            lsum += ps[0] + ps[1];
            bf16x8_t pf[2];
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                u32x4_t w = {pack_bf16x2(s[8 * st + 0], s[8 * st + 1]), pack_bf16x2(s[8 * st + 2], s[8 * st + 3]),
                             pack_bf16x2(s[8 * st + 4], s[8 * st + 5]), pack_bf16x2(s[8 * st + 6], s[8 * st + 7])};
                pf[st] = __builtin_bit_cast(bf16x8_t, w);
            }
            const char* vp = vs + kt * 32 * 64 + v_lane_off;
#pragma unroll
            for (int st = 0; st < 2; ++st) {
#pragma unroll
                for (int t = 0; t < DT; ++t) {
                    const char* v0 = vp + t * (KC * 64) + st * 16 * 64;
                    o[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_pair(v0, v0 + 8 * 64), pf[st], o[t], 0, 0, 0);
                }
            }
        }
        if (lone_key) {
            const int key = nk - 1;                                   // row of the chunk
            const char* kp = ks + key * (HDP * 2) + h2 * 16;
            float s0 = 0.f, s1 = 0.f;
#pragma unroll
            for (int st = 0; st < KS; ++st) {
                const u32x4_t kq = *(const u32x4_t*)(kp + st * 32), qq = __builtin_bit_cast(u32x4_t, qf[st]);
                asm("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(s0) : "v"(qq[0]), "v"(kq[0]));
                asm("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(s1) : "v"(qq[1]), "v"(kq[1]));
                asm("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(s0) : "v"(qq[2]), "v"(kq[2]));
                asm("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(s1) : "v"(qq[3]), "v"(kq[3]));
            }
            asm("s_nop 2" : "+v"(s0), "+v"(s1));                      // DOT result -> ordinary VALU read (not inserted inside inline asm)
            float sd = s0 + s1;                                       // this lane's d chunks; the others sit in lane ^ 32
            {
                const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(sd), __float_as_uint(sd), false, false);
                sd = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
            }
            const float mx = sd * a.scale_log2;
            if (!__all(mx - m <= 8.0f)) {
                const float mn = fmaxf(m, mx);
                const float alpha = __builtin_amdgcn_exp2f(m - mn);
                m = mn;
                lsum *= alpha;
#pragma unroll
                for (int t = 0; t < DT; ++t)
#pragma unroll
                    for (int i = 0; i < 16; ++i) o[t][i] *= alpha;
            }
            const float pk = __builtin_amdgcn_exp2f(mx - m);
            lsum += h2 ? 0.f : pk;                                    // the halves' sums are added at the end: count the key once
#pragma unroll
            for (int t = 0; t < DT; ++t) {                            // o[t][4 g + e]: d = 32 t + 8 g + 4 h2 + e
                const char* vrow = vs + t * (KC * 64) + key * 64 + 8 * h2;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const u32x2_t v = *(const u32x2_t*)(vrow + g * 16);
                    o[t][4 * g + 0] = fmaf(pk, bf16lo_to_f32(v[0]), o[t][4 * g + 0]);
                    o[t][4 * g + 1] = fmaf(pk, bf16hi_to_f32(v[0]), o[t][4 * g + 1]);
                    o[t][4 * g + 2] = fmaf(pk, bf16lo_to_f32(v[1]), o[t][4 * g + 2]);
                    o[t][4 * g + 3] = fmaf(pk, bf16hi_to_f32(v[1]), o[t][4 * g + 3]);
                }
            }
        }
    }
    if (!active) return;
    const float l = lsum + __shfl_xor(lsum, 32, 64);
    const float inv = 1.0f / l;
    const int q = q0 + r;
    if (q < L) {
        ov_bf16* op = a.out + ((int64_t)b * L + q) * a.ldo + h * hd + 4 * h2;
#pragma unroll
        for (int t = 0; t < DT; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int d = t * 32 + 8 * g + 4 * h2;
                if (d < hd) {
                    const u32x2_t w = {pack_bf16x2(o[t][4 * g] * inv, o[t][4 * g + 1] * inv),
                                       pack_bf16x2(o[t][4 * g + 2] * inv, o[t][4 * g + 3] * inv)};
                    *(u32x2_t*)(op + t * 32 + 8 * g) = w;
                }
            }
    }
}

}  // namespace

namespace {
int attention_impl(const ov_bf16* qkv, int64_t ld_qkv, ov_bf16* out, int64_t ld_out, int B, int L, int H, int hd, float scale,
                   const float* out_amax, float* amax_next, float* lse, ov_stream_t stream);
}

extern "C" int ov_attention(const ov_bf16* qkv, int64_t ld_qkv, ov_bf16* out, int64_t ld_out, int B, int L,
                            int H, int hd, float scale, ov_stream_t stream) {
    return attention_impl(qkv, ld_qkv, out, ld_out, B, L, H, hd, scale, nullptr, nullptr, nullptr, stream);
}

// ov_attention that also keeps the softmax statistics for ov_attention_backward_saved: lse[B*H][Lp] (Lp = L rounded up to 32) = the
// log-sum-exp of every query row's scaled scores, in log2 units (m + log2 l of the online softmax).  Only for the shapes the resident
// backward kernel takes (head_dim 64, L <= 288); OV_ERR_UNSUPPORTED otherwise -- call ov_attention and let the backward recompute.
extern "C" int ov_attention_lse(const ov_bf16* qkv, int64_t ld_qkv, ov_bf16* out, int64_t ld_out, float* lse, int B, int L, int H, int hd,
                                float scale, ov_stream_t stream) {
    if (!lse || ((uintptr_t)lse & 3)) return OV_ERR_INVALID;
    return attention_impl(qkv, ld_qkv, out, ld_out, B, L, H, hd, scale, nullptr, nullptr, lse, stream);
}

// Same attention, output written as e4m3 bytes out8[B*L, H*64] (ld_out in bytes) with the static scale 2 * (*out_amax) / 448
// (fp8 path: the out-proj GEMM reads it with that scalar scale).  head_dim 64 only; OV_ERR_UNSUPPORTED otherwise.
extern "C" int ov_attention_fp8out(const ov_bf16* qkv, int64_t ld_qkv, unsigned char* out8, int64_t ld_out, int B, int L, int H,
                                   int hd, float scale, const float* out_amax, float* out_amax_next, ov_stream_t stream) {
    if (!out_amax) return OV_ERR_INVALID;
    if (hd != 64) return OV_ERR_UNSUPPORTED;
    return attention_impl(qkv, ld_qkv, (ov_bf16*)out8, ld_out, B, L, H, hd, scale, out_amax, out_amax_next, nullptr, stream);
}

namespace {
int attention_impl(const ov_bf16* qkv, int64_t ld_qkv, ov_bf16* out, int64_t ld_out, int B, int L, int H, int hd, float scale,
                   const float* out_amax, float* amax_next, float* lse, ov_stream_t stream) {
    if (!qkv || !out || B <= 0 || L <= 0 || H <= 0) return OV_ERR_INVALID;
    if (hd <= 0 || hd % 8 || hd > 96) return OV_ERR_UNSUPPORTED;
    if (ld_qkv % 8 || ld_out % 8 || ld_qkv < 3 * H * hd || ld_out < H * hd) return OV_ERR_INVALID;
    if (((uintptr_t)qkv | (uintptr_t)out) & 15) return OV_ERR_INVALID;
    const bool out8 = out_amax != nullptr;
    if (lse != nullptr && (hd != 64 || (L + 31) / 32 * 32 > 288 || out8)) return OV_ERR_UNSUPPORTED;   // the resident backward's shapes only
    if (hd != 64) {                                      // So400m (72) / H (80): generic padded-head kernel
        AttnArgs g;
        g.qkv = qkv; g.ldq = ld_qkv; g.out = out; g.ldo = ld_out;
        g.B = B; g.L = L; g.H = H; g.nqt = (L + 31) / 32; g.mode = 0;
        g.scale_log2 = scale * 1.4426950408889634f;
        const int lpad = g.nqt * 32;
        // up to 320 (padded) keys: the whole head resident in LDS (123 KB at 320) and one wave per query tile (<= 10), so a head is
        // staged once by one workgroup (L = 257 used to take two 256-key chunks and a second workgroup for the 257th query row: So400m's
        // attention 8.0 ms per step); longer sequences: 256-key chunks, 8 waves
        const bool resident = lpad <= 320;
        g.KC = resident ? lpad : 256;
        const int nwg = resident ? g.nqt : 8;
        const size_t smem = (size_t)g.KC * 96 * 4;       // K (KC x 192 B) + V (3 x KC x 64 B)
        static OvPerDeviceOnce attr3;
        const int dev_attr3 = ov_current_device();
        if (attr3.need(dev_attr3)) {
            hipError_t e = hipFuncSetAttribute((const void*)attn_fwd_generic<96>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                               160 * 1024);
            if (e != hipSuccess) return OV_ERR_HIP - (int)e;
            attr3.mark(dev_attr3);
        }
        hipLaunchKernelGGL(attn_fwd_generic<96>, dim3((unsigned)(B * H), (unsigned)((g.nqt + nwg - 1) / nwg)), dim3(nwg * 64), smem,
                           (hipStream_t)stream, g, hd);
        OV_LAUNCH_CHECK();
        return OV_OK;
    }
    AttnArgs a;
    a.qkv = qkv; a.ldq = ld_qkv; a.out = out; a.ldo = ld_out;
    a.B = B; a.L = L; a.H = H;
    a.nqt = (L + 31) / 32;
    a.scale_log2 = scale * 1.4426950408889634f;
    { static int mode = -1; if (mode < 0) { const char* e = getenv("OVHIP_ATTN_MODE"); mode = e ? atoi(e) : 0; } a.mode = mode; }
    const int lp = a.nqt * 32;
    static int force_v1 = -1;
    if (force_v1 < 0) { const char* e = getenv("OVHIP_ATTN_V1"); force_v1 = (e && e[0] == '1') ? 1 : 0; }
    if (lp <= 320 && (!force_v1 || lse) && (int64_t)L * ld_qkv * 2 < 0x7fffffffLL && (int64_t)L * ld_out * 2 < 0x7fffffffLL) {   // (lse: only this kernel keeps it)
        AttnPArgs p;
        p.qkv = qkv; p.ldq = ld_qkv; p.out = out; p.ldo = ld_out;
        p.L = L; p.H = H; p.nqt = a.nqt; p.KC = lp; p.nheads = B * H; p.scale_log2 = a.scale_log2; p.out_amax = out_amax; p.amax_next = amax_next; p.lse = lse;
        { static int lk = -1; if (lk < 0) { const char* e = getenv("OVHIP_ATTN_LONEKEY"); lk = (e && e[0] == '0') ? 0 : 1; } p.lone_valu = lk; }
        static OvPerDeviceOnce attr2;
        const int dev_attr2 = ov_current_device();
        if (attr2.need(dev_attr2)) {
            hipError_t e = hipFuncSetAttribute((const void*)attn_fwd_hd64_persist<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                               160 * 1024);
            if (e == hipSuccess)
                e = hipFuncSetAttribute((const void*)attn_fwd_hd64_persist<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e == hipSuccess)
                e = hipFuncSetAttribute((const void*)attn_fwd_hd64_persist<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e == hipSuccess)
                e = hipFuncSetAttribute((const void*)attn_fwd_hd64_persist<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e == hipSuccess)
                e = hipFuncSetAttribute((const void*)attn_fwd_hd64_persist<false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e == hipSuccess)
                e = hipFuncSetAttribute((const void*)attn_fwd_hd64_persist<true, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) return OV_ERR_HIP - (int)e;
            attr2.mark(dev_attr2);
        }
        const int ncu = ov_num_cus();
        const size_t smem = (size_t)2 * lp * 256;
        int per_cu = (int)((160 * 1024) / smem);                  // workgroups per CU by LDS ...
        const int by_waves = (a.nqt <= 8 ? 8 : 12) / a.nqt;       // ... and by waves (2 per SIMD for the DEEP variant, else 3)
        if (per_cu > by_waves) per_cu = by_waves;
        if (per_cu < 1) per_cu = 1;
        const int cap = ncu * per_cu;
        const int grid = p.nheads < cap ? p.nheads : cap;
        const dim3 pg((unsigned)grid), pb(a.nqt * 64);
        if (a.nqt <= 8) {
            if (out8) hipLaunchKernelGGL((attn_fwd_hd64_persist<true, true>), pg, pb, smem, (hipStream_t)stream, p);
            else if (lse) hipLaunchKernelGGL((attn_fwd_hd64_persist<true, false, true>), pg, pb, smem, (hipStream_t)stream, p);
            else hipLaunchKernelGGL((attn_fwd_hd64_persist<true, false>), pg, pb, smem, (hipStream_t)stream, p);
        } else {
            if (out8) hipLaunchKernelGGL((attn_fwd_hd64_persist<false, true>), pg, pb, smem, (hipStream_t)stream, p);
            else if (lse) hipLaunchKernelGGL((attn_fwd_hd64_persist<false, false, true>), pg, pb, smem, (hipStream_t)stream, p);
            else hipLaunchKernelGGL((attn_fwd_hd64_persist<false, false>), pg, pb, smem, (hipStream_t)stream, p);
        }
        OV_LAUNCH_CHECK();
        return OV_OK;
    }
    if (lse != nullptr) return OV_ERR_UNSUPPORTED;       // (OVHIP_ATTN_V1 / row pitches beyond 2 GiB: no kept lse)
    if (lp > 320 && !force_v1) {
        AttnSArgs sa;
        sa.qkv = qkv; sa.ldq = ld_qkv; sa.out = out; sa.ldo = ld_out;
        sa.L = L; sa.H = H; sa.nqt = a.nqt; sa.nqb = (a.nqt + 7) / 8; sa.nheads = B * H; sa.nchunks = (L + 63) / 64;
        sa.scale_log2 = a.scale_log2;
        sa.out_amax = out_amax;
        sa.amax_next = amax_next;
        { static int lk = -1; if (lk < 0) { const char* e = getenv("OVHIP_ATTN_LONEKEY"); lk = (e && e[0] == '0') ? 0 : 1; } sa.lone_valu = lk; }
        static OvPerDeviceOnce attr4;
        const int dev_attr4 = ov_current_device();
        if (attr4.need(dev_attr4)) {
            hipError_t e = hipFuncSetAttribute((const void*)attn_fwd_hd64_stream<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e == hipSuccess)
                e = hipFuncSetAttribute((const void*)attn_fwd_hd64_stream<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) return OV_ERR_HIP - (int)e;
            attr4.mark(dev_attr4);
        }
        const int64_t heads8 = ((int64_t)sa.nheads + 7) / 8 * 8;             // whole rounds of 8 XCDs; surplus ids exit at once
        const int64_t nwg = heads8 * sa.nqb;
        if (nwg > 0x7fffffffLL) return OV_ERR_UNSUPPORTED;
        if (out8) hipLaunchKernelGGL(attn_fwd_hd64_stream<true>, dim3((unsigned)nwg), dim3(512), 4 * 16384, (hipStream_t)stream, sa);
        else hipLaunchKernelGGL(attn_fwd_hd64_stream<false>, dim3((unsigned)nwg), dim3(512), 4 * 16384, (hipStream_t)stream, sa);
        OV_LAUNCH_CHECK();
        return OV_OK;
    }
    if (out8) return OV_ERR_UNSUPPORTED;                // the fallback kernel has no e4m3 epilogue
    int nw;
    if (lp <= 320) { a.KC = lp; nw = a.nqt; }          // whole K/V of a head resident: one chunk
    else { a.KC = 256; nw = 8; }
    const int gy = (a.nqt + nw - 1) / nw;
    const size_t smem = (size_t)a.KC * 256;
    static OvPerDeviceOnce attr_set;
        const int dev_attr_set = ov_current_device();
    if (attr_set.need(dev_attr_set)) {
        hipError_t e = hipFuncSetAttribute((const void*)attn_fwd_hd64, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           160 * 1024);
        if (e != hipSuccess) return OV_ERR_HIP - (int)e;
        attr_set.mark(dev_attr_set);
    }
    hipLaunchKernelGGL(attn_fwd_hd64, dim3((unsigned)(B * H), (unsigned)gy), dim3(nw * 64), smem,
                       (hipStream_t)stream, a);
    OV_LAUNCH_CHECK();
    return OV_OK;
}
}  // namespace
