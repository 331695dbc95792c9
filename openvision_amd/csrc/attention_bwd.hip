// attention_bwd.hip — backward of the unmasked softmax attention inside nn.MultiheadAttention (gfx950, head_dim 64).
//
// The reference obtains it from autograd through nn.MultiheadAttention (open_clip/transformer.py:225,239-252; torch
// nn/functional.py scaled-dot-product path).  With S = scale Q K^T, P = softmax(S), O = P V and an upstream dO:
//     dV = P^T dO,   dP = dO V^T,   dS = P * (dP - delta),  delta[q] = sum_d dO[q,d] O[q,d],   dQ = scale dS K,   dK = scale dS^T Q.
// L <= 288: one workgroup per (image, head); Q, K, V and dO of the head live in LDS, every wave owns one 32-row tile
// (longer sequences: the two streaming kernels at the end of this file, same tile arithmetic):
//   pass 1  (wave = query tile)  log-sum-exp of its rows (the forward does not keep it) and delta        -> LDS
//   pass 2  (wave = query tile)  S^T, dP^T tiles against every key tile, dQ^T += K^T dS^T                -> dQ
//   pass 3  (wave = key tile)    S, dP tiles against every query tile, dV^T += dO^T P, dK^T += Q^T dS   -> dK, dV
// S and dP are recomputed in pass 3 (7 tile products per tile pair instead of the minimal 5) so that no accumulator is shared
// between waves: no atomics, results are deterministic.  All products are v_mfma_f32_32x32x16_bf16; P and dS are rounded to bf16
// for the second products exactly as the forward rounds P.  The LDS image of each tensor is [d half][row][32 d] (64-byte rows):
// it serves both the row-major 16-byte fragment reads (contraction over d) and ds_read_b64_tr_b16 (contraction over rows).
// 16-byte chunks are XOR-swizzled by (row >> 2) & 3 so that the ds_read_b128 lane groups are bank-conflict free.
// First version: correct and deterministic, not yet tuned (plain staging, serial LDS waits).
#include "common.h"

namespace {

struct AttnBwdArgs {
    const ov_bf16* qkv; int64_t ldq;          // [B*L, 3*H*64]  (q | k | v)
    const ov_bf16* out; int64_t ldo;          // forward output [B*L, H*64]
    const ov_bf16* dout; int64_t lddo;        // upstream gradient [B*L, H*64]
    ov_bf16* dqkv; int64_t lddq;              // [B*L, 3*H*64]  (dq | dk | dv)
    int L, H, KC;
    float scale, scale_log2;
    const float* lse_in;                      // optional: [B*H][KC] row lse kept by the forward (ov_attention_lse): pass 1 is skipped
};

__device__ __forceinline__ bf16x8_t join8(u32x2_t a, u32x2_t b) {
    const u32x4_t w = {a[0], a[1], b[0], b[1]};
    return __builtin_bit_cast(bf16x8_t, w);
}
// A-operand fragments of X^T for one 32-row tile: (d half 0, rows 0-15), (half 1, rows 0-15), (half 0, rows 16-31), (half 1,
// rows 16-31); each is two ds_read_b64_tr_b16 (rows r.. and 8 rows further, offset 512 B), transposed by the LDS unit.  All eight
// reads and their wait sit in ONE asm statement: the destination registers are complete when the statement ends.
struct TrFrags { bf16x8_t f[4]; };
__device__ __forceinline__ TrFrags tr_frags(unsigned a0, unsigned b0, unsigned a1, unsigned b1) {   // d half 0 / 1; a: rows +0/+16, b: +8/+24
    u32x2_t v0, v1, v2, v3, v4, v5, v6, v7;
    asm volatile("ds_read_b64_tr_b16 %0, %8\n\tds_read_b64_tr_b16 %1, %9 offset:512\n\t"
                 "ds_read_b64_tr_b16 %2, %10\n\tds_read_b64_tr_b16 %3, %11 offset:512\n\t"
                 "ds_read_b64_tr_b16 %4, %8 offset:1024\n\tds_read_b64_tr_b16 %5, %9 offset:1536\n\t"
                 "ds_read_b64_tr_b16 %6, %10 offset:1024\n\tds_read_b64_tr_b16 %7, %11 offset:1536\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3), "=&v"(v4), "=&v"(v5), "=&v"(v6), "=&v"(v7)
                 : "v"(a0), "v"(b0), "v"(a1), "v"(b1) : "memory");
    TrFrags t;
    t.f[0] = join8(v0, v1); t.f[1] = join8(v2, v3); t.f[2] = join8(v4, v5); t.f[3] = join8(v6, v7);
    return t;
}
__device__ __forceinline__ bf16x8_t pack8(const f32x16_t& t, int s) {
    const u32x4_t w = {pack_bf16x2(t[8 * s + 0], t[8 * s + 1]), pack_bf16x2(t[8 * s + 2], t[8 * s + 3]),
                       pack_bf16x2(t[8 * s + 4], t[8 * s + 5]), pack_bf16x2(t[8 * s + 6], t[8 * s + 7])};
    return __builtin_bit_cast(bf16x8_t, w);
}
__device__ __forceinline__ float swap_halves(float v) { return __shfl_xor(v, 32, 64); }

template <bool SAVED>
__global__ __launch_bounds__(640) void attn_bwd_hd64(const AttnBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h2 = lane >> 5;
    const int L = a.L, KC = a.KC, HD = a.H * 64;
    const int b = blockIdx.x / a.H, h = blockIdx.x - b * a.H;
    const int img_bytes = KC * 128;
    char* qimg = smem;
    char* kimg = smem + img_bytes;
    char* vimg = smem + 2 * img_bytes;
    char* dimg = smem + 3 * img_bytes;
    float* lse = (float*)(smem + 4 * img_bytes);
    float* dlt = lse + KC;

    // ---- stage the four [KC, 64] tiles of this head: piece p of an image = (d half, row, 16-byte chunk)
    {
        const int npiece = KC * 8;
        const ov_bf16* qbase = a.qkv + (int64_t)b * L * a.ldq + h * 64;
        const ov_bf16* dbase = a.dout + (int64_t)b * L * a.lddo + h * 64;
        for (int q = tid; q < 4 * npiece; q += blockDim.x) {
            const int t = q / npiece, p = q - t * npiece;
            const int dh = p / (KC * 4), pp = p - dh * KC * 4;
            int row = pp >> 2;
            row = row < L ? row : L - 1;
            const int col = (dh * 4 + (pp & 3)) * 8;
            const ov_bf16* src = t < 3 ? qbase + (int64_t)row * a.ldq + t * HD + col : dbase + (int64_t)row * a.lddo + col;
            // 16-byte chunk c of LDS row `pos` sits at chunk c ^ ((pos >> 2) & 3): the 16-lane groups of ds_read_b128 (rows 0-3, 12-15,
            // 20-27 / 4-11, 16-19, 28-31) then cover all 16 slots of the 256-byte bank row; ds_read_b64_tr_b16 stays conflict-free
            const int pos = pp >> 2;
            *(u32x4_t*)(smem + t * img_bytes + dh * KC * 64 + pos * 64 + (((pp & 3) ^ ((pos >> 2) & 3)) << 4)) = *(const u32x4_t*)src;
        }
    }
    __syncthreads();

    // row-major fragment (contraction over d): row = tile * 32 + r, d = 16 st + 8 h2 .. + 8
    auto frag = [&](const char* img, int tile, int st) {
        return *(const bf16x8_t*)(img + (st >> 1) * KC * 64 + (tile * 32 + r) * 64 + (((((st & 1) << 1) | h2) ^ ((r >> 2) & 3)) << 4));
    };
    const int vi = lane & 15, vg = (lane >> 4) & 1;
    // transposing reads: lane row 4 h2 + (vi >> 2) (+ 8 k), logical chunk 2 vg + ((vi & 3) >> 1), 8-byte half vi & 1 of it; the chunk
    // swizzle of that row is (h2 + 2 k) & 3: h2 for the row offsets 0 / 16 (immediates 0, 1024), h2 ^ 2 for 8 / 24 (512, 1536)
    const unsigned tr_row = (unsigned)((4 * h2 + (vi >> 2)) * 64 + 8 * (vi & 1));
    const unsigned tr_ch = (unsigned)(2 * vg + ((vi & 3) >> 1));
    const unsigned tr_lane_a = tr_row + ((tr_ch ^ (unsigned)h2) << 4), tr_lane_b = tr_row + ((tr_ch ^ (unsigned)h2 ^ 2u) << 4);
    auto tr_tile = [&](const char* img, int tile) {          // the four X^T fragments of rows tile*32 .. +32
        const unsigned base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const char*)img + (unsigned)(tile * 2048);
        return tr_frags(base + tr_lane_a, base + tr_lane_b, base + (unsigned)(KC * 64) + tr_lane_a, base + (unsigned)(KC * 64) + tr_lane_b);
    };
    const int nt = KC >> 5;                       // tiles = waves

    // ================= passes 1 and 2: this wave's QUERY tile =================
    {
        const int i = wave;
        const int query = i * 32 + r;
        bf16x8_t qB[4], dB[4];
#pragma unroll
        for (int st = 0; st < 4; ++st) qB[st] = frag(qimg, i, st);      // (dB is read behind pass 1: 16 registers fewer across it)
        // pass 1: lse (log2 units) of row `query`; a lane sees the keys (t&3) + 8 (t>>2) + 4 h2 of each tile
        // (SAVED: the forward kept it -- no score pass, no exponentials here)
        float m = -INFINITY, l = 0.f;
        for (int j = 0; j < (SAVED ? 0 : nt); ++j) {
            f32x16_t s;
#pragma unroll
            for (int t = 0; t < 16; ++t) s[t] = 0.f;
#pragma unroll
            for (int st = 0; st < 4; ++st) s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag(kimg, j, st), qB[st], s, 0, 0, 0);
            float mx = -INFINITY;
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const int key = j * 32 + (t & 3) + 8 * (t >> 2) + 4 * h2;
                s[t] = key < L ? s[t] * a.scale_log2 : -INFINITY;
                mx = fmaxf(mx, s[t]);
            }
            if (mx > -INFINITY) {
                const float mn = fmaxf(m, mx);
                float ps = 0.f;
#pragma unroll
                for (int t = 0; t < 16; ++t) ps += __builtin_amdgcn_exp2f(s[t] - mn);
                l = l * (m > -INFINITY ? __builtin_amdgcn_exp2f(m - mn) : 0.f) + ps;
                m = mn;
            }
        }
#pragma unroll
        for (int st = 0; st < 4; ++st) dB[st] = frag(dimg, i, st);
        float lse2;
        if (SAVED) {
            // the forward writes entries [0, L) only (padded query lanes re-store row L - 1 at index L - 1): entries [L, KC) are
            // whatever the recycled buffer held, possibly NaN / Inf -- never read them (p = 0 is selected for q >= L further down)
            lse2 = query < L ? a.lse_in[(int64_t)blockIdx.x * KC + query] : 0.f;
        } else {
            const float mo = swap_halves(m), lo = swap_halves(l);
            const float mn = fmaxf(m, mo);                       // finite: key 0 is always valid for one of the halves
            l = l * (m > -INFINITY ? __builtin_amdgcn_exp2f(m - mn) : 0.f) + lo * (mo > -INFINITY ? __builtin_amdgcn_exp2f(mo - mn) : 0.f);
            m = mn;
            lse2 = m + __builtin_amdgcn_logf(l);                 // v_log_f32 = log2
        }
        // delta = sum_d dO[q, d] O[q, d]: this lane takes d half h2
        float delta = 0.f;
        {
            const int qrow = query < L ? query : L - 1;
            const ov_bf16* op = a.out + ((int64_t)b * L + qrow) * a.ldo + h * 64 + 32 * h2;
            const char* dp = dimg + h2 * KC * 64 + query * 64;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const u32x4_t ov = *(const u32x4_t*)(op + 8 * c);
                const u32x4_t dv = *(const u32x4_t*)(dp + ((c ^ ((query >> 2) & 3)) << 4));
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    delta = fmaf(bf16lo_to_f32(ov[e]), bf16lo_to_f32(dv[e]), delta);
                    delta = fmaf(bf16hi_to_f32(ov[e]), bf16hi_to_f32(dv[e]), delta);
                }
            }
            delta += swap_halves(delta);
        }
        if (h2 == 0) { lse[query] = lse2; dlt[query] = delta; }

        // pass 2: dQ^T[d, query] += K_j^T[d, key] dS^T[key, query]
        f32x16_t dq0, dq1;
#pragma unroll
        for (int t = 0; t < 16; ++t) { dq0[t] = 0.f; dq1[t] = 0.f; }
        for (int j = 0; j < nt; ++j) {
            f32x16_t s, dp;
#pragma unroll
            for (int t = 0; t < 16; ++t) { s[t] = 0.f; dp[t] = 0.f; }
#pragma unroll
            for (int st = 0; st < 4; ++st) s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag(kimg, j, st), qB[st], s, 0, 0, 0);
#pragma unroll
            for (int st = 0; st < 4; ++st) dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag(vimg, j, st), dB[st], dp, 0, 0, 0);
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const int key = j * 32 + (t & 3) + 8 * (t >> 2) + 4 * h2;
                const float p = key < L ? __builtin_amdgcn_exp2f(fmaf(s[t], a.scale_log2, -lse2)) : 0.f;
                s[t] = p * (dp[t] - delta);                      // dS^T
            }
            const bf16x8_t b0 = pack8(s, 0), b1 = pack8(s, 1);
            const TrFrags kt = tr_tile(kimg, j);
            dq0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kt.f[0], b0, dq0, 0, 0, 0);
            dq1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kt.f[1], b0, dq1, 0, 0, 0);
            dq0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kt.f[2], b1, dq0, 0, 0, 0);
            dq1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kt.f[3], b1, dq1, 0, 0, 0);
        }
        if (query < L) {                                         // rows d = 8 g + 4 h2 + e (dq0), + 32 (dq1)
            ov_bf16* op = a.dqkv + ((int64_t)b * L + query) * a.lddq + h * 64 + 4 * h2;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const u32x2_t w0 = {pack_bf16x2(dq0[4 * g] * a.scale, dq0[4 * g + 1] * a.scale), pack_bf16x2(dq0[4 * g + 2] * a.scale, dq0[4 * g + 3] * a.scale)};
                const u32x2_t w1 = {pack_bf16x2(dq1[4 * g] * a.scale, dq1[4 * g + 1] * a.scale), pack_bf16x2(dq1[4 * g + 2] * a.scale, dq1[4 * g + 3] * a.scale)};
                *(u32x2_t*)(op + 8 * g) = w0;
                *(u32x2_t*)(op + 32 + 8 * g) = w1;
            }
        }
    }
    __syncthreads();                                             // every row's lse / delta is in LDS

    // ================= pass 3: this wave's KEY tile =================
    {
        const int j = wave;
        const int key = j * 32 + r;
        bf16x8_t kB[4], vB[4];
#pragma unroll
        for (int st = 0; st < 4; ++st) { kB[st] = frag(kimg, j, st); vB[st] = frag(vimg, j, st); }
        f32x16_t dk0, dk1, dv0, dv1;
#pragma unroll
        for (int t = 0; t < 16; ++t) { dk0[t] = 0.f; dk1[t] = 0.f; dv0[t] = 0.f; dv1[t] = 0.f; }
        for (int i = 0; i < nt; ++i) {
            f32x16_t s, dp;
#pragma unroll
            for (int t = 0; t < 16; ++t) { s[t] = 0.f; dp[t] = 0.f; }
#pragma unroll
            for (int st = 0; st < 4; ++st) s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag(qimg, i, st), kB[st], s, 0, 0, 0);
#pragma unroll
            for (int st = 0; st < 4; ++st) dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag(dimg, i, st), vB[st], dp, 0, 0, 0);
            // s[t], dp[t]: query i*32 + (t&3) + 8 (t>>2) + 4 h2 (register), key = lane column
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const int q = i * 32 + (t & 3) + 8 * (t >> 2) + 4 * h2;
                const float p = q < L ? __builtin_amdgcn_exp2f(fmaf(s[t], a.scale_log2, -lse[q])) : 0.f;
                s[t] = p;                                        // P
                dp[t] = p * (dp[t] - dlt[q]);                    // dS
            }
            const bf16x8_t p0 = pack8(s, 0), p1 = pack8(s, 1), g0 = pack8(dp, 0), g1 = pack8(dp, 1);
            const TrFrags dt = tr_tile(dimg, i);
            dv0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dt.f[0], p0, dv0, 0, 0, 0);
            dv1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dt.f[1], p0, dv1, 0, 0, 0);
            dv0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dt.f[2], p1, dv0, 0, 0, 0);
            dv1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dt.f[3], p1, dv1, 0, 0, 0);
            const TrFrags qt = tr_tile(qimg, i);
            dk0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qt.f[0], g0, dk0, 0, 0, 0);
            dk1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qt.f[1], g0, dk1, 0, 0, 0);
            dk0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qt.f[2], g1, dk0, 0, 0, 0);
            dk1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qt.f[3], g1, dk1, 0, 0, 0);
        }
        if (key < L) {
            ov_bf16* kp = a.dqkv + ((int64_t)b * L + key) * a.lddq + HD + h * 64 + 4 * h2;
            ov_bf16* vp = kp + HD;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const u32x2_t k0 = {pack_bf16x2(dk0[4 * g] * a.scale, dk0[4 * g + 1] * a.scale), pack_bf16x2(dk0[4 * g + 2] * a.scale, dk0[4 * g + 3] * a.scale)};
                const u32x2_t k1 = {pack_bf16x2(dk1[4 * g] * a.scale, dk1[4 * g + 1] * a.scale), pack_bf16x2(dk1[4 * g + 2] * a.scale, dk1[4 * g + 3] * a.scale)};
                const u32x2_t v0 = {pack_bf16x2(dv0[4 * g], dv0[4 * g + 1]), pack_bf16x2(dv0[4 * g + 2], dv0[4 * g + 3])};
                const u32x2_t v1 = {pack_bf16x2(dv1[4 * g], dv1[4 * g + 1]), pack_bf16x2(dv1[4 * g + 2], dv1[4 * g + 3])};
                *(u32x2_t*)(kp + 8 * g) = k0;
                *(u32x2_t*)(kp + 32 + 8 * g) = k1;
                *(u32x2_t*)(vp + 8 * g) = v0;
                *(u32x2_t*)(vp + 32 + 8 * g) = v1;
            }
        }
    }
}


// ---- long sequences (L > 288: S/8@384 has 2305 tokens) and head dims 72 / 80: two streaming kernels, same tile arithmetic ---------
// A workgroup owns eight 32-row tiles (its fragments come straight from global memory into registers) and streams the OTHER side
// through LDS in 256-row chunks of two images:
//   attn_bwd_stream_q   own = query tiles; chunks of K (pass 1: lse) then K and V (pass 2: dQ); writes lse, delta to global
//   attn_bwd_stream_kv  own = key tiles; chunks of Q and dO with their lse / delta; writes dK, dV
// NDH = 32-wide d slices of an image (2: head_dim 64; 3: head dims 72 / 80 zero-padded to 96, So400m and H/14).
constexpr int CH = 256;                       // chunk rows

struct AttnBwdSArgs {
    AttnBwdArgs a;
    float* lse;                               // [B*H, Lpad] log2-domain row lse
    float* dlt;                               // [B*H, Lpad]
    int Lpad, nblk, hd;
};

template <int NDH>
struct Stream {
    static constexpr int IMG = NDH * CH * 64;                     // bytes of one chunk image
    // stage rows [row0, row0 + 256) (clamped to L - 1) of one or two [L, hd] column blocks into the images; columns >= hd are zeros
    static __device__ __forceinline__ void stage(char* img0, char* img1, const ov_bf16* src0, const ov_bf16* src1, int64_t ld0, int64_t ld1,
                                                 int row0, int L, int hd, int tid, int nthreads, int nimg) {
        constexpr int NP = CH * NDH * 4;                          // 16-byte pieces per image
        for (int q = tid; q < nimg * NP; q += nthreads) {
            const int t = q / NP, p = q - t * NP;
            const int dh = p / (CH * 4), pp = p - dh * CH * 4;
            const int pos = pp >> 2;
            int row = row0 + pos;
            row = row < L ? row : L - 1;
            const int col = (dh * 4 + (pp & 3)) * 8;
            u32x4_t v = {0u, 0u, 0u, 0u};
            if (col < hd) v = *(const u32x4_t*)(t == 0 ? src0 + (int64_t)row * ld0 + col : src1 + (int64_t)row * ld1 + col);
            *(u32x4_t*)((t == 0 ? img0 : img1) + dh * CH * 64 + pos * 64 + (((pp & 3) ^ ((pos >> 2) & 3)) << 4)) = v;
        }
    }
    // this lane's B-operand fragments of row `row` of a [L, hd] column block: d = 16 st + 8 h2 .. + 8 (zeros past hd)
    static __device__ __forceinline__ void load_own(bf16x8_t (&f)[2 * NDH], const ov_bf16* base, int64_t ld, int row, int h2, int hd) {
#pragma unroll
        for (int st = 0; st < 2 * NDH; ++st) {
            u32x4_t v = {0u, 0u, 0u, 0u};
            if (16 * st + 8 * h2 < hd) v = *(const u32x4_t*)(base + (int64_t)row * ld + 16 * st + 8 * h2);
            f[st] = __builtin_bit_cast(bf16x8_t, v);
        }
    }
    // store rows d = 32 dh + 8 g + 4 h2 + e of the accumulators (lane = output row of the tensor) scaled by `sc`
    static __device__ __forceinline__ void store(ov_bf16* op, const f32x16_t (&acc)[NDH], float sc, int h2, int hd) {
#pragma unroll
        for (int dh = 0; dh < NDH; ++dh)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const int d0 = 32 * dh + 8 * gq + 4 * h2;
                if (d0 < hd) {
                    const u32x2_t w = {pack_bf16x2(acc[dh][4 * gq] * sc, acc[dh][4 * gq + 1] * sc),
                                       pack_bf16x2(acc[dh][4 * gq + 2] * sc, acc[dh][4 * gq + 3] * sc)};
                    *(u32x2_t*)(op + d0) = w;
                }
            }
    }
};

// two A-operand fragments (contraction steps 0 and 1) of one d slice of X^T for a 32-row tile; reads and wait in ONE asm statement
__device__ __forceinline__ void tr_half(unsigned a0, unsigned b0, bf16x8_t& f0, bf16x8_t& f1) {
    u32x2_t v0, v1, v2, v3;
    asm volatile("ds_read_b64_tr_b16 %0, %4\n\tds_read_b64_tr_b16 %1, %5 offset:512\n\t"
                 "ds_read_b64_tr_b16 %2, %4 offset:1024\n\tds_read_b64_tr_b16 %3, %5 offset:1536\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3) : "v"(a0), "v"(b0) : "memory");
    f0 = join8(v0, v1);
    f1 = join8(v2, v3);
}

#define OV_CHUNK_HELPERS()                                                                                                            \
    auto frag = [&](const char* img, int tile, int st) {                                                                              \
        return *(const bf16x8_t*)(img + (st >> 1) * CH * 64 + (tile * 32 + r) * 64 + (((((st & 1) << 1) | h2) ^ ((r >> 2) & 3)) << 4)); \
    };                                                                                                                                \
    const int vi = lane & 15, vg = (lane >> 4) & 1;                                                                                   \
    const unsigned tr_row = (unsigned)((4 * h2 + (vi >> 2)) * 64 + 8 * (vi & 1));                                                     \
    const unsigned tr_ch = (unsigned)(2 * vg + ((vi & 3) >> 1));                                                                      \
    const unsigned tr_lane_a = tr_row + ((tr_ch ^ (unsigned)h2) << 4), tr_lane_b = tr_row + ((tr_ch ^ (unsigned)h2 ^ 2u) << 4);     \
    /* acc[dh] += X^T(d slice dh, rows of `tile`) . T  with T packed as b0 (rows 0-15) / b1 (rows 16-31) */                           \
    auto tr_mma = [&](const char* img, int tile, bf16x8_t b0, bf16x8_t b1, f32x16_t (&acc)[NDH]) {                                    \
        const unsigned base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const char*)img + (unsigned)(tile * 2048);      \
        _Pragma("unroll") for (int dh = 0; dh < NDH; ++dh) {                                                                          \
            bf16x8_t f0, f1;                                                                                                          \
            tr_half(base + (unsigned)(dh * CH * 64) + tr_lane_a, base + (unsigned)(dh * CH * 64) + tr_lane_b, f0, f1);                \
            acc[dh] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f0, b0, acc[dh], 0, 0, 0);                                              \
            acc[dh] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f1, b1, acc[dh], 0, 0, 0);                                              \
        }                                                                                                                             \
    };

template <int NDH>
__global__ __launch_bounds__(512) void attn_bwd_stream_q(const AttnBwdSArgs g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef Stream<NDH> St;
    const AttnBwdArgs& a = g.a;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h2 = lane >> 5;
    const int L = a.L, hd = g.hd, HD = a.H * hd;
    const int bh = blockIdx.x / g.nblk, blk = blockIdx.x - bh * g.nblk;
    const int b = bh / a.H, h = bh - b * a.H;
    char* img0 = smem;
    char* img1 = smem + St::IMG;
    OV_CHUNK_HELPERS()
    const ov_bf16* qbase = a.qkv + (int64_t)b * L * a.ldq + h * hd;
    const ov_bf16* dbase = a.dout + (int64_t)b * L * a.lddo + h * hd;
    const int query = blk * CH + wave * 32 + r;
    const int qrow = query < L ? query : L - 1;
    bf16x8_t qB[2 * NDH], dB[2 * NDH];
    St::load_own(qB, qbase, a.ldq, qrow, h2, hd);
    St::load_own(dB, dbase, a.lddo, qrow, h2, hd);
    const int nchunk = (L + CH - 1) / CH;

    // pass 1: lse over all keys
    float m = -INFINITY, l = 0.f;
    for (int ck = 0; ck < nchunk; ++ck) {
        __syncthreads();
        St::stage(img0, img1, qbase + HD, qbase + HD, a.ldq, a.ldq, ck * CH, L, hd, tid, blockDim.x, 1);
        __syncthreads();
        const int ntile = (min(L - ck * CH, CH) + 31) >> 5;
        for (int j = 0; j < ntile; ++j) {
            f32x16_t s;
#pragma unroll
            for (int t = 0; t < 16; ++t) s[t] = 0.f;
#pragma unroll
            for (int st = 0; st < 2 * NDH; ++st) s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag(img0, j, st), qB[st], s, 0, 0, 0);
            float mx = -INFINITY;
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const int key = ck * CH + j * 32 + (t & 3) + 8 * (t >> 2) + 4 * h2;
                s[t] = key < L ? s[t] * a.scale_log2 : -INFINITY;
                mx = fmaxf(mx, s[t]);
            }
            if (mx > -INFINITY) {
                const float mn = fmaxf(m, mx);
                float ps = 0.f;
#pragma unroll
                for (int t = 0; t < 16; ++t) ps += __builtin_amdgcn_exp2f(s[t] - mn);
                l = l * (m > -INFINITY ? __builtin_amdgcn_exp2f(m - mn) : 0.f) + ps;
                m = mn;
            }
        }
    }
    {
        const float mo = swap_halves(m), lo = swap_halves(l);
        const float mn = fmaxf(m, mo);
        l = l * (m > -INFINITY ? __builtin_amdgcn_exp2f(m - mn) : 0.f) + lo * (mo > -INFINITY ? __builtin_amdgcn_exp2f(mo - mn) : 0.f);
        m = mn;
    }
    const float lse2 = m + __builtin_amdgcn_logf(l);
    float delta = 0.f;
    {   // delta = sum_d dO[q, d] O[q, d]: the two lane halves take alternate 8-element chunks
        const ov_bf16* op = a.out + ((int64_t)b * L + qrow) * a.ldo + h * hd;
        const ov_bf16* dp = dbase + (int64_t)qrow * a.lddo;
        for (int cc = h2; cc < (hd >> 3); cc += 2) {
            const u32x4_t ov = *(const u32x4_t*)(op + 8 * cc);
            const u32x4_t dv = *(const u32x4_t*)(dp + 8 * cc);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                delta = fmaf(bf16lo_to_f32(ov[e]), bf16lo_to_f32(dv[e]), delta);
                delta = fmaf(bf16hi_to_f32(ov[e]), bf16hi_to_f32(dv[e]), delta);
            }
        }
        delta += swap_halves(delta);
    }
    if (h2 == 0 && query < g.Lpad) { g.lse[(int64_t)bh * g.Lpad + query] = lse2; g.dlt[(int64_t)bh * g.Lpad + query] = delta; }

    // pass 2: dQ
    f32x16_t dq[NDH];
#pragma unroll
    for (int dh = 0; dh < NDH; ++dh)
#pragma unroll
        for (int t = 0; t < 16; ++t) dq[dh][t] = 0.f;
    for (int ck = 0; ck < nchunk; ++ck) {
        __syncthreads();
        St::stage(img0, img1, qbase + HD, qbase + 2 * HD, a.ldq, a.ldq, ck * CH, L, hd, tid, blockDim.x, 2);
        __syncthreads();
        const int ntile = (min(L - ck * CH, CH) + 31) >> 5;
        for (int j = 0; j < ntile; ++j) {
            f32x16_t s, dp;
#pragma unroll
            for (int t = 0; t < 16; ++t) { s[t] = 0.f; dp[t] = 0.f; }
#pragma unroll
            for (int st = 0; st < 2 * NDH; ++st) s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag(img0, j, st), qB[st], s, 0, 0, 0);
#pragma unroll
            for (int st = 0; st < 2 * NDH; ++st) dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag(img1, j, st), dB[st], dp, 0, 0, 0);
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const int key = ck * CH + j * 32 + (t & 3) + 8 * (t >> 2) + 4 * h2;
                const float p = key < L ? __builtin_amdgcn_exp2f(fmaf(s[t], a.scale_log2, -lse2)) : 0.f;
                s[t] = p * (dp[t] - delta);
            }
            tr_mma(img0, j, pack8(s, 0), pack8(s, 1), dq);
        }
    }
    if (query < L) St::store(a.dqkv + ((int64_t)b * L + query) * a.lddq + h * hd, dq, a.scale, h2, hd);
}

template <int NDH>
__global__ __launch_bounds__(512) void attn_bwd_stream_kv(const AttnBwdSArgs g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef Stream<NDH> St;
    const AttnBwdArgs& a = g.a;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h2 = lane >> 5;
    const int L = a.L, hd = g.hd, HD = a.H * hd;
    const int bh = blockIdx.x / g.nblk, blk = blockIdx.x - bh * g.nblk;
    const int b = bh / a.H, h = bh - b * a.H;
    char* img0 = smem;
    char* img1 = smem + St::IMG;
    float* lse_s = (float*)(smem + 2 * St::IMG);
    float* dlt_s = lse_s + CH;
    OV_CHUNK_HELPERS()
    const ov_bf16* qbase = a.qkv + (int64_t)b * L * a.ldq + h * hd;
    const ov_bf16* dbase = a.dout + (int64_t)b * L * a.lddo + h * hd;
    const int key = blk * CH + wave * 32 + r;
    const int krow = key < L ? key : L - 1;
    bf16x8_t kB[2 * NDH], vB[2 * NDH];
    St::load_own(kB, qbase + HD, a.ldq, krow, h2, hd);
    St::load_own(vB, qbase + 2 * HD, a.ldq, krow, h2, hd);
    f32x16_t dk[NDH], dv[NDH];
#pragma unroll
    for (int dh = 0; dh < NDH; ++dh)
#pragma unroll
        for (int t = 0; t < 16; ++t) { dk[dh][t] = 0.f; dv[dh][t] = 0.f; }
    const int nchunk = (L + CH - 1) / CH;
    for (int ck = 0; ck < nchunk; ++ck) {
        __syncthreads();
        St::stage(img0, img1, qbase, dbase, a.ldq, a.lddo, ck * CH, L, hd, tid, blockDim.x, 2);
        for (int q = tid; q < CH; q += blockDim.x) {
            const int qq = ck * CH + q;
            const int qc = qq < g.Lpad ? qq : g.Lpad - 1;
            lse_s[q] = g.lse[(int64_t)bh * g.Lpad + qc];
            dlt_s[q] = g.dlt[(int64_t)bh * g.Lpad + qc];
        }
        __syncthreads();
        const int ntile = (min(L - ck * CH, CH) + 31) >> 5;
        for (int i = 0; i < ntile; ++i) {
            f32x16_t s, dp;
#pragma unroll
            for (int t = 0; t < 16; ++t) { s[t] = 0.f; dp[t] = 0.f; }
#pragma unroll
            for (int st = 0; st < 2 * NDH; ++st) s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag(img0, i, st), kB[st], s, 0, 0, 0);
#pragma unroll
            for (int st = 0; st < 2 * NDH; ++st) dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag(img1, i, st), vB[st], dp, 0, 0, 0);
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const int ql = i * 32 + (t & 3) + 8 * (t >> 2) + 4 * h2;
                const float p = ck * CH + ql < L ? __builtin_amdgcn_exp2f(fmaf(s[t], a.scale_log2, -lse_s[ql])) : 0.f;
                s[t] = p;
                dp[t] = p * (dp[t] - dlt_s[ql]);
            }
            tr_mma(img1, i, pack8(s, 0), pack8(s, 1), dv);
            tr_mma(img0, i, pack8(dp, 0), pack8(dp, 1), dk);
        }
    }
    if (key < L) {
        ov_bf16* kp = a.dqkv + ((int64_t)b * L + key) * a.lddq + HD + h * hd;
        St::store(kp, dk, a.scale, h2, hd);
        St::store(kp + HD, dv, 1.0f, h2, hd);
    }
}
#undef OV_CHUNK_HELPERS

}  // namespace

extern "C" size_t ov_attention_backward_workspace_bytes(int B, int L, int H, int hd) {
    if (B <= 0 || L <= 0 || H <= 0 || (hd == 64 && L <= 288)) return 0;       // the resident kernel needs none
    const int64_t lpad = (int64_t)(L + CH - 1) / CH * CH;
    return (size_t)2 * B * H * lpad * sizeof(float);
}

namespace {
template <int NDH>
int launch_stream(const AttnBwdSArgs& g, int B, hipStream_t st) {
    static OvPerDeviceOnce attr;
    const int dev = ov_current_device();
    const size_t smem = (size_t)2 * Stream<NDH>::IMG + (size_t)2 * CH * sizeof(float);
    if (attr.need(dev)) {
        hipError_t e = hipFuncSetAttribute((const void*)attn_bwd_stream_q<NDH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)attn_bwd_stream_kv<NDH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return ov_hip(e);
        attr.mark(dev);
    }
    const dim3 grid((unsigned)(B * g.a.H * g.nblk));
    hipLaunchKernelGGL(attn_bwd_stream_q<NDH>, grid, dim3(512), smem, st, g);
    OV_LAUNCH_CHECK();
    hipLaunchKernelGGL(attn_bwd_stream_kv<NDH>, grid, dim3(512), smem, st, g);
    OV_LAUNCH_CHECK();
    return OV_OK;
}
}  // namespace

// lse (or NULL): the forward's row statistics from ov_attention_lse, [B*H][L rounded up to 32]; used by the resident kernel
// (head_dim 64, L <= 288), ignored by the streaming kernels
extern "C" int ov_attention_backward_saved(const ov_bf16* qkv, int64_t ld_qkv, const ov_bf16* out, int64_t ld_out, const ov_bf16* dout,
                                           int64_t ld_dout, ov_bf16* dqkv, int64_t ld_dqkv, const float* lse, int B, int L, int H, int hd,
                                           float scale, void* workspace, size_t workspace_bytes, ov_stream_t stream) {
    if (!qkv || !out || !dout || !dqkv || B <= 0 || L <= 0 || H <= 0) return OV_ERR_INVALID;
    if (hd <= 0 || hd % 8 || hd > 96) return OV_ERR_UNSUPPORTED;
    if (ld_qkv % 8 || ld_out % 8 || ld_dout % 8 || ld_dqkv % 8 || ld_qkv < 3 * H * hd || ld_dqkv < 3 * H * hd || ld_out < H * hd ||
        ld_dout < H * hd)
        return OV_ERR_UNSUPPORTED;
    if (((uintptr_t)qkv | (uintptr_t)out | (uintptr_t)dout | (uintptr_t)dqkv) & 15) return OV_ERR_INVALID;
    if ((int64_t)B * H > 0x7fffffffLL) return OV_ERR_UNSUPPORTED;
    AttnBwdArgs a;
    a.qkv = qkv; a.ldq = ld_qkv; a.out = out; a.ldo = ld_out; a.dout = dout; a.lddo = ld_dout; a.dqkv = dqkv; a.lddq = ld_dqkv;
    a.L = L; a.H = H; a.KC = (L + 31) / 32 * 32;
    a.scale = scale; a.scale_log2 = scale * 1.4426950408889634f;
    a.lse_in = lse;
    if (hd == 64 && L <= 288) {                                 // Q, K, V, dO of a head resident in LDS
        static OvPerDeviceOnce attr;
        const int dev = ov_current_device();
        if (attr.need(dev)) {
            hipError_t e = hipFuncSetAttribute((const void*)attn_bwd_hd64<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e == hipSuccess) e = hipFuncSetAttribute((const void*)attn_bwd_hd64<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) return ov_hip(e);
            attr.mark(dev);
        }
        const size_t smem = (size_t)4 * a.KC * 128 + (size_t)2 * a.KC * sizeof(float);
        const dim3 grid((unsigned)(B * H)), blk((unsigned)(a.KC / 32 * 64));
        if (lse != nullptr) hipLaunchKernelGGL(attn_bwd_hd64<true>, grid, blk, smem, (hipStream_t)stream, a);
        else hipLaunchKernelGGL(attn_bwd_hd64<false>, grid, blk, smem, (hipStream_t)stream, a);
        OV_LAUNCH_CHECK();
        return OV_OK;
    }
    if (!workspace || ((uintptr_t)workspace & 15)) return OV_ERR_INVALID;
    if (workspace_bytes < ov_attention_backward_workspace_bytes(B, L, H, hd)) return OV_ERR_WORKSPACE;
    AttnBwdSArgs g;
    g.a = a;
    g.hd = hd;
    g.nblk = (L + CH - 1) / CH;
    g.Lpad = g.nblk * CH;
    g.lse = (float*)workspace;
    g.dlt = g.lse + (size_t)B * H * g.Lpad;
    if ((int64_t)B * H * g.nblk > 0x7fffffffLL) return OV_ERR_UNSUPPORTED;
    return hd <= 64 ? launch_stream<2>(g, B, (hipStream_t)stream) : launch_stream<3>(g, B, (hipStream_t)stream);
}

extern "C" int ov_attention_backward(const ov_bf16* qkv, int64_t ld_qkv, const ov_bf16* out, int64_t ld_out, const ov_bf16* dout,
                                     int64_t ld_dout, ov_bf16* dqkv, int64_t ld_dqkv, int B, int L, int H, int hd, float scale,
                                     void* workspace, size_t workspace_bytes, ov_stream_t stream) {
    return ov_attention_backward_saved(qkv, ld_qkv, out, ld_out, dout, ld_dout, dqkv, ld_dqkv, nullptr, B, L, H, hd, scale, workspace,
                                       workspace_bytes, stream);
}
