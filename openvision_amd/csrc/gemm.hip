// gemm.hip — bf16 MFMA GEMM with fused epilogues for gfx950 (MI355X).
//
//   C[M,N] = epilogue(A[M,K] . W[N,K]^T + bias)       A, W, C bf16 row-major (K contiguous), fp32 accumulate
//
// Replaces the aten addmm behind every nn.Linear / `@ proj` on the path (reference
// open_clip/transformer.py:225 in_proj/out_proj, :232-236 c_fc/c_proj, :645-646 proj; model.py:278-282).
//
// Structure (one workgroup = 8 waves = one 256x256 output tile, BK = 64):
//   * operands staged HBM/L2 -> LDS with global_load_lds_dwordx4 (no VGPR round trip), two stages;
//     the LDS image is lane-linear, so the bank swizzle (16-B chunk ^= row & 7) is applied to the
//     per-lane SOURCE address and again on the ds_read_b128 side;
//   * each wave owns a 128(m) x 64(n) sub-tile = 8 x 4 v_mfma_f32_16x16x32_bf16 accumulators;
//     W is fed as the MFMA A operand so a lane ends up holding 4 consecutive n for one m;
//   * epilogue: bias / GELU in fp32 on the accumulators, bf16 pack, transpose through (swizzled) LDS,
//     then 16-byte row-contiguous stores with the residual added on the way out;
//   * workgroup id -> tile map is XCD-aware: each XCD's L2 sees a contiguous run of tiles that walk n
//     fastest, so the 32 tiles resident on an XCD share A row-panels and the whole of W.
#include "common.h"
#include <stdlib.h>

namespace {

constexpr int BM = 256, BN = 256, BK = 64;
constexpr int NTHREADS = 512;
constexpr int TILE_BYTES = BM * BK * 2;        // one operand tile: 32 KiB
constexpr int STAGE_BYTES = 2 * TILE_BYTES;    // A + W
constexpr int SMEM_BYTES = 2 * STAGE_BYTES;    // double buffered: 128 KiB

struct GemmArgs {
    const ov_bf16* A; const ov_bf16* W; const float* bias; ov_bf16* C; const ov_bf16* R;
    int64_t lda, ldw, ldc, ldr, M;
    int N, K, tiles_m, tiles_n, out_group, resid_mod, resid_off;
    const float* colsum;            // LN fold: column sums of W' (NULL = plain GEMM)
    const float* rowstats;          // LN fold: {mean, rstd} per row of A
    unsigned long long* stamps;     // diagnostics only (ov_debug_gemm_stamps): [block][tile slot][4] s_memtime values
    int stamp_slots;
};

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// ---- shared epilogue: acc[i][j][r] = C[m0 + wm*128 + i*16 + fr][n0 + wn*64 + j*16 + fq*4 + r] -------------------
template <int EPI>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& g, f32x4_t (&acc)[8][4], char* smem, int64_t m0, int n0,
                                              int wave, int lane) {
    const int wm = wave >> 2, wn = wave & 3;
    const int fr = lane & 15, fq = lane >> 4;
    __syncthreads();                                        // last tile's LDS reads are done
    char* ep = smem + wave * 16384;                         // this wave's 128 x 64 bf16 image
    float bv[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int nn = n0 + wn * 64 + j * 16 + fq * 4;
        if (g.bias != nullptr && nn < g.N) {
            const float4 b4 = *(const float4*)(g.bias + nn);
            bv[j][0] = b4.x; bv[j][1] = b4.y; bv[j][2] = b4.z; bv[j][3] = b4.w;
        } else {
            bv[j][0] = bv[j][1] = bv[j][2] = bv[j][3] = 0.f;
        }
    }
    const bool fold = (EPI != OV_EPI_BIAS_RESIDUAL) && g.colsum != nullptr;
    float sv[4][4];
    if (fold) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int nn = n0 + wn * 64 + j * 16 + fq * 4;
            if (nn < g.N) {
                const float4 s4 = *(const float4*)(g.colsum + nn);
                sv[j][0] = s4.x; sv[j][1] = s4.y; sv[j][2] = s4.z; sv[j][3] = s4.w;
            } else {
                sv[j][0] = sv[j][1] = sv[j][2] = sv[j][3] = 0.f;
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int ml = i * 16 + fr;
        float rmean = 0.f, rrstd = 1.f;
        if (fold) {
            int64_t m = m0 + wm * 128 + i * 16 + fr;
            m = m < g.M ? m : g.M - 1;
            const float2 st = *(const float2*)(g.rowstats + 2 * m);
            rmean = st.x; rrstd = st.y;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            // same arithmetic (packed fp32) as epilogue_2pass: results must not depend on which kernel variant ran
            f32x2_t v01 = f32x2_t{acc[i][j][0], acc[i][j][1]};
            f32x2_t v23 = f32x2_t{acc[i][j][2], acc[i][j][3]};
            if (fold) {
                const f32x2_t nm = {-rmean, -rmean}, rs = {rrstd, rrstd};
                v01 = __builtin_elementwise_fma(f32x2_t{sv[j][0], sv[j][1]}, nm, v01);
                v23 = __builtin_elementwise_fma(f32x2_t{sv[j][2], sv[j][3]}, nm, v23);
                v01 = __builtin_elementwise_fma(v01, rs, f32x2_t{bv[j][0], bv[j][1]});
                v23 = __builtin_elementwise_fma(v23, rs, f32x2_t{bv[j][2], bv[j][3]});
            } else {
                v01 += f32x2_t{bv[j][0], bv[j][1]};
                v23 += f32x2_t{bv[j][2], bv[j][3]};
            }
            if (EPI == OV_EPI_BIAS_GELU_ERF) { v01 = gelu_erf_f2(v01); v23 = gelu_erf_f2(v23); }
            if (EPI == OV_EPI_BIAS_GELU_TANH) { v01 = gelu_tanh_f2(v01); v23 = gelu_tanh_f2(v23); }
            u32x2_t p = {pack_bf16x2(v01[0], v01[1]), pack_bf16x2(v23[0], v23[1])};
            const int c = j * 2 + (fq >> 1);
            *(u32x2_t*)(ep + ml * 128 + ((c ^ (ml & 7)) << 4) + (fq & 1) * 8) = p;
        }
    }
    __syncthreads();
    const int er = lane >> 3, ec = lane & 7;
    const int n = n0 + wn * 64 + ec * 8;
#pragma unroll 4
    for (int it = 0; it < 16; ++it) {
        const int row = it * 8 + er;
        u32x4_t v = *(const u32x4_t*)(ep + row * 128 + ((ec ^ (row & 7)) << 4));
        const int64_t m = m0 + wm * 128 + row;
        if (m < g.M && n < g.N) {
            if (EPI == OV_EPI_BIAS_RESIDUAL) {
                const int64_t rrow = g.resid_mod ? (m % g.resid_mod) + g.resid_off : m;
                const u32x4_t rv = *(const u32x4_t*)(g.R + rrow * g.ldr + n);
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    v[e] = pack_bf16x2(bf16lo_to_f32(v[e]) + bf16lo_to_f32(rv[e]),
                                       bf16hi_to_f32(v[e]) + bf16hi_to_f32(rv[e]));
            }
            const int64_t orow = g.out_group ? m + m / g.out_group + 1 : m;
            *(u32x4_t*)(g.C + orow * g.ldc + n) = v;
        }
    }
}

template <int EPI>
__global__ __launch_bounds__(NTHREADS, 2) void gemm_bf16_256x256(const GemmArgs g) {
    __shared__ __attribute__((aligned(16))) char smem[SMEM_BYTES];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // ---- XCD-aware, bijective workgroup -> tile map (blocks are dealt round-robin over 8 XCDs) ----
    const int nwg = g.tiles_m * g.tiles_n;
    const int bid = blockIdx.x;
    const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    const int wgid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
    const int tm = wgid / g.tiles_n, tn = wgid - tm * g.tiles_n;
    const int64_t m0 = (int64_t)tm * BM;
    const int n0 = tn * BN;

    // ---- per-thread staging sources: 4 x 16 B of A and of W per K-tile ------------------------------
    // LDS chunk q = j*512 + tid holds tile row q>>3, logical 16-B chunk (q&7) ^ (row&7).
    const int srow = tid >> 3;
    const int schunk = (tid & 7) ^ (srow & 7);
    const ov_bf16* asrc[4];
    const ov_bf16* wsrc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        int64_t ar = m0 + j * 64 + srow;
        ar = ar < g.M ? ar : g.M - 1;                 // clamp: tail rows load a valid row, never stored
        int wr = n0 + j * 64 + srow;
        wr = wr < g.N ? wr : g.N - 1;
        asrc[j] = g.A + ar * g.lda + schunk * 8;
        wsrc[j] = g.W + (int64_t)wr * g.ldw + schunk * 8;
    }
    auto stage = [&](int buf, int k0) {
        char* sa = smem + buf * STAGE_BYTES + wave * 1024;
        char* sw = sa + TILE_BYTES;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            __builtin_amdgcn_global_load_lds((gptr_t)(asrc[j] + k0), (lptr_t)(sa + j * 8192), 16, 0, 0);
#pragma unroll
        for (int j = 0; j < 4; ++j)
            __builtin_amdgcn_global_load_lds((gptr_t)(wsrc[j] + k0), (lptr_t)(sw + j * 8192), 16, 0, 0);
    };

    // ---- fragment addressing ------------------------------------------------------------------------
    const int wm = wave >> 2, wn = wave & 3;
    const int fr = lane & 15, fq = lane >> 4;
    const int a_off = (wm * 128 + fr) * 128;
    const int w_off = TILE_BYTES + (wn * 64 + fr) * 128;
    const int sw0 = ((fq) ^ (fr & 7)) << 4;
    const int sw1 = ((4 + fq) ^ (fr & 7)) << 4;

    f32x4_t acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    const int nt = g.K / BK;
    stage(0, 0);
    for (int t = 0; t < nt; ++t) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of tile t have landed
        __syncthreads();                                    // everyone's have; buffer (t+1)&1 is free
        if (t + 1 < nt) stage((t + 1) & 1, (t + 1) * BK);
        const char* s = smem + (t & 1) * STAGE_BYTES;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int sw = kk ? sw1 : sw0;
            bf16x8_t af[8], wf[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) wf[j] = *(const bf16x8_t*)(s + w_off + j * 2048 + sw);
#pragma unroll
            for (int i = 0; i < 8; ++i) af[i] = *(const bf16x8_t*)(s + a_off + i * 2048 + sw);
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], af[i], acc[i][j], 0, 0, 0);
        }
    }

    gemm_epilogue<EPI>(g, acc, smem, m0, n0, wave, lane);
}

// =====================================================================================================
// Ping-pong kernel (default).  Same tile and epilogue as above, different main loop:
//   * each K-tile (BK = 64) is staged as FOUR 16-KiB pieces -- (A,k 0-31) (W,k 0-31) (A,k 32-63) (W,k 32-63) --
//     one piece per phase, one K-tile ahead, with a COUNTED s_waitcnt vmcnt(4) at phases 1 and 3 (never 0 in the
//     loop) and raw s_barrier, so two pieces stay in flight across every barrier;
//   * a K-tile is four phases of 16 MFMAs: (k-half, m-half) = (0,0) (0,1) (1,0) (1,1); each phase is a LOAD
//     segment (4-8 ds_read_b128 + 2 global_load_lds) and a COMPUTE segment (16 MFMAs), separated by barriers;
//   * waves 4-7 (the lower 128 rows) run ONE barrier behind waves 0-3, so on every SIMD one wave is in its COMPUTE
//     segment while its partner is in its LOAD segment: the matrix pipe sees back-to-back MFMA clusters instead of
//     both waves stalling on LDS at once.
// LDS map (128 KiB): buffer b in {0,1} at b*64 KiB, then [A k0][W k0][A k1][W k1] x 16 KiB; a piece is 256 rows x 64 B,
// 16-B chunk c of row r stored at chunk c ^ f((r >> 2) & 3), f = {0,2,3,1}: conflict-free for the 16x16x32 operand reads.
// Hazards (interval = span between two consecutive barriers; wave group 1 lags group 0 by one interval):
//   RAW  a piece issued in phase p of tile T is waited for (counted vmcnt) in phase 3 (p<2) or phase 1 of T+1 (p>=2),
//        and first read >= one barrier after EVERY wave has passed that wait;
//   WAR  a slot is re-staged >= 4 intervals after its last ds_read was issued (those reads are consumed by the
//        MFMAs of the reader's next COMPUTE segment, i.e. before its next barrier).
constexpr int PIECE_BYTES = 256 * 64;          // 16 KiB

__device__ __forceinline__ int swz4(int row) { return (0x78 >> (((row >> 2) & 3) * 2)) & 3; }

template <int EPI>
__global__ __launch_bounds__(NTHREADS, 2) void gemm_bf16_pp(const GemmArgs g) {
    __shared__ __attribute__((aligned(16))) char smem[SMEM_BYTES];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    const int nwg = g.tiles_m * g.tiles_n;
    const int bid = blockIdx.x;
    const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    const int wgid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
    const int tm = wgid / g.tiles_n, tn = wgid - tm * g.tiles_n;
    const int64_t m0 = (int64_t)tm * BM;
    const int n0 = tn * BN;

    // staging: LDS chunk q = i*512 + tid of a piece holds row q>>2, logical chunk (q&3) ^ f(row)
    const int srow = tid >> 2;
    const int schunk = (tid & 3) ^ swz4(srow);
    const ov_bf16* asrc[2];
    const ov_bf16* wsrc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        int64_t ar = m0 + i * 128 + srow;
        ar = ar < g.M ? ar : g.M - 1;
        int wr = n0 + i * 128 + srow;
        wr = wr < g.N ? wr : g.N - 1;
        asrc[i] = g.A + ar * g.lda + schunk * 8;
        wsrc[i] = g.W + (int64_t)wr * g.ldw + schunk * 8;
    }
    char* const sbase = smem + wave * 1024;
    // piece j of the K-tile starting at k0 -> buffer buf
    auto stage_piece = [&](int buf, int j, int k0) {
        char* dst = sbase + buf * STAGE_BYTES + j * PIECE_BYTES;
        const int kk = k0 + (j >> 1) * 32;
        if (j & 1) {
            __builtin_amdgcn_global_load_lds((gptr_t)(wsrc[0] + kk), (lptr_t)dst, 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gptr_t)(wsrc[1] + kk), (lptr_t)(dst + 8192), 16, 0, 0);
        } else {
            __builtin_amdgcn_global_load_lds((gptr_t)(asrc[0] + kk), (lptr_t)dst, 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gptr_t)(asrc[1] + kk), (lptr_t)(dst + 8192), 16, 0, 0);
        }
    };

    const int wm = wave >> 2, wn = wave & 3;
    const int fr = lane & 15, fq = lane >> 4;
    const int lsw = (fq ^ swz4(fr)) << 4;
    const int a_lane = (wm * 128 + fr) * 64 + lsw;                 // + piece(A,kh) + i*1024
    const int w_lane = PIECE_BYTES + (wn * 64 + fr) * 64 + lsw;    // + piece pair(kh) + j*1024

    f32x4_t acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    const int nt = g.K / BK;
#pragma unroll
    for (int j = 0; j < 4; ++j) stage_piece(0, j, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (wm == 1) __builtin_amdgcn_s_barrier();                     // stagger the lower wave group by one interval

    bf16x8_t af[4], wf[4];
    for (int t = 0; t < nt; ++t) {
        const char* s = smem + (t & 1) * STAGE_BYTES;
        const bool more = (t + 1 < nt);
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int kh = p >> 1, mh = p & 1;
            // ---------------- LOAD segment ----------------
            const char* sp = s + kh * (2 * PIECE_BYTES);
            if (mh == 0) {
#pragma unroll
                for (int j = 0; j < 4; ++j) wf[j] = *(const bf16x8_t*)(sp + w_lane + j * 1024);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) af[i] = *(const bf16x8_t*)(sp + a_lane + (mh * 4 + i) * 1024);
            if (more) stage_piece((t + 1) & 1, p, (t + 1) * BK);
            if (p & 1) {
                if (more) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            // ---------------- COMPUTE segment ----------------
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[mh * 4 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], af[i], acc[mh * 4 + i][j], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if (wm == 0) __builtin_amdgcn_s_barrier();                     // re-align the two wave groups
    gemm_epilogue<EPI>(g, acc, smem, m0, n0, wave, lane);
}

// =====================================================================================================
// Persistent ping-pong kernel (default).  One workgroup per CU walks a static, XCD-contiguous list of output tiles.
// On top of the ping-pong main loop above:
//   * the LAST K-tile of a tile stages K-tile 0 of the NEXT tile into the free LDS buffer, so the next main loop
//     starts with its operands already on chip (no per-tile prologue latency, no relaunch);
//   * the epilogue transposes through the 64 KiB buffer the last K-tile occupied (8 KiB per wave, two 64-row passes,
//     wave-local), leaving the prefetched buffer untouched; its global stores are issued and the next tile's MFMAs
//     start while they drain, so store bursts of different CUs no longer line up in time;
//   * the two wave groups re-align for the epilogue (both halves of every SIMD share the VALU work) and re-stagger
//     by one barrier afterwards.
template <int EPI>
__device__ __forceinline__ void epilogue_2pass(const GemmArgs& g, f32x4_t (&acc)[8][4], char* ep, int64_t m0, int n0,
                                               int wave, int lane, bool drain_loads_before_stores, bool resid_folded) {
    const int wm = wave >> 2, wn = wave & 3;
    const int fr = lane & 15, fq = lane >> 4;
    float bv[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int nn = n0 + wn * 64 + j * 16 + fq * 4;
        if (g.bias != nullptr && nn < g.N) {
            const float4 b4 = *(const float4*)(g.bias + nn);
            bv[j][0] = b4.x; bv[j][1] = b4.y; bv[j][2] = b4.z; bv[j][3] = b4.w;
        } else {
            bv[j][0] = bv[j][1] = bv[j][2] = bv[j][3] = 0.f;
        }
    }
    const bool fold = (EPI != OV_EPI_BIAS_RESIDUAL) && g.colsum != nullptr;
    float sv[4][4];
    if (fold) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int nn = n0 + wn * 64 + j * 16 + fq * 4;
            if (nn < g.N) {
                const float4 s4 = *(const float4*)(g.colsum + nn);
                sv[j][0] = s4.x; sv[j][1] = s4.y; sv[j][2] = s4.z; sv[j][3] = s4.w;
            } else {
                sv[j][0] = sv[j][1] = sv[j][2] = sv[j][3] = 0.f;
            }
        }
    }
    const int er = lane >> 3, ec = lane & 7;
    const int n = n0 + wn * 64 + ec * 8;
    // Every load of the epilogue is issued before the first store: a load placed behind stores would have to wait for
    // their write acknowledgements (vmcnt is in order), which costs the residual epilogue ~10 us per tile.  Row statistics
    // of all 8 row groups up front; the residual rows of BOTH passes are fetched (unpredicated, clamped addresses -- a
    // predicated load makes hipcc serialise the loads behind vmcnt(0)) before any output row is stored.
    float rmean[8], rrstd[8];
    if (fold) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            int64_t m = m0 + wm * 128 + i * 16 + fr;
            m = m < g.M ? m : g.M - 1;
            const float2 st = *(const float2*)(g.rowstats + 2 * m);
            rmean[i] = st.x; rrstd[i] = st.y;
        }
    }
    u32x4_t vout[2][8];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int ml = i * 16 + fr;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                f32x2_t v01 = f32x2_t{acc[h * 4 + i][j][0], acc[h * 4 + i][j][1]};
                f32x2_t v23 = f32x2_t{acc[h * 4 + i][j][2], acc[h * 4 + i][j][3]};
                if (fold) {      // rstd * (acc - mean * colsum) + cvec, as two explicit FMAs (identical in every kernel variant)
                    const f32x2_t nm = {-rmean[h * 4 + i], -rmean[h * 4 + i]}, rs = {rrstd[h * 4 + i], rrstd[h * 4 + i]};
                    v01 = __builtin_elementwise_fma(f32x2_t{sv[j][0], sv[j][1]}, nm, v01);
                    v23 = __builtin_elementwise_fma(f32x2_t{sv[j][2], sv[j][3]}, nm, v23);
                    v01 = __builtin_elementwise_fma(v01, rs, f32x2_t{bv[j][0], bv[j][1]});
                    v23 = __builtin_elementwise_fma(v23, rs, f32x2_t{bv[j][2], bv[j][3]});
                } else {
                    v01 += f32x2_t{bv[j][0], bv[j][1]};
                    v23 += f32x2_t{bv[j][2], bv[j][3]};
                }
                if (EPI == OV_EPI_BIAS_GELU_ERF) { v01 = gelu_erf_f2(v01); v23 = gelu_erf_f2(v23); }
                if (EPI == OV_EPI_BIAS_GELU_TANH) { v01 = gelu_tanh_f2(v01); v23 = gelu_tanh_f2(v23); }
                u32x2_t pk = {pack_bf16x2(v01[0], v01[1]), pack_bf16x2(v23[0], v23[1])};
                const int c = j * 2 + (fq >> 1);
                *(u32x2_t*)(ep + ml * 128 + ((c ^ (ml & 7)) << 4) + (fq & 1) * 8) = pk;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // wave-local hand-over (DS ops of one wave are in order)
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int row = it * 8 + er;
            vout[h][it] = *(const u32x4_t*)(ep + row * 128 + ((ec ^ (row & 7)) << 4));
        }
        if (EPI == OV_EPI_BIAS_RESIDUAL && !resid_folded) {
            u32x4_t rv[8];
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                int64_t m = m0 + wm * 128 + h * 64 + it * 8 + er;
                m = m < g.M ? m : g.M - 1;
                const int nc = n < g.N ? n : g.N - 8;
                const int64_t rrow = g.resid_mod ? (m % g.resid_mod) + g.resid_off : m;
                rv[it] = *(const u32x4_t*)(g.R + rrow * g.ldr + nc);
            }
#pragma unroll
            for (int it = 0; it < 8; ++it)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    vout[h][it][e] = pack_bf16x2(bf16lo_to_f32(vout[h][it][e]) + bf16lo_to_f32(rv[it][e]),
                                                 bf16hi_to_f32(vout[h][it][e]) + bf16hi_to_f32(rv[it][e]));
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // pass-h reads done before pass-(h+1) writes reuse the image
        // Without residual loads nothing is fetched after this point, so a pass's rows are stored right away (they drain
        // under the other pass's arithmetic); with them, both passes' loads come first (see above).
        if (EPI != OV_EPI_BIAS_RESIDUAL || h == 1) {
            if (drain_loads_before_stores && (EPI == OV_EPI_BIAS_RESIDUAL || h == 0))
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // next tile's K-tile 0 has landed; nothing older than the stores
#pragma unroll
            for (int hh = (EPI == OV_EPI_BIAS_RESIDUAL ? 0 : h); hh <= h; ++hh)
#pragma unroll
                for (int it = 0; it < 8; ++it) {
                    const int64_t m = m0 + wm * 128 + hh * 64 + it * 8 + er;
                    if (m < g.M && n < g.N) {
                        const int64_t orow = g.out_group ? m + m / g.out_group + 1 : m;
                        *(u32x4_t*)(g.C + orow * g.ldc + n) = vout[hh][it];
                    }
                }
        }
    }
}

template <int EPI>
__global__ __launch_bounds__(NTHREADS, 2) void gemm_bf16_persist(const GemmArgs g) {
    __shared__ __attribute__((aligned(16))) char smem[SMEM_BYTES];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // ---- static persistent schedule: XCD x owns a contiguous run of tiles (n fastest), its workgroups stride it ----
    const int nwg = g.tiles_m * g.tiles_n;
    const int G = gridDim.x, bid = blockIdx.x;
    const int xcd = bid & 7, li = bid >> 3;
    const int q8 = nwg >> 3, r8 = nwg & 7;
    const int xstart = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
    const int xcnt = q8 + (xcd < r8 ? 1 : 0);
    const int nper = (G - xcd + 7) >> 3;
    int tcur = li;
    if (tcur >= xcnt) return;

    const int srow = tid >> 2;
    const int schunk = (tid & 3) ^ swz4(srow);
    const ov_bf16* asrc[2];
    const ov_bf16* wsrc[2];
    const ov_bf16* nasrc[2];
    const ov_bf16* nwsrc[2];
    int64_t m0, nm0 = 0;
    int n0, nn0 = 0;
    auto set_tile = [&](int trel, const ov_bf16* (&as)[2], const ov_bf16* (&ws)[2], int64_t& mm, int& nn) {
        const int wg = xstart + trel;
        const int tm = wg / g.tiles_n, tn = wg - tm * g.tiles_n;
        mm = (int64_t)tm * BM;
        nn = tn * BN;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            int64_t ar = mm + i * 128 + srow;
            ar = ar < g.M ? ar : g.M - 1;
            int wr = nn + i * 128 + srow;
            wr = wr < g.N ? wr : g.N - 1;
            as[i] = g.A + ar * g.lda + schunk * 8;
            ws[i] = g.W + (int64_t)wr * g.ldw + schunk * 8;
        }
    };
    char* const sbase = smem + wave * 1024;
    auto stage_piece = [&](const ov_bf16* const (&as)[2], const ov_bf16* const (&ws)[2], int buf, int j, int k0) {
        char* dst = sbase + buf * STAGE_BYTES + j * PIECE_BYTES;
        const int kk = k0 + (j >> 1) * 32;
        if (j & 1) {
            __builtin_amdgcn_global_load_lds((gptr_t)(ws[0] + kk), (lptr_t)dst, 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gptr_t)(ws[1] + kk), (lptr_t)(dst + 8192), 16, 0, 0);
        } else {
            __builtin_amdgcn_global_load_lds((gptr_t)(as[0] + kk), (lptr_t)dst, 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gptr_t)(as[1] + kk), (lptr_t)(dst + 8192), 16, 0, 0);
        }
    };

    const int wm = wave >> 2, wn = wave & 3;
    const int fr = lane & 15, fq = lane >> 4;
    const int lsw = (fq ^ swz4(fr)) << 4;
    const int a_lane = (wm * 128 + fr) * 64 + lsw;
    const int w_lane = PIECE_BYTES + (wn * 64 + fr) * 64 + lsw;
    const int nt = g.K / BK;
    set_tile(tcur, asrc, wsrc, m0, n0);
#pragma unroll
    for (int j = 0; j < 4; ++j) stage_piece(asrc, wsrc, 0, j, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (wm == 1) __builtin_amdgcn_s_barrier();                     // stagger the lower wave group by one interval

    int par = 0;
    int titer = 0;
    auto stamp = [&](int k) {
        if (g.stamps != nullptr && tid == 0 && titer < g.stamp_slots)
            g.stamps[((size_t)bid * g.stamp_slots + titer) * 4 + k] = __builtin_amdgcn_s_memtime();
    };
    for (;;) {
        stamp(0);
        const int tnext = tcur + nper;
        const bool has_next = tnext < xcnt;
        if (has_next) set_tile(tnext, nasrc, nwsrc, nm0, nn0);

        f32x4_t acc[8][4];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

        bf16x8_t af[4], wf[4];
        for (int t = 0; t < nt; ++t) {
            const char* s = smem + ((par + t) & 1) * STAGE_BYTES;
            const bool last = (t == nt - 1);
            const bool more = !last || has_next;
            const int nbuf = (par + t + 1) & 1;
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const int kh = p >> 1, mh = p & 1;
                const char* sp = s + kh * (2 * PIECE_BYTES);
                if (mh == 0) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) wf[j] = *(const bf16x8_t*)(sp + w_lane + j * 1024);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) af[i] = *(const bf16x8_t*)(sp + a_lane + (mh * 4 + i) * 1024);
                if (more) {
                    if (!last) stage_piece(asrc, wsrc, nbuf, p, (t + 1) * BK);
                    else stage_piece(nasrc, nwsrc, nbuf, p, 0);
                }
                if (p & 1) {
                    // K-tile 0 of every tile was fully waited for (prologue / previous epilogue): no wait at t == 0, p == 1
                    if (more) { if (t > 0 || p == 3) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
                    else if (p == 1 && t > 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[mh * 4 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], af[i], acc[mh * 4 + i][j], 0, 0, 0);
                __builtin_amdgcn_s_setprio(0);
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        stamp(1);
        if (wm == 0) __builtin_amdgcn_s_barrier();                 // re-align: every wave is past its last COMPUTE segment
        stamp(2);
        char* ep = smem + ((par + nt - 1) & 1) * STAGE_BYTES + wave * 8192;
        epilogue_2pass<EPI>(g, acc, ep, m0, n0, wave, lane, has_next, false);
        stamp(3);
        ++titer;
        if (!has_next) break;
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();                               // next K-tile 0 visible to all; epilogue image is dead
        if (wm == 1) __builtin_amdgcn_s_barrier();                  // re-stagger
        par = (par + nt) & 1;
        asrc[0] = nasrc[0]; asrc[1] = nasrc[1]; wsrc[0] = nwsrc[0]; wsrc[1] = nwsrc[1];
        m0 = nm0; n0 = nn0;
        tcur = tnext;
    }
}

int num_cus() {
    static int n = 0;
    if (n == 0) {
        int dev = 0;
        hipDeviceProp_t p;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess && p.multiProcessorCount > 0)
            n = p.multiProcessorCount;
        else
            n = 256;
    }
    return n;
}

thread_local const float* g_colsum = nullptr;      // set by ov_gemm_ln around its call into ov_gemm
thread_local const float* g_rowstats = nullptr;
unsigned long long* g_stamps = nullptr;
int g_stamp_slots = 0;

int gemm_variant() {       // 0 = persistent ping-pong (default), 1 = v1 two-stage, 2 = non-persistent ping-pong
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("OVHIP_GEMM_VARIANT");
        v = (e && e[0] >= '0' && e[0] <= '2') ? e[0] - '0' : 0;
    }
    return v;
}

template <int EPI>
int launch(GemmArgs a, hipStream_t st) {
    int var = gemm_variant();
    const int nwg = a.tiles_m * a.tiles_n;
    // fewer tiles than CUs (pooled heads, the tower's tail images): persistence buys nothing, use the plain launch
    if (var == 0 && nwg < num_cus()) var = 2;
    if (var == 1) {
        hipLaunchKernelGGL(gemm_bf16_256x256<EPI>, dim3(nwg), dim3(NTHREADS), 0, st, a);
    } else if (var == 2) {
        hipLaunchKernelGGL(gemm_bf16_pp<EPI>, dim3(nwg), dim3(NTHREADS), 0, st, a);
    } else {
        const int ncu = num_cus();
        hipLaunchKernelGGL(gemm_bf16_persist<EPI>, dim3(nwg < ncu ? nwg : ncu), dim3(NTHREADS), 0, st, a);
    }
    OV_LAUNCH_CHECK();
    return OV_OK;
}

}  // namespace

extern "C" int ov_gemm(const ov_bf16* A, int64_t lda, const ov_bf16* W, int64_t ldw, const float* bias,
                       ov_bf16* C, int64_t ldc, int64_t M, int N, int K, int epilogue,
                       const ov_bf16* R, int64_t ldr, int out_group, int resid_mod, int resid_off,
                       ov_stream_t stream) {
    if (!A || !W || !C || M <= 0 || N <= 0 || K <= 0) return OV_ERR_INVALID;
    if (K % BK || N % 8 || lda % 8 || ldw % 8 || ldc % 8) return OV_ERR_UNSUPPORTED;
    if (lda < K || ldw < K || ldc < N) return OV_ERR_INVALID;
    if (((uintptr_t)A | (uintptr_t)W | (uintptr_t)C) & 15) return OV_ERR_INVALID;
    if (bias && ((uintptr_t)bias & 15)) return OV_ERR_INVALID;
    if (epilogue == OV_EPI_BIAS_RESIDUAL) {
        if (!R || ldr % 8 || ldr < N || ((uintptr_t)R & 15)) return OV_ERR_INVALID;
    }
    if (out_group < 0 || resid_mod < 0 || resid_off < 0) return OV_ERR_INVALID;
    const int64_t tiles_m = (M + BM - 1) / BM;
    const int64_t tiles_n = (N + BN - 1) / BN;
    if (tiles_m * tiles_n > 0x7fffffffLL) return OV_ERR_UNSUPPORTED;
    GemmArgs a{A, W, bias, C, R, lda, ldw, ldc, ldr, M, N, K, (int)tiles_m, (int)tiles_n,
               out_group, resid_mod, resid_off, g_colsum, g_rowstats, g_stamps, g_stamp_slots};
    hipStream_t st = (hipStream_t)stream;
    switch (epilogue) {
        case OV_EPI_BIAS: return launch<OV_EPI_BIAS>(a, st);
        case OV_EPI_BIAS_GELU_ERF: return launch<OV_EPI_BIAS_GELU_ERF>(a, st);
        case OV_EPI_BIAS_GELU_TANH: return launch<OV_EPI_BIAS_GELU_TANH>(a, st);
        case OV_EPI_BIAS_RESIDUAL: return launch<OV_EPI_BIAS_RESIDUAL>(a, st);
        default: return OV_ERR_INVALID;
    }
}

// Diagnostics: when set, the persistent kernel's thread 0 of every workgroup records s_memtime at tile start / main-loop
// end / after the re-align barrier / epilogue end into buf[block][slot][4] (slot = tile iteration < slots).  NULL = off.
extern "C" int ov_debug_gemm_stamps(unsigned long long* buf, int slots) {
    g_stamps = buf;
    g_stamp_slots = buf ? slots : 0;
    return OV_OK;
}

extern "C" int ov_gemm_ln(const ov_bf16* X, int64_t ldx, const ov_bf16* Wg, int64_t ldw, const float* cvec, const float* colsum,
                          const float* rowstats, ov_bf16* C, int64_t ldc, int64_t M, int N, int K, int epilogue,
                          ov_stream_t stream) {
    if (!colsum || !rowstats || !cvec) return OV_ERR_INVALID;
    if (epilogue == OV_EPI_BIAS_RESIDUAL) return OV_ERR_UNSUPPORTED;
    if ((((uintptr_t)colsum | (uintptr_t)cvec) & 15) || ((uintptr_t)rowstats & 7)) return OV_ERR_INVALID;
    g_colsum = colsum;
    g_rowstats = rowstats;
    const int rc = ov_gemm(X, ldx, Wg, ldw, cvec, C, ldc, M, N, K, epilogue, nullptr, 0, 0, 0, 0, stream);
    g_colsum = nullptr;
    g_rowstats = nullptr;
    return rc;
}
