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
//   * epilogue: bias / LN fold / GELU in fp32 on the accumulators, bf16 pack, then either a transposition through a (swizzled)
//     wave-local LDS image and row-contiguous 16-byte stores (bias epilogue of the persistent kernel, every epilogue of the
//     non-persistent kernels) or a v_permlane16_swap exchange and row-per-lane 16-byte stores (GELU and residual epilogues of the
//     persistent kernel); the residual is added on the way out.  All forms perform the same arithmetic in the same order;
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
    unsigned long long* wstamps;    // diagnostics only: per-wave epilogue timeline [block][tile slot][wave][8]
    int ngroup;                     // persistent kernel: n-tiles per group of the XCD tile walk (== tiles_n: plain n-fastest walk)
    int64_t batch_a, batch_w, batch_c;   // gemm_bf16_pp with gridDim.y > 1: element strides of A, W, C per batch entry (split-K partials)
    int stagger, stagger_classes;        // persistent kernel: start delay (shader cycles) per class (bid >> 3) % classes (0 = off)
    int epi_prio;                        // persistent kernel: n > 0: waves 4-7 (the arbitration losers) run epilogue passes < n at s_setprio 1
    ov_bf16* C2; int64_t ldc2;           // ov_gemm_keep: second output = the GELU epilogue's pre-activation (acc + bias), bf16
    float* psum;                         // gemm_bf16_pp_tn: [gridDim.y][M] column sums of P over the split's rows (NULL = off)
    int st_plain;                        // persistent kernel, bias / GELU epilogues: plain instead of streaming output stores (small outputs)
    int half_ok;                         // persistent kernel: half tiles allowed (OVHIP_GEMM_HALF=0 switches them off)
    int rev;                             // persistent kernel: walk the row tiles from the last to the first (launcher: c_proj)
    int rotmask;                         // persistent kernel, plain walk, half last n-tile (see HALF TILES): tiles_n - 1 when the workgroup
                                         // stride is a multiple of tiles_n -- the n index is then rotated by the workgroup's tile count, so
                                         // that every workgroup alternates between full and half tiles (0 = off)
    float* rowpart;                      // residual epilogue (persistent direct form, skinny kernel): {sum, sum of squares} of every
                                         // 32-column group of every OUTPUT row, [M][N / 32][2] fp32 (common.h: row statistics); NULL = off
};

template <int V> struct IntC { static constexpr int value = V; };
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
    const bool fold = (EPI < OV_EPI_BIAS_RESIDUAL) && g.colsum != nullptr;
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
            if (EPI == OV_EPI_BIAS_GELU_ERF) gelu_erf_f2x2(v01, v23);
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
            if (EPI >= OV_EPI_BIAS_RESIDUAL) {
                const int64_t rrow = g.resid_mod ? (m % g.resid_mod) + g.resid_off : m;
                const u32x4_t rv = *(const u32x4_t*)(g.R + rrow * g.ldr + n);
                v = epi_combine<EPI>(v, rv);
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
// Experimental (OVHIP_GEMM_VARIANT=3): FOUR waves per workgroup, one per SIMD, each owning a 128 x 128 quarter of the 256 x 256 tile
// (8 x 8 accumulator fragments = 256 registers; launch bound 1 wave per SIMD = 512 registers).  Per K-tile a wave reads 32
// fragments for 128 MFMAs where the ping-pong kernels read 24 for 64: a third less LDS traffic per flop, no partner wave whose
// load segment interferes with the MFMA stream, ONE barrier per K-tile.  The wave software-pipelines its own operands: while the
// 64 MFMAs of k-half 0 run it issues the 16 LDS-DMAs of the next K-tile (whole 128-byte lines, 8 rows per instruction: v1's LDS
// image) and the 16 fragment reads of k-half 1; half-way through k-half 1 it waits for its DMAs, meets the other waves at the
// barrier and reads the next K-tile's k-half 0 fragments under the remaining MFMAs.  Non-persistent prototype with the shared
// epilogue (each wave plays two of the epilogue's "waves"); measures what the main loop is worth before anything is built on it.
template <int EPI>
__global__ __launch_bounds__(256, 1) void gemm_bf16_w4(const GemmArgs g) {
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

    // staging: instruction j (0..7) of this wave fills rows (4 j + wave) * 8 + (lane >> 3), stored chunk lane & 7 = logical chunk
    // (lane & 7) ^ (row & 7) (v1's image: 128-byte rows, conflict-free for the 16x16x32 operand reads)
    const ov_bf16* asrc[8];
    const ov_bf16* wsrc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int row = (4 * j + wave) * 8 + (lane >> 3);
        const int ch = (lane & 7) ^ (row & 7);
        int64_t ar = m0 + row;
        ar = ar < g.M ? ar : g.M - 1;
        int wr = n0 + row;
        wr = wr < g.N ? wr : g.N - 1;
        asrc[j] = g.A + ar * g.lda + ch * 8;
        wsrc[j] = g.W + (int64_t)wr * g.ldw + ch * 8;
    }
    char* const sdst = smem + wave * 1024;
    auto dma = [&](int buf, int q, int k0) {      // q 0..7: A rows block q; 8..15: W
        char* d = sdst + buf * STAGE_BYTES + (q >> 3) * TILE_BYTES + (q & 7) * 4096;
        const ov_bf16* src = (q < 8 ? asrc[q & 7] : wsrc[q & 7]) + k0;
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)d, 16, 0, 0);
    };
    const int wm = wave >> 1, wn = wave & 1;
    const int fr = lane & 15, fq = lane >> 4;
    const int a_off = (wm * 128 + fr) * 128;
    const int w_off = TILE_BYTES + (wn * 128 + fr) * 128;
    const int sw0 = (fq ^ (fr & 7)) << 4, sw1 = ((4 + fq) ^ (fr & 7)) << 4;
    auto frag = [&](const char* st, int f, int kh) {      // f 0..7: A fragment i = f; 8..15: W fragment j = f - 8
        return *(const bf16x8_t*)(st + (f < 8 ? a_off : w_off) + (f & 7) * 2048 + (kh ? sw1 : sw0));
    };

    f32x4_t acc0[8][4], acc1[8][4];               // columns j 0-3 / 4-7 of the wave's 8 x 8 fragment grid
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) { acc0[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f}; acc1[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f}; }

    const int nt = g.K / BK;
#pragma unroll
    for (int q = 0; q < 16; ++q) dma(0, q, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    bf16x8_t cur[16], nxt[16];
#pragma unroll
    for (int f = 0; f < 16; ++f) cur[f] = frag(smem, f, 0);

    // accumulators pinned to AGPRs ("+a"): left to itself hipcc keeps them in VGPRs and parks the FRAGMENTS in AGPRs, which MFMA
    // operands cannot come from -- 170 v_accvgpr moves and their s_nops per K-tile
    auto mma = [&](f32x4_t& c, const bf16x8_t& wfrag, const bf16x8_t& afrag) {
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(wfrag), "v"(afrag));
    };
    auto ktile = [&](int t, auto morec) {
        constexpr bool MORE = decltype(morec)::value;
        const char* st = smem + (t & 1) * STAGE_BYTES;
        // ---- k-half 0: 64 MFMAs; under them the next K-tile's 16 DMAs and this K-tile's k-half 1 fragments
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            nxt[q] = frag(st, q, 1);
            if (MORE && q < 8) { dma((t + 1) & 1, 2 * q, (t + 1) * BK); dma((t + 1) & 1, 2 * q + 1, (t + 1) * BK); }   // all out in the first half
            const int i = q >> 1;
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                if (q & 1) mma(acc1[i][jj], cur[12 + jj], cur[i]);
                else mma(acc0[i][jj], cur[8 + jj], cur[i]);
            }
        }
        // ---- k-half 1: 64 MFMAs; half-way the DMAs have landed for everybody, then the next K-tile's k-half 0 fragments
        const char* sn = smem + ((t + 1) & 1) * STAGE_BYTES;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            if (MORE && q == 10) {                                 // as late as the fragment reads below allow: the DMAs get >= 1.1 k cycles
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            }
            if (MORE && q >= 10) {                                 // 16 reads over six groups: 3 3 3 3 2 2
                const int f0 = q < 14 ? 3 * (q - 10) : 12 + 2 * (q - 14), nf = q < 14 ? 3 : 2;
#pragma unroll
                for (int f = 0; f < 3; ++f)
                    if (f < nf) cur[f0 + f] = frag(sn, f0 + f, 0);
            }
            const int i = q >> 1;
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                if (q & 1) mma(acc1[i][jj], nxt[12 + jj], nxt[i]);
                else mma(acc0[i][jj], nxt[8 + jj], nxt[i]);
            }
        }
    };
    for (int t = 0; t + 1 < nt; ++t) ktile(t, IntC<1>{});
    ktile(nt - 1, IntC<0>{});
    // each wave plays the epilogue's waves (wm, 2 wn) and (wm, 2 wn + 1): same images, same arithmetic
    gemm_epilogue<EPI>(g, acc0, smem, m0, n0, wm * 4 + wn * 2, lane);
    gemm_epilogue<EPI>(g, acc1, smem, m0, n0, wm * 4 + wn * 2 + 1, lane);
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
__global__ __launch_bounds__(NTHREADS, 2) void gemm_bf16_pp(const GemmArgs g_in) {
    __shared__ __attribute__((aligned(16))) char smem[SMEM_BYTES];
    GemmArgs g = g_in;
    if (gridDim.y > 1) {                        // batch entry (wave-uniform): independent operands and output per blockIdx.y
        g.A += (int64_t)blockIdx.y * g.batch_a;
        g.W += (int64_t)blockIdx.y * g.batch_w;
        g.C += (int64_t)blockIdx.y * g.batch_c;
    }
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
// TN ping-pong kernel (weight gradients):  C[i, j] = sum_m P[m, i] * Q[m, j]  with BOTH operands stored row-major over the
// contraction index m (dW = dY^T X: P = dY [M, N], Q = X [M, K]) -- no explicit transposes.  Same tile, phases, barriers and
// epilogue as gemm_bf16_pp; what differs is the LDS image and the fragment reads:
//   * a piece is 32 contraction rows x 256 columns (512-byte rows, 16 KiB), LDS-DMA'd row-major; the 16-byte chunk c of row m
//     is stored at chunk c ^ (key(m) << 1), key(m) = (m & 3) | ((m >> 3) & 1) << 2 (applied on the per-lane SOURCE address, the
//     DMA writes LDS lane-linearly);
//   * an MFMA operand fragment (8 consecutive m of one column per lane) is two ds_read_b64_tr_b16: each returns, per 16-lane
//     group, a 4-row x 16-column block column-major (lane i gets column i of the four rows).  With the key above the 32 lanes of
//     a half-wave (row groups 8g .. 8g+3 of two g) hit 32 distinct bank pairs: conflict-free.
// blockIdx.y = split-K range z over the contraction rows [z * chunk, min(Mc, (z + 1) * chunk)); partials are bf16, summed in
// fp32 by the caller -- the same products in the same order as the transposed path (bitwise the same results).
// GemmArgs reuse: A = P (lda), W = Q (ldw), M = columns of P (output rows), N = columns of Q (output columns), K = Mc
// (contraction rows, % 64 == 0), batch_a = chunk (rows, % 64 == 0), batch_c = elements between partial outputs.
// psum != NULL: the workgroups of output column tile 0 also produce the column sums of P (the bias gradient sum_m dY[m, i]) as one
// more MFMA per phase and wave against an all-ones operand -- the P fragments are in registers anyway, so the separate pass over dY
// (8 ms of the L/14 training step) disappears; wave wn takes fragment wn of the four, fp32 accumulators like every other product.
__device__ __forceinline__ int tn_key(int m) { return (m & 3) | (((m >> 3) & 1) << 2); }

template <int OFF>
__device__ __forceinline__ u32x2_t tn_tr_read(unsigned addr) {
    u32x2_t v;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}
__device__ __forceinline__ bf16x8_t tn_join(u32x2_t a, u32x2_t b) {
    const u32x4_t w = {a[0], a[1], b[0], b[1]};
    return __builtin_bit_cast(bf16x8_t, w);
}

__global__ __launch_bounds__(NTHREADS, 2) void gemm_bf16_pp_tn(const GemmArgs g_in) {
    __shared__ __attribute__((aligned(16))) char smem[SMEM_BYTES];
    GemmArgs g = g_in;
    const int64_t kbeg = (int64_t)blockIdx.y * g.batch_a;
    int64_t kend = kbeg + g.batch_a;
    if (kend > g.K) kend = g.K;
    const int nt = (int)((kend - kbeg) / BK);                       // >= 1 (launcher)
    g.C += (int64_t)blockIdx.y * g.batch_c;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    const int nwg = g.tiles_m * g.tiles_n;
    const int bid = blockIdx.x;
    const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    const int wgid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
    const int tm = wgid / g.tiles_n, tn = wgid - tm * g.tiles_n;
    const int64_t m0 = (int64_t)tm * BM;                            // first column of P = first output row
    const int n0 = tn * BN;                                         // first column of Q = first output column

    // staging: DMA i (0, 1) of a piece fills LDS chunk q = i*512 + tid = row q >> 5 (0..31), stored chunk q & 31
    const ov_bf16* psrc[2];
    const ov_bf16* qsrc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int mr = i * 16 + (tid >> 5);
        const int c = (tid & 31) ^ (tn_key(mr) << 1);               // source chunk of the stored position
        int64_t pc = m0 + c * 8;
        pc = pc + 8 <= g.M ? pc : g.M - 8;                           // columns past the matrix: any valid chunk (never stored)
        int qc = n0 + c * 8;
        qc = qc + 8 <= g.N ? qc : g.N - 8;
        psrc[i] = g.A + (kbeg + mr) * g.lda + pc;
        qsrc[i] = g.W + (kbeg + mr) * g.ldw + qc;
    }
    const int64_t pstep = 32 * g.lda, qstep = 32 * g.ldw;           // half a K-tile of contraction rows
    char* const sbase = smem + wave * 1024;
    auto stage_piece = [&](int buf, int j, int kt) {               // piece j of K-tile kt -> buffer buf
        char* dst = sbase + buf * STAGE_BYTES + j * PIECE_BYTES;
        const int64_t h = 2 * kt + (j >> 1);
        if (j & 1) {
            __builtin_amdgcn_global_load_lds((gptr_t)(qsrc[0] + h * qstep), (lptr_t)dst, 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gptr_t)(qsrc[1] + h * qstep), (lptr_t)(dst + 8192), 16, 0, 0);
        } else {
            __builtin_amdgcn_global_load_lds((gptr_t)(psrc[0] + h * pstep), (lptr_t)dst, 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gptr_t)(psrc[1] + h * pstep), (lptr_t)(dst + 8192), 16, 0, 0);
        }
    };

    // fragment addressing: lane (g4 = lane >> 4, q = (lane & 15) >> 2, p = lane & 3) supplies block row q, columns 4p..4p+3
    const int wm = wave >> 2, wn = wave & 3;
    const int g4 = lane >> 4, fq4 = (lane & 15) >> 2, fp = lane & 3;
    const int mrow = 8 * g4 + fq4;                                  // first read: rows 8 g4 .. + 3; second read: + 4 (same key)
    const int key = tn_key(mrow);
    const unsigned lbase = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    // block b of 16 columns inside the wave's range sits at chunk pair ((range_first_pair + b) ^ key): XOR on byte-offset bits 5-7
    const unsigned a_lane = lbase + mrow * 512 + wm * 256 + (key << 5) + (fp >> 1) * 16 + (fp & 1) * 8;
    const unsigned w_lane = lbase + PIECE_BYTES + mrow * 512 + (((wn * 4) ^ key) << 5) + (fp >> 1) * 16 + (fp & 1) * 8;

    f32x4_t acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

#pragma unroll
    for (int j = 0; j < 4; ++j) stage_piece(0, j, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (wm == 1) __builtin_amdgcn_s_barrier();                     // stagger the lower wave group by one interval

    u32x2_t wr[8], ar[8];                                          // 4 W-role / 4 A-role fragments, two transposed reads each
    const bool colsums = g.psum != nullptr && tn == 0;             // block-uniform
    f32x4_t accs[2] = {f32x4_t{0.f, 0.f, 0.f, 0.f}, f32x4_t{0.f, 0.f, 0.f, 0.f}};
    const u32x4_t ones_w = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
    const bf16x8_t ones = __builtin_bit_cast(bf16x8_t, ones_w);
    for (int t = 0; t < nt; ++t) {
        const unsigned soff = (unsigned)((t & 1) * STAGE_BYTES);
        const bool more = (t + 1 < nt);
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int kh = p >> 1, mh = p & 1;
            const unsigned po = soff + kh * (2 * PIECE_BYTES);
            if (mh == 0) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const unsigned ad = (w_lane + po) ^ (unsigned)(j << 5);
                    wr[2 * j] = tn_tr_read<0>(ad);
                    wr[2 * j + 1] = tn_tr_read<2048>(ad);
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const unsigned ad = (a_lane + po) ^ (unsigned)((mh * 4 + i) << 5);
                ar[2 * i] = tn_tr_read<0>(ad);
                ar[2 * i + 1] = tn_tr_read<2048>(ad);
            }
            if (more) stage_piece((t + 1) & 1, p, t + 1);
            if (p & 1) {
                if (more) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(wr[0]), "+v"(wr[1]), "+v"(wr[2]), "+v"(wr[3]), "+v"(wr[4]), "+v"(wr[5]), "+v"(wr[6]), "+v"(wr[7]),
                           "+v"(ar[0]), "+v"(ar[1]), "+v"(ar[2]), "+v"(ar[3]), "+v"(ar[4]), "+v"(ar[5]), "+v"(ar[6]), "+v"(ar[7]));
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[mh * 4 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tn_join(wr[2 * j], wr[2 * j + 1]),
                                                                                 tn_join(ar[2 * i], ar[2 * i + 1]), acc[mh * 4 + i][j], 0, 0, 0);
            if (colsums) {      // D[r][c] = sum_k 1 * P[k, c]: every row of the 16 x 16 result holds the 16 column sums
                switch (wn) {
                    case 0: accs[mh] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, tn_join(ar[0], ar[1]), accs[mh], 0, 0, 0); break;
                    case 1: accs[mh] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, tn_join(ar[2], ar[3]), accs[mh], 0, 0, 0); break;
                    case 2: accs[mh] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, tn_join(ar[4], ar[5]), accs[mh], 0, 0, 0); break;
                    default: accs[mh] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, tn_join(ar[6], ar[7]), accs[mh], 0, 0, 0); break;
                }
            }
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if (wm == 0) __builtin_amdgcn_s_barrier();                     // re-align the two wave groups
    if (colsums && lane < 16) {                                    // row 0 of the result: lane c holds the sum of column c
#pragma unroll
        for (int mh = 0; mh < 2; ++mh) {
            const int64_t ni = m0 + wm * 128 + (mh * 4 + wn) * 16 + lane;
            if (ni < g.M) g.psum[(int64_t)blockIdx.y * g.M + ni] = accs[mh][0];
        }
    }
    gemm_epilogue<OV_EPI_BIAS>(g, acc, smem, m0, n0, wave, lane);
}

// =====================================================================================================
// Persistent ping-pong kernel (default).  One workgroup per CU walks a static, XCD-contiguous list of output tiles.
// On top of the ping-pong main loop above:
//   * the LAST K-tile of a tile stages K-tile 0 of the NEXT tile into the free LDS buffer, and the epilogue opens by
//     staging K-tile 1 of the next tile into the buffer that has just been consumed -- so every LDS-DMA the next main
//     loop waits for in its first two K-tiles is OLDER than this epilogue's stores (vmcnt retires in order: a load
//     issued behind stores waits for their write acknowledgements, several thousand cycles under a store burst);
//   * the tile's bias / LN-fold column sums / row statistics are LDS-DMA'd into a small parameter block one tile
//     ahead, so the epilogue starts without a single global-memory round trip;
//   * the epilogue streams the accumulators through a 2-KiB wave-local transposition image, one 16-row MFMA fragment
//     row per pass: pack + ds_write of pass i, the row-contiguous ds_read_b128 of pass i, the 16-byte global stores of
//     pass i-1.  DS operations of one wave execute in order, so the passes need no waits beyond data use;
//   * the two wave groups re-align for the epilogue (both halves of every SIMD share the VALU work) and re-stagger
//     by one barrier afterwards.
constexpr int PRM_OFF = SMEM_BYTES;                  // two 4-KiB parameter blocks: bias[256] f32 | colsum[256] f32 | rowstats[256][2] f32
constexpr int IMG_OFF = PRM_OFF + 2 * 4096;          // DIRECT == false only: 8 wave-local 2-KiB transposition images (16 rows x 64 n bf16)
constexpr int SMEM_PERSIST = IMG_OFF + 8 * 2048;     // 152 KiB of the CU's 160 (136 KiB without the images)

// Epilogue of the persistent kernel.  After the MFMAs a lane (fr = lane & 15, fq = lane >> 4) holds, for each of its 8 fragment rows i
// and 4 column blocks j, four consecutive n of ONE output row: acc[i][j][0..3] = C[i*16 + fr][j*16 + fq*4 + 0..3].  Packed to bf16
// that is 8 bytes per (i, j).  Two v_permlane16_swap per pair of column blocks (j0, j0 + 1) exchange the 8 bytes of block j0 + 1 in
// the even 16-lane rows with the 8 bytes of block j0 in the odd rows, after which every lane owns 16 CONTIGUOUS bytes of its row:
//     fq 0: n  0- 7 | fq 2: n  8-15 | fq 1: n 16-23 | fq 3: n 24-31      (+ 32 for the pair j0 = 2)
// so one global_store_dwordx4 writes 64 contiguous bytes of each of the wave's 16 rows -- no transposition through LDS (the 8-pass
// LDS image of the previous version cost 4 ds_write_b64 + 2 ds_read_b128 and an LDS round trip per pass).
// MAPPED = the row maps out_group / resid_mod are in use (the patch embedding's GEMM only; a kernel template parameter chosen by the
// launcher).  Without them the output / residual row IS the tile row: no unsigned division by the map parameters (hipcc if-converts
// `g.out_group ? m + m / g.out_group + 1 : m` and computes the software division on every store), and row pointers are one
// 64-bit multiply per lane, stepped by whole rows, instead of one per store -- ~250 of the ~500 (bias) to ~1200 (residual) VALU
// instructions of the epilogue were such address arithmetic.
template <int N> __device__ __forceinline__ void wait_vmcnt() {          // literal counts only (the hand-counted waits of the epilogues)
    static_assert(N == 8 || N == 14 || N == 22 || N == 24 || N == 26 || N == 28, "add the literal");
    if (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (N == 14) asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
    else if (N == 22) asm volatile("s_waitcnt vmcnt(22)" ::: "memory");
    else if (N == 24) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
    else if (N == 26) asm volatile("s_waitcnt vmcnt(26)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(28)" ::: "memory");
}

// STATS (residual epilogue only): every pass also forms the row statistics' partial sums of its 16 rows x 64 columns (two 32-column
// groups per wave: common.h) and leaves them in an LDS image of the tile; gemm_bf16_persist writes the image out behind the tile
// barrier.  The next LayerNorm's row pass (a re-read of the whole residual stream, 1.5 ms of the L/14 step) shrinks to a pass over
// these sums.
template <int EPI, bool FOLD, bool MAPPED, bool KEEP = false, bool STATS = false, int ST = OVHIP_ST_DIRECT>
__device__ __forceinline__ void epilogue_stream(const GemmArgs& g, f32x4_t (&acc)[8][4], const char* prm,
                                                int64_t m0, int n0, int wave, int lane, bool edge, unsigned long long* wst,
                                                bool htile, char* stats_lds = nullptr) {
    // `htile` (a HALF TILE: at most 128 valid columns, see gemm_bf16_persist): the waves form a 4 x 2 grid of 64 x 64 sub-tiles, only
    // acc[0..3] hold results and only passes 0-3 store.  Such a tile is an `edge` tile: every wait below is then a full drain.
    const int wm = wave >> 2;
    const int rb = htile ? (wave >> 1) * 64 : wm * 128;            // the wave's first row / column inside the tile
    const int cw = htile ? (wave & 1) * 64 : (wave & 3) * 64;
    // every per-lane constant of the epilogue is rebuilt per tile from an opaque lane id: as invariants of the tile loop hipcc carried
    // them through the main loop (or spilled them)
    const int lane_f = fresh_lane();
    const int fr = lane_f & 15, fq = lane_f >> 4;
    const int cb = (fq & 1) * 16 + (fq >> 1) * 8;                 // column of this lane's 16-byte chunk inside a 32-column pair
    const int n_lo = n0 + cw + cb;                           // pair 0; pair 1 is + 32
    const bool ncol0 = n_lo + 8 <= g.N, ncol1 = n_lo + 40 <= g.N;
    // Residual rows (unpredicated, clamped addresses -- a predicated load makes hipcc serialise the loads behind
    // vmcnt(0)).  Every load of the epilogue is issued before its first store: rows of passes 0-3 now, rows of passes
    // 4-7 once four accumulator rows have been retired (register room), and only then the first store.
    u32x4_t rv[8][2];
    const int nc0 = ncol0 ? n_lo : g.N - 8, nc1 = ncol1 ? n_lo + 32 : g.N - 8;
    // !MAPPED: this lane's row of pass 0 (clamped for the residual read: rows past M re-read row M - 1, never stored)
    const int64_t mbase = m0 + rb + fr;
    ov_bf16* const cbase = !MAPPED ? g.C + mbase * g.ldc + n_lo : nullptr;
    ov_bf16* const cbase2 = KEEP ? g.C2 + mbase * g.ldc2 + n_lo : nullptr;
    auto load_resid = [&](int i) {      // inline asm: the waits below are counted by hand (hipcc would use vmcnt(0))
        if (!MAPPED) {
            int64_t m = mbase + i * 16;
            m = m < g.M ? m : g.M - 1;
            const ov_bf16* src = g.R + m * g.ldr;
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(rv[i][0]) : "v"(src + nc0));
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(rv[i][1]) : "v"(src + nc1));
            return;
        }
        unsigned m = (unsigned)m0 + rb + i * 16 + fr;
        m = m < (unsigned)g.M ? m : (unsigned)g.M - 1;
        const unsigned rrow = g.resid_mod ? (m % (unsigned)g.resid_mod) + g.resid_off : m;
        const ov_bf16* src = g.R + (int64_t)rrow * g.ldr;
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(rv[i][0]) : "v"(src + nc0));
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(rv[i][1]) : "v"(src + nc1));
    };
    if (EPI >= OV_EPI_BIAS_RESIDUAL) {
#pragma unroll
        for (int i = 0; i < 4; ++i) load_resid(i);
    }
    // Parameter block reads are inline asm: as plain LDS loads hipcc orders them behind every LDS-DMA in flight
    // (s_waitcnt vmcnt(0)), i.e. behind the next tile's K-tile 1 that was issued a moment ago.
    f32x4_t bq[4], sq[4];
    f32x2_t stq[8];
    const unsigned pa = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const char*)(prm + (cw + fq * 4) * 4);
    const unsigned ra = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const char*)(prm + 2048 + (rb + fr) * 8);
    // (no branch between a read and its wait: a merge point would make hipcc copy the destination registers early)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(bq[j]) : "v"(pa), "n"(j * 64));
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(sq[j]) : "v"(pa), "n"(1024 + j * 64));
    }
    const unsigned ra2 = htile ? ra - 512u : ra;                   // half tile: passes 4-7 re-read the rows of passes 0-3 (in range, unused)
#pragma unroll
    for (int i = 0; i < 8; ++i) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(stq[i]) : "v"(i < 4 ? ra : ra2), "n"(i * 128));
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(bq[0]), "+v"(bq[1]), "+v"(bq[2]), "+v"(bq[3]), "+v"(sq[0]), "+v"(sq[1]), "+v"(sq[2]), "+v"(sq[3]));
    asm volatile("" : "+v"(stq[0]), "+v"(stq[1]), "+v"(stq[2]), "+v"(stq[3]), "+v"(stq[4]), "+v"(stq[5]), "+v"(stq[6]), "+v"(stq[7]));
    if (wst != nullptr && lane == 0) wst[2] = __builtin_amdgcn_s_memtime();
    const bool has_bias = g.bias != nullptr;
    float bv[4][4], sv[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            bv[j][e] = has_bias ? bq[j][e] : 0.f;
            sv[j][e] = FOLD ? sq[j][e] : 0.f;
        }
    u32x4_t vo[8][2];
    auto put = [&](int i) {
        const unsigned m = (unsigned)m0 + rb + i * 16 + fr;
        const bool live = i < 4 || !htile;
        ov_bf16* dst;
        if (!MAPPED) {
            dst = cbase + (int64_t)(i * 16) * g.ldc;
        } else {
            const unsigned orow = g.out_group ? m + m / (unsigned)g.out_group + 1 : m;
            dst = g.C + (int64_t)orow * g.ldc + n_lo;
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            u32x4_t o = vo[i][h];
            if (EPI >= OV_EPI_BIAS_RESIDUAL) o = epi_combine<EPI>(o, rv[i][h]);
            if (m < (unsigned)g.M && (h ? ncol1 : ncol0) && live) store16<(EPI >= OV_EPI_BIAS_RESIDUAL) ? OVHIP_ST_RESID : ST>(dst + h * 32, o);
            if (STATS) {
                // this lane's 8 consecutive columns -> octet; fq 0 | 2 | 1 | 3 hold octets 0 | 1 | 2 | 3 of the 32-column group
                float s8, q8;
                stat_octet(o, s8, q8);
                s8 = add_rowpair_even_first(add_halves_lo_first(s8));       // (oct0 + oct1) + (oct2 + oct3)
                q8 = add_rowpair_even_first(add_halves_lo_first(q8));
                // into the tile's [256 rows][8 groups] fp32-pair image in LDS; the workgroup writes it out in 64-byte row pieces
                // behind the tile barrier (16 scattered 8-byte stores per pass measured slower than the pass over x they replace)
                if (fq == 0 && live) *(float2*)(stats_lds + ((rb + i * 16 + fr) * 8 + (cw >> 5) + h) * 8) = make_float2(s8, q8);
            }
        }
    };
    // vmcnt is counted by hand around the asm residual loads: `edge` tiles issue fewer than 2 stores per pass, so they
    // fall back to a full drain
    auto pin_rows = [&](int i0, int i1) {            // names the rows' registers behind the wait: no consumer can be scheduled above it
#pragma unroll
        for (int k = i0; k < i1; ++k) asm volatile("" : "+v"(rv[k][0]), "+v"(rv[k][1]));
    };
#define OV_WAIT_RESID(I0, I1, YOUNGER)                                        \
    do {                                                                      \
        if (edge) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           \
        else wait_vmcnt<(YOUNGER)>();                                         \
        pin_rows(I0, I1);                                                     \
    } while (0)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        if (wm == 1 && g.epi_prio > 0 && i == g.epi_prio) __builtin_amdgcn_s_setprio(0);
        if (EPI >= OV_EPI_BIAS_RESIDUAL && i == 4) {
#pragma unroll
            for (int k = 4; k < 8; ++k) load_resid(k);
        }
        f32x2_t nm = {0.f, 0.f}, rs = {1.f, 1.f};
        if (FOLD) {
            nm = f32x2_t{-stq[i][0], -stq[i][0]};
            rs = f32x2_t{stq[i][1], stq[i][1]};
        }
        u32x2_t pk[4], pk2[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            f32x2_t v01 = f32x2_t{acc[i][j][0], acc[i][j][1]};
            f32x2_t v23 = f32x2_t{acc[i][j][2], acc[i][j][3]};
            if (FOLD) {      // rstd * (acc - mean * colsum) + cvec, as two explicit FMAs (identical in every kernel variant)
                v01 = __builtin_elementwise_fma(f32x2_t{sv[j][0], sv[j][1]}, nm, v01);
                v23 = __builtin_elementwise_fma(f32x2_t{sv[j][2], sv[j][3]}, nm, v23);
                v01 = __builtin_elementwise_fma(v01, rs, f32x2_t{bv[j][0], bv[j][1]});
                v23 = __builtin_elementwise_fma(v23, rs, f32x2_t{bv[j][2], bv[j][3]});
            } else {
                v01 += f32x2_t{bv[j][0], bv[j][1]};
                v23 += f32x2_t{bv[j][2], bv[j][3]};
            }
            if (KEEP) pk2[j] = u32x2_t{pack_bf16x2(v01[0], v01[1]), pack_bf16x2(v23[0], v23[1])};
            if (EPI == OV_EPI_BIAS_GELU_ERF) gelu_erf_f2x2(v01, v23);
            if (EPI == OV_EPI_BIAS_GELU_TANH) { v01 = gelu_tanh_f2(v01); v23 = gelu_tanh_f2(v23); }
            pk[j] = u32x2_t{pack_bf16x2(v01[0], v01[1]), pack_bf16x2(v23[0], v23[1])};
        }
        if (KEEP) {          // the pre-activation rows, same lane -> chunk map as the output (never MAPPED, never residual)
            const unsigned m = (unsigned)m0 + rb + i * 16 + fr;
            ov_bf16* dst2 = cbase2 + (int64_t)(i * 16) * g.ldc2;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const u32x2_t s0 = __builtin_amdgcn_permlane16_swap(pk2[2 * h][0], pk2[2 * h + 1][0], false, false);
                const u32x2_t s1 = __builtin_amdgcn_permlane16_swap(pk2[2 * h][1], pk2[2 * h + 1][1], false, false);
                if (m < (unsigned)g.M && (h ? ncol1 : ncol0) && (i < 4 || !htile)) store16<ST>(dst2 + h * 32, u32x4_t{s0[0], s1[0], s0[1], s1[1]});
            }
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            // odd 16-lane rows of block 2h <-> even rows of block 2h + 1 (one swap per dword)
            const u32x2_t s0 = __builtin_amdgcn_permlane16_swap(pk[2 * h][0], pk[2 * h + 1][0], false, false);
            const u32x2_t s1 = __builtin_amdgcn_permlane16_swap(pk[2 * h][1], pk[2 * h + 1][1], false, false);
            vo[i][h] = u32x4_t{s0[0], s1[0], s0[1], s1[1]};
        }
        if (EPI >= OV_EPI_BIAS_RESIDUAL) {
            // in flight behind the rows waited for: i == 3 -> after the second batch of loads at i == 4.  Stores start once every
            // load is out: at i == 4 the rows of passes 4-7 (8 loads) are younger than those of passes 0-3; later, per pass, the
            // remaining row loads plus the stores of all earlier passes = 14 every time
            if (i == 4) { OV_WAIT_RESID(0, 4, 8); put(0); put(1); put(2); put(3); }
            if (i >= 4) { OV_WAIT_RESID(i, i + 1, 14); put(i); }
        } else {
            put(i);
        }
        if (wst != nullptr && lane == 0 && (i == 1 || i == 4)) wst[i == 1 ? 3 : 4] = __builtin_amdgcn_s_memtime();
    }
    if (wst != nullptr && lane == 0) wst[5] = __builtin_amdgcn_s_memtime();
#undef OV_WAIT_RESID
}

// The same epilogue with the 16-byte stores made row-contiguous through a wave-local LDS image (8 lanes x 16 B = one 128-B line):
// coalesced stores (16 TA cycles per instruction against ~70 for the row-per-lane form above), at the price of the LDS round trip.
template <int EPI, bool FOLD, bool MAPPED, int ST = OVHIP_ST_LDS>
__device__ __forceinline__ void epilogue_stream_lds(const GemmArgs& g, f32x4_t (&acc)[8][4], char* img, const char* prm,
                                                int64_t m0, int n0, int wave, int lane, bool edge, unsigned long long* wst, bool htile) {
    const int wm = wave >> 2;
    const int rb = htile ? (wave >> 1) * 64 : wm * 128;            // (see epilogue_stream)
    const int cw = htile ? (wave & 1) * 64 : (wave & 3) * 64;
    const int fr = lane & 15, fq = lane >> 4;
    const int er = lane >> 3, ec = lane & 7;
    const int n = n0 + cw + ec * 8;
    const bool ncol = n < g.N;
    // Residual rows (unpredicated, clamped addresses -- a predicated load makes hipcc serialise the loads behind
    // vmcnt(0)).  Every load of the epilogue is issued before its first store: rows of passes 0-3 now, rows of passes
    // 4-7 once four accumulator rows have been retired (register room), and only then the first store.
    u32x4_t rv[8][2];
    const int nc = ncol ? n : g.N - 8;
    auto load_resid = [&](int i) {      // inline asm: the waits below are counted by hand (hipcc would use vmcnt(0))
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            unsigned m = (unsigned)m0 + rb + i * 16 + it * 8 + er;
            m = m < (unsigned)g.M ? m : (unsigned)g.M - 1;
            const unsigned rrow = g.resid_mod ? (m % (unsigned)g.resid_mod) + g.resid_off : m;
            const ov_bf16* src = g.R + (int64_t)rrow * g.ldr + nc;
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(rv[i][it]) : "v"(src));
        }
    };
    if (EPI >= OV_EPI_BIAS_RESIDUAL) {
#pragma unroll
        for (int i = 0; i < 4; ++i) load_resid(i);
    }
    // Parameter block reads are inline asm: as plain LDS loads hipcc orders them behind every LDS-DMA in flight
    // (s_waitcnt vmcnt(0)), i.e. behind the next tile's K-tile 1 that was issued a moment ago.
    f32x4_t bq[4], sq[4];
    f32x2_t stq[8];
    const unsigned pa = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const char*)(prm + (cw + fq * 4) * 4);
    const unsigned ra = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const char*)(prm + 2048 + (rb + fr) * 8);
    // (no branch between a read and its wait: a merge point would make hipcc copy the destination registers early)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(bq[j]) : "v"(pa), "n"(j * 64));
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(sq[j]) : "v"(pa), "n"(1024 + j * 64));
    }
    const unsigned ra2 = htile ? ra - 512u : ra;                   // half tile: passes 4-7 re-read the rows of passes 0-3 (in range, unused)
#pragma unroll
    for (int i = 0; i < 8; ++i) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(stq[i]) : "v"(i < 4 ? ra : ra2), "n"(i * 128));
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(bq[0]), "+v"(bq[1]), "+v"(bq[2]), "+v"(bq[3]), "+v"(sq[0]), "+v"(sq[1]), "+v"(sq[2]), "+v"(sq[3]));
    asm volatile("" : "+v"(stq[0]), "+v"(stq[1]), "+v"(stq[2]), "+v"(stq[3]), "+v"(stq[4]), "+v"(stq[5]), "+v"(stq[6]), "+v"(stq[7]));
    if (wst != nullptr && lane == 0) wst[2] = __builtin_amdgcn_s_memtime();
    const bool has_bias = g.bias != nullptr;
    float bv[4][4], sv[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            bv[j][e] = has_bias ? bq[j][e] : 0.f;
            sv[j][e] = FOLD ? sq[j][e] : 0.f;
        }
    char* const wr = img + fr * 128 + (fq & 1) * 8;
    const int wsw = fr & 7;
    const char* const rd = img + er * 128 + ((ec ^ er) << 4);       // rows er and er + 8 share (row & 7)
    ov_bf16* const cbase_lds = !MAPPED ? g.C + (m0 + rb + er) * g.ldc + n : nullptr;
    u32x4_t vo[8][2];
    auto put = [&](int i) {
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            u32x4_t o = vo[i][it];
            if (EPI >= OV_EPI_BIAS_RESIDUAL) o = epi_combine<EPI>(o, rv[i][it]);
            const unsigned m = (unsigned)m0 + rb + i * 16 + it * 8 + er;
            const bool live = i < 4 || !htile;
            if (!MAPPED) {   // no row map: the lane's row pointer of pass 0, stepped by whole rows
                if (m < (unsigned)g.M && ncol && live) store16<(EPI >= OV_EPI_BIAS_RESIDUAL) ? OVHIP_ST_RESID : ST>(cbase_lds + (int64_t)(i * 16 + it * 8) * g.ldc, o);
                continue;
            }
            if (m < (unsigned)g.M && ncol && live) {
                const unsigned orow = g.out_group ? m + m / (unsigned)g.out_group + 1 : m;
                store16<(EPI >= OV_EPI_BIAS_RESIDUAL) ? OVHIP_ST_RESID : ST>(g.C + (int64_t)orow * g.ldc + n, o);
            }
        }
    };
    // vmcnt is counted by hand around the asm residual loads: `edge` tiles issue fewer than 2 stores per pass, so they
    // fall back to a full drain
    auto wait_resid = [&](int i0, int i1, int younger) {
        if (edge) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (younger == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
#pragma unroll
        for (int k = i0; k < i1; ++k) asm volatile("" : "+v"(rv[k][0]), "+v"(rv[k][1]));
    };
#pragma unroll
    for (int i = 0; i <= 8; ++i) {
        if (wm == 1 && g.epi_prio > 0 && i == g.epi_prio) __builtin_amdgcn_s_setprio(0);
        if (EPI >= OV_EPI_BIAS_RESIDUAL && i == 4) {
#pragma unroll
            for (int k = 4; k < 8; ++k) load_resid(k);
        }
        if (i < 8) {
            f32x2_t nm = {0.f, 0.f}, rs = {1.f, 1.f};
            if (FOLD) {
                nm = f32x2_t{-stq[i][0], -stq[i][0]};
                rs = f32x2_t{stq[i][1], stq[i][1]};
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                f32x2_t v01 = f32x2_t{acc[i][j][0], acc[i][j][1]};
                f32x2_t v23 = f32x2_t{acc[i][j][2], acc[i][j][3]};
                if (FOLD) {      // rstd * (acc - mean * colsum) + cvec, as two explicit FMAs (identical in every kernel variant)
                    v01 = __builtin_elementwise_fma(f32x2_t{sv[j][0], sv[j][1]}, nm, v01);
                    v23 = __builtin_elementwise_fma(f32x2_t{sv[j][2], sv[j][3]}, nm, v23);
                    v01 = __builtin_elementwise_fma(v01, rs, f32x2_t{bv[j][0], bv[j][1]});
                    v23 = __builtin_elementwise_fma(v23, rs, f32x2_t{bv[j][2], bv[j][3]});
                } else {
                    v01 += f32x2_t{bv[j][0], bv[j][1]};
                    v23 += f32x2_t{bv[j][2], bv[j][3]};
                }
                if (EPI == OV_EPI_BIAS_GELU_ERF) gelu_erf_f2x2(v01, v23);
                if (EPI == OV_EPI_BIAS_GELU_TANH) { v01 = gelu_tanh_f2(v01); v23 = gelu_tanh_f2(v23); }
                const u32x2_t pk = {pack_bf16x2(v01[0], v01[1]), pack_bf16x2(v23[0], v23[1])};
                *(u32x2_t*)(wr + (((j * 2 + (fq >> 1)) ^ wsw) << 4)) = pk;
            }
        }
        // rows of pass i-1 (read one pass ago) leave while pass i is in the LDS pipe; with residual rows the first
        // four passes are held back until the second half of the residual loads is out
        if (EPI >= OV_EPI_BIAS_RESIDUAL) {
            // in flight behind the rows waited for: i == 4: rows of passes 4-7 (8 loads); later: the remaining row loads
            // plus the stores of all earlier passes = 14 every time
            if (i == 4) { wait_resid(0, 4, 8); put(0); put(1); put(2); put(3); }
            else if (i > 4) { wait_resid(i - 1, i, 14); put(i - 1); }
        } else if (i > 0) {
            put(i - 1);
        }
        if (i < 8) {
            vo[i][0] = *(const u32x4_t*)(rd);
            vo[i][1] = *(const u32x4_t*)(rd + 1024);
        }
        if (wst != nullptr && lane == 0 && (i == 1 || i == 4)) wst[i == 1 ? 3 : 4] = __builtin_amdgcn_s_memtime();
    }
    if (wst != nullptr && lane == 0) wst[5] = __builtin_amdgcn_s_memtime();
}

struct TileSrc { const ov_bf16* a0; const ov_bf16* a1; const ov_bf16* w0; const ov_bf16* w1; };   // per-lane staging sources

template <int EPI, bool FOLD, bool DIRECT, bool MAPPED, bool KEEP = false, bool STATS = false>
__global__ __launch_bounds__(NTHREADS, 2) void gemm_bf16_persist(const GemmArgs g) {
    __shared__ __attribute__((aligned(16))) char smem[DIRECT ? IMG_OFF + (STATS ? 256 * 64 : 0) : SMEM_PERSIST];   // STATS: + the tile's statistics image
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // ---- static persistent schedule: XCD x owns a contiguous run of tiles (n fastest), its workgroups stride it ----
    // Plain walk (ngroup == tiles_n): the linear n-fastest tile list is cut into 8 contiguous runs.  Grouped walk (wide N):
    // XCD x owns whole row panels [p0, p0 + pc) and walks them n-group by n-group -- (group, panel, n within group) -- so
    // the W slice of a group (ngroup x 256 rows x K) stays in that XCD's 4 MiB L2 while the A panels stream past it;
    // the plain walk re-fetches all of W from the Infinity Cache every round (8 MB > L2 at N = 4096, K = 1024).
    const int nwg = g.tiles_m * g.tiles_n;
    const int G = gridDim.x, bid = blockIdx.x;
    const int xcd = bid & 7, li = bid >> 3;
    const bool grouped = g.ngroup < g.tiles_n;
    int xstart, xcnt, p0 = 0, pc = 1;
    if (grouped) {
        const int pq = g.tiles_m >> 3, pr = g.tiles_m & 7;
        p0 = xcd * pq + (xcd < pr ? xcd : pr);
        pc = pq + (xcd < pr ? 1 : 0);
        xstart = 0;
        xcnt = pc * g.tiles_n;
    } else if (g.rotmask) {              // rotated plain walk: runs of whole row panels (the rotation permutes the tiles of a panel)
        const int pq = g.tiles_m >> 3, pr = g.tiles_m & 7;
        xstart = (xcd * pq + (xcd < pr ? xcd : pr)) * g.tiles_n;
        xcnt = (pq + (xcd < pr ? 1 : 0)) * g.tiles_n;
    } else {
        const int q8 = nwg >> 3, r8 = nwg & 7;
        xstart = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
        xcnt = q8 + (xcd < r8 ? 1 : 0);
    }
    const int nper = (G - xcd + 7) >> 3;
    int tcur = li;
    if (tcur >= xcnt) return;
    if (g.stagger > 0) {
        // De-synchronise the CUs of an XCD: in lockstep every CU reaches its epilogue at the same moment and the 4 MB an XCD then
        // writes at once queue on its fabric link; classes start a fraction of a tile period apart so the bursts interleave.
        const int cls = li % g.stagger_classes;
        if (cls) {
            const unsigned long long t0 = __builtin_amdgcn_s_memtime();
            const unsigned long long want = (unsigned long long)cls * (unsigned)g.stagger;
            while (__builtin_amdgcn_s_memtime() - t0 < want) __builtin_amdgcn_s_sleep(16);
        }
    }

    TileSrc cur, nxt;
    int64_t m0, nm0 = 0;
    int n0, nn0 = 0;
    // The walk as a mixed-radix counter (n within group | panel | group), stepped by the workgroup's stride `nper` with carries: no
    // integer division per tile (hipcc expands each into ~25 VALU instructions in front of the tile's first K-tile).  Plain walk =
    // the same counter with radices (tiles_n | unbounded | -): digit 1 is then the row tile itself.
    const int rad0 = grouped ? g.ngroup : g.tiles_n;
    const int rad1 = grouped ? pc : 0x7fffffff;
    int d0, d1, d2, s0, s1, s2;                       // digits of the NEXT tile to set up / of the stride
    int rotk = 0;                                     // tiles set up so far (rotated walk)
    {
        const int base = grouped ? li : xstart + li;
        d0 = base % rad0;
        const int q = base / rad0;
        d1 = grouped ? q % rad1 : q;
        d2 = grouped ? q / rad1 : 0;
        s0 = nper % rad0;
        const int qs = nper / rad0;
        s1 = grouped ? qs % rad1 : qs;
        s2 = grouped ? qs / rad1 : 0;
    }
    auto set_tile = [&](TileSrc& ts, int64_t& mm, int& nn) {      // sets up the tile the counter points at, then steps the counter
        // the lane's staging row / chunk, rebuilt per tile from an opaque lane id: as loop invariants hipcc kept the 64-bit per-lane
        // offsets (row * ld + chunk) live through the whole tile loop -- and spilled them in the variant with row statistics
        const int tid_f = wave * 64 + fresh_lane();
        const int srow = tid_f >> 2;
        const int schunk = (tid_f & 3) ^ swz4(srow);
        // (rotated walk: this workgroup's k-th tile takes n index (d0 + k) mod tiles_n; its d0 never changes, the stride being a
        // multiple of tiles_n, and the tiles_n workgroups that share a row panel in a round hold distinct d0)
        const int tmf = p0 + d1;
        const int tm = g.rev ? g.tiles_m - 1 - tmf : tmf, tn = g.rotmask ? ((d0 + rotk) & g.rotmask) : d2 * rad0 + d0;
        ++rotk;
        d0 += s0;
        if (d0 >= rad0) { d0 -= rad0; ++d1; }
        d1 += s1;
        if (d1 >= rad1) { d1 -= rad1; ++d2; }
        d2 += s2;
        mm = (int64_t)tm * BM;
        nn = tn * BN;
        int64_t ar0 = mm + srow, ar1 = mm + 128 + srow;
        ar0 = ar0 < g.M ? ar0 : g.M - 1;
        ar1 = ar1 < g.M ? ar1 : g.M - 1;
        int wr0 = nn + srow, wr1 = nn + 128 + srow;
        wr0 = wr0 < g.N ? wr0 : g.N - 1;
        wr1 = wr1 < g.N ? wr1 : g.N - 1;
        ts.a0 = g.A + ar0 * g.lda + schunk * 8;
        ts.a1 = g.A + ar1 * g.lda + schunk * 8;
        ts.w0 = g.W + (int64_t)wr0 * g.ldw + schunk * 8;
        ts.w1 = g.W + (int64_t)wr1 * g.ldw + schunk * 8;
    };
    auto advance = [&](TileSrc& ts) { ts.a0 += BK; ts.a1 += BK; ts.w0 += BK; ts.w1 += BK; };
    char* const sbase = smem + wave * 1024;
    // piece p of the K-tile `ts` points at -> LDS buffer at byte offset boff.  Pieces 2, 3 are the k 32-63 half: the +64 B
    // ride in the instruction offset, which the DMA also adds to its LDS address (hence the -64 on M0).
    auto stage_piece = [&](const TileSrc& ts, int boff, int p) {
        char* dst = sbase + boff + p * PIECE_BYTES;
        const ov_bf16* s0 = (p & 1) ? ts.w0 : ts.a0;
        const ov_bf16* s1 = (p & 1) ? ts.w1 : ts.a1;
#ifndef OVHIP_NT_A
#define OVHIP_NT_A 0          /* experiment: 2 = non-temporal (streaming) cache policy for the A pieces (W keeps the default) */
#endif
        if (p & 1) {
            if (p >> 1) {
                __builtin_amdgcn_global_load_lds((gptr_t)s0, (lptr_t)(dst - 64), 16, 64, 0);
                __builtin_amdgcn_global_load_lds((gptr_t)s1, (lptr_t)(dst + 8192 - 64), 16, 64, 0);
            } else {
                __builtin_amdgcn_global_load_lds((gptr_t)s0, (lptr_t)dst, 16, 0, 0);
                __builtin_amdgcn_global_load_lds((gptr_t)s1, (lptr_t)(dst + 8192), 16, 0, 0);
            }
        } else {
            if (p >> 1) {
                __builtin_amdgcn_global_load_lds((gptr_t)s0, (lptr_t)(dst - 64), 16, 64, OVHIP_NT_A);
                __builtin_amdgcn_global_load_lds((gptr_t)s1, (lptr_t)(dst + 8192 - 64), 16, 64, OVHIP_NT_A);
            } else {
                __builtin_amdgcn_global_load_lds((gptr_t)s0, (lptr_t)dst, 16, 0, OVHIP_NT_A);
                __builtin_amdgcn_global_load_lds((gptr_t)s1, (lptr_t)(dst + 8192), 16, 0, OVHIP_NT_A);
            }
        }
    };
    // tile parameters -> LDS block `slot` (lane-linear DMA images): wave 0 the 256 bias values, wave 1 the column sums,
    // every wave 32 rows of {mean, rstd} (dword granules, so a clamped row never shifts its neighbours)
    auto stage_params = [&](int slot, int64_t mm, int nn) {
        char* dst = smem + PRM_OFF + slot * 4096;
        const int lane = fresh_lane();                  // (shadows the kernel's: rebuilt per tile, see fresh_lane)
        int c = nn + lane * 4;
        c = c + 4 <= g.N ? c : g.N - 4;
        if (wave == 0 && g.bias != nullptr) __builtin_amdgcn_global_load_lds((gptr_t)(g.bias + c), (lptr_t)dst, 16, 0, 0);
        if (FOLD) {
            if (wave == 1) __builtin_amdgcn_global_load_lds((gptr_t)(g.colsum + c), (lptr_t)(dst + 1024), 16, 0, 0);
            const int f = wave * 64 + lane;
            int64_t r = mm + (f >> 1);
            r = r < g.M ? r : g.M - 1;
            __builtin_amdgcn_global_load_lds((gptr_t)(g.rowstats + 2 * r + (f & 1)), (lptr_t)(dst + 2048 + wave * 256), 4, 0, 0);
        }
    };

    const int wm = wave >> 2;
    // HALF TILES.  A tile with at most 128 valid columns (the last n-tile of N = 384, 1152, ...: S/8's D = 384 leaves a quarter of the
    // out-proj / c_proj MFMAs on padding otherwise) is computed by the waves as a 4 x 2 grid of 64 x 64 sub-tiles instead of 2 x 4 of
    // 128 x 64: only the m-half-0 phases of a K-tile hold MFMAs (acc[0..3]), staging, waits and barriers are unchanged (same DMA
    // count, so every counted wait stands), and each output element sums its products in the same order as in a full tile.
    bool htile;
    int a_lane, w_lane;
    auto set_lanes = [&](int nn) {
        htile = g.half_ok && g.N - nn <= 128;
        const int lf = fresh_lane();
        const int fr = lf & 15, fq = lf >> 4;
        const int lsw = (fq ^ swz4(fr)) << 4;
        a_lane = ((htile ? (wave >> 1) * 64 : wm * 128) + fr) * 64 + lsw;
        w_lane = PIECE_BYTES + ((htile ? (wave & 1) : (wave & 3)) * 64 + fr) * 64 + lsw;
    };
    const int nt = g.K / BK;                                       // >= 3 (launcher)
    set_tile(cur, m0, n0);
    set_lanes(n0);
#pragma unroll
    for (int j = 0; j < 4; ++j) stage_piece(cur, 0, j);
    advance(cur);
#pragma unroll
    for (int j = 0; j < 4; ++j) stage_piece(cur, STAGE_BYTES, j);
    advance(cur);                                                  // invariant at tile start: `cur` points at K-tile 2
    stage_params(0, m0, n0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (wm == 1) __builtin_amdgcn_s_barrier();                     // stagger the lower wave group by one interval

    int cb = 0, pslot = 0;   // LDS buffer (byte offset) of the K-tile being consumed; parameter block of the current tile
    bool strict = true;      // K-tile 1 wait of this tile must not count on 16 younger stores (first tile / after an edge tile)
    bool has_next = false;
    int titer = 0;
#ifndef OVHIP_STAMPS
#define OVHIP_STAMPS 0       /* 1: the s_memtime diagnostics of tools/gemm_stamps.py / gemm_wave_stamps.py -- a VARIANT build (tools/build_variant.py */
#endif                       /* stamps gemm.hip -DOVHIP_STAMPS=1, OVHIP_LIB=libovhip_stamps.so); compiled out of the product: their address arithmetic cost registers */
    auto stamp = [&](int k) {
        if (OVHIP_STAMPS && g.stamps != nullptr && tid == 0 && titer < g.stamp_slots)
            g.stamps[((size_t)bid * g.stamp_slots + titer) * 8 + k] = __builtin_amdgcn_s_memtime();
    };
    f32x4_t acc[8][4];
    bf16x8_t af[4], wf[4];
    // One K-tile = four phases (k-half x m-half), each a LOAD segment (LDS fragment reads, one piece of the K-tile after
    // next by LDS-DMA) and a COMPUTE segment (16 MFMAs) between barriers.  KIND selects what is staged and waited for, at
    // compile time, so the steady-state body carries no branch:
    //   0: K-tile 0 of a tile -- K-tile 1 is already in flight (previous epilogue / prologue), nothing is staged; its
    //      wait leaves the previous epilogue's 16 stores outstanding;   1: K-tile 1 -- stages K-tile 2, first wait at p 3;
    //   2: steady state;   3: last K-tile -- stages K-tile 0 of the next tile, if any.
    auto ktile = [&](auto kind) {
        constexpr int KIND = decltype(kind)::value;
        const char* s = smem + cb;
        const int nb = cb ^ STAGE_BYTES;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int kh = p >> 1, mh = p & 1;
            const char* sp = s + kh * (2 * PIECE_BYTES);
            if (mh == 0) {
#pragma unroll
                for (int j = 0; j < 4; ++j) wf[j] = *(const bf16x8_t*)(sp + w_lane + j * 1024);
            }
            if (mh == 0 || !htile) {
#pragma unroll
                for (int i = 0; i < 4; ++i) af[i] = *(const bf16x8_t*)(sp + a_lane + (mh * 4 + i) * 1024);
            }
            if (KIND == 1 || KIND == 2) stage_piece(cur, nb, p);
            if (KIND == 3) { if (has_next) stage_piece(nxt, nb, p); }
            if (p & 1) {
                if (KIND == 0) {
                    if (p == 3) {
                        if (strict) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
                    }
                } else if (KIND == 1) {
                    if (p == 3) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                } else if (KIND == 2) {
                    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                } else {
                    if (has_next) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                    else if (p == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
            if (mh == 0 || !htile) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        // a tile's first products start from a literal zero C operand: no 128 v_mov per tile to clear the accumulators
                        if (KIND == 0 && kh == 0)
                            acc[mh * 4 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], af[i], f32x4_t{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                        else
                            acc[mh * 4 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], af[i], acc[mh * 4 + i][j], 0, 0, 0);
                    }
            }
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    for (;;) {
        stamp(0);
        const int tnext = tcur + nper;
        has_next = tnext < xcnt;
        ktile(IntC<0>{});
        cb ^= STAGE_BYTES;
        stamp(4);
        ktile(IntC<1>{});
        cb ^= STAGE_BYTES;
        advance(cur);
        stamp(5);
        for (int t = 2; t < nt - 1; ++t) {
            ktile(IntC<2>{});
            cb ^= STAGE_BYTES;
            advance(cur);
            if (OVHIP_STAMPS && g.stamps != nullptr) {             // diagnostics only: where inside the main loop the time goes
                if (t == 3) stamp(6);
                else if (t == 7) stamp(7);
            }
        }
        // the next tile's staging pointers only now, in front of the K-tile that first uses them: set up at the top of the tile they
        // were 8 registers live (or spilled and reloaded) through the whole main loop
        if (has_next) set_tile(nxt, nm0, nn0);
        ktile(IntC<3>{});
        stamp(1);
        unsigned long long* wst = (OVHIP_STAMPS && g.wstamps != nullptr && titer < g.stamp_slots)
                                      ? g.wstamps + (((size_t)bid * g.stamp_slots + titer) * 8 + wave) * 8 : nullptr;
        if (wst != nullptr && lane == 0) wst[0] = __builtin_amdgcn_s_memtime();
        if (wm == 0) __builtin_amdgcn_s_barrier();                 // re-align: every wave is past its last COMPUTE segment
        stamp(2);
        if (has_next) {
            // the buffer of the last K-tile is free: K-tile 1 of the next tile and its parameters go out ahead of the stores
            advance(nxt);
#pragma unroll
            for (int j = 0; j < 4; ++j) stage_piece(nxt, cb, j);
            advance(nxt);
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");       // K-tile 0 of the next tile (issued a K-tile ago) has landed
            stage_params(pslot ^ 1, nm0, nn0);
        }
        __builtin_amdgcn_sched_barrier(0);
        const bool edge = (m0 + BM > g.M) || (n0 + BN > g.N);      // an edge tile issues fewer than 16 stores per wave
        if (wst != nullptr && lane == 0) wst[1] = __builtin_amdgcn_s_memtime();
        if (g.epi_prio && wm == 1) __builtin_amdgcn_s_setprio(1);
        // Store policy by the size of the output (launcher): streaming (nt) stores keep a wide activation (qkv, hidden: 0.4-0.9 GB at
        // the benchmark shapes) from displacing the operands in L2; an output that fits the 256 MB Infinity Cache is stored plainly so
        // that its consumer finds it there (Ti/16: 6.68 -> 6.42 ms per step).  The residual forms always store plainly (x is re-read
        // at once by the row statistics and the next GEMM).  Two inlined copies of the epilogue, one wave-uniform branch per tile.
        constexpr bool DUAL = EPI < OV_EPI_BIAS_RESIDUAL;
        if (DIRECT) {
            if (DUAL && g.st_plain) epilogue_stream<EPI, FOLD, MAPPED, KEEP, STATS, 0>(g, acc, smem + PRM_OFF + pslot * 4096, m0, n0, wave, lane, edge, wst, htile, smem + IMG_OFF);
            else epilogue_stream<EPI, FOLD, MAPPED, KEEP, STATS, OVHIP_ST_DIRECT>(g, acc, smem + PRM_OFF + pslot * 4096, m0, n0, wave, lane, edge, wst, htile, smem + IMG_OFF);
        } else {
            if (DUAL && g.st_plain) epilogue_stream_lds<EPI, FOLD, MAPPED, 0>(g, acc, smem + IMG_OFF + wave * 2048, smem + PRM_OFF + pslot * 4096, m0, n0, wave, lane, edge, wst, htile);
            else epilogue_stream_lds<EPI, FOLD, MAPPED, OVHIP_ST_LDS>(g, acc, smem + IMG_OFF + wave * 2048, smem + PRM_OFF + pslot * 4096, m0, n0, wave, lane, edge, wst, htile);
        }
        if (g.epi_prio && wm == 1) __builtin_amdgcn_s_setprio(0);
        stamp(3);
        ++titer;
        auto flush_stats = [&]() {
            // the tile's statistics image -> rowpart: wave w takes rows 32 w .. 32 w + 31, a lane one 16-byte quarter (two groups) of a
            // row's 64 bytes: 2 stores of 16 rows x 64 contiguous bytes.  (They sit behind the tile's output stores in the in-order
            // queue: K-tile 0's vmcnt(16) then retires two of those early -- harmless; an edge tile makes the next wait strict anyway.)
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int ln = fresh_lane();            // (not a loop invariant of the tile loop: see fresh_lane)
                const int row = wave * 32 + it * 16 + (ln >> 2), qd = ln & 3;
                const u32x4_t v = *(const u32x4_t*)(smem + IMG_OFF + row * 64 + qd * 16);
                const int64_t m = m0 + row;
                const int grp = (n0 >> 5) + qd * 2;
                if (m < g.M && grp * 32 < g.N) *(u32x4_t*)(g.rowpart + (m * (g.N >> 5) + grp) * 2) = v;
            }
        };
        if (!has_next) {
            if (STATS) { __builtin_amdgcn_s_barrier(); flush_stats(); }
            break;
        }
        strict = edge;
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();                               // next K-tile 0 visible to all
        if (STATS) flush_stats();
        if (wst != nullptr && lane == 0) wst[6] = __builtin_amdgcn_s_memtime();
        if (wm == 1) __builtin_amdgcn_s_barrier();                  // re-stagger
        cb ^= STAGE_BYTES;
        pslot ^= 1;
        cur = nxt;
        m0 = nm0; n0 = nn0;
        set_lanes(n0);
        tcur = tnext;
    }
}

// =====================================================================================================
// Skinny kernel: the small-M path (ov-zero-shot-test.py:167-195 encodes ONE image per call: M = 257 rows at L/14).  There the 256 x 256
// kernels put 8-32 workgroups on 256 CUs and every one of them walks its whole K extent one K-tile ahead of the HBM latency of the
// weights (4.7 ms per L/14 image, profiles/r02_probes.log).  Here a workgroup is 4 waves on a 64 x 64 tile (80-320 workgroups for
// the L/14 shapes at M = 257), operands arrive through a ring of FOUR LDS stages by LDS-DMA in whole 128-byte lines, three K-tiles
// ahead (counted vmcnt, one raw barrier per K-tile), so the weight stream is pulled by all CUs at once and its latency is covered.
// NO split-K: every output element accumulates its K products in the same order, through the same v_mfma_f32_16x16x32_bf16 (W as
// the A operand, k 0-31 then 32-63 of each K-tile), as in the 256 x 256 kernels, and the epilogue applies the same functions -- a
// row's result is bitwise the same whichever kernel computed it (batch 1 against batch 9: test_batch_invariance...).
// LDS image of a stage: [A 64 rows x 128 B][W 64 rows x 128 B], 16-byte chunk c of row r stored at chunk c ^ (r & 7) (on the DMA source
// side and on the read side), conflict-free for the 16x16x32 operand reads (the v1 kernel's image).
constexpr int SK_BM = 64, SK_BN = 64, SK_STAGES = 4;
constexpr int SK_STAGE_BYTES = (SK_BM + SK_BN) * 128;            // 16 KiB

template <int EPI>
__global__ __launch_bounds__(256) void gemm_bf16_skinny(const GemmArgs g) {
    __shared__ __attribute__((aligned(16))) char smem[SK_STAGES * SK_STAGE_BYTES];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tn = blockIdx.x % g.tiles_n, tm = blockIdx.x / g.tiles_n;          // n fastest: neighbours share the A rows in L2
    const int64_t m0 = (int64_t)tm * SK_BM;
    const int n0 = tn * SK_BN;

    // staging: wave w issues, per K-tile, A instructions 2 w, 2 w + 1 and W instructions 2 w, 2 w + 1 (8 rows x 128 B each)
    const ov_bf16* asrc[2];
    const ov_bf16* wsrc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int row = (2 * wave + j) * 8 + (lane >> 3);
        const int ch = (lane & 7) ^ (row & 7);
        int64_t ar = m0 + row;
        ar = ar < g.M ? ar : g.M - 1;                 // clamp: tail rows load a valid row, never stored
        int wr = n0 + row;
        wr = wr < g.N ? wr : g.N - 1;
        asrc[j] = g.A + ar * g.lda + ch * 8;
        wsrc[j] = g.W + (int64_t)wr * g.ldw + ch * 8;
    }
    auto stage = [&](int t) {
        char* dst = smem + (t & (SK_STAGES - 1)) * SK_STAGE_BYTES + wave * 2048;
        const int k0 = t * BK;
        __builtin_amdgcn_global_load_lds((gptr_t)(asrc[0] + k0), (lptr_t)dst, 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gptr_t)(asrc[1] + k0), (lptr_t)(dst + 1024), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gptr_t)(wsrc[0] + k0), (lptr_t)(dst + SK_BM * 128), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gptr_t)(wsrc[1] + k0), (lptr_t)(dst + SK_BM * 128 + 1024), 16, 0, 0);
    };
    // wave (wm, wn) owns rows wm * 32 .. + 32, columns wn * 32 .. + 32 of the tile: 2 x 2 accumulator fragments
    const int wm = wave >> 1, wn = wave & 1;
    const int fr = lane & 15, fq = lane >> 4;
    const int a_off = (wm * 32 + fr) * 128;
    const int w_off = SK_BM * 128 + (wn * 32 + fr) * 128;
    const int sw0 = (fq ^ (fr & 7)) << 4, sw1 = ((4 + fq) ^ (fr & 7)) << 4;          // rows fr and fr + 16 share (row & 7)

    f32x4_t acc[2][2];
    const int nt = g.K / BK;
    for (int t = 0; t < 3 && t < nt; ++t) stage(t);
    for (int t = 0; t < nt; ++t) {
        // K-tile t has landed once at most min(2, nt - 1 - t) younger K-tiles (4 DMAs each) are still in flight
        const int ahead = nt - 1 - t;
        if (ahead >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (ahead == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                     // everyone's pieces of K-tile t are in; everyone has read K-tile t - 1
        if (t + 3 < nt) stage(t + 3);                       // into the stage K-tile t - 1 occupied
        const char* s = smem + (t & (SK_STAGES - 1)) * SK_STAGE_BYTES;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int sw = kk ? sw1 : sw0;
            bf16x8_t af[2], wf[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) wf[j] = *(const bf16x8_t*)(s + w_off + j * 2048 + sw);
#pragma unroll
            for (int i = 0; i < 2; ++i) af[i] = *(const bf16x8_t*)(s + a_off + i * 2048 + sw);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    if (t == 0 && kk == 0)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], af[i], f32x4_t{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                    else
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], af[i], acc[i][j], 0, 0, 0);
                }
        }
    }
    // ---- epilogue: acc[i][j][e] = C[m0 + wm*32 + i*16 + fr][n0 + wn*32 + j*16 + fq*4 + e]; the arithmetic of gemm_epilogue, per element
    const bool fold = (EPI < OV_EPI_BIAS_RESIDUAL) && g.colsum != nullptr;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int64_t m = m0 + wm * 32 + i * 16 + fr;
        const int64_t mc = m < g.M ? m : g.M - 1;
        float rmean = 0.f, rrstd = 1.f;
        if (fold) {
            const float2 st = *(const float2*)(g.rowstats + 2 * mc);
            rmean = st.x; rrstd = st.y;
        }
        float gs = 0.f, gq = 0.f;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int nn = n0 + wn * 32 + j * 16 + fq * 4;
            const bool ok = m < g.M && nn < g.N;           // N % 8 == 0 and nn % 4 == 0: the 4 columns are in or out together
            const int nc = nn < g.N ? nn : g.N - 4;
            f32x2_t b01 = {0.f, 0.f}, b23 = {0.f, 0.f};
            if (g.bias != nullptr) {
                const float4 b4 = *(const float4*)(g.bias + nc);
                b01 = f32x2_t{b4.x, b4.y}; b23 = f32x2_t{b4.z, b4.w};
            }
            f32x2_t v01 = f32x2_t{acc[i][j][0], acc[i][j][1]};
            f32x2_t v23 = f32x2_t{acc[i][j][2], acc[i][j][3]};
            if (fold) {
                const float4 s4 = *(const float4*)(g.colsum + nc);
                const f32x2_t nm = {-rmean, -rmean}, rs = {rrstd, rrstd};
                v01 = __builtin_elementwise_fma(f32x2_t{s4.x, s4.y}, nm, v01);
                v23 = __builtin_elementwise_fma(f32x2_t{s4.z, s4.w}, nm, v23);
                v01 = __builtin_elementwise_fma(v01, rs, b01);
                v23 = __builtin_elementwise_fma(v23, rs, b23);
            } else {
                v01 += b01;
                v23 += b23;
            }
            if (EPI == OV_EPI_BIAS_GELU_ERF) gelu_erf_f2x2(v01, v23);
            if (EPI == OV_EPI_BIAS_GELU_TANH) { v01 = gelu_tanh_f2(v01); v23 = gelu_tanh_f2(v23); }
            u32x2_t pk = {pack_bf16x2(v01[0], v01[1]), pack_bf16x2(v23[0], v23[1])};
            if (EPI >= OV_EPI_BIAS_RESIDUAL) {
                const int64_t rrow = g.resid_mod ? (mc % g.resid_mod) + g.resid_off : mc;
                const u32x2_t rv = *(const u32x2_t*)(g.R + rrow * g.ldr + nc);
                const u32x4_t o = epi_combine<EPI>(u32x4_t{pk[0], pk[1], pk[0], pk[1]}, u32x4_t{rv[0], rv[1], rv[0], rv[1]});
                pk = u32x2_t{o[0], o[1]};
            }
            if (ok) {
                const int64_t orow = g.out_group ? m + m / g.out_group + 1 : m;
                *(u32x2_t*)(g.C + orow * g.ldc + nn) = pk;
            }
            if (EPI == OV_EPI_BIAS_RESIDUAL && g.rowpart != nullptr) {      // wave-uniform
                // the lane's quad (n fq*4 .. +3 of block j) -> octets (16-lane rows 0+1, 2+3) -> block (halves) -> group = block 0 + block 1
                float sb, qb;
                stat_quad(pk[0], pk[1], sb, qb);
                sb = add_halves_lo_first(add_rowpair_even_first(sb));
                qb = add_halves_lo_first(add_rowpair_even_first(qb));
                if (j == 0) { gs = sb; gq = qb; }
                else {
                    gs = __fadd_rn(gs, sb);
                    gq = __fadd_rn(gq, qb);
                    const int nb = n0 + wn * 32;                             // this wave's 32-column group
                    if (fq == 0 && m < g.M && nb < g.N)
                        *(float2*)(g.rowpart + (m * (g.N >> 5) + (nb >> 5)) * 2) = make_float2(gs, gq);
                }
            }
        }
    }
}

// The skinny kernel takes over while the shape gives the 256 x 256 kernels at most this many tiles (OVHIP_GEMM_SKINNY_TILES; 0 =
// never).  Measured crossover on the four L/14 block shapes (tools/dbg/skinny_cross.py): QKV / c_fc win with the skinny kernel up to
// 60 / 80 big tiles (M = 1028) and lose at 108 / 144 (M = 2056), out_proj / c_proj win at 68 (M = 4112) and lose at 132 (M = 8224).
int gemm_skinny_tiles() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("OVHIP_GEMM_SKINNY_TILES"); v = e ? atoi(e) : 96; }
    return v;
}

int num_cus() { return ov_num_cus(); }

thread_local const float* g_colsum = nullptr;      // set by ov_gemm_ln around its call into ov_gemm
thread_local const float* g_rowstats = nullptr;
thread_local ov_bf16* g_keep = nullptr;            // set by ov_gemm_keep around its call into ov_gemm
thread_local int64_t g_ldkeep = 0;
thread_local float* g_rowpart = nullptr;           // set by ov_gemm_rowparts around its call into ov_gemm
unsigned long long* g_stamps = nullptr;
unsigned long long* g_wstamps = nullptr;
int g_stamp_slots = 0;

int gemm_variant() {       // 0 = default (persistent ping-pong; skinny kernel for small M), 1 = v1 two-stage, 2 = non-persistent ping-pong,
    static int v = -1;     // 3 = four-wave prototype, 4 = skinny kernel for EVERY shape (tests: bitwise against the default)
    if (v < 0) {
        const char* e = getenv("OVHIP_GEMM_VARIANT");
        v = (e && e[0] >= '0' && e[0] <= '4') ? e[0] - '0' : 0;
    }
    return v;
}

// n-tiles per group of the persistent walk: groups of 4 once W no longer fits an XCD's L2 next to the streaming A panels
// (OVHIP_GEMM_NGROUP: 0 = never group, n = force groups of n where tiles_n is a multiple)
int gemm_ngroup(int tiles_m, int tiles_n, int K) {
    static int force = -2;
    if (force == -2) {
        const char* e = getenv("OVHIP_GEMM_NGROUP");
        force = e ? atoi(e) : -1;
    }
    int gsz = 4;
    if (force == 0) return tiles_n;
    if (force > 0) gsz = force;
    const int64_t wbytes = (int64_t)tiles_n * BN * K * 2;
    if (force < 0 && wbytes <= (3 << 20)) return tiles_n;
    if (tiles_n <= gsz || tiles_n % gsz || tiles_m < 16) return tiles_n;
    return gsz;
}

// start stagger of the persistent kernel, cycles per class (OVHIP_GEMM_STAGGER; experiment knob, 0 = off)
int gemm_stagger(int K) {
    static int v = -1;
    if (v < 0) { const char* e = getenv("OVHIP_GEMM_STAGGER"); v = e ? atoi(e) : 0; }
    (void)K;
    return v;
}
int gemm_stagger_classes() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("OVHIP_GEMM_STAGGER_CLASSES"); v = e ? atoi(e) : 4; if (v < 1) v = 1; }
    return v;
}

int gemm_epi_prio() {
    static int v = -1;
    // default 4: waves 4-7 (younger, they lose every VALU / LDS arbitration against waves 0-3 and finish the epilogue 1.4 k cycles
    // later, while waves 0-3 wait at the tile barrier) run the first four passes at priority 1: both groups then finish together.
    // Measured in the model: QKV 9.59 -> 9.41 ms, c_fc 12.55 -> 12.24 ms per step (-2 %); all eight passes at priority 1 just swaps
    // the roles (no gain).
    if (v < 0) { const char* e = getenv("OVHIP_GEMM_EPI_PRIO"); v = e ? atoi(e) : 4; }
    return v;
}

// Per-launch policy of the persistent kernel: half tiles and the rotated walk, store policy by output size, reverse row walk.
inline int env_flag(const char* name, int dflt) {                 // "0" / "1" (anything else: the default)
    const char* e = getenv(name);
    return (e && (e[0] == '0' || e[0] == '1')) ? e[0] - '0' : dflt;
}
template <int EPI>
void persist_policy(GemmArgs& a, int grid_x) {
    static const int rot_env = env_flag("OVHIP_GEMM_ROTATE", 1), half_env = env_flag("OVHIP_GEMM_HALF", 1), rev_env = env_flag("OVHIP_GEMM_REVERSE", 1);
    static const int64_t nt_min = [] { const char* e = getenv("OVHIP_GEMM_NT_MIN_MB"); return (int64_t)(e ? atoi(e) : 192) << 20; }();
    a.half_ok = half_env;
    // outputs of at least OVHIP_GEMM_NT_MIN_MB (192) MB are streamed, smaller ones stored plainly: their consumer finds them in the
    // 256 MB Infinity Cache (gemm_bf16_persist, "Store policy")
    a.st_plain = (int64_t)a.M * a.N * 2 < nt_min ? 1 : 0;
    // Row tiles from the last to the first for the residual GEMM that reads a wide hidden activation (c_proj: K >= 2 N): its
    // producer (c_fc) wrote the rows in ascending order, so the ones it wrote last are those the Infinity Cache still holds.
    // S/8@384 (906 MB hidden per layer) 35.03 -> 34.80 ms per step, L/14 (537 MB) 44.78 -> 44.70: small, free, bitwise the same
    // results.  OVHIP_GEMM_REVERSE=0 switches it off.
    a.rev = (rev_env && EPI == OV_EPI_BIAS_RESIDUAL && a.K >= 2 * a.N && a.out_group == 0 && a.resid_mod == 0) ? 1 : 0;
    // a half last n-tile under the plain walk with a workgroup stride that is a multiple of tiles_n: rotate (GemmArgs::rotmask)
    const int rem = a.N % BN, nper = grid_x >> 3;
    a.rotmask = 0;
    if (rot_env && half_env && rem > 0 && rem <= 128 && a.ngroup >= a.tiles_n && a.tiles_n > 1 && (a.tiles_n & (a.tiles_n - 1)) == 0 &&
        grid_x % 8 == 0 && nper % a.tiles_n == 0)
        a.rotmask = a.tiles_n - 1;
}

template <int EPI>
int launch(GemmArgs a, hipStream_t st) {
    int var = gemm_variant();
    if (a.C2 == nullptr && (var == 4 || (var == 0 && (int64_t)a.tiles_m * a.tiles_n <= gemm_skinny_tiles()))) {
        GemmArgs b = a;
        b.tiles_m = (int)((a.M + SK_BM - 1) / SK_BM);
        b.tiles_n = (a.N + SK_BN - 1) / SK_BN;
        if ((int64_t)b.tiles_m * b.tiles_n > 0x7fffffffLL) return OV_ERR_UNSUPPORTED;
        hipLaunchKernelGGL(gemm_bf16_skinny<EPI>, dim3((unsigned)(b.tiles_m * b.tiles_n)), dim3(256), 0, st, b);     // (writes a.rowpart itself)
        OV_LAUNCH_CHECK();
        return OV_OK;
    }
    if (var == 4) var = 0;
    // row-statistic partial sums: fused into the persistent kernel's direct residual epilogue; every other kernel is followed by the
    // stand-alone pass over its output (the same arithmetic, common.h: bitwise the same sums)
    float* const rowpart = a.rowpart;
    bool stats_fused = false;
    const int nwg = a.tiles_m * a.tiles_n;
    // fewer tiles than CUs (pooled heads, the tower's tail images): persistence buys nothing, use the plain launch
    // the persistent kernel also wins on grids somewhat smaller than the chip (one tile per workgroup, but its epilogue starts
    // without global round trips); tiny grids (the tower's tail images) stay on the plain launch
    static int min_persist_env = -2;
    if (min_persist_env == -2) { const char* e = getenv("OVHIP_GEMM_MINPERSIST"); min_persist_env = e ? atoi(e) : -1; }
    const int min_persist = min_persist_env >= 0 ? min_persist_env : num_cus();
    if (var == 0 && (nwg < min_persist || a.K < 3 * BK)) var = 2;
    // row maps (the patch embedding's GEMM) exist in the persistent kernel for the bias and residual epilogues only
    if (var == 0 && (a.out_group != 0 || a.resid_mod != 0) && EPI != OV_EPI_BIAS && EPI != OV_EPI_BIAS_RESIDUAL) var = 2;
    if (a.C2 != nullptr && var != 0) {
        // grids below the persistent kernel's threshold (the tower's tail images): the pre-activation from a second, bias-only
        // product -- the same fp32 accumulators and the same rounding as the fused form
        GemmArgs b = a;
        b.C = a.C2; b.ldc = a.ldc2; b.C2 = nullptr;
        hipLaunchKernelGGL(gemm_bf16_pp<OV_EPI_BIAS>, dim3(nwg), dim3(NTHREADS), 0, st, b);
        OV_LAUNCH_CHECK();
        a.C2 = nullptr;
        var = 2;
    }
    if (var == 3) {
        hipLaunchKernelGGL(gemm_bf16_w4<EPI>, dim3(nwg), dim3(256), 0, st, a);
    } else if (var == 1) {
        hipLaunchKernelGGL(gemm_bf16_256x256<EPI>, dim3(nwg), dim3(NTHREADS), 0, st, a);
    } else if (var == 2) {
        hipLaunchKernelGGL(gemm_bf16_pp<EPI>, dim3(nwg), dim3(NTHREADS), 0, st, a);
    } else {
        const int ncu = num_cus();
        const dim3 grid(nwg < ncu ? nwg : ncu), blk(NTHREADS);
        persist_policy<EPI>(a, (int)grid.x);
        // Epilogue form.  Bias-only epilogue (QKV, projections): LDS-transposed coalesced stores measure faster in the model (9.5-9.8
        // against 10.0-10.3 ms per step for the QKV GEMMs); OVHIP_GEMM_EPI_DIRECT=1 selects the direct form there too.  GELU (LN-folded
        // c_fc): by store policy, below.  Residual: LDS-transposed (below); with row maps, kept pre-activations or row statistics: direct.
        constexpr bool CAN_FOLD = EPI < OV_EPI_BIAS_RESIDUAL;
        const bool mapped = a.out_group != 0 || a.resid_mod != 0;          // row maps: the patch embedding's GEMM only
        if (EPI == OV_EPI_BIAS) {
            static int direct0 = -1;
            if (direct0 < 0) { const char* e = getenv("OVHIP_GEMM_EPI_DIRECT"); direct0 = (e && e[0] == '1') ? 1 : 0; }
            if (a.colsum != nullptr) {          // LN fold: never with row maps
                if (direct0) hipLaunchKernelGGL((gemm_bf16_persist<OV_EPI_BIAS, true, true, false>), grid, blk, 0, st, a);
                else hipLaunchKernelGGL((gemm_bf16_persist<OV_EPI_BIAS, true, false, false>), grid, blk, 0, st, a);
            } else if (mapped) {
                hipLaunchKernelGGL((gemm_bf16_persist<OV_EPI_BIAS, false, false, true>), grid, blk, 0, st, a);
            } else {
                if (direct0) hipLaunchKernelGGL((gemm_bf16_persist<OV_EPI_BIAS, false, true, false>), grid, blk, 0, st, a);
                else hipLaunchKernelGGL((gemm_bf16_persist<OV_EPI_BIAS, false, false, false>), grid, blk, 0, st, a);
            }
        } else if (a.C2 != nullptr) {             // ov_gemm_keep (GELU epilogues, unfolded weights: checked by the caller)
            constexpr int E = (EPI == OV_EPI_BIAS_GELU_ERF || EPI == OV_EPI_BIAS_GELU_TANH) ? EPI : OV_EPI_BIAS_GELU_ERF;
            hipLaunchKernelGGL((gemm_bf16_persist<E, false, true, false, true>), grid, blk, 0, st, a);
        } else if (CAN_FOLD && a.colsum != nullptr) {
            // GELU with the LN fold (vision / text c_fc): OVHIP_GEMM_GELU_LDS=1 selects the LDS-transposed form (whole-line stores)
            // Default: the LDS-transposed form when the output is streamed (whole 128-byte lines leave L2; the direct form's 16-byte
            // streaming stores leave as partial lines: WRITE_SIZE 0.69 against 0.54 GB per L/14 launch) -- L/14 45.1 -> 44.9 ms, S/8
            // 35.7 -> 35.4; the direct form for small, plainly stored outputs (Ti/16: 0.71 against 0.73 ms).  OVHIP_GEMM_GELU_LDS=0|1 forces.
            static int gelu_env = -2;
            if (gelu_env == -2) { const char* e = getenv("OVHIP_GEMM_GELU_LDS"); gelu_env = e ? (e[0] == '1' ? 1 : 0) : -1; }
            const int gelu_lds = gelu_env >= 0 ? gelu_env : (a.st_plain ? 0 : 1);
            constexpr int EF = CAN_FOLD ? EPI : OV_EPI_BIAS_GELU_ERF;      // (the residual epilogues never come here: not instantiated)
            if (gelu_lds) hipLaunchKernelGGL((gemm_bf16_persist<EF, true, false, false>), grid, blk, 0, st, a);
            else hipLaunchKernelGGL((gemm_bf16_persist<EF, true, true, false>), grid, blk, 0, st, a);
        } else if (EPI == OV_EPI_BIAS_RESIDUAL && mapped) {
            hipLaunchKernelGGL((gemm_bf16_persist<OV_EPI_BIAS_RESIDUAL, false, true, true>), grid, blk, 0, st, a);
        } else if (EPI == OV_EPI_BIAS_RESIDUAL && rowpart != nullptr) {
            hipLaunchKernelGGL((gemm_bf16_persist<OV_EPI_BIAS_RESIDUAL, false, true, false, false, true>), grid, blk, 0, st, a);
            stats_fused = true;
        } else {
            // Residual epilogue: the LDS-transposed form (whole-line stores, 16 instead of ~70 TA cycles per store instruction) since the
            // register diet of round 3 made it fit (246 VGPRs, zero scratch): out-proj 4.26 -> 4.10 ms per L/14 step, step -0.5 %
            // (Ti/16 -1.8 %); OVHIP_GEMM_RESID_LDS=0 selects the direct form again.
            static const int resid_lds = env_flag("OVHIP_GEMM_RESID_LDS", 1);
            if constexpr (EPI >= OV_EPI_BIAS_RESIDUAL) {       // (residual and GELU-gradient epilogues: both read a second operand row-wise)
                if (resid_lds) hipLaunchKernelGGL((gemm_bf16_persist<EPI, false, false, false>), grid, blk, 0, st, a);
                else hipLaunchKernelGGL((gemm_bf16_persist<EPI, false, true, false>), grid, blk, 0, st, a);
            } else {
                hipLaunchKernelGGL((gemm_bf16_persist<EPI, false, true, false>), grid, blk, 0, st, a);
            }
        }
    }
    OV_LAUNCH_CHECK();
    if (rowpart != nullptr && !stats_fused) return ov_rowparts(a.C, a.ldc, rowpart, a.M, a.N, (ov_stream_t)st);
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
    if (epilogue >= OV_EPI_BIAS_RESIDUAL) {
        if (!R || ldr % 8 || ldr < N || ((uintptr_t)R & 15)) return OV_ERR_INVALID;
    }
    if (epilogue > OV_EPI_BIAS_RESIDUAL && (out_group || resid_mod)) return OV_ERR_UNSUPPORTED;      // row maps: bias / residual only
    if (out_group < 0 || resid_mod < 0 || resid_off < 0) return OV_ERR_INVALID;
    if (g_rowpart != nullptr && (epilogue != OV_EPI_BIAS_RESIDUAL || out_group || resid_mod || N % 32)) return OV_ERR_UNSUPPORTED;
    const int64_t tiles_m = (M + BM - 1) / BM;
    const int64_t tiles_n = (N + BN - 1) / BN;
    if (tiles_m * tiles_n > 0x7fffffffLL || M > 0x7fff0000LL) return OV_ERR_UNSUPPORTED;    // 32-bit row indices in the kernels
    GemmArgs a{A, W, bias, C, R, lda, ldw, ldc, ldr, M, N, K, (int)tiles_m, (int)tiles_n,
               out_group, resid_mod, resid_off, g_colsum, g_rowstats, g_stamps, g_stamp_slots, g_wstamps, gemm_ngroup((int)tiles_m, (int)tiles_n, K), 0, 0, 0,
               gemm_stagger(K), gemm_stagger_classes(), gemm_epi_prio(), g_keep, g_ldkeep, nullptr, 0, 1, 0, 0, g_rowpart};
    hipStream_t st = (hipStream_t)stream;
    switch (epilogue) {
        case OV_EPI_BIAS: return launch<OV_EPI_BIAS>(a, st);
        case OV_EPI_BIAS_GELU_ERF: return launch<OV_EPI_BIAS_GELU_ERF>(a, st);
        case OV_EPI_BIAS_GELU_TANH: return launch<OV_EPI_BIAS_GELU_TANH>(a, st);
        case OV_EPI_BIAS_RESIDUAL: return launch<OV_EPI_BIAS_RESIDUAL>(a, st);
        case OV_EPI_GELU_GRAD_ERF: return launch<OV_EPI_GELU_GRAD_ERF>(a, st);
        case OV_EPI_GELU_GRAD_TANH: return launch<OV_EPI_GELU_GRAD_TANH>(a, st);
        default: return OV_ERR_INVALID;
    }
}

// ov_gemm with a GELU epilogue that also keeps the pre-activation: C = gelu(A W^T + bias), C2 = bf16(A W^T + bias) -- the training
// forward's c_fc, so that the backward does not run this product a second time.
extern "C" int ov_gemm_keep(const ov_bf16* A, int64_t lda, const ov_bf16* W, int64_t ldw, const float* bias, ov_bf16* C, int64_t ldc,
                            ov_bf16* C2, int64_t ldc2, int64_t M, int N, int K, int epilogue, ov_stream_t stream) {
    if (epilogue != OV_EPI_BIAS_GELU_ERF && epilogue != OV_EPI_BIAS_GELU_TANH) return OV_ERR_INVALID;
    if (!C2 || ldc2 % 8 || ldc2 < N || ((uintptr_t)C2 & 15)) return OV_ERR_INVALID;
    g_keep = C2; g_ldkeep = ldc2;
    const int rc = ov_gemm(A, lda, W, ldw, bias, C, ldc, M, N, K, epilogue, nullptr, 0, 0, 0, 0, stream);
    g_keep = nullptr; g_ldkeep = 0;
    return rc;
}

// ov_gemm with the residual epilogue that also leaves the row statistics' partial sums of its OUTPUT: rowparts[m][N / 32] = {sum, sum
// of squares} of the bf16 values C[m][32 g .. 32 g + 31] (fp32, fixed association: common.h) -- what ov_rowstats_finalize turns into
// the next LayerNorm's {mean, rstd} without re-reading the residual stream.  N % 32 == 0, no row maps.
extern "C" int ov_gemm_rowparts(const ov_bf16* A, int64_t lda, const ov_bf16* W, int64_t ldw, const float* bias, ov_bf16* C, int64_t ldc,
                                int64_t M, int N, int K, const ov_bf16* R, int64_t ldr, float* rowparts, ov_stream_t stream) {
    if (!rowparts || ((uintptr_t)rowparts & 7)) return OV_ERR_INVALID;
    g_rowpart = rowparts;
    const int rc = ov_gemm(A, lda, W, ldw, bias, C, ldc, M, N, K, OV_EPI_BIAS_RESIDUAL, R, ldr, 0, 0, 0, stream);
    g_rowpart = nullptr;
    return rc;
}

// `batch` independent products C_z = A_z . W_z^T (bf16 out, no bias) in ONE launch of the non-persistent kernel, blockIdx.y = z.
// With A_z / W_z = column ranges of one pair of operands this is split-K with bf16 partials (ov_linear_backward's dW).
extern "C" int ov_gemm_batched(const ov_bf16* A, int64_t lda, int64_t stride_a, const ov_bf16* W, int64_t ldw, int64_t stride_w,
                               ov_bf16* C, int64_t ldc, int64_t stride_c, int64_t M, int N, int K, int batch, ov_stream_t stream) {
    if (!A || !W || !C || M <= 0 || N <= 0 || K <= 0 || batch <= 0 || batch > 65535) return OV_ERR_INVALID;
    if (K % BK || N % 8 || lda % 8 || ldw % 8 || ldc % 8 || stride_a % 8 || stride_w % 8 || stride_c % 8) return OV_ERR_UNSUPPORTED;
    if (lda < K || ldw < K || ldc < N) return OV_ERR_INVALID;
    if (((uintptr_t)A | (uintptr_t)W | (uintptr_t)C) & 15) return OV_ERR_INVALID;
    const int64_t tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
    if (tiles_m * tiles_n > 0x7fffffffLL || M > 0x7fff0000LL) return OV_ERR_UNSUPPORTED;
    GemmArgs a{A, W, nullptr, C, nullptr, lda, ldw, ldc, 0, M, N, K, (int)tiles_m, (int)tiles_n, 0, 0, 0, nullptr, nullptr, nullptr, 0, nullptr,
               (int)tiles_n, stride_a, stride_w, stride_c, 0, 1, 0};
    hipLaunchKernelGGL(gemm_bf16_pp<OV_EPI_BIAS>, dim3((unsigned)(tiles_m * tiles_n), (unsigned)batch), dim3(NTHREADS), 0,
                       (hipStream_t)stream, a);
    OV_LAUNCH_CHECK();
    return OV_OK;
}

// `batch` split-K partials of C = P^T Q (both operands row-major over the contraction rows): partial z contracts rows
// [z * chunk, min(Mc, (z + 1) * chunk)) into C + z * stride_c (bf16 [NI, NJ], no bias).  Mc % 64 == 0, chunk % 64 == 0, NI % 8 == 0,
// NJ % 8 == 0, ldp / ldq / ldc % 8 == 0.  The weight-gradient product of ov_linear_backward without explicit transposes.
// psum (or NULL): fp32 [batch][NI], psum[z][i] = sum of P[m, i] over the rows of partial z (the bias gradient's partial sums).
extern "C" int ov_gemm_tn_batched(const ov_bf16* P, int64_t ldp, const ov_bf16* Q, int64_t ldq, ov_bf16* C, int64_t ldc, int64_t stride_c,
                                  int64_t Mc, int NI, int NJ, int64_t chunk, int batch, float* psum, ov_stream_t stream) {
    if (!P || !Q || !C || Mc <= 0 || NI <= 0 || NJ <= 0 || chunk <= 0 || batch <= 0 || batch > 65535) return OV_ERR_INVALID;
    if (Mc % BK || chunk % BK || NI % 8 || NJ % 8 || ldp % 8 || ldq % 8 || ldc % 8 || stride_c % 8) return OV_ERR_UNSUPPORTED;
    if (ldp < NI || ldq < NJ || ldc < NJ || (int64_t)(batch - 1) * chunk >= Mc || (int64_t)batch * chunk < Mc) return OV_ERR_INVALID;
    if (((uintptr_t)P | (uintptr_t)Q | (uintptr_t)C) & 15) return OV_ERR_INVALID;
    const int64_t tiles_m = (NI + BM - 1) / BM, tiles_n = (NJ + BN - 1) / BN;
    if (tiles_m * tiles_n > 0x7fffffffLL) return OV_ERR_UNSUPPORTED;
    GemmArgs a{P, Q, nullptr, C, nullptr, ldp, ldq, ldc, 0, NI, NJ, 0, (int)tiles_m, (int)tiles_n, 0, 0, 0, nullptr, nullptr, nullptr, 0, nullptr,
               (int)tiles_n, chunk, 0, stride_c, 0, 1, 0};
    a.K = 0;                    // (int field: the contraction length does not fit the struct's K for huge M; carried below)
    if (Mc > 0x7fffffffLL) return OV_ERR_UNSUPPORTED;
    a.K = (int)Mc;
    a.psum = psum;
    hipLaunchKernelGGL(gemm_bf16_pp_tn, dim3((unsigned)(tiles_m * tiles_n), (unsigned)batch), dim3(NTHREADS), 0, (hipStream_t)stream, a);
    OV_LAUNCH_CHECK();
    return OV_OK;
}

// Diagnostics: when set, the persistent kernel's thread 0 of every workgroup records s_memtime at tile start / main-loop
// end / after the re-align barrier / epilogue end / end of K-tiles 0, 1, 3, 7 into buf[block][slot][8] (slot = tile iteration < slots).  NULL = off.
extern "C" int ov_debug_gemm_stamps(unsigned long long* buf, int slots) {
    g_stamps = buf;
    g_stamp_slots = buf ? slots : 0;
    return OV_OK;
}
// Per-wave epilogue timeline: wbuf[block][slot][wave][8] = main loop end / epilogue_stream entry / parameters read / passes 1 and 4 done /
// last store issued / past the tile-boundary barrier.  Needs ov_debug_gemm_stamps(buf, slots) as well (it carries `slots`).
extern "C" int ov_debug_gemm_wave_stamps(unsigned long long* wbuf) {
    g_wstamps = wbuf;
    return OV_OK;
}

extern "C" int ov_gemm_ln(const ov_bf16* X, int64_t ldx, const ov_bf16* Wg, int64_t ldw, const float* cvec, const float* colsum,
                          const float* rowstats, ov_bf16* C, int64_t ldc, int64_t M, int N, int K, int epilogue,
                          ov_stream_t stream) {
    if (!colsum || !rowstats || !cvec) return OV_ERR_INVALID;
    if (epilogue >= OV_EPI_BIAS_RESIDUAL) return OV_ERR_UNSUPPORTED;
    if ((((uintptr_t)colsum | (uintptr_t)cvec) & 15) || ((uintptr_t)rowstats & 7)) return OV_ERR_INVALID;
    g_colsum = colsum;
    g_rowstats = rowstats;
    const int rc = ov_gemm(X, ldx, Wg, ldw, cvec, C, ldc, M, N, K, epilogue, nullptr, 0, 0, 0, 0, stream);
    g_colsum = nullptr;
    g_rowstats = nullptr;
    return rc;
}
