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
};

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

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

    // ---- epilogue --------------------------------------------------------------------------------
    // acc[i][j][r] = C[m0 + wm*128 + i*16 + fr][n0 + wn*64 + j*16 + fq*4 + r]
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
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int ml = i * 16 + fr;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float x = acc[i][j][r] + bv[j][r];
                if (EPI == OV_EPI_BIAS_GELU_ERF) x = gelu_erf_f(x);
                if (EPI == OV_EPI_BIAS_GELU_TANH) x = gelu_tanh_f(x);
                v[r] = x;
            }
            u32x2_t p = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
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
int launch(const GemmArgs& a, hipStream_t st) {
    const int nwg = a.tiles_m * a.tiles_n;
    hipLaunchKernelGGL(gemm_bf16_256x256<EPI>, dim3(nwg), dim3(NTHREADS), 0, st, a);
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
               out_group, resid_mod, resid_off};
    hipStream_t st = (hipStream_t)stream;
    switch (epilogue) {
        case OV_EPI_BIAS: return launch<OV_EPI_BIAS>(a, st);
        case OV_EPI_BIAS_GELU_ERF: return launch<OV_EPI_BIAS_GELU_ERF>(a, st);
        case OV_EPI_BIAS_GELU_TANH: return launch<OV_EPI_BIAS_GELU_TANH>(a, st);
        case OV_EPI_BIAS_RESIDUAL: return launch<OV_EPI_BIAS_RESIDUAL>(a, st);
        default: return OV_ERR_INVALID;
    }
}
