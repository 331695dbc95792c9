// gemm_fp8.hip — fp8 (OCP e4m3fn) GEMM on the MX-scaled matrix instruction of gfx950, for BASELINE.json config #5
// ("ViT-L/14@224 fp8 weights/activations on CDNA4 fp8 MFMA").
//
//   C[M,N] = epilogue( rowscale[m] * colscale[n] * (A[M,K] . W[N,K]^T) + bias[n] )     A, W fp8 row-major (K contiguous),
//                                                                                      fp32 accumulate, C (and R) bf16
//
// Same structure as the persistent bf16 kernel (gemm.hip): 256x256 output tile, 8 waves (2 M x 4 N), each a 128 x 64
// sub-tile of 8 x 4 accumulators, operands L2 -> LDS by global_load_lds_dwordx4, ping-pong wave groups, persistent XCD tile
// walk, next tile's first two K-tiles and parameter block staged around the epilogue, streamed 8-pass epilogue.  What changes:
//   * v_mfma_scale_f32_16x16x128_f8f6f4 with unit e8m0 scales (127): twice the bf16 rate per clock.  A lane feeds 32 bytes of
//     its operand row; with unit block scales ANY 32 bytes of the 128-byte K-tile will do as long as A and W use the same map
//     (measured with exact integer data, tools/fp8/probe.py), so a lane takes the two 16-byte chunks 2 fq, 2 fq + 1.
//   * a K-tile is 128 bytes of K for both operands = 128 fp8 elements: 64 KiB of LDS, four 16-KiB DMA pieces and 24
//     ds_read_b128 per wave as in the bf16 kernel, 32 MFMAs x 32 cycles = the same 1024 matrix-pipe cycles -- at twice the K.
//   * every MFMA needs the whole 128-byte K extent, so the pieces are cut by ROW set (A m-half 0 | W n-half 0 | W n-half 1 |
//     A m-half 1, in issue order) and the four phases of a K-tile are (m-half, n-half) = (0,0) (0,1) (1,1) (1,0): phase 0 needs
//     pieces 0 and 1, phase 1 piece 2, phase 2 piece 3 -- every piece keeps >= 3 phases between its DMA issue and first use.
//   * LDS rows are 128 bytes (8 chunks): chunk ^= row & 7, on the DMA source side and on the ds_read side (conflict-free).
//   * per-row (activation) and per-column (weight) dequantisation scales ride in the parameter block and are applied in fp32
//     in the epilogue: v = (acc * rowscale[m]) * colscale[n] + bias[n].
#include "common.h"
#include <stdlib.h>

#ifndef OVHIP_ST_FP8OUT
#define OVHIP_ST_FP8OUT OVHIP_ST_LDS     /* store policy of the e4m3 output (64-byte row pieces: half lines) */
#endif

namespace {

constexpr int BM = 256, BN = 256, BKB = 128;       // K-tile: 128 bytes of K per row
constexpr int NTHREADS = 512;
constexpr int PIECE = 128 * 128;                   // 128 rows x 128 B = 16 KiB
constexpr int STAGE = 4 * PIECE;                   // 64 KiB
constexpr int IMG_OFF = 2 * STAGE;                 // 8 wave-local 2-KiB transposition images
constexpr int PRM_OFF = IMG_OFF + 8 * 2048;        // two 4-KiB parameter blocks: bias[256] | colscale[256] | rowscale[256] (f32)
constexpr int SMEM_TOTAL = PRM_OFF + 2 * 4096;

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
typedef int i32x8_t __attribute__((ext_vector_type(8)));

struct Fp8Args {
    const unsigned char* A; const unsigned char* W;
    const float* bias; const float* rowscale; const float* colscale;
    ov_bf16* C; const ov_bf16* R;
    int64_t lda, ldw, ldc, ldr, M;
    int N, K, tiles_m, tiles_n;
    const float* amax_in;      // MODE 2: every row of A was quantised with the scale HEADROOM * (*amax_in) / 448 (rowscale unused)
    const float* amax_out;     // MODE 1: C is written as e4m3 bytes with that static scale of its own
    float* amax_next;          // MODE 1, optional: running maximum of |C| before quantisation (the NEXT call's scale: delayed scaling)
};

template <int V> struct IntC { static constexpr int value = V; };

// Maximum over the wave without index registers (DPP patterns + the two permlane swaps): __shfl_xor's six bpermute indices are
// loop invariants of the whole kernel -- hipcc hoisted them to the kernel entry and spilled all six here.
__device__ __forceinline__ float wave_max_noidx(float v) {
    auto dpp = [](float x, auto ctrl) {
        return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), decltype(ctrl)::value, 0xf, 0xf, false));
    };
    v = fmaxf(v, dpp(v, IntC<0xB1>{}));        // quad_perm [1, 0, 3, 2]
    v = fmaxf(v, dpp(v, IntC<0x4E>{}));        // quad_perm [2, 3, 0, 1]
    v = fmaxf(v, dpp(v, IntC<0x141>{}));       // row_half_mirror: all 8
    v = fmaxf(v, dpp(v, IntC<0x140>{}));       // row_mirror: all 16
    const u32x2_t a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));        // rows {0, 1} and {2, 3}
    const u32x2_t b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}

constexpr float HEADROOM = 2.0f;   // static activation scales leave one binade above the calibrated maximum (e4m3 is floating point:
                                   // head room costs range at the bottom, not precision)


// MODE 0: per-row scales from the parameter block, bf16 output.  MODE 1: same, output quantised to e4m3 with the static scale
// HEADROOM * amax_out / 448 (feeds the next fp8 GEMM without a bf16 round trip).  MODE 2: one static input scale, bf16 output.
template <int EPI, int MODE>
__device__ __forceinline__ void epilogue_fp8(const Fp8Args& g, f32x4_t (&acc)[8][4], char* img, const char* prm, int64_t m0, int n0,
                                             int wave, int lane_in, bool edge, float rs_const, float inv_out, float next_thr) {
    (void)lane_in;
    const int lane = fresh_lane();       // opaque: the epilogue's per-lane addresses are recomputed per tile, not carried (and spilled)
                                         // through the main loop as loop invariants
    const int wm = wave >> 2, wn = wave & 3;
    const int fr = lane & 15, fq = lane >> 4;
    const int er = lane >> 3, ec = lane & 7;
    const int n = n0 + wn * 64 + ec * 8;
    const bool ncol = n < g.N;
    u32x4_t rv[8][2];
    const int nc = ncol ? n : g.N - 8;
    auto load_resid = [&](int i) {
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            unsigned m = (unsigned)m0 + wm * 128 + i * 16 + it * 8 + er;
            m = m < (unsigned)g.M ? m : (unsigned)g.M - 1;
            const ov_bf16* src = g.R + (int64_t)m * g.ldr + nc;
            rv[i][it] = *(const u32x4_t*)src;       // compiler-managed wait: this kernel still spills, and an inline-asm load's
                                                    // destination must never be spilled before its data has landed
        }
    };
    if (EPI == OV_EPI_BIAS_RESIDUAL) { load_resid(0); load_resid(1); }
    f32x4_t bq[4], cq[4];
    float rsq[8];
    const unsigned pa = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const char*)(prm + (wn * 64 + fq * 4) * 4);
    const unsigned ra = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const char*)(prm + 2048 + (wm * 128 + fr) * 4);
    // the reads and their wait form ONE asm statement: the destinations are valid when it ends, whatever the allocator does
    asm volatile(
        "ds_read_b128 %0, %16\n\tds_read_b128 %1, %16 offset:64\n\tds_read_b128 %2, %16 offset:128\n\tds_read_b128 %3, %16 offset:192\n\t"
        "ds_read_b128 %4, %16 offset:1024\n\tds_read_b128 %5, %16 offset:1088\n\tds_read_b128 %6, %16 offset:1152\n\t"
        "ds_read_b128 %7, %16 offset:1216\n\t"
        "ds_read_b32 %8, %17\n\tds_read_b32 %9, %17 offset:64\n\tds_read_b32 %10, %17 offset:128\n\tds_read_b32 %11, %17 offset:192\n\t"
        "ds_read_b32 %12, %17 offset:256\n\tds_read_b32 %13, %17 offset:320\n\tds_read_b32 %14, %17 offset:384\n\t"
        "ds_read_b32 %15, %17 offset:448\n\ts_waitcnt lgkmcnt(0)"
        : "=&v"(bq[0]), "=&v"(bq[1]), "=&v"(bq[2]), "=&v"(bq[3]), "=&v"(cq[0]), "=&v"(cq[1]), "=&v"(cq[2]), "=&v"(cq[3]),
          "=&v"(rsq[0]), "=&v"(rsq[1]), "=&v"(rsq[2]), "=&v"(rsq[3]), "=&v"(rsq[4]), "=&v"(rsq[5]), "=&v"(rsq[6]), "=&v"(rsq[7])
        : "v"(pa), "v"(ra)
        : "memory");
    char* const wr = img + fr * 128 + (fq & 1) * 8;
    const int wsw = fr & 7;
    const char* const rd = img + er * 128 + ((ec ^ er) << 4);
    u32x4_t vo[8][2];
    float lmax = 0.f;
    auto put = [&](int i) {
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            u32x4_t o = vo[i][it];
            if (EPI == OV_EPI_BIAS_RESIDUAL) {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    o[e] = pack_bf16x2(bf16lo_to_f32(o[e]) + bf16lo_to_f32(rv[i][it][e]),
                                       bf16hi_to_f32(o[e]) + bf16hi_to_f32(rv[i][it][e]));
            }
            const unsigned m = (unsigned)m0 + wm * 128 + i * 16 + it * 8 + er;
            // (the residual form writes the residual stream, re-read at once: plain stores -- common.h, OVHIP_ST_RESID)
            if (m < (unsigned)g.M && ncol) store16<(EPI == OV_EPI_BIAS_RESIDUAL) ? OVHIP_ST_RESID : OVHIP_ST_LDS>(g.C + (int64_t)m * g.ldc + n, o);
        }
    };
#pragma unroll
    for (int i = 0; i <= 8; ++i) {
        // residual rows two passes ahead of their use (pass i stores row i - 1): 16 registers in flight, compiler-managed waits
        if (EPI == OV_EPI_BIAS_RESIDUAL && i >= 1 && i + 1 < 8 && (i & 1)) { load_resid(i + 1); load_resid(i + 2 < 8 ? i + 2 : 7); }
        if (i < 8) {
            const f32x2_t rs = {MODE == 2 ? rs_const : rsq[i], MODE == 2 ? rs_const : rsq[i]};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                f32x2_t v01 = f32x2_t{acc[i][j][0], acc[i][j][1]} * rs;
                f32x2_t v23 = f32x2_t{acc[i][j][2], acc[i][j][3]} * rs;
                const f32x2_t b01 = {bq[j][0], bq[j][1]};          // (no bias: the kernel zeroed the slots' bias rows at its start)
                const f32x2_t b23 = {bq[j][2], bq[j][3]};
                v01 = __builtin_elementwise_fma(v01, f32x2_t{cq[j][0], cq[j][1]}, b01);
                v23 = __builtin_elementwise_fma(v23, f32x2_t{cq[j][2], cq[j][3]}, b23);
                if (EPI == OV_EPI_BIAS_GELU_ERF) gelu_erf_f2x2(v01, v23);
                if (EPI == OV_EPI_BIAS_GELU_TANH) { v01 = gelu_tanh_f2(v01); v23 = gelu_tanh_f2(v23); }
                if (MODE == 1) {
                    // 4 consecutive n -> 4 e4m3 bytes; image rows are 64 B (16 dwords), chunk ^= (row >> 1) & 3
                    lmax = fmaxf(fmaxf(lmax, fabsf(v01[0])), fabsf(v01[1]));     // two v_max3_f32 with |x| source modifiers
                    lmax = fmaxf(fmaxf(lmax, fabsf(v23[0])), fabsf(v23[1]));
                    const f32x2_t io = {inv_out, inv_out};
                    v01 *= io; v23 *= io;
                    int pk8 = 0;
                    pk8 = __builtin_amdgcn_cvt_pk_fp8_f32(__builtin_amdgcn_fmed3f(v01[0], -448.f, 448.f), __builtin_amdgcn_fmed3f(v01[1], -448.f, 448.f), pk8, false);
                    pk8 = __builtin_amdgcn_cvt_pk_fp8_f32(__builtin_amdgcn_fmed3f(v23[0], -448.f, 448.f), __builtin_amdgcn_fmed3f(v23[1], -448.f, 448.f), pk8, true);
                    *(int*)(img + fr * 64 + ((j ^ ((fr >> 1) & 3)) << 4) + fq * 4) = pk8;
                } else {
                    const u32x2_t pk = {pack_bf16x2(v01[0], v01[1]), pack_bf16x2(v23[0], v23[1])};
                    *(u32x2_t*)(wr + (((j * 2 + (fq >> 1)) ^ wsw) << 4)) = pk;
                }
            }
        }
        if (MODE == 1) {
            // 16 rows x 64 bytes: lane -> row lane >> 2, 16-byte chunk lane & 3 (one read, one 16-byte store per pass)
            if (i > 0) {
                const unsigned m = (unsigned)m0 + wm * 128 + (i - 1) * 16 + (lane >> 2);
                const int n8 = n0 + wn * 64 + (lane & 3) * 16;
                if (m < (unsigned)g.M && n8 < g.N) store16<OVHIP_ST_FP8OUT>((unsigned char*)g.C + (int64_t)m * g.ldc + n8, vo[i - 1][0]);
            }
            if (i < 8) vo[i][0] = *(const u32x4_t*)(img + (lane >> 2) * 64 + (((lane & 3) ^ ((lane >> 3) & 3)) << 4));
        } else {
            if (i > 0) put(i - 1);
            if (i < 8) {
                vo[i][0] = *(const u32x4_t*)(rd);
                vo[i][1] = *(const u32x4_t*)(rd + 1024);
            }
        }
    }
    if (MODE == 1 && g.amax_next != nullptr) {       // delayed scaling: this tile's maximum feeds the next call's scale
        lmax = wave_max_noidx(lmax);
        // next_thr = the value at kernel start: a load here would make the compiler wait for this epilogue's stores to be acknowledged
        if (lane == 0 && lmax > next_thr) atomicMax((unsigned*)g.amax_next, __float_as_uint(lmax));
    }
}

template <int EPI, int MODE>
__global__ __launch_bounds__(NTHREADS, 2) void gemm_fp8_persist(const Fp8Args g) {
    __shared__ __attribute__((aligned(16))) char smem[SMEM_TOTAL];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // ---- static persistent schedule: the linear n-fastest tile list cut into 8 contiguous XCD runs ----
    const int nwg = g.tiles_m * g.tiles_n;
    const int G = gridDim.x, bid = blockIdx.x;
    const int xcd = bid & 7, li = bid >> 3;
    const int q8 = nwg >> 3, r8 = nwg & 7;
    const int xstart = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
    const int xcnt = q8 + (xcd < r8 ? 1 : 0);
    const int nper = (G - xcd + 7) >> 3;
    int tcur = li;
    if (tcur >= xcnt) return;

    // ---- DMA geometry.  A piece = 128 rows x 128 B = 16 wave-instructions of 8 rows; wave w issues instructions w and w + 8.
    // Lane i of an instruction lands at row 8 g + (i >> 3), physical chunk i & 7, which holds logical chunk (i & 7) ^ (row & 7).
    // Per-lane byte offsets of this lane's 2 DMA rows of every piece from the tile's A / W base: the same for every tile.  Rows
    // past M / N are clamped at issue time: off = min(off, last_row * ld + chunk), exact because chunk < ld.
    // DMA offsets are NOT kept in registers: every stage_piece rebuilds its two from an opaque lane id (7 VALU instructions) -- as
    // loop invariants hipcc carried them (zero-extended to 64 bits, through every K-tile variant and the epilogue) and spilled ~30
    // registers, and a spill reload inside the K loop is a vector-memory operation the hand-counted vmcnt waits do not know about
    struct TileBase { const unsigned char* a; const unsigned char* w; unsigned lima, limw; };   // wave-uniform
    TileBase cur, nxt;
    int64_t m0, nm0 = 0;
    int n0, nn0 = 0;
    auto set_tile = [&](int trel, TileBase& tb, int64_t& mm, int& nn) {
        const int wg = xstart + trel;
        const int tm = wg / g.tiles_n, tn = wg - tm * g.tiles_n;
        mm = (int64_t)tm * BM;
        nn = tn * BN;
        tb.a = g.A + mm * g.lda;
        tb.w = g.W + (int64_t)nn * g.ldw;
        const int64_t la = g.M - 1 - mm, lw = g.N - 1 - nn;
        tb.lima = (unsigned)((la < BM ? la : BM) * g.lda);
        tb.limw = (unsigned)((lw < BN ? lw : BN) * g.ldw);
    };
    auto advance = [&](TileBase& tb) { tb.a += BKB; tb.w += BKB; };
    char* const sbase = smem + wave * 1024;
    // phase p stages piece: 0 = A m-half 0, 1 = W n-half 0, 2 = W n-half 1, 3 = A m-half 1 (LDS order: A0 | W0 | W1 | A1)
    auto stage_piece = [&](const TileBase& tb, int boff, int p) {
        char* dst = sbase + boff + p * PIECE;
        const bool isa = (p == 0 || p == 3);
        const unsigned char* b = isa ? tb.a : tb.w;
        const int ln = fresh_lane();
        const int dr = ln >> 3;                                  // DMA row within the 8-row instruction
        const unsigned dch16 = (unsigned)(((ln & 7) ^ dr) << 4); // logical 16-byte chunk this lane fetches (the image is XOR-swizzled)
        const unsigned lim = (isa ? tb.lima : tb.limw) + dch16;
        // instruction w of the piece = its rows 8 w .. 8 w + 7; A piece (m-half h): tile rows h * 64 + r (r < 64), + 128 for w + 8;
        // W piece (n-half h): tile rows (r >> 5) * 64 + h * 32 + (r & 31) with r = 8 w + dr, + 128 for w + 8
        const unsigned ld = (unsigned)(isa ? g.lda : g.ldw);
        const unsigned urow = isa ? (unsigned)(8 * wave + (p == 3 ? 64 : 0)) : (unsigned)((wave >> 2) * 64 + 8 * (wave & 3) + (p == 2 ? 32 : 0));
        unsigned o0 = ((unsigned)dr + urow) * ld + dch16;
        unsigned o1 = o0 + 128u * ld;
        o0 = o0 < lim ? o0 : lim;
        o1 = o1 < lim ? o1 : lim;
        __builtin_amdgcn_global_load_lds((gptr_t)(b + o0), (lptr_t)dst, 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gptr_t)(b + o1), (lptr_t)(dst + 8192), 16, 0, 0);
    };
    auto stage_params = [&](int slot, int64_t mm, int nn) {
        char* dst = smem + PRM_OFF + slot * 4096;
        const int lane = fresh_lane();
        int c = nn + lane * 4;
        c = c + 4 <= g.N ? c : g.N - 4;
        if (wave == 0 && g.bias != nullptr) __builtin_amdgcn_global_load_lds((gptr_t)(g.bias + c), (lptr_t)dst, 16, 0, 0);
        if (wave == 1) __builtin_amdgcn_global_load_lds((gptr_t)(g.colscale + c), (lptr_t)(dst + 1024), 16, 0, 0);
        if (MODE != 2 && wave >= 4) {                                // 256 row scales: 4 waves x 64 dwords (clamped rows stay in bounds)
            int64_t r = mm + (wave - 4) * 64 + lane;
            r = r < g.M ? r : g.M - 1;
            __builtin_amdgcn_global_load_lds((gptr_t)(g.rowscale + r), (lptr_t)(dst + 2048 + (wave - 4) * 256), 4, 0, 0);
        }
    };

    const float rs_const = MODE == 2 ? HEADROOM * (1.0f / 448.0f) * *g.amax_in : 0.f;
    const float inv_out = MODE == 1 ? 448.0f / (HEADROOM * fmaxf(*g.amax_out, 1e-30f)) : 0.f;
    const float next_thr = (MODE == 1 && g.amax_next != nullptr) ? *g.amax_next : 0.f;
    const int wm = wave >> 2, wn = wave & 3;
    const int fr = lane & 15, fq = lane >> 4;
    // fragment reads: operand row `row` of a piece, logical chunks 2 fq and 2 fq + 1 -> physical chunk ^ (row & 7) = ^ (fr & 7)
    // fragment addressing, rebuilt per tile from an opaque lane id (not carried through the epilogue in registers)
    int c0, c1, a_row, w_row;
    auto frag_consts = [&]() {
        const int ln = fresh_lane();
        const int r = ln & 15, q = ln >> 4;
        c0 = ((2 * q) ^ (r & 7)) << 4;
        c1 = ((2 * q + 1) ^ (r & 7)) << 4;
        a_row = (wm * 64 + r) * 128;                                 // + (i & 3) * 16 * 128 within piece A(mh)
        w_row = (wn * 32 + r) * 128;                                 // + (j & 1) * 16 * 128 within piece W(nh)
    };
    const int nt = g.K / BKB;                                        // >= 3 (launcher)

    if (g.bias == nullptr && tid < 256) {                             // no bias: both parameter slots read zeros (never DMA'd over)
        *(float*)(smem + PRM_OFF + tid * 4) = 0.f;
        *(float*)(smem + PRM_OFF + 4096 + tid * 4) = 0.f;
    }
    set_tile(tcur, cur, m0, n0);
#pragma unroll
    for (int p = 0; p < 4; ++p) stage_piece(cur, 0, p);
    advance(cur);
#pragma unroll
    for (int p = 0; p < 4; ++p) stage_piece(cur, STAGE, p);
    advance(cur);                                                    // invariant at tile start: `cur` points at K-tile 2
    stage_params(0, m0, n0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (wm == 1) __builtin_amdgcn_s_barrier();                       // stagger the lower wave group by one interval

    int cb = 0, pslot = 0, titer = 0;
    bool strict = true, has_next = false;
    f32x4_t acc[8][4];
    i32x8_t af[4], wf[2];                                            // A fragments of the current m-half, W fragments of the current n-half
    auto frag = [](const char* p, int ca, int cb2) {                 // 32 operand bytes = two 16-byte chunks of the row
        return __builtin_bit_cast(i32x8_t, __builtin_shufflevector(*(const u32x4_t*)(p + ca), *(const u32x4_t*)(p + cb2), 0, 1, 2, 3, 4, 5, 6, 7));
    };
    // KIND: 0 = K-tile 0 of a tile (nothing staged; K-tile 1 + parameters are older than the previous epilogue's 16 stores),
    // 1 = K-tile 1 (stages K-tile 2; K-tile 1 has landed as a whole), 2 = steady state, 3 = last (stages the next tile's K-tile 0)
    auto ktile = [&](auto kind) {
        constexpr int KIND = decltype(kind)::value;
        const char* s = smem + cb;
        const int nb = cb ^ STAGE;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int mh = p >> 1, nh = (p == 1 || p == 2) ? 1 : 0;
            if (p == 0 || p == 2) {                                  // new m-half: 4 A fragments (piece A0 at 0, A1 at 3 * PIECE)
                const char* ap = s + (mh ? 3 * PIECE : 0) + a_row;
#pragma unroll
                for (int i = 0; i < 4; ++i) af[i] = frag(ap + i * 2048, c0, c1);
            }
            if (p != 2) {                                            // n-half changes at p = 0 (nh 0), 1 (nh 1), 3 (nh 0 again): 2 W fragments
                const char* wp = s + (1 + nh) * PIECE + w_row;
#pragma unroll
                for (int j = 0; j < 2; ++j) wf[j] = frag(wp + j * 2048, c0, c1);
            }
            if (KIND == 1 || KIND == 2) stage_piece(cur, nb, p);
            if (KIND == 3) { if (has_next) stage_piece(nxt, nb, p); }
            // piece j of the next K-tile is issued in phase j and first read in phase 0 (j = 0, 1), 1 (j = 2), 2 (j = 3)
            if (KIND == 0) {
                if (p == 3) {
                    if (strict) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    else if (MODE == 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");      // byte output: 8 stores per wave and tile
                    else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
                }
            } else if (KIND == 1) {
                if (p == 3) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            } else if (KIND == 2) {
                if (p != 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            } else {
                if (has_next) { if (p != 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
                else if (p == 0) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
                else if (p == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[mh * 4 + i][nh * 2 + j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(
                        wf[j], af[i], acc[mh * 4 + i][nh * 2 + j], 0, 0, 0, 127, 0, 127);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    for (;;) {
        const int tnext = tcur + nper;
        has_next = tnext < xcnt;
        if (has_next) set_tile(tnext, nxt, nm0, nn0);
        ++titer;
        frag_consts();
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

        ktile(IntC<0>{});
        cb ^= STAGE;
        ktile(IntC<1>{});
        cb ^= STAGE;
        advance(cur);
        for (int t = 2; t < nt - 1; ++t) {
            ktile(IntC<2>{});
            cb ^= STAGE;
            advance(cur);
        }
        ktile(IntC<3>{});
        if (wm == 0) __builtin_amdgcn_s_barrier();                   // re-align: every wave is past its last COMPUTE segment
        if (has_next) {
            advance(nxt);
#pragma unroll
            for (int p = 0; p < 4; ++p) stage_piece(nxt, cb, p);
            advance(nxt);
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");         // the next tile's K-tile 0 (issued a K-tile ago) has landed
            stage_params(pslot ^ 1, nm0, nn0);
        }
        __builtin_amdgcn_sched_barrier(0);
        const bool edge = (m0 + BM > g.M) || (n0 + BN > g.N);
        epilogue_fp8<EPI, MODE>(g, acc, smem + IMG_OFF + wave * 2048, smem + PRM_OFF + pslot * 4096, m0, n0, wave, lane, edge, rs_const,
                                inv_out, next_thr);
        if (!has_next) break;
        strict = edge;
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        if (wm == 1) __builtin_amdgcn_s_barrier();
        cb ^= STAGE;
        pslot ^= 1;
        cur = nxt;
        m0 = nm0; n0 = nn0;
        tcur = tnext;
    }
}

int num_cus_fp8() { return ov_num_cus(); }

template <int EPI, int MODE>
int launch_fp8(const Fp8Args& a, hipStream_t st) {
    const int nwg = a.tiles_m * a.tiles_n, ncu = num_cus_fp8();
    hipLaunchKernelGGL((gemm_fp8_persist<EPI, MODE>), dim3(nwg < ncu ? nwg : ncu), dim3(NTHREADS), 0, st, a);
    OV_LAUNCH_CHECK();
    return OV_OK;
}

int check_fp8(const void* A, int64_t lda, const void* W, int64_t ldw, const void* C, int64_t ldc, int64_t M, int N, int K) {
    if (!A || !W || !C || M <= 0 || N <= 0 || K <= 0) return OV_ERR_INVALID;
    if (K % BKB || K < 3 * BKB || N % 8 || lda % 16 || ldw % 16 || ldc % 8) return OV_ERR_UNSUPPORTED;
    if (lda < K || ldw < K || ldc < N) return OV_ERR_INVALID;
    if (((uintptr_t)A | (uintptr_t)W | (uintptr_t)C) & 15) return OV_ERR_INVALID;
    const int64_t tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
    if (tiles_m * tiles_n > 0x7fffffffLL || M > 0x7fff0000LL) return OV_ERR_UNSUPPORTED;
    if ((int64_t)BM * lda > 0x7fffffffLL || (int64_t)BN * ldw > 0x7fffffffLL) return OV_ERR_UNSUPPORTED;   // 32-bit in-tile offsets
    return OV_OK;
}

}  // namespace

extern "C" int ov_gemm_fp8(const unsigned char* A, int64_t lda, const unsigned char* W, int64_t ldw, const float* rowscale,
                           const float* colscale, const float* bias, ov_bf16* C, int64_t ldc, int64_t M, int N, int K,
                           int epilogue, const ov_bf16* R, int64_t ldr, ov_stream_t stream) {
    int rc = check_fp8(A, lda, W, ldw, C, ldc, M, N, K);
    if (rc) return rc;
    if (!rowscale || !colscale || ((uintptr_t)colscale & 15)) return OV_ERR_INVALID;
    if (bias && ((uintptr_t)bias & 15)) return OV_ERR_INVALID;
    if (epilogue == OV_EPI_BIAS_RESIDUAL && (!R || ldr % 8 || ldr < N || ((uintptr_t)R & 15))) return OV_ERR_INVALID;
    const Fp8Args a{A, W, bias, rowscale, colscale, C, R, lda, ldw, ldc, ldr, M, N, K, (int)((M + BM - 1) / BM), (int)((N + BN - 1) / BN),
                    nullptr, nullptr, nullptr};
    hipStream_t st = (hipStream_t)stream;
    switch (epilogue) {
        case OV_EPI_BIAS: return launch_fp8<OV_EPI_BIAS, 0>(a, st);
        case OV_EPI_BIAS_GELU_ERF: return launch_fp8<OV_EPI_BIAS_GELU_ERF, 0>(a, st);
        case OV_EPI_BIAS_GELU_TANH: return launch_fp8<OV_EPI_BIAS_GELU_TANH, 0>(a, st);
        case OV_EPI_BIAS_RESIDUAL: return launch_fp8<OV_EPI_BIAS_RESIDUAL, 0>(a, st);
        default: return OV_ERR_UNSUPPORTED;
    }
}

// The same GEMM with a STATIC scale on one side (calibrated activation maximum, device scalar; scale = 2 * amax / 448):
//   out_amax != NULL: C is written as e4m3 bytes C8[M, N] (ldc in bytes, % 16, N % 16 == 0) quantised with out_amax's scale
//                     -- epilogue OV_EPI_BIAS_GELU_ERF / _TANH (the c_fc -> c_proj hand-over without a bf16 round trip);
//                     out_amax_next (optional): running maximum of |C| before quantisation, for the next call (delayed scaling);
//   in_amax  != NULL: every row of A carries in_amax's scale (rowscale ignored) -- epilogue OV_EPI_BIAS_RESIDUAL.
extern "C" int ov_gemm_fp8_static(const unsigned char* A, int64_t lda, const unsigned char* W, int64_t ldw, const float* rowscale,
                                  const float* in_amax, const float* colscale, const float* bias, void* C, int64_t ldc,
                                  const float* out_amax, float* out_amax_next, int64_t M, int N, int K, int epilogue,
                                  const ov_bf16* R, int64_t ldr, ov_stream_t stream) {
    int rc = check_fp8(A, lda, W, ldw, C, ldc, M, N, K);
    if (rc) return rc;
    if (!colscale || ((uintptr_t)colscale & 15) || (bias && ((uintptr_t)bias & 15))) return OV_ERR_INVALID;
    if ((in_amax != nullptr) == (out_amax != nullptr)) return OV_ERR_INVALID;          // exactly one static side
    const Fp8Args a{A, W, bias, rowscale, colscale, (ov_bf16*)C, R, lda, ldw, ldc, ldr, M, N, K, (int)((M + BM - 1) / BM),
                    (int)((N + BN - 1) / BN), in_amax, out_amax, out_amax_next};
    hipStream_t st = (hipStream_t)stream;
    if (out_amax) {
        if (!rowscale || N % 16 || ldc % 16) return OV_ERR_INVALID;
        if (epilogue == OV_EPI_BIAS_GELU_ERF) return launch_fp8<OV_EPI_BIAS_GELU_ERF, 1>(a, st);
        if (epilogue == OV_EPI_BIAS_GELU_TANH) return launch_fp8<OV_EPI_BIAS_GELU_TANH, 1>(a, st);
        return OV_ERR_UNSUPPORTED;
    }
    if (epilogue != OV_EPI_BIAS_RESIDUAL) return OV_ERR_UNSUPPORTED;
    if (!R || ldr % 8 || ldr < N || ((uintptr_t)R & 15)) return OV_ERR_INVALID;
    return launch_fp8<OV_EPI_BIAS_RESIDUAL, 2>(a, st);
}
