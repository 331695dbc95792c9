// common.h — shared device helpers for the gfx950 kernels (wave64, bf16 bit tricks, vector types).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/ovhip.h"

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
typedef short s16x4_t __attribute__((ext_vector_type(4)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));

#define OV_WAVE 64

#define OV_LAUNCH_CHECK()                                            \
    do {                                                             \
        hipError_t e__ = hipGetLastError();                          \
        if (e__ != hipSuccess) return OV_ERR_HIP - (int)e__;         \
    } while (0)

static inline int ov_hip(hipError_t e) { return e == hipSuccess ? OV_OK : OV_ERR_HIP - (int)e; }

// ---- per-device host state: function attributes (hipFuncSetAttribute) and device properties belong to ONE device; a process
// may drive several (one host thread each).  Device ids beyond OV_MAX_DEVICES - 1 share the last slot's "not yet done" answer
// (the attribute is then set on every launch: correct, just slower).
#define OV_MAX_DEVICES 64
static inline int ov_current_device() {
    int d = 0;
    return hipGetDevice(&d) == hipSuccess && d >= 0 ? d : 0;
}
struct OvPerDeviceOnce {
    unsigned long long done = 0;      // bit d: set for device d (benign race: the guarded calls are idempotent)
    bool need(int dev) const { return dev >= OV_MAX_DEVICES || !((__atomic_load_n(&done, __ATOMIC_ACQUIRE) >> dev) & 1ull); }
    void mark(int dev) { if (dev < OV_MAX_DEVICES) __atomic_fetch_or(&done, 1ull << dev, __ATOMIC_RELEASE); }
};
static inline int ov_num_cus() {
    static int cache[OV_MAX_DEVICES] = {};
    const int dev = ov_current_device();
    int n = dev < OV_MAX_DEVICES ? __atomic_load_n(&cache[dev], __ATOMIC_RELAXED) : 0;
    if (n == 0) {
        hipDeviceProp_t p;
        n = (hipGetDeviceProperties(&p, dev) == hipSuccess && p.multiProcessorCount > 0) ? p.multiProcessorCount : 256;
        if (dev < OV_MAX_DEVICES) __atomic_store_n(&cache[dev], n, __ATOMIC_RELAXED);
    }
    return n;
}

__device__ __forceinline__ float bf16_bits_to_f32(unsigned short h) {
    return __uint_as_float(((unsigned int)h) << 16);
}
__device__ __forceinline__ float bf16lo_to_f32(unsigned int w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf16hi_to_f32(unsigned int w) { return __uint_as_float(w & 0xffff0000u); }

// round-to-nearest-even f32 -> bf16 (plain cast: hipcc emits v_cvt_pk_bf16_f32 on gfx950, NaN-safe)
__device__ __forceinline__ unsigned int pack_bf16x2(float lo, float hi) {
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    bf16x2_t v = {(__bf16)lo, (__bf16)hi};
    return __builtin_bit_cast(unsigned int, v);
}
__device__ __forceinline__ unsigned short f32_to_bf16_bits(float x) {
    __bf16 b = (__bf16)x;
    return __builtin_bit_cast(unsigned short, b);
}
__device__ __forceinline__ float round_bf16(float x) { return bf16_bits_to_f32(f32_to_bf16_bits(x)); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// Exact-erf GELU, 0.5 x (1 + erf(x / sqrt 2)), with erf from Abramowitz & Stegun 7.1.26
// (|abs err| <= 1.5e-7, far below the bf16 output resolution): 1 v_rcp + 1 v_exp + ~10 FMA/MUL instead of
// the ~35-instruction branchy erff().  erfc(z) = poly(t) * exp(-z^2), t = 1 / (1 + p z), z >= 0.
__device__ __forceinline__ float gelu_erf_f(float x) {
    const float z = fabsf(x) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
    float p = fmaf(t, 1.061405429f, -1.453152027f);
    p = fmaf(t, p, 1.421413741f);
    p = fmaf(t, p, -0.284496736f);
    p = fmaf(t, p, 0.254829592f);
    p *= t;
    const float erfc_abs = p * __builtin_amdgcn_exp2f(-1.4426950408889634f * z * z);   // erfc(|x|/sqrt2)
    const float hx = 0.5f * x;
    // x >= 0: 1 + erf = 2 - erfc ; x < 0: 1 + erf = erfc
    return x >= 0.f ? hx * (2.0f - erfc_abs) : hx * erfc_abs;
}
__device__ __forceinline__ float gelu_tanh_f(float x) {
    // 0.5 x (1 + tanh(u)), u = sqrt(2/pi) (x + 0.044715 x^3)   ==  x * sigmoid(2u) = x / (1 + exp(-2u))
    const float u = 0.7978845608028654f * x * fmaf(0.044715f * x, x, 1.0f);
    return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-2.8853900817779268f * u));
}

// ---- two-lane (v_pk_*_f32) forms: the GEMM epilogues are VALU-issue bound, packed fp32 halves the instruction count ----
typedef float f32x2_t __attribute__((ext_vector_type(2)));

// Exact-erf GELU x * Phi(x) for the bf16 epilogue, transcendental-free: Phi(x) = 0.5 + x_c * q(x_c^2), x_c = clamp(x, -4, 4),
// q = degree-7 near-minimax (Chebyshev-node) polynomial; |Phi error| <= 3.6e-5, |GELU error| <= 1.4e-4 on [-4, 4] and
// <= 6.6e-5 * |x| beyond (tools/fit_gelu.py) -- an order of magnitude inside the bf16 output rounding (2^-9 relative).
// 2 v_med3 + 10 packed fp32 ops per PAIR of elements, against 17 + 4 quarter-rate v_rcp/v_exp for the A&S 7.1.26 form.
__device__ __forceinline__ f32x2_t gelu_erf_f2(f32x2_t x) {
    const f32x2_t xc = {__builtin_amdgcn_fmed3f(x[0], -4.0f, 4.0f), __builtin_amdgcn_fmed3f(x[1], -4.0f, 4.0f)};
    const f32x2_t u = xc * xc;
    f32x2_t q = __builtin_elementwise_fma(u, f32x2_t{-1.832668545e-09f, -1.832668545e-09f}, f32x2_t{1.370619420e-07f, 1.370619420e-07f});
    q = __builtin_elementwise_fma(q, u, f32x2_t{-4.476110938e-06f, -4.476110938e-06f});
    q = __builtin_elementwise_fma(q, u, f32x2_t{8.535522916e-05f, 8.535522916e-05f});
    q = __builtin_elementwise_fma(q, u, f32x2_t{-1.079715229e-03f, -1.079715229e-03f});
    q = __builtin_elementwise_fma(q, u, f32x2_t{9.774306659e-03f, 9.774306659e-03f});
    q = __builtin_elementwise_fma(q, u, f32x2_t{-6.634450104e-02f, -6.634450104e-02f});
    q = __builtin_elementwise_fma(q, u, f32x2_t{3.989241898e-01f, 3.989241898e-01f});
    const f32x2_t phi = __builtin_elementwise_fma(xc, q, f32x2_t{0.5f, 0.5f});
    return x * phi;
}
// Two pairs at once, the two Horner chains interleaved statement by statement: back-to-back dependent v_pk_fma_f32
// cost a wait state each (hipcc pads them with s_nop), two independent chains issue without gaps.  Same arithmetic.
__device__ __forceinline__ void gelu_erf_f2x2(f32x2_t& x, f32x2_t& y) {
    const f32x2_t xc = {__builtin_amdgcn_fmed3f(x[0], -4.0f, 4.0f), __builtin_amdgcn_fmed3f(x[1], -4.0f, 4.0f)};
    const f32x2_t yc = {__builtin_amdgcn_fmed3f(y[0], -4.0f, 4.0f), __builtin_amdgcn_fmed3f(y[1], -4.0f, 4.0f)};
    const f32x2_t u = xc * xc, v = yc * yc;
    f32x2_t q = __builtin_elementwise_fma(u, f32x2_t{-1.832668545e-09f, -1.832668545e-09f}, f32x2_t{1.370619420e-07f, 1.370619420e-07f});
    f32x2_t r = __builtin_elementwise_fma(v, f32x2_t{-1.832668545e-09f, -1.832668545e-09f}, f32x2_t{1.370619420e-07f, 1.370619420e-07f});
    q = __builtin_elementwise_fma(q, u, f32x2_t{-4.476110938e-06f, -4.476110938e-06f});
    r = __builtin_elementwise_fma(r, v, f32x2_t{-4.476110938e-06f, -4.476110938e-06f});
    q = __builtin_elementwise_fma(q, u, f32x2_t{8.535522916e-05f, 8.535522916e-05f});
    r = __builtin_elementwise_fma(r, v, f32x2_t{8.535522916e-05f, 8.535522916e-05f});
    q = __builtin_elementwise_fma(q, u, f32x2_t{-1.079715229e-03f, -1.079715229e-03f});
    r = __builtin_elementwise_fma(r, v, f32x2_t{-1.079715229e-03f, -1.079715229e-03f});
    q = __builtin_elementwise_fma(q, u, f32x2_t{9.774306659e-03f, 9.774306659e-03f});
    r = __builtin_elementwise_fma(r, v, f32x2_t{9.774306659e-03f, 9.774306659e-03f});
    q = __builtin_elementwise_fma(q, u, f32x2_t{-6.634450104e-02f, -6.634450104e-02f});
    r = __builtin_elementwise_fma(r, v, f32x2_t{-6.634450104e-02f, -6.634450104e-02f});
    q = __builtin_elementwise_fma(q, u, f32x2_t{3.989241898e-01f, 3.989241898e-01f});
    r = __builtin_elementwise_fma(r, v, f32x2_t{3.989241898e-01f, 3.989241898e-01f});
    const f32x2_t phx = __builtin_elementwise_fma(xc, q, f32x2_t{0.5f, 0.5f});
    const f32x2_t phy = __builtin_elementwise_fma(yc, r, f32x2_t{0.5f, 0.5f});
    x = x * phx;
    y = y * phy;
}
__device__ __forceinline__ f32x2_t gelu_tanh_f2(f32x2_t x) {
    const f32x2_t u = x * 0.7978845608028654f * __builtin_elementwise_fma(x * 0.044715f, x, f32x2_t{1.0f, 1.0f});
    const f32x2_t a = u * -2.8853900817779268f;
    const f32x2_t den = f32x2_t{__builtin_amdgcn_exp2f(a[0]), __builtin_amdgcn_exp2f(a[1])} + 1.0f;
    return x * f32x2_t{__builtin_amdgcn_rcpf(den[0]), __builtin_amdgcn_rcpf(den[1])};
}

// d/dx of the exact-erf GELU, Phi(x) + x phi(x), for the backward's fused epilogue (OV_EPI_GELU_GRAD_ERF), transcendental-free like
// gelu_erf_f2: gelu'(x) = 0.5 + x_c s(x_c^2), x_c = clamp(x, -4, 4), s = degree-7 Chebyshev-node fit (tools/fit_gelu_grad.py);
// |error| <= 4.3e-4 everywhere -- the factor multiplies a bf16 gradient whose own rounding is 2^-9 relative.  Two pairs at once,
// the Horner chains interleaved (see gelu_erf_f2x2).
__device__ __forceinline__ void gelu_erf_grad_f2x2(f32x2_t& x, f32x2_t& y) {
    const f32x2_t xc = {__builtin_amdgcn_fmed3f(x[0], -4.0f, 4.0f), __builtin_amdgcn_fmed3f(x[1], -4.0f, 4.0f)};
    const f32x2_t yc = {__builtin_amdgcn_fmed3f(y[0], -4.0f, 4.0f), __builtin_amdgcn_fmed3f(y[1], -4.0f, 4.0f)};
    const f32x2_t u = xc * xc, v = yc * yc;
    f32x2_t q = __builtin_elementwise_fma(u, f32x2_t{-1.933266631e-08f, -1.933266631e-08f}, f32x2_t{1.391892320e-06f, 1.391892320e-06f});
    f32x2_t r = __builtin_elementwise_fma(v, f32x2_t{-1.933266631e-08f, -1.933266631e-08f}, f32x2_t{1.391892320e-06f, 1.391892320e-06f});
    q = __builtin_elementwise_fma(q, u, f32x2_t{-4.282898866e-05f, -4.282898866e-05f});
    r = __builtin_elementwise_fma(r, v, f32x2_t{-4.282898866e-05f, -4.282898866e-05f});
    q = __builtin_elementwise_fma(q, u, f32x2_t{7.424731401e-04f, 7.424731401e-04f});
    r = __builtin_elementwise_fma(r, v, f32x2_t{7.424731401e-04f, 7.424731401e-04f});
    q = __builtin_elementwise_fma(q, u, f32x2_t{-8.057965438e-03f, -8.057965438e-03f});
    r = __builtin_elementwise_fma(r, v, f32x2_t{-8.057965438e-03f, -8.057965438e-03f});
    q = __builtin_elementwise_fma(q, u, f32x2_t{5.721020480e-02f, 5.721020480e-02f});
    r = __builtin_elementwise_fma(r, v, f32x2_t{5.721020480e-02f, 5.721020480e-02f});
    q = __builtin_elementwise_fma(q, u, f32x2_t{-2.640489873e-01f, -2.640489873e-01f});
    r = __builtin_elementwise_fma(r, v, f32x2_t{-2.640489873e-01f, -2.640489873e-01f});
    q = __builtin_elementwise_fma(q, u, f32x2_t{7.976477333e-01f, 7.976477333e-01f});
    r = __builtin_elementwise_fma(r, v, f32x2_t{7.976477333e-01f, 7.976477333e-01f});
    x = __builtin_elementwise_fma(xc, q, f32x2_t{0.5f, 0.5f});
    y = __builtin_elementwise_fma(yc, r, f32x2_t{0.5f, 0.5f});
}
// d/dx of the tanh-form GELU: sg + 2 x sg (1 - sg) u'(x), sg = sigmoid(2u), u = sqrt(2/pi) (x + 0.044715 x^3)
__device__ __forceinline__ f32x2_t gelu_tanh_grad_f2(f32x2_t x) {
    const f32x2_t x2 = x * x;
    const f32x2_t u = x * 0.7978845608028654f * __builtin_elementwise_fma(x2, f32x2_t{0.044715f, 0.044715f}, f32x2_t{1.0f, 1.0f});
    const f32x2_t a = u * -2.8853900817779268f;
    const f32x2_t den = f32x2_t{__builtin_amdgcn_exp2f(a[0]), __builtin_amdgcn_exp2f(a[1])} + 1.0f;
    const f32x2_t sg = {__builtin_amdgcn_rcpf(den[0]), __builtin_amdgcn_rcpf(den[1])};
    const f32x2_t du = __builtin_elementwise_fma(x2, f32x2_t{3.0f * 0.044715f * 0.7978845608028654f, 3.0f * 0.044715f * 0.7978845608028654f},
                                                 f32x2_t{0.7978845608028654f, 0.7978845608028654f});
    return __builtin_elementwise_fma(x * 2.0f * sg * (f32x2_t{1.0f, 1.0f} - sg), du, sg);
}

// The epilogues that read a second operand R (same rows and columns as C): out = combine(bf16(acc + bias), R), per packed bf16 pair.
//   OV_EPI_BIAS_RESIDUAL (3): o + r        OV_EPI_GELU_GRAD_ERF (4) / _TANH (5): o * gelu'(r)
template <int EPI>
__device__ __forceinline__ u32x4_t epi_combine(u32x4_t o, u32x4_t r) {
    u32x4_t out;
    if (EPI == 3) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
            out[e] = pack_bf16x2(bf16lo_to_f32(o[e]) + bf16lo_to_f32(r[e]), bf16hi_to_f32(o[e]) + bf16hi_to_f32(r[e]));
    } else {
#pragma unroll
        for (int e = 0; e < 4; e += 2) {
            f32x2_t g0 = {bf16lo_to_f32(r[e]), bf16hi_to_f32(r[e])}, g1 = {bf16lo_to_f32(r[e + 1]), bf16hi_to_f32(r[e + 1])};
            if (EPI == 4) gelu_erf_grad_f2x2(g0, g1);
            else { g0 = gelu_tanh_grad_f2(g0); g1 = gelu_tanh_grad_f2(g1); }
            const f32x2_t v0 = f32x2_t{bf16lo_to_f32(o[e]), bf16hi_to_f32(o[e])} * g0;
            const f32x2_t v1 = f32x2_t{bf16lo_to_f32(o[e + 1]), bf16hi_to_f32(o[e + 1])} * g1;
            out[e] = pack_bf16x2(v0[0], v0[1]);
            out[e + 1] = pack_bf16x2(v1[0], v1[1]);
        }
    }
    return out;
}

// The lane id, recomputed where it is needed (two VALU instructions) and opaque to CSE: as ONE value defined at kernel entry it is live
// through every K-tile variant and epilogue, and in the kernels at the register limit (fp8 GEMM, persistent GEMM with row statistics) it was what got spilled: a scratch reload
// inside the epilogue is a vector-memory operation the hand-counted vmcnt waits do not know about.
__device__ __forceinline__ int fresh_lane() {
    int l = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    asm volatile("" : "+v"(l));
    return l;
}

// ---- row statistics in partial sums (round 3): {sum x, sum x^2} over every 32-column group of a row, computed on the bf16-ROUNDED
// values with ONE sequence of operations wherever they are produced -- the persistent GEMM's residual epilogue, the skinny GEMM's, or
// the stand-alone pass over x (ov_rowparts) -- so that a row's statistics are bitwise the same whichever kernel wrote the row:
//   quad  (4 consecutive n = two packed dwords): two chained v_dot2c_f32_bf16 from 0 (sum: against {1, 1}; squares: against itself --
//                                                 bf16 products are exact in fp32, the adds round as the instruction rounds them)
//   octet (8 consecutive n)                     = quad(n 0-3) + quad(n 4-7)
//   block (16 n = one MFMA fragment column)     = octet 0 + octet 1
//   group (32 n)                                = block 0 + block 1
// __fadd_rn: no contraction, no re-association.  The dot instructions are inline asm (one statement: the two accumulation chains
// interleaved, the wait states a DOT result needs before an ordinary VALU read appended -- the assembler does not look inside asm).
__device__ __forceinline__ void stat_quad(unsigned w0, unsigned w1, float& s, float& q) {
    float ss = 0.f, qq = 0.f;
    const unsigned ones = 0x3f803f80u;       // bf16 {1, 1}
    asm("v_dot2c_f32_bf16 %0, %2, %4\n\tv_dot2c_f32_bf16 %1, %2, %2\n\tv_dot2c_f32_bf16 %0, %3, %4\n\tv_dot2c_f32_bf16 %1, %3, %3\n\ts_nop 2"
        : "+v"(ss), "+v"(qq) : "v"(w0), "v"(w1), "v"(ones));
    s = ss;
    q = qq;
}
__device__ __forceinline__ void stat_octet(u32x4_t w, float& s, float& q) {
    float s0, q0, s1, q1;
    stat_quad(w[0], w[1], s0, q0);
    stat_quad(w[2], w[3], s1, q1);
    s = __fadd_rn(s0, s1);
    q = __fadd_rn(q0, q1);
}
// x[lane] + x[lane ^ 32] in the order (lower half) + (upper half), in every lane
__device__ __forceinline__ float add_halves_lo_first(float x) {
    const u32x2_t t = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __fadd_rn(__uint_as_float(t[0]), __uint_as_float(t[1]));
}
// x[even 16-lane row] + x[odd 16-lane row] of each pair of rows, in that order, in every lane of the pair
__device__ __forceinline__ float add_rowpair_even_first(float x) {
    const u32x2_t t = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __fadd_rn(__uint_as_float(t[0]), __uint_as_float(t[1]));
}

// 16-byte output store with a compile-time cache policy: 0 = plain, 1 = sc1 (write-through; the line is not kept in the XCD's L2 --
// MI355X_MICROARCH.md, store flavours), 2 = nt (streaming), 4 = sc0 sc1.  A tile's 128 KiB of output otherwise displace the W slice /
// A panels the next K-tiles are about to be fetched from (32 CUs x 128 KiB = the whole 4 MiB L2 of an XCD per round of tiles).
#ifndef OVHIP_ST_LDS
#define OVHIP_ST_LDS 2          /* policy of the LDS-transposed (whole-line) epilogue's stores */
#endif
#ifndef OVHIP_ST_DIRECT
#define OVHIP_ST_DIRECT 2       /* policy of the direct (row-per-lane, half-line) epilogue's stores */
#endif
#ifndef OVHIP_ST_RESID
#define OVHIP_ST_RESID 0        /* residual epilogues: the output is the residual stream itself, re-read at once -- plain (45.46 -> 45.35 ms at L/14) */
#endif
template <int POLICY>
__device__ __forceinline__ void store16(void* dst, u32x4_t v) {
    if (POLICY == 1) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" :: "v"(dst), "v"(v) : "memory");
    else if (POLICY == 2) asm volatile("global_store_dwordx4 %0, %1, off nt\n\ts_nop 1" :: "v"(dst), "v"(v) : "memory");
    else if (POLICY == 4) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" :: "v"(dst), "v"(v) : "memory");
    else *(u32x4_t*)dst = v;
}

