// LDS-DMA issue-rate probe (gfx950): what does staging one 64-KiB K-tile per CU cost the TA when every global_load_lds_dwordx4
// touches 16 rows x 64 B (the GEMM's pieces: half a cache line per row) against 8 rows x 128 B (whole lines)?
// Build: hipcc --offload-arch=gfx950 -O3 -o dma_probe tools/dbg/dma_probe.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <int MODE>   // 0: 16 rows x 64 B per instruction (A kh0, W kh0, A kh1, W kh1), 1: 8 rows x 128 B, 2: as 0 but both k-halves of the same rows back to back
__global__ __launch_bounds__(512, 2) void probe(const unsigned short* __restrict__ A, const unsigned short* __restrict__ W, long lda, long ldw,
                                                 int rows_a, int iters, unsigned long long* out) {
    __shared__ __attribute__((aligned(16))) char smem[131072];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // c_fc-like traffic: the 4 workgroups of a group (same XCD) share an A panel [256, K] per tile and hold one W block [256, K] each
    // (the group's W slice 4 x 256 x K stays in the XCD's L2); a tile = K / 64 iterations, then the next A panel
    const int xcd = blockIdx.x & 7, li = blockIdx.x >> 3, grp = li >> 2, mem = li & 3;
    const int gid = xcd * 8 + grp;
    const unsigned short* ap[2];
    const unsigned short* wp[2];
    if (MODE != 1) {          // instruction i of a piece covers rows i*128 + tid/4, 64 B (32 elements) per row, chunk tid & 3
        for (int i = 0; i < 2; ++i) {
            ap[i] = A + (long)(i * 128 + (tid >> 2)) * lda + (tid & 3) * 8;
            wp[i] = W + (long)(mem * 256 + i * 128 + (tid >> 2)) * ldw + (tid & 3) * 8;
        }
    } else {                  // instruction i covers rows i*64 + tid/8, 128 B (64 elements) per row, chunk tid & 7: 4 instructions per operand
        for (int i = 0; i < 2; ++i) {
            ap[i] = A + (long)(i * 64 + (tid >> 3)) * lda + (tid & 7) * 8;
            wp[i] = W + (long)(mem * 256 + i * 64 + (tid >> 3)) * ldw + (tid & 7) * 8;
        }
    }
    char* dst = smem + wave * 1024;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    const int kt = (int)(lda / 64);                   // K-tiles per tile
    for (int it = 0; it < iters; ++it) {
        const int tile = it / kt;
        const long k = (long)(((long)gid * (iters / kt) + tile) % (rows_a / 256)) * 256 * lda + (it - tile * kt) * 64;   // A: panel + K offset
        const int kw = (it - tile * kt) * 64;
        char* d = dst + (it & 1) * 65536;
        if (MODE == 2) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int kh = 0; kh < 2; ++kh)
                    __builtin_amdgcn_global_load_lds((gptr_t)(ap[i] + k + kh * 32), (lptr_t)(d + kh * 32768 + i * 8192), 16, 0, 0);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int kh = 0; kh < 2; ++kh)
                    __builtin_amdgcn_global_load_lds((gptr_t)(wp[i] + kw + kh * 32), (lptr_t)(d + kh * 32768 + 16384 + i * 8192), 16, 0, 0);
        } else if (MODE == 0) {
#pragma unroll
            for (int kh = 0; kh < 2; ++kh) {      // pieces: (A, kh), (W, kh): 2 instructions each
                __builtin_amdgcn_global_load_lds((gptr_t)(ap[0] + k + kh * 32), (lptr_t)(d + kh * 32768), 16, 0, 0);
                __builtin_amdgcn_global_load_lds((gptr_t)(ap[1] + k + kh * 32), (lptr_t)(d + kh * 32768 + 8192), 16, 0, 0);
                __builtin_amdgcn_global_load_lds((gptr_t)(wp[0] + kw + kh * 32), (lptr_t)(d + kh * 32768 + 16384), 16, 0, 0);
                __builtin_amdgcn_global_load_lds((gptr_t)(wp[1] + kw + kh * 32), (lptr_t)(d + kh * 32768 + 24576), 16, 0, 0);
            }
        } else {
#pragma unroll
            for (int q = 0; q < 2; ++q) {         // rows q*128 .. : 2 instructions of 64 rows each per operand and q
                __builtin_amdgcn_global_load_lds((gptr_t)(ap[0] + (long)q * 128 * lda + k), (lptr_t)(d + q * 32768), 16, 0, 0);
                __builtin_amdgcn_global_load_lds((gptr_t)(ap[1] + (long)q * 128 * lda + k), (lptr_t)(d + q * 32768 + 8192), 16, 0, 0);
                __builtin_amdgcn_global_load_lds((gptr_t)(wp[0] + (long)q * 128 * ldw + kw), (lptr_t)(d + q * 32768 + 16384), 16, 0, 0);
                __builtin_amdgcn_global_load_lds((gptr_t)(wp[1] + (long)q * 128 * ldw + kw), (lptr_t)(d + q * 32768 + 24576), 16, 0, 0);
            }
        }
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    __syncthreads();
    if (tid == 0) out[blockIdx.x] = t1 - t0 + (smem[(t1 & 127)] == 77 ? 1 : 0);
}

int main() {
    const int rows_a = 65536, K = 1024, iters = 16 * (K / 64);      // 16 tiles of K / 64 K-tiles per workgroup
    unsigned short *A, *W; unsigned long long* out;
    hipMalloc(&A, (size_t)rows_a * K * 2); hipMalloc(&W, (size_t)1024 * K * 2); hipMalloc(&out, 256 * 8);
    hipMemset(A, 1, (size_t)rows_a * K * 2); hipMemset(W, 1, (size_t)1024 * K * 2);
    std::vector<unsigned long long> h(256);
    for (int mode = 0; mode < 3; ++mode)
        for (int rep = 0; rep < 3; ++rep) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(probe<0>, dim3(256), dim3(512), 0, 0, A, W, (long)K, (long)K, rows_a, iters, out);
            else if (mode == 2) hipLaunchKernelGGL(probe<2>, dim3(256), dim3(512), 0, 0, A, W, (long)K, (long)K, rows_a, iters, out);
            else hipLaunchKernelGGL(probe<1>, dim3(256), dim3(512), 0, 0, A, W, (long)K, (long)K, rows_a, iters, out);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            hipMemcpy(h.data(), out, 256 * 8, hipMemcpyDeviceToHost);
            double avg = 0; for (auto v : h) avg += v; avg /= 256;
            printf("mode %d (%s): %.1f us, %.0f cycles per 64-KiB K-tile per CU (s_memtime), %.2f TB/s\n", mode, mode == 1 ? "8 rows x 128 B" : mode == 2 ? "16 rows x 64 B, k-halves back to back" : "16 rows x 64 B",
                   ms * 1e3, avg / iters, 256.0 * iters * 65536 / (ms * 1e-3) / 1e12);
        }
    return 0;
}
