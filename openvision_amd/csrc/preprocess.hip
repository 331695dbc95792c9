// preprocess.hip — device-side image front-end of encode_image, gfx950.
//
// Replaces the per-image CPU transform in front of the towers:
//     transforms.Resize -> convert("RGB") -> ToTensor -> Normalize      (reference ov-zero-shot-test.py:72-77;
//     open_clip/transform.py:355-392: Resize [+ CenterCrop] -> _convert_to_rgb -> ToTensor -> normalize)
// torchvision's Resize on a PIL image is PIL.Image.resize, i.e. Pillow's two-pass fixed-point convolution
// (src/libImaging/Resample.c, 8 bits per channel: 22-bit coefficients, int32 accumulation started at 1 << 21, >> 22, clipped,
// the horizontal pass rounded to uint8 before the vertical one).  The coefficient tables come from the host planner
// (openvision_amd/preprocess.py, same double arithmetic as precompute_coeffs / normalize_coeffs_8bpc); the kernels below do the
// integer convolution exactly, so the uint8 result equals Pillow's bit for bit, then ToTensor (x / 255) and Normalize
// ((x - mean) / std) in IEEE fp32 like torch.
//
// HBM-bound byte work: every source byte is read once from HBM (neighbouring lanes share taps through L1/L2), the uint8
// intermediate [H, Wr, 3] is written and read once, the CHW output is written once.
#include "common.h"

namespace {

// horizontal pass: out[y][xx][c] = clip8((1 << 21) + sum_x in[y][xmin + x][c] * k[xx][x]) for xx in [0, Wr)
__global__ __launch_bounds__(256) void resample_h_u8(const unsigned char* __restrict__ in, int H, int W,
                                                    const int* __restrict__ bounds, const int* __restrict__ kk, int ksize,
                                                    unsigned char* __restrict__ out, int Wr) {
    const int xx = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    if (xx >= Wr) return;
    const int xmin = bounds[2 * xx], cnt = bounds[2 * xx + 1];
    const int* k = kk + (int64_t)xx * ksize;
    const unsigned char* p = in + ((int64_t)y * W + xmin) * 3;
    int s0 = 1 << 21, s1 = 1 << 21, s2 = 1 << 21;
    for (int x = 0; x < cnt; ++x) {
        const int w = k[x];
        s0 += (int)p[3 * x] * w;
        s1 += (int)p[3 * x + 1] * w;
        s2 += (int)p[3 * x + 2] * w;
    }
    auto clip8 = [](int v) { v >>= 22; return (unsigned char)(v < 0 ? 0 : (v > 255 ? 255 : v)); };
    unsigned char* o = out + ((int64_t)y * Wr + xx) * 3;
    o[0] = clip8(s0); o[1] = clip8(s1); o[2] = clip8(s2);
}

// vertical pass over the cropped window + ToTensor + Normalize: dst[c][yo][xo], yo/xo in [0, S)
template <bool OUT_BF16>
__global__ __launch_bounds__(256) void resample_v_norm(const unsigned char* __restrict__ tmp, int Wr, int row0,
                                                       const int* __restrict__ bounds, const int* __restrict__ kk, int ksize,
                                                       int x0, int y0, int So_h, int So_w, float m0, float m1, float m2,
                                                       float d0, float d1, float d2, void* __restrict__ dst) {
    const int xo = blockIdx.x * blockDim.x + threadIdx.x;
    const int yo = blockIdx.y;
    if (xo >= So_w) return;
    const int yy = yo + y0;
    const int ymin = bounds[2 * yy] - row0, cnt = bounds[2 * yy + 1];
    const int* k = kk + (int64_t)yy * ksize;
    const unsigned char* p = tmp + ((int64_t)ymin * Wr + xo + x0) * 3;
    int s0 = 1 << 21, s1 = 1 << 21, s2 = 1 << 21;
    for (int y = 0; y < cnt; ++y) {
        const int w = k[y];
        const unsigned char* q = p + (int64_t)y * Wr * 3;
        s0 += (int)q[0] * w;
        s1 += (int)q[1] * w;
        s2 += (int)q[2] * w;
    }
    auto clip8 = [](int v) { v >>= 22; return v < 0 ? 0 : (v > 255 ? 255 : v); };
    const float v0 = ((float)clip8(s0) / 255.0f - m0) / d0;
    const float v1 = ((float)clip8(s1) / 255.0f - m1) / d1;
    const float v2 = ((float)clip8(s2) / 255.0f - m2) / d2;
    const int64_t plane = (int64_t)So_h * So_w, o = (int64_t)yo * So_w + xo;
    if (OUT_BF16) {
        ov_bf16* d = (ov_bf16*)dst;
        d[o] = f32_to_bf16_bits(v0); d[plane + o] = f32_to_bf16_bits(v1); d[2 * plane + o] = f32_to_bf16_bits(v2);
    } else {
        float* d = (float*)dst;
        d[o] = v0; d[plane + o] = v1; d[2 * plane + o] = v2;
    }
}

}  // namespace

extern "C" int ov_preprocess_image(const unsigned char* src, int H, int W, const int* bounds_x, const int* coef_x, int ksize_x,
                                   int Wr, const int* bounds_y, const int* coef_y, int ksize_y, int Hr, int row0, int nrows,
                                   unsigned char* tmp, int crop_x, int crop_y, int out_h, int out_w, const float* mean,
                                   const float* stdv, void* out, int out_dtype, ov_stream_t stream) {
    if (!src || !bounds_x || !coef_x || !bounds_y || !coef_y || !tmp || !out || !mean || !stdv) return OV_ERR_INVALID;
    if (H <= 0 || W <= 0 || Wr <= 0 || Hr <= 0 || ksize_x <= 0 || ksize_y <= 0 || out_h <= 0 || out_w <= 0) return OV_ERR_INVALID;
    if (row0 < 0 || nrows <= 0 || row0 + nrows > H) return OV_ERR_INVALID;
    if (crop_x < 0 || crop_y < 0 || crop_x + out_w > Wr || crop_y + out_h > Hr) return OV_ERR_INVALID;
    if (out_dtype != OV_F32 && out_dtype != OV_BF16) return OV_ERR_INVALID;
    if (nrows > 65535 || out_h > 65535) return OV_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    // horizontal pass only over the source rows the vertical pass reads ([row0, row0 + nrows), as Pillow's ybox)
    hipLaunchKernelGGL(resample_h_u8, dim3((Wr + 255) / 256, nrows), dim3(256), 0, st, src + (int64_t)row0 * W * 3, nrows, W,
                       bounds_x, coef_x, ksize_x, tmp, Wr);
    OV_LAUNCH_CHECK();
    const dim3 grid((out_w + 255) / 256, out_h);
    if (out_dtype == OV_BF16)
        hipLaunchKernelGGL(resample_v_norm<true>, grid, dim3(256), 0, st, tmp, Wr, row0, bounds_y, coef_y, ksize_y, crop_x, crop_y,
                           out_h, out_w, mean[0], mean[1], mean[2], stdv[0], stdv[1], stdv[2], out);
    else
        hipLaunchKernelGGL(resample_v_norm<false>, grid, dim3(256), 0, st, tmp, Wr, row0, bounds_y, coef_y, ksize_y, crop_x, crop_y,
                           out_h, out_w, mean[0], mean[1], mean[2], stdv[0], stdv[1], stdv[2], out);
    OV_LAUNCH_CHECK();
    return OV_OK;
}
