// Strategy preprocessing of a page on the device: the transforms the reference applies before a read
// (ocr_agent/tools.py:503-546) in the form they take where OpenCV is absent - the PIL fallbacks, which are what
// tests/golden/preprocess_kats.json pins - and the image processor's bicubic resize, all in the library's exact integer /
// dyadic arithmetic so that the pixels equal PIL's bit for bit:
//   high_contrast  ImageEnhance.Contrast(2.0): blend(gray(mean L), image, 2) = clip(2 v - mean) per byte (tools.py:514-516;
//                  Pillow ImageEnhance.py Contrast, libImaging/Blend.c)
//   binarize       convert("L").point(v > 128) (tools.py:530-531; libImaging/Convert.c rgb2l: (19595 R + 38470 G + 7471 B + 0x8000) >> 16)
//   sharpen        ImageFilter.SHARPEN: 3x3 (-2 .. 32 .. -2) / 16, border pixels copied (tools.py:544-546; libImaging/Filter.c)
//   resize         Image.resize(BICUBIC): two separable passes with 22-bit fixed-point coefficients, uint8 between the passes
//                  (HF image_processing_pil_qwen2_vl.py:152-183 -> libImaging/Resample.c); coefficient tables come from the host
//                  (handwritten-ocr_amd/gpupre.py restates precompute_coeffs)
// Images are uint8 [H][W][3].  These are byte-granular HBM-bound kernels: one pass each, 4 bytes per lane.
#include "common.h"
#include "hwocr.h"

namespace {

__device__ __forceinline__ int luma(int r, int g, int b) { return (r * 19595 + g * 38470 + b * 7471 + 0x8000) >> 16; }

__global__ __launch_bounds__(256) void luma_sum_kernel(const unsigned char* rgb, long npix, unsigned long long* sum) {
  unsigned long long s = 0;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < npix; i += (long)gridDim.x * 256)
    s += luma(rgb[3 * i], rgb[3 * i + 1], rgb[3 * i + 2]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if ((threadIdx.x & 63) == 0) atomicAdd(sum, s);
}

__global__ __launch_bounds__(256) void contrast_kernel(const unsigned char* src, unsigned char* dst, long nbytes, int mean,
                                                       float factor) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nbytes; i += (long)gridDim.x * 256) {
    const float t = (float)((float)mean + factor * (float)((int)src[i] - mean));  // Blend.c, alpha outside [0, 1]
    dst[i] = t <= 0.0f ? 0 : (t >= 255.0f ? 255 : (unsigned char)t);
  }
}

// the same blend with the mean taken from the device: mean = int(sum / npix + 0.5) in double, which is Python's
// int(int(sum) / n + 0.5) (both operands below 2^53, one correctly rounded division) - no host round trip between the two kernels
__global__ __launch_bounds__(256) void contrast_dev_kernel(const unsigned char* src, unsigned char* dst, long nbytes,
                                                           const unsigned long long* sum, long npix, float factor) {
  const int mean = (int)((double)*sum / (double)npix + 0.5);
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nbytes; i += (long)gridDim.x * 256) {
    const float t = (float)((float)mean + factor * (float)((int)src[i] - mean));
    dst[i] = t <= 0.0f ? 0 : (t >= 255.0f ? 255 : (unsigned char)t);
  }
}

__global__ __launch_bounds__(256) void binarize_kernel(const unsigned char* rgb, unsigned char* dst, long npix) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < npix; i += (long)gridDim.x * 256) {
    const unsigned char v = luma(rgb[3 * i], rgb[3 * i + 1], rgb[3 * i + 2]) > 128 ? 255 : 0;
    dst[3 * i] = dst[3 * i + 1] = dst[3 * i + 2] = v;  // mode L, as the processor's convert("RGB") replicates it
  }
}

__global__ __launch_bounds__(256) void sharpen_kernel(const unsigned char* src, unsigned char* dst, int H, int W) {
  const long n = (long)H * W * 3;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const int c = i % 3;
    const long p = i / 3;
    const int x = p % W, y = p / W;
    if (x == 0 || y == 0 || x == W - 1 || y == H - 1) {
      dst[i] = src[i];
      continue;
    }
    float ss = 0.5f;  // every term is a multiple of 1/8 below 2^12: exact in fp32 in any order
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
      for (int dx = -1; dx <= 1; ++dx)
        ss += (float)src[((long)(y + dy) * W + (x + dx)) * 3 + c] * ((dx == 0 && dy == 0) ? 2.0f : -0.125f);
    dst[i] = ss <= 0.0f ? 0 : (ss >= 255.0f ? 255 : (unsigned char)ss);
  }
}

// one pass of Resample.c: out[o][j] = clip8((2^21 + sum_k in[b0 + k][j] * coef[o][k]) >> 22) along one axis.
// horizontal: in [rows][n_in][3] -> out [rows][n_out][3]; vertical: in [n_in][cols*3] -> out [n_out][cols*3]
__global__ __launch_bounds__(256) void resample_kernel(const unsigned char* in, unsigned char* out, const int* bounds,
                                                       const int* coef, int ksize, int n_in, int n_out, long lines,
                                                       int horizontal) {
  const long total = lines * n_out * (horizontal ? 3 : 1);
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    long line;
    int o, c = 0;
    if (horizontal) {
      c = i % 3;
      o = (i / 3) % n_out;
      line = i / (3L * n_out);
    } else {
      line = i % lines;  // a byte column of the [n_in][lines] image: consecutive lanes read consecutive bytes
      o = i / lines;
    }
    const int b0 = bounds[2 * o], n = bounds[2 * o + 1];
    const int* k = coef + (long)o * ksize;
    int ss = 1 << 21;
    if (horizontal) {
      const unsigned char* row = in + (line * n_in + b0) * 3 + c;
      for (int t = 0; t < n; ++t) ss += (int)row[3 * t] * k[t];
      ss >>= 22;
      out[(line * n_out + o) * 3 + c] = ss < 0 ? 0 : (ss > 255 ? 255 : ss);
    } else {
      const unsigned char* col = in + (long)b0 * lines + line;
      for (int t = 0; t < n; ++t) ss += (int)col[(long)t * lines] * k[t];
      ss >>= 22;
      out[(long)o * lines + line] = ss < 0 ? 0 : (ss > 255 ? 255 : ss);
    }
  }
}

inline int blocks_for(long n) {
  const long b = (n + 255) / 256;
  return (int)(b < 1 ? 1 : (b > 8192 ? 8192 : b));
}

}  // namespace

extern "C" int hwocr_img_luma_sum(const void* rgb, long npix, unsigned long long* sum, hipStream_t stream) {
  (void)hipGetLastError();
  if (!rgb || !sum || npix <= 0) return HWOCR_EINVAL;
  if (hipMemsetAsync(sum, 0, sizeof(unsigned long long), stream) != hipSuccess) return HWOCR_ELAUNCH;
  hipLaunchKernelGGL(luma_sum_kernel, dim3(blocks_for(npix) > 1024 ? 1024 : blocks_for(npix)), dim3(256), 0, stream,
                     (const unsigned char*)rgb, npix, sum);
  return hwocr_launch_status();
}

extern "C" int hwocr_img_contrast(const void* src, void* dst, long nbytes, int mean, float factor, hipStream_t stream) {
  (void)hipGetLastError();
  if (!src || !dst || nbytes <= 0 || mean < 0 || mean > 255) return HWOCR_EINVAL;
  hipLaunchKernelGGL(contrast_kernel, dim3(blocks_for(nbytes)), dim3(256), 0, stream, (const unsigned char*)src,
                     (unsigned char*)dst, nbytes, mean, factor);
  return hwocr_launch_status();
}

extern "C" int hwocr_img_contrast_dev(const void* src, void* dst, long nbytes, const unsigned long long* sum, long npix,
                                      float factor, hipStream_t stream) {
  (void)hipGetLastError();
  if (!src || !dst || !sum || nbytes <= 0 || npix <= 0) return HWOCR_EINVAL;
  hipLaunchKernelGGL(contrast_dev_kernel, dim3(blocks_for(nbytes)), dim3(256), 0, stream, (const unsigned char*)src,
                     (unsigned char*)dst, nbytes, sum, npix, factor);
  return hwocr_launch_status();
}

extern "C" int hwocr_img_binarize(const void* rgb, void* dst, long npix, hipStream_t stream) {
  (void)hipGetLastError();
  if (!rgb || !dst || npix <= 0) return HWOCR_EINVAL;
  hipLaunchKernelGGL(binarize_kernel, dim3(blocks_for(npix)), dim3(256), 0, stream, (const unsigned char*)rgb,
                     (unsigned char*)dst, npix);
  return hwocr_launch_status();
}

extern "C" int hwocr_img_sharpen(const void* src, void* dst, int H, int W, hipStream_t stream) {
  (void)hipGetLastError();
  if (!src || !dst || src == dst || H < 1 || W < 1) return HWOCR_EINVAL;
  hipLaunchKernelGGL(sharpen_kernel, dim3(blocks_for((long)H * W * 3)), dim3(256), 0, stream, (const unsigned char*)src,
                     (unsigned char*)dst, H, W);
  return hwocr_launch_status();
}

extern "C" int hwocr_img_resize_bicubic(const void* src, void* tmp, void* dst, int H, int W, int out_h, int out_w,
                                        const int* h_bounds, const int* h_coef, int h_ksize, const int* v_bounds,
                                        const int* v_coef, int v_ksize, hipStream_t stream) {
  (void)hipGetLastError();
  if (!src || !tmp || !dst || !h_bounds || !h_coef || !v_bounds || !v_coef || H < 1 || W < 1 || out_h < 1 || out_w < 1 ||
      h_ksize < 1 || v_ksize < 1)
    return HWOCR_EINVAL;
  // horizontal pass over all H rows -> tmp [H][out_w][3], then vertical -> dst [out_h][out_w][3]
  hipLaunchKernelGGL(resample_kernel, dim3(blocks_for((long)H * out_w * 3)), dim3(256), 0, stream, (const unsigned char*)src,
                     (unsigned char*)tmp, h_bounds, h_coef, h_ksize, W, out_w, (long)H, 1);
  hipLaunchKernelGGL(resample_kernel, dim3(blocks_for((long)out_h * out_w * 3)), dim3(256), 0, stream,
                     (const unsigned char*)tmp, (unsigned char*)dst, v_bounds, v_coef, v_ksize, H, out_h, (long)out_w * 3, 0);
  return hwocr_launch_status();
}
