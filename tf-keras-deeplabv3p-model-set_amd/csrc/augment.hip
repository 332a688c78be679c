// Byte-level augmentations of SegmentationGenerator.__getitem__ on the device (reference deeplabv3p/data.py:72-104,
// common/data_utils.py): the two flips, the crop branch of random_crop, and the four PIL ImageEnhance adjustments
// (random_brightness / random_chroma / random_contrast / random_sharpness, data_utils.py:83-239).  The random draws stay
// on the host (the reference draws them with np.random / random); these kernels apply a batch of drawn decisions to a
// batch of uint8 RGB images (N,H,W,3) and uint8 label maps (N,H,W), bit for bit what PIL / NumPy produce:
//   ImageEnhance.X(img).enhance(f) = Image.blend(degenerate, img, f):
//     0 <= f <= 1:  out = (uint8)(d + f * (v - d))            (float32, truncation)
//     otherwise  :  out = clip(d + f * (v - d)) to [0, 255], then truncation
//   degenerate:  Brightness 0;  Color L(pixel) = (19595 R + 38470 G + 7471 B + 0x8000) >> 16;
//                Contrast int(mean(L over the image) + 0.5);  Sharpness ImageFilter.SMOOTH = 3x3 (1 1 1 / 1 5 1 / 1 1 1) / 13,
//                + 0.5 then truncation (= (2 S + 13) / 26 in integers), border pixels copied.
// Pinned against PIL itself by tests/golden/make_pil_enhance.py.
#include "common.h"

__device__ __forceinline__ unsigned char pil_blend(int d, int v, float f, bool interp) {
  const float t = (float)d + f * (float)(v - d);
  if (interp) return (unsigned char)t;
  return t <= 0.f ? 0 : (t >= 255.f ? 255 : (unsigned char)t);
}
__device__ __forceinline__ int pil_luma(int r, int g, int b) { return (r * 19595 + g * 38470 + b * 7471 + 0x8000) >> 16; }

// sum of L over each image (integer atomics: exact, order-independent)
__global__ __launch_bounds__(256) void aug_luma_sum_kernel(const unsigned char* img, unsigned long long* sums, long long P) {
  const int n = blockIdx.y;
  const unsigned char* p = img + (size_t)n * P * 3;
  unsigned long long s = 0;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < P; i += (long long)gridDim.x * 256)
    s += (unsigned)pil_luma(p[3 * i], p[3 * i + 1], p[3 * i + 2]);
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if ((threadIdx.x & 63) == 0) atomicAdd(&sums[n], s);
}

template <int OP>
__global__ __launch_bounds__(256) void aug_enhance_kernel(const unsigned char* img, unsigned char* out, const float* factor,
                                                          const unsigned long long* sums, int H, int W) {
  const int n = blockIdx.y;
  const long long P = (long long)H * W;
  const unsigned char* src = img + (size_t)n * P * 3;
  unsigned char* dst = out + (size_t)n * P * 3;
  const float f = factor[n];
  const bool interp = f >= 0.f && f <= 1.f;
  int mean = 0;
  if (OP == 2) mean = (int)((double)sums[n] / (double)P + 0.5);
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < P; i += (long long)gridDim.x * 256) {
    const int r = src[3 * i], g = src[3 * i + 1], b = src[3 * i + 2];
    int dr = 0, dg = 0, db = 0;
    if (OP == 1) dr = dg = db = pil_luma(r, g, b);
    if (OP == 2) dr = dg = db = mean;
    if (OP == 3) {
      const int y = (int)(i / W), x = (int)(i - (long long)y * W);
      if (y == 0 || x == 0 || y == H - 1 || x == W - 1) {
        dr = r; dg = g; db = b;
      } else {
        int s[3] = {0, 0, 0};
#pragma unroll
        for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
          for (int dx = -1; dx <= 1; ++dx) {
            const unsigned char* q = src + ((long long)(y + dy) * W + (x + dx)) * 3;
            const int k = (dy == 0 && dx == 0) ? 5 : 1;
            s[0] += k * q[0]; s[1] += k * q[1]; s[2] += k * q[2];
          }
        dr = (2 * s[0] + 13) / 26; dg = (2 * s[1] + 13) / 26; db = (2 * s[2] + 13) / 26;
      }
    }
    dst[3 * i] = pil_blend(dr, r, f, interp);
    dst[3 * i + 1] = pil_blend(dg, g, f, interp);
    dst[3 * i + 2] = pil_blend(db, b, f, interp);
  }
}

extern "C" int dl3p_aug_enhance_u8(const unsigned char* img, unsigned char* out, const float* factor, int op,
                                   unsigned long long* sums, int N, int H, int W, void* stream) {
  DL3P_CHECK_ARG(img && out && factor && N > 0 && H > 0 && W > 0 && op >= 0 && op <= 3, "dl3p_aug_enhance_u8: bad arguments");
  DL3P_CHECK_ARG(op != 3 || img != out, "dl3p_aug_enhance_u8: sharpness cannot run in place");
  DL3P_CHECK_ARG(op != 2 || sums, "dl3p_aug_enhance_u8: contrast needs the N-entry sums workspace");
  hipStream_t st = (hipStream_t)stream;
  const long long P = (long long)H * W;
  long long gx = ceil_div_ll(P, 256 * 8);
  if (gx > 1024) gx = 1024;
  if (gx < 1) gx = 1;
  const dim3 grid((unsigned)gx, N), block(256);
  if (op == 2) {
    hipError_t e = hipMemsetAsync(sums, 0, sizeof(unsigned long long) * N, st);
    DL3P_CHECK_ARG(e == hipSuccess, "dl3p_aug_enhance_u8: memset failed: %s", hipGetErrorString(e));
    hipLaunchKernelGGL(aug_luma_sum_kernel, grid, block, 0, st, img, sums, P);
  }
  if (op == 0) hipLaunchKernelGGL(aug_enhance_kernel<0>, grid, block, 0, st, img, out, factor, sums, H, W);
  else if (op == 1) hipLaunchKernelGGL(aug_enhance_kernel<1>, grid, block, 0, st, img, out, factor, sums, H, W);
  else if (op == 2) hipLaunchKernelGGL(aug_enhance_kernel<2>, grid, block, 0, st, img, out, factor, sums, H, W);
  else hipLaunchKernelGGL(aug_enhance_kernel<3>, grid, block, 0, st, img, out, factor, sums, H, W);
  DL3P_CHECK_LAUNCH("dl3p_aug_enhance_u8");
  return DL3P_OK;
}

// flips (flags[n] bit 0: horizontal = cv2.flip(.., 1), bit 1: vertical = cv2.flip(.., 0)) and a crop window, in one
// gather: out[n, y, x] = in[n, fy(y0[n] + y), fx(x0[n] + x)] for an (h, w) window at (y0, x0) (NULL: the whole image)
__global__ __launch_bounds__(256) void aug_flip_crop_kernel(const unsigned char* img, unsigned char* out, const unsigned char* label,
                                                            unsigned char* label_out, const int* flags, const int* yx, int H,
                                                            int W, int h, int w) {
  const int n = blockIdx.y;
  const int fl = flags ? flags[n] : 0;
  const int y0 = yx ? yx[2 * n] : 0, x0 = yx ? yx[2 * n + 1] : 0;
  const long long P = (long long)h * w;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < P; i += (long long)gridDim.x * 256) {
    const int y = (int)(i / w), x = (int)(i - (long long)y * w);
    // the reference flips first, then crops: the window is taken from the flipped image
    int sy = y0 + y, sx = x0 + x;
    if (fl & 2) sy = H - 1 - sy;
    if (fl & 1) sx = W - 1 - sx;
    const long long s = ((long long)n * H + sy) * W + sx, d = (long long)n * P + i;
    if (img) {
      out[3 * d] = img[3 * s]; out[3 * d + 1] = img[3 * s + 1]; out[3 * d + 2] = img[3 * s + 2];
    }
    if (label) label_out[d] = label[s];
  }
}

extern "C" int dl3p_aug_flip_crop_u8(const unsigned char* img, unsigned char* out, const unsigned char* label,
                                     unsigned char* label_out, const int* flags, const int* yx, int N, int H, int W, int h,
                                     int w, void* stream) {
  DL3P_CHECK_ARG((img || label) && N > 0 && H > 0 && W > 0 && h > 0 && w > 0 && h <= H && w <= W, "dl3p_aug_flip_crop_u8: bad arguments");
  DL3P_CHECK_ARG((!img || (out && out != img)) && (!label || (label_out && label_out != label)), "dl3p_aug_flip_crop_u8: needs distinct outputs");
  DL3P_CHECK_ARG(yx || (h == H && w == W), "dl3p_aug_flip_crop_u8: a window smaller than the image needs its offsets");
  const long long P = (long long)h * w;
  long long gx = ceil_div_ll(P, 256 * 8);
  if (gx > 1024) gx = 1024;
  if (gx < 1) gx = 1;
  hipLaunchKernelGGL(aug_flip_crop_kernel, dim3((unsigned)gx, N), dim3(256), 0, (hipStream_t)stream, img, out, label, label_out,
                     flags, yx, H, W, h, w);
  DL3P_CHECK_LAUNCH("dl3p_aug_flip_crop_u8");
  return DL3P_OK;
}

// GridMask (random_gridmask, common/data_utils.py:276-361; Grid.__call__ with mode = 1): image and label are multiplied by
// 1 - rotate(grid) where grid is a square of ones of edge hh = ceil(sqrt(h^2 + w^2)) with zero row / column bands of width l
// every d pixels, rotated by PIL's Image.rotate (NEAREST, zero fill) and centre-cropped.  Nothing is materialised here: the
// factor of pixel (y, x) is evaluated from the draws -- source pixel through Pillow's 16.16 fixed-point affine map (or its
// transpose fast paths at 0 / 90 / 180 / 270 degrees), then the band test.  params[n] = {apply, hh, d, l, st_h, st_w, kind,
// a0 .. a5, top, left} as oracle/np_augment.gridmask_params writes them (the host draws d, st_h, st_w, r like the reference).
__global__ __launch_bounds__(256) void aug_gridmask_kernel(unsigned char* img, unsigned char* label, const int* params, int H, int W) {
  const int n = blockIdx.y;
  const int* q = params + 16 * n;
  if (!q[0]) return;
  const int hh = q[1], d = q[2], l = q[3], st_h = q[4], st_w = q[5], kind = q[6];
  const long long a0 = q[7], a1 = q[8], a2 = q[9], a3 = q[10], a4 = q[11], a5 = q[12];
  const int top = q[13], left = q[14];
  const long long P = (long long)H * W;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < P; i += (long long)gridDim.x * 256) {
    const int y = (int)(i / W), x = (int)(i - (long long)y * W);
    const int Y = y + top, X = x + left;
    long long xin, yin;
    if (kind == 0) { xin = X; yin = Y; }
    else if (kind == 1) { yin = X; xin = hh - 1 - Y; }          // ROTATE_90:  rot[Y][X] = m[X][hh-1-Y]
    else if (kind == 2) { yin = hh - 1 - Y; xin = hh - 1 - X; }
    else if (kind == 3) { yin = hh - 1 - X; xin = Y; }          // ROTATE_270: rot[Y][X] = m[hh-1-X][Y]
    else { xin = (a2 + a1 * Y + a0 * X) >> 16; yin = (a5 + a4 * Y + a3 * X) >> 16; }
    int rot = 0;                                                 // outside the source: fill 0
    if (xin >= 0 && xin < hh && yin >= 0 && yin < hh) {
      int ry = ((int)yin - st_h) % d, rx = ((int)xin - st_w) % d;
      if (ry < 0) ry += d;
      if (rx < 0) rx += d;
      rot = (ry < l || rx < l) ? 0 : 1;
    }
    if (rot) {                                                   // factor 1 - rot = 0
      const long long o = (long long)n * P + i;
      if (img) { img[3 * o] = 0; img[3 * o + 1] = 0; img[3 * o + 2] = 0; }
      if (label) label[o] = 0;
    }
  }
}

extern "C" int dl3p_aug_gridmask_u8(unsigned char* img, unsigned char* label, const int* params, int N, int H, int W, void* stream) {
  DL3P_CHECK_ARG((img || label) && params && N > 0 && H > 0 && W > 0, "dl3p_aug_gridmask_u8: bad arguments");
  const long long P = (long long)H * W;
  long long gx = ceil_div_ll(P, 256 * 8);
  if (gx > 1024) gx = 1024;
  if (gx < 1) gx = 1;
  hipLaunchKernelGGL(aug_gridmask_kernel, dim3((unsigned)gx, N), dim3(256), 0, (hipStream_t)stream, img, label, params, H, W);
  DL3P_CHECK_LAUNCH("dl3p_aug_gridmask_u8");
  return DL3P_OK;
}

// random_grayscale + random_blur (common/data_utils.py:105-124, 152-172; applied in that order by deeplabv3p/data.py:95-99).  UNPINNED
// restatements of OpenCV's published 8-bit arithmetic (cv2 is not in this image: oracle/np_augment.py restates the same formulas
// and the tests hold the kernel to that restatement, not to OpenCV itself):
//   cv2.cvtColor(BGR2GRAY) on uint8:  gray = (c0 * 1868 + c1 * 9617 + c2 * 4899 + (1 << 13)) >> 14   (the reference hands an RGB array
//     to a BGR conversion: c0 is the array's first channel), then GRAY2BGR replicates it;
//   cv2.GaussianBlur(img, (5, 5), 0) on uint8: sigma <= 0 with a 5-tap kernel takes the fixed table [1 4 6 4 1] / 16, separable,
//     in 8.8 fixed point with ONE rounding at the end: out = (sum_ij w_i w_j p(reflect-101 border) + 128) >> 8.
// flags[n]: bit 0 grayscale, bit 1 blur; 0 copies the image.
__global__ __launch_bounds__(256) void aug_gray_blur_kernel(const unsigned char* img, unsigned char* out, const int* flags, int H, int W) {
  const int n = blockIdx.y;
  const int f = flags[n];
  const long long P = (long long)H * W;
  const unsigned char* src = img + (size_t)n * P * 3;
  unsigned char* dst = out + (size_t)n * P * 3;
  auto gray = [](int c0, int c1, int c2) { return (c0 * 1868 + c1 * 9617 + c2 * 4899 + (1 << 13)) >> 14; };
  auto refl = [](int i, int n_) { return i < 0 ? -i : (i >= n_ ? 2 * n_ - 2 - i : i); };      // BORDER_REFLECT_101
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < P; i += (long long)gridDim.x * 256) {
    const int y = (int)(i / W), x = (int)(i - (long long)y * W);
    int o0, o1, o2;
    if (!(f & 2)) {
      o0 = src[3 * i]; o1 = src[3 * i + 1]; o2 = src[3 * i + 2];
      if (f & 1) o0 = o1 = o2 = gray(o0, o1, o2);
    } else {
      const int wv[5] = {1, 4, 6, 4, 1};
      int s0 = 0, s1 = 0, s2 = 0;
#pragma unroll
      for (int dy = -2; dy <= 2; ++dy) {
        const int yy = H > 1 ? refl(y + dy, H) : 0;
#pragma unroll
        for (int dx = -2; dx <= 2; ++dx) {
          const int xx = W > 1 ? refl(x + dx, W) : 0;
          const unsigned char* q = src + ((long long)yy * W + xx) * 3;
          int c0 = q[0], c1 = q[1], c2 = q[2];
          if (f & 1) c0 = c1 = c2 = gray(c0, c1, c2);
          const int wgt = wv[dy + 2] * wv[dx + 2];
          s0 += wgt * c0; s1 += wgt * c1; s2 += wgt * c2;
        }
      }
      o0 = (s0 + 128) >> 8; o1 = (s1 + 128) >> 8; o2 = (s2 + 128) >> 8;
    }
    dst[3 * i] = (unsigned char)o0; dst[3 * i + 1] = (unsigned char)o1; dst[3 * i + 2] = (unsigned char)o2;
  }
}

extern "C" int dl3p_aug_gray_blur_u8(const unsigned char* img, unsigned char* out, const int* flags, int N, int H, int W, void* stream) {
  DL3P_CHECK_ARG(img && out && out != img && flags && N > 0 && H > 2 && W > 2, "dl3p_aug_gray_blur_u8: bad arguments (H, W >= 3, distinct output)");
  const long long P = (long long)H * W;
  long long gx = ceil_div_ll(P, 256 * 4);
  if (gx > 2048) gx = 2048;
  if (gx < 1) gx = 1;
  hipLaunchKernelGGL(aug_gray_blur_kernel, dim3((unsigned)gx, N), dim3(256), 0, (hipStream_t)stream, img, out, flags, H, W);
  DL3P_CHECK_LAUNCH("dl3p_aug_gray_blur_u8");
  return DL3P_OK;
}
