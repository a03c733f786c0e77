// Device pyramid producers: what SetSourceFrame / SetTargetFrame do through OpenCV
// (phovo/include/CPhotoconsistencyOdometryAnalytic.h:115-189,466-491):
//   convertTo(fp64, 1./255), cv::resize(level 0 -> level L, INTER_LINEAR), optional GaussianBlur x2,
//   cv::Scharr dx/dy with the per-level scale.
// The arithmetic follows oracle/phovo_oracle.c operation for operation (floating-point contraction
// is switched off in this file) so that device pyramids are BIT-IDENTICAL to the oracle's; whether
// they are bit-identical to OpenCV's cannot be checked in this image (OpenCV is absent) -- callers
// that need OpenCV's exact planes can hand them over with phovo_engine_set_level_planes().
// These kernels are HBM-bound streaming kernels outside the reference's timed region
// (apps/PhotoconsistencyFrameAlignment/PhotoconsistencyFrameAlignment.cpp:94-101).

#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>

#include "phovo_internal.hpp"

#pragma clang fp contract(off)

namespace phovo_hip {

namespace {

__device__ __forceinline__ double load_px(const uint8_t *src, size_t i, double)
{
  return (double)src[i] * (1. / 255);          // convertTo(..., 1./255)  :471,484
}
__device__ __forceinline__ double load_px(const double *src, size_t i, double) { return src[i]; }
// 16-bit depth (TUM / Kinect PNG): double(u16) * scale, as `imgDepth * depthScalingFactor` does on the host
// (apps/PhotoconsistencyVisualOdometry/PhotoconsistencyVisualOdometry.cpp:208,220)
__device__ __forceinline__ double load_px(const uint16_t *src, size_t i, double scale) { return (double)src[i] * scale; }

__device__ __forceinline__ int reflect101(int p, int len)
{
  if (len == 1) return 0;
  while (p < 0 || p >= len) {
    if (p < 0) p = -p;
    else p = 2 * len - 2 - p;
  }
  return p;
}

// cv::resize by 2^-level from level 0.  level 1: 2x2 area mean (((a+b)+c)+d)*0.25;
// level >= 2: bilinear with the two central taps, weights 0.5 -- rows first, then columns.
// blockIdx.z = frame of the batch: frame f reads src + f*src_frame_stride and writes dst + f*dst_frame_stride.
template <typename SrcT>
__global__ __launch_bounds__(256) void k_resize_level(const SrcT *src_base, size_t src_frame_stride, double scl,
                                                      int w, int h, int level, int lw, int lh,
                                                      double *dst_base, size_t dst_frame_stride)
{
  const int dx = blockIdx.x * blockDim.x + threadIdx.x;
  const int dy = blockIdx.y;
  if (dx >= lw || dy >= lh) return;
  const SrcT *src = src_base + (size_t)blockIdx.z * src_frame_stride;
  double *dst = dst_base + (size_t)blockIdx.z * dst_frame_stride;
  auto at = [&](int y, int x) { return load_px(src, (size_t)y * w + x, scl); };
  double out;
  if (level == 0) {
    out = at(dy, dx);
  } else if (level == 1) {
    const int sx = dx * 2, sy = dy * 2;
    if (sx + 1 < w && sy + 1 < h) {
      const double a = at(sy, sx), b = at(sy, sx + 1);
      const double c = at(sy + 1, sx), d = at(sy + 1, sx + 1);
      out = (((a + b) + c) + d) * 0.25;
    } else {
      double sum = 0; int count = 0;
      for (int yy = 0; yy < 2; yy++) {
        if (sy + yy >= h) break;
        for (int xx = 0; xx < 2; xx++) {
          if (sx + xx >= w) break;
          sum += at(sy + yy, sx + xx);
          count++;
        }
      }
      out = count ? sum / count : 0.0;
    }
  } else {
    const int s = 1 << level, half = s / 2 - 1;
    int sy = dy * s + half; double wy1 = 0.5;
    if (sy >= h - 1) { sy = h - 1; wy1 = 0.0; }
    int sx = dx * s + half; double wx1 = 0.5;
    if (sx >= w - 1) { sx = w - 1; wx1 = 0.0; }
    double top, bot;
    if (wx1 != 0.0) top = at(sy, sx) * 0.5 + at(sy, sx + 1) * 0.5;
    else top = at(sy, sx) * 1.0;
    if (wy1 != 0.0) {
      if (wx1 != 0.0) bot = at(sy + 1, sx) * 0.5 + at(sy + 1, sx + 1) * 0.5;
      else bot = at(sy + 1, sx) * 1.0;
      out = top * 0.5 + bot * 0.5;
    } else {
      out = top * 1.0 + top * 0.0;
    }
  }
  dst[(size_t)dy * lw + dx] = out;
}

// cv::Scharr (1,0) and (0,1), scale on the smoothing kernel, BORDER_REFLECT_101  (:181-187).
// blockIdx.z = frame; the three planes of frame f sit at base + f*frame_stride + {img,gx,gy}_off.
__global__ __launch_bounds__(256) void k_scharr(const double *base, size_t frame_stride, size_t img_off,
                                                size_t gx_off, size_t gy_off, int w, int h, double scale)
{
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  const int y = blockIdx.y;
  if (x >= w || y >= h) return;
  const double *img = base + (size_t)blockIdx.z * frame_stride + img_off;
  double *gx = const_cast<double *>(base) + (size_t)blockIdx.z * frame_stride + gx_off;
  double *gy = const_cast<double *>(base) + (size_t)blockIdx.z * frame_stride + gy_off;
  const double k3 = 3.0 * scale, k10 = 10.0 * scale;
  const int xl = reflect101(x - 1, w), xr = reflect101(x + 1, w);
  const int yu = reflect101(y - 1, h), yd = reflect101(y + 1, h);
  const int rows[3] = {yu, y, yd};
  double tx[3], ty[3];
#pragma unroll
  for (int j = 0; j < 3; j++) {
    const double *row = img + (size_t)rows[j] * w;
    const double a = row[xl], b = row[x], c = row[xr];
    tx[j] = ((-1.0 * a) + (0.0 * b)) + (1.0 * c);
    ty[j] = ((k3 * a) + (k10 * b)) + (k3 * c);
  }
  gx[(size_t)y * w + x] = (k10 * tx[1]) + (k3 * (tx[2] + tx[0]));
  gy[(size_t)y * w + x] = 0.0 + 1.0 * (ty[2] - ty[0]);
}

// One separable Gaussian pass (rows, then columns) -- cv::GaussianBlur(k x k, sigma 3)  (:146-147).
__global__ __launch_bounds__(256) void k_blur_rows(const double *img, int w, int h, int ksize,
                                                   const double *kern, double *tmp)
{
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  const int y = blockIdx.y;
  if (x >= w || y >= h) return;
  const int r = ksize / 2;
  const double *row = img + (size_t)y * w;
  double s0 = kern[0] * row[reflect101(x - r, w)];
  for (int k = 1; k < ksize; k++) s0 += kern[k] * row[reflect101(x - r + k, w)];
  tmp[(size_t)y * w + x] = s0;
}

__global__ __launch_bounds__(256) void k_blur_cols(const double *tmp, int w, int h, int ksize,
                                                   const double *kern, double *img)
{
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  const int y = blockIdx.y;
  if (x >= w || y >= h) return;
  const int r = ksize / 2;
  double s0 = kern[r] * tmp[(size_t)y * w + x];
  for (int k = 1; k <= r; k++)
    s0 += kern[r + k] * (tmp[(size_t)reflect101(y + k, h) * w + x] + tmp[(size_t)reflect101(y - k, h) * w + x]);
  img[(size_t)y * w + x] = s0;
}

__global__ __launch_bounds__(256) void k_fill_i32(int *dst, size_t n, int value)
{
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) dst[i] = value;
}

// Rounding to the storage type: fp64 -> fp32 is one IEEE round-to-nearest-even; fp16 goes through fp32.
__device__ __forceinline__ void put(double *p, double v) { *p = v; }
__device__ __forceinline__ void put(float *p, double v) { *p = (float)v; }
__device__ __forceinline__ void put(__half *p, double v) { *p = __float2half_rn((float)v); }
__device__ __forceinline__ double get(const double *p) { return *p; }
__device__ __forceinline__ double get(const float *p) { return (double)*p; }
__device__ __forceinline__ double get(const __half *p) { return (double)__half2float(*p); }

template <typename T>
__global__ __launch_bounds__(256) void k_store_plane(const double *src_base, size_t src_frame_stride, int n,
                                                     unsigned char *dst_base, size_t dst_frame_bytes)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double *src = src_base + (size_t)blockIdx.z * src_frame_stride;
  T *dst = reinterpret_cast<T *>(dst_base + (size_t)blockIdx.z * dst_frame_bytes);
  put(dst + i, src[i]);
}

template <typename T>
__global__ __launch_bounds__(256) void k_load_plane(const unsigned char *src, int n, double *dst)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = get(reinterpret_cast<const T *>(src) + i);
}

// Tap records of the bilinear extension: rec[k] = {I[k], GX[k], GY[k], 0} in the storage type T of the three planes.
template <typename T>
__global__ __launch_bounds__(256) void k_build_tap_records(unsigned char *pool, size_t frame_bytes, size_t off_i, size_t off_gx,
                                                           size_t off_gy, size_t rec_off, int n)
{
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  unsigned char *frame = pool + (size_t)blockIdx.z * frame_bytes;
  const T *pi = reinterpret_cast<const T *>(frame + off_i), *pgx = reinterpret_cast<const T *>(frame + off_gx);
  const T *pgy = reinterpret_cast<const T *>(frame + off_gy);
  T *rec = reinterpret_cast<T *>(frame + rec_off) + (size_t)4 * k;
  rec[0] = pi[k];
  rec[1] = pgx[k];
  rec[2] = pgy[k];
  rec[3] = T(0);
}

inline dim3 grid3d(int w, int h, int frames)
{
  return dim3((unsigned)((w + 255) / 256), (unsigned)h, (unsigned)frames);
}

}  // namespace

hipError_t pyr_intensity_level(const uint8_t *gray, size_t src_frame_stride, int frames, int w, int h, int level,
                               int lw, int lh, double *dst, size_t dst_frame_stride, hipStream_t stream)
{
  hipLaunchKernelGGL(k_resize_level<uint8_t>, grid3d(lw, lh, frames), dim3(256), 0, stream, gray, src_frame_stride,
                     1.0, w, h, level, lw, lh, dst, dst_frame_stride);
  return hipGetLastError();
}

hipError_t pyr_depth_level(const double *depth, size_t src_frame_stride, int frames, int w, int h, int level,
                           int lw, int lh, double *dst, size_t dst_frame_stride, hipStream_t stream)
{
  hipLaunchKernelGGL(k_resize_level<double>, grid3d(lw, lh, frames), dim3(256), 0, stream, depth, src_frame_stride,
                     1.0, w, h, level, lw, lh, dst, dst_frame_stride);
  return hipGetLastError();
}

hipError_t pyr_depth_level_u16(const uint16_t *depth, size_t src_frame_stride, double scale, int frames, int w, int h,
                               int level, int lw, int lh, double *dst, size_t dst_frame_stride, hipStream_t stream)
{
  hipLaunchKernelGGL(k_resize_level<uint16_t>, grid3d(lw, lh, frames), dim3(256), 0, stream, depth, src_frame_stride,
                     scale, w, h, level, lw, lh, dst, dst_frame_stride);
  return hipGetLastError();
}

hipError_t pyr_scharr(const double *base, size_t frame_stride, size_t img_off, size_t gx_off, size_t gy_off,
                      int frames, int w, int h, double scale, hipStream_t stream)
{
  hipLaunchKernelGGL(k_scharr, grid3d(w, h, frames), dim3(256), 0, stream, base, frame_stride, img_off, gx_off,
                     gy_off, w, h, scale);
  return hipGetLastError();
}

size_t storage_elem_size(int storage, bool is_depth)
{
  if (storage == PHOVO_STORAGE_F32) return 4;
  if (storage == PHOVO_STORAGE_F16) return is_depth ? 4 : 2;
  return 8;
}

hipError_t pyr_store_plane(const double *src, size_t src_frame_stride, int frames, int n, unsigned char *dst,
                           size_t dst_frame_bytes, int storage, bool is_depth, hipStream_t stream)
{
  const dim3 grid((unsigned)((n + 255) / 256), 1, (unsigned)frames);
  const size_t es = storage_elem_size(storage, is_depth);
  if (es == 8) hipLaunchKernelGGL(k_store_plane<double>, grid, dim3(256), 0, stream, src, src_frame_stride, n, dst, dst_frame_bytes);
  else if (es == 4) hipLaunchKernelGGL(k_store_plane<float>, grid, dim3(256), 0, stream, src, src_frame_stride, n, dst, dst_frame_bytes);
  else hipLaunchKernelGGL(k_store_plane<__half>, grid, dim3(256), 0, stream, src, src_frame_stride, n, dst, dst_frame_bytes);
  return hipGetLastError();
}

hipError_t pyr_load_plane(const unsigned char *src, int n, double *dst, int storage, bool is_depth, hipStream_t stream)
{
  const dim3 grid((unsigned)((n + 255) / 256));
  const size_t es = storage_elem_size(storage, is_depth);
  if (es == 8) hipLaunchKernelGGL(k_load_plane<double>, grid, dim3(256), 0, stream, src, n, dst);
  else if (es == 4) hipLaunchKernelGGL(k_load_plane<float>, grid, dim3(256), 0, stream, src, n, dst);
  else hipLaunchKernelGGL(k_load_plane<__half>, grid, dim3(256), 0, stream, src, n, dst);
  return hipGetLastError();
}

hipError_t pyr_build_tap_records(unsigned char *pool, size_t frame_bytes, const size_t plane_off[PLANES_PER_FRAME],
                                 size_t rec_off, int frames, int n, int storage, hipStream_t stream)
{
  const dim3 grid((unsigned)((n + 255) / 256), 1, (unsigned)frames);
  const size_t oi = plane_off[PLANE_I], ogx = plane_off[PLANE_GX], ogy = plane_off[PLANE_GY];
  if (storage == PHOVO_STORAGE_F64) hipLaunchKernelGGL(k_build_tap_records<double>, grid, dim3(256), 0, stream, pool, frame_bytes, oi, ogx, ogy, rec_off, n);
  else if (storage == PHOVO_STORAGE_F32) hipLaunchKernelGGL(k_build_tap_records<float>, grid, dim3(256), 0, stream, pool, frame_bytes, oi, ogx, ogy, rec_off, n);
  else hipLaunchKernelGGL(k_build_tap_records<__half>, grid, dim3(256), 0, stream, pool, frame_bytes, oi, ogx, ogy, rec_off, n);
  return hipGetLastError();
}

hipError_t pyr_gaussian_blur(double *img, double *tmp, int w, int h, int ksize,
                             const double *d_kernel, hipStream_t stream)
{
  hipLaunchKernelGGL(k_blur_rows, grid3d(w, h, 1), dim3(256), 0, stream, img, w, h, ksize, d_kernel, tmp);
  hipLaunchKernelGGL(k_blur_cols, grid3d(w, h, 1), dim3(256), 0, stream, tmp, w, h, ksize, d_kernel, img);
  return hipGetLastError();
}

hipError_t fill_i32(int *dst, size_t n, int value, hipStream_t stream)
{
  if (n == 0) return hipSuccess;
  size_t blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(k_fill_i32, dim3((unsigned)blocks), dim3(256), 0, stream, dst, n, value);
  return hipGetLastError();
}

}  // namespace phovo_hip
