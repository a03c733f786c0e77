// phovo_warp_image: the display-side forward warp that both reference apps call right after Optimize()
// (free function phovo::warpImage, phovo/include/CPhotoconsistencyOdometry.h:73-134; callers
// apps/PhotoconsistencyFrameAlignment/PhotoconsistencyFrameAlignment.cpp:108 and
// apps/PhotoconsistencyVisualOdometry/PhotoconsistencyVisualOdometry.cpp:248-250).
//
// The reference walks the source image in raster order and overwrites warped(tr, tc), so the LARGEST source index
// that lands on a target pixel is the one that stays.  Same resolution as in the alignment kernels: pass 1 takes
// atomicMax(owner[target], source index), pass 2 copies intensity[owner].  Semantics kept: depth > 0 gate (:107, not the
// min/max-depth gate of the alignment), static_cast<int> = truncation toward zero (:119-122, not round()), a true
// IEEE division (:119), zeros where nothing lands (:98).  Compiled with -ffp-contract=off so that every product and sum
// rounds where the reference's (and the oracle's) does; the comparison with the oracle is bit-exact on the u8 output.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <string>

#include "phovo_hip.h"
#include "phovo_internal.hpp"

namespace phovo_hip {
namespace {

struct WarpArgs {
  const uint8_t *intensity;   // [h][w] tightly packed on the device
  const double *depth;        // [h][w]
  int *owner;                 // [h*w], -1 = nobody landed here
  uint8_t *warped;            // [h][w]
  int w, h;
  double fx, fy, ox, oy, inv_fx, inv_fy;
  double rt[12];              // rows 0..2 of Rt
};

__global__ __launch_bounds__(256) void k_warp_scatter(const WarpArgs A)
{
  const int n = A.w * A.h;
  // consecutive lanes take consecutive pixels: coalesced 8-byte depth reads
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const double dz = A.depth[i];
    if (!(dz > 0)) continue;                                            // :107
    const int r = i / A.w, c = i - r * A.w;
    const double px = (c - A.ox) * dz * A.inv_fx;                       // :111
    const double py = (r - A.oy) * dz * A.inv_fy;                       // :112
    double tp[3];
#pragma unroll
    for (int a = 0; a < 3; a++)                                         // Rt * point3D  :116
      tp[a] = ((A.rt[4 * a + 0] * px + A.rt[4 * a + 1] * py) + A.rt[4 * a + 2] * dz) + A.rt[4 * a + 3] * 1.0;
    const double tcd = ((tp[0] * A.fx) / tp[2]) + A.ox;                 // :119-120
    const double trd = ((tp[1] * A.fy) / tp[2]) + A.oy;                 // :121-122
    // non-finite or beyond int: undefined in the reference's cast, treated as out of bounds here and in the oracle
    if (!(tcd > -2147483648.0 && tcd < 2147483647.0 && trd > -2147483648.0 && trd < 2147483647.0)) continue;
    const int tc = (int)tcd, tr = (int)trd;                             // truncation toward zero
    if (tr >= 0 && tr < A.h && tc >= 0 && tc < A.w) atomicMax(&A.owner[tr * A.w + tc], i);   // last raster writer  :128
  }
}

__global__ __launch_bounds__(256) void k_warp_gather(const WarpArgs A)
{
  const int n = A.w * A.h;
  for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x) {
    const int o = A.owner[k];
    A.warped[k] = o >= 0 ? A.intensity[o] : (uint8_t)0;                 // zeros where nothing landed  :98
  }
}

struct DeviceBuffers {      // freed on every exit path
  void *p[4] = {nullptr, nullptr, nullptr, nullptr};
  hipStream_t stream = nullptr;
  ~DeviceBuffers()
  {
    for (void *q : p)
      if (q) (void)hipFree(q);
    if (stream) (void)hipStreamDestroy(stream);
  }
};

#define PHOVO_WARP_CHECK(expr)                                                                  \
  do {                                                                                          \
    hipError_t _e = (expr);                                                                     \
    if (_e != hipSuccess)                                                                       \
      return fail(PHOVO_E_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));              \
  } while (0)

}  // namespace
}  // namespace phovo_hip

using namespace phovo_hip;

extern "C" int phovo_warp_image(int device, const uint8_t *intensity, size_t intensity_stride_bytes,
                                const double *depth, size_t depth_stride_bytes, int w, int h, const double rt[16],
                                const double k[9], int level, uint8_t *warped, size_t warped_stride_bytes)
{
  if (!intensity || !depth || !rt || !k || !warped) return fail(PHOVO_E_INVALID_ARGUMENT, "phovo_warp_image: null argument");
  if (w <= 0 || h <= 0 || (long long)w * h > (1ll << 30)) return fail(PHOVO_E_SHAPE, "phovo_warp_image: bad image size");
  if (level < 0 || level >= PHOVO_MAX_LEVELS) return fail(PHOVO_E_INVALID_ARGUMENT, "phovo_warp_image: bad level");
  if (intensity_stride_bytes < (size_t)w || depth_stride_bytes < (size_t)w * sizeof(double) ||
      warped_stride_bytes < (size_t)w)
    return fail(PHOVO_E_SHAPE, "phovo_warp_image: stride shorter than a row");
  PHOVO_WARP_CHECK(hipSetDevice(device));

  const size_t n = (size_t)w * (size_t)h;
  DeviceBuffers b;
  PHOVO_WARP_CHECK(hipStreamCreateWithFlags(&b.stream, hipStreamNonBlocking));
  PHOVO_WARP_CHECK(hipMalloc(&b.p[0], n));
  PHOVO_WARP_CHECK(hipMalloc(&b.p[1], n * sizeof(double)));
  PHOVO_WARP_CHECK(hipMalloc(&b.p[2], n * sizeof(int)));
  PHOVO_WARP_CHECK(hipMalloc(&b.p[3], n));

  WarpArgs a{};
  a.intensity = static_cast<const uint8_t *>(b.p[0]);
  a.depth = static_cast<const double *>(b.p[1]);
  a.owner = static_cast<int *>(b.p[2]);
  a.warped = static_cast<uint8_t *>(b.p[3]);
  a.w = w;
  a.h = h;
  const double s = std::pow(2, level);          // :88-93 (1.f/fx is exact in double: 1.f == 1.0)
  a.fx = k[0] / s;
  a.fy = k[4] / s;
  a.inv_fx = 1.f / a.fx;
  a.inv_fy = 1.f / a.fy;
  a.ox = k[2] / s;
  a.oy = k[5] / s;
  for (int i = 0; i < 12; i++) a.rt[i] = rt[i];

  PHOVO_WARP_CHECK(hipMemcpy2DAsync(b.p[0], (size_t)w, intensity, intensity_stride_bytes, (size_t)w, (size_t)h,
                                    hipMemcpyHostToDevice, b.stream));
  PHOVO_WARP_CHECK(hipMemcpy2DAsync(b.p[1], (size_t)w * sizeof(double), depth, depth_stride_bytes,
                                    (size_t)w * sizeof(double), (size_t)h, hipMemcpyHostToDevice, b.stream));
  PHOVO_WARP_CHECK(hipMemsetAsync(b.p[2], 0xFF, n * sizeof(int), b.stream));      // every owner = -1
  const unsigned blocks = (unsigned)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
  hipLaunchKernelGGL(k_warp_scatter, dim3(blocks), dim3(256), 0, b.stream, a);
  PHOVO_WARP_CHECK(hipGetLastError());
  hipLaunchKernelGGL(k_warp_gather, dim3(blocks), dim3(256), 0, b.stream, a);
  PHOVO_WARP_CHECK(hipGetLastError());
  PHOVO_WARP_CHECK(hipMemcpy2DAsync(warped, warped_stride_bytes, b.p[3], (size_t)w, (size_t)w, (size_t)h,
                                    hipMemcpyDeviceToHost, b.stream));
  PHOVO_WARP_CHECK(hipStreamSynchronize(b.stream));
  return PHOVO_OK;
}
