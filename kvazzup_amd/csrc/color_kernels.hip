// kvazzup_amd/csrc/color_kernels.hip -- row f1 of SURVEY.md 8: I420 -> RGB32 (and, further down, RGB32 -> I420), the step uvgComm runs on every decoded
// picture before display (YUVtoRGB32::process, /root/reference/src/media/processing/yuvtorgb32.cpp:29-64, which
// calls yuv420_to_rgb_i_{avx2_mt,avx2,sse41,c}, yuvconversions.cpp:72-493).
//
// The reference has TWO arithmetics, and the filter picks by CPU features and width:
//   * the SSE4.1 / AVX2 converters (width % 16 == 0 on any x86-64 of the last decade), yuvconversions.cpp:72-420:
//       u = U - 128, v = V - 128
//       byte 2 = clamp(Y + v + (v>>2) + (v>>3) + (v>>5))
//       byte 1 = clamp(Y - ((u>>2) + (u>>4) + (u>>5) + (v>>1) + (v>>3) + (v>>4) + (v>>5)))
//       byte 0 = clamp(Y + u + (u>>1) + (u>>2) + (u>>6))          byte 3 = 0
//   * the scalar fallback yuv420_to_rgb_i_c, :423-493, which reads the planes the other way round and never
//     writes byte 3:
//       cr = U - 128, cb = V - 128
//       byte 0 = clamp(Y + cr + (cr>>2) + (cr>>3) + (cr>>5))
//       byte 1 = clamp(Y - ((cb>>2) + (cb>>4) + (cb>>5)) - ((cr>>1) + (cr>>3) + (cr>>4) + (cr>>5)))
//       byte 2 = clamp(Y + cb + (cb>>1) + (cb>>2) + (cb>>6))      byte 3 untouched
// Both are reproduced bit for bit (tests/test_gpu_color.py runs the reference's own object code beside this).
// Pure streaming: 1.5 bytes in, 4 out per sample -- the one kernel on the path that is HBM bound.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/kvazzup_amd.h"

namespace kvzx {

__device__ __forceinline__ int clamp255(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }

__device__ __forceinline__ uint32_t pixel_simd(int y, int u, int v)
{
  const int r = y + v + (v >> 2) + (v >> 3) + (v >> 5);
  const int g = y - ((u >> 2) + (u >> 4) + (u >> 5) + (v >> 1) + (v >> 3) + (v >> 4) + (v >> 5));
  const int b = y + u + (u >> 1) + (u >> 2) + (u >> 6);
  return (uint32_t)clamp255(b) | ((uint32_t)clamp255(g) << 8) | ((uint32_t)clamp255(r) << 16);
}

// width % 8 == 0: a thread converts 8 x 2 samples (two 8-byte luma loads, one 4-byte load per chroma plane, four
// 16-byte stores)
__global__ __launch_bounds__(256) void k_i420_to_rgb32_simd(const uint8_t *py, const uint8_t *pu, const uint8_t *pv, int ypitch, int cpitch,
                                                            uint8_t *out, int w, int h)
{
  const int tx = blockIdx.x * blockDim.x + threadIdx.x, ty = blockIdx.y;
  if (tx * 8 >= w || ty * 2 >= h) return;
  const uint32_t u4 = *(const uint32_t *)(pu + (size_t)ty * cpitch + tx * 4), v4 = *(const uint32_t *)(pv + (size_t)ty * cpitch + tx * 4);
#pragma unroll
  for (int r = 0; r < 2; r++) {
    const uint2 y8 = *(const uint2 *)(py + (size_t)(ty * 2 + r) * ypitch + tx * 8);
    uint32_t px[8];
#pragma unroll
    for (int i = 0; i < 8; i++) {
      const int y = (int)(((i < 4 ? y8.x : y8.y) >> (8 * (i & 3))) & 255u);
      const int u = (int)((u4 >> (8 * (i >> 1))) & 255u) - 128, v = (int)((v4 >> (8 * (i >> 1))) & 255u) - 128;
      px[i] = pixel_simd(y, u, v);
    }
    uint4 *o = (uint4 *)(out + ((size_t)(ty * 2 + r) * w + tx * 8) * 4);
    o[0] = make_uint4(px[0], px[1], px[2], px[3]);
    o[1] = make_uint4(px[4], px[5], px[6], px[7]);
  }
}

// any even size: a thread converts 2 x 2 samples with the scalar fallback's arithmetic; byte 3 is not written
__global__ __launch_bounds__(256) void k_i420_to_rgb32_c(const uint8_t *py, const uint8_t *pu, const uint8_t *pv, int ypitch, int cpitch,
                                                         uint8_t *out, int w, int h)
{
  const int cx = blockIdx.x * blockDim.x + threadIdx.x, cy = blockIdx.y;
  if (cx * 2 >= w || cy * 2 >= h) return;
  const int cr = (int)pu[(size_t)cy * cpitch + cx] - 128, cb = (int)pv[(size_t)cy * cpitch + cx] - 128;
  const int t0 = cr + (cr >> 2) + (cr >> 3) + (cr >> 5);
  const int t1 = -((cb >> 2) + (cb >> 4) + (cb >> 5)) - ((cr >> 1) + (cr >> 3) + (cr >> 4) + (cr >> 5));
  const int t2 = cb + (cb >> 1) + (cb >> 2) + (cb >> 6);
  for (int r = 0; r < 2; r++)
    for (int i = 0; i < 2; i++) {
      const int x = cx * 2 + i, yy = cy * 2 + r, y = py[(size_t)yy * ypitch + x];
      uint8_t *o = out + ((size_t)yy * w + x) * 4;
      o[0] = (uint8_t)clamp255(y + t0); o[1] = (uint8_t)clamp255(y + t1); o[2] = (uint8_t)clamp255(y + t2);
    }
}

// variant: 0 = the filter's choice (SIMD arithmetic when width % 16 == 0), 1 = scalar, 2 = SIMD (needs width % 8 == 0)
int convert_i420_to_rgb32(const uint8_t *y, const uint8_t *u, const uint8_t *v, int ypitch, int cpitch, uint8_t *rgb, int w, int h, int variant, hipStream_t st)
{
  if (!y || !u || !v || !rgb || w < 2 || h < 2 || (w & 1) || (h & 1)) return 0;
  const bool simd = variant == 2 || (variant == 0 && (w % 16) == 0);
  if (simd) {
    if ((w % 8) || (ypitch % 8) || (cpitch % 4)) return 0;
    hipLaunchKernelGGL(k_i420_to_rgb32_simd, dim3((w / 8 + 255) / 256, h / 2), dim3(256), 0, st, y, u, v, ypitch, cpitch, rgb, w, h);
  } else {
    hipLaunchKernelGGL(k_i420_to_rgb32_c, dim3((w / 2 + 255) / 256, h / 2), dim3(256), 0, st, y, u, v, ypitch, cpitch, rgb, w, h);
  }
  return hipGetLastError() == hipSuccess ? 1 : 0;
}


// ---------------------------------------------------------------------------------------------------------------
// RGB32 -> I420, the other half of row f1 (rgb_to_yuv420_i_c / rgb_to_yuv420_i_sse41, yuvconversions.cpp:496-797).  Again the
// reference holds two different arithmetics; both are reproduced bit for bit, quirks included:
//   * rgb_to_yuv420_i_c (:770-797): Y = (76 * byte0 + 150 * byte1 + 29 * byte2 + 128) >> 8 -- the R weight on byte 0 -- while the
//     chroma takes byte 0 as BLUE: U = ((sum over the 2x2 block of 127 * byte0 - 84 * byte1 - 43 * byte2) + 512 >> 10) + 128,
//     V = ((sum of -21 * byte0 - 106 * byte1 + 127 * byte2) + 512 >> 10) + 128, stored modulo 256, rows top to bottom;
//   * rgb_to_yuv420_i_sse41 (:634-767): b = byte0, g = byte1, r = byte2; Y = clamp((76 r + 150 g + 29 b) >> 8), no rounding;
//     per column of a 2x2 block t = clamp(((-43 rs - 84 gs + 127 bs) + 255 * 255) >> 9) with the sums over the two rows, then
//     U = (t_left + t_right) >> 1 (V likewise with 127, -106, -21); and the picture comes out UPSIDE DOWN (row r -> row h - 1 - r in
//     all three planes).  Needs width % 4 == 0.
// A thread converts 4 x 2 pixels: two 16-byte loads, two 4-byte luma stores, one 2-byte store per chroma plane.
// ---------------------------------------------------------------------------------------------------------------
template <bool SSE>
__global__ __launch_bounds__(256) void k_rgb32_to_i420(const uint8_t *in, uint8_t *out, int w, int h)
{
  const int tx = blockIdx.x * blockDim.x + threadIdx.x, ty = blockIdx.y;
  if (tx * 4 >= w || ty * 2 >= h) return;
  const uint4 p0 = *(const uint4 *)(in + ((size_t)(2 * ty) * w + tx * 4) * 4), p1 = *(const uint4 *)(in + ((size_t)(2 * ty + 1) * w + tx * 4) * 4);
  const uint32_t px[2][4] = {{p0.x, p0.y, p0.z, p0.w}, {p1.x, p1.y, p1.z, p1.w}};
  uint8_t *oy = out, *ou = out + (size_t)w * h, *ov = ou + (size_t)(w / 2) * (h / 2);
  int us[4], vs[4];
#pragma unroll
  for (int i = 0; i < 4; i++) { us[i] = 0; vs[i] = 0; }
#pragma unroll
  for (int r = 0; r < 2; r++) {
    uint32_t y4 = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int b0 = px[r][i] & 255, b1 = (px[r][i] >> 8) & 255, b2 = (px[r][i] >> 16) & 255;
      if (SSE) {
        y4 |= (uint32_t)clamp255((76 * b2 + 150 * b1 + 29 * b0) >> 8) << (8 * i);
        us[i] += -43 * b2 - 84 * b1 + 127 * b0; vs[i] += 127 * b2 - 106 * b1 - 21 * b0;
      } else {
        y4 |= (uint32_t)(((76 * b0 + 150 * b1 + 29 * b2 + 128) >> 8) & 255) << (8 * i);
        us[i] += 127 * b0 - 84 * b1 - 43 * b2; vs[i] += -21 * b0 - 106 * b1 + 127 * b2;
      }
    }
    const int orow = SSE ? h - 1 - (2 * ty + r) : 2 * ty + r;
    *(uint32_t *)(oy + (size_t)orow * w + tx * 4) = y4;
  }
  int u0, u1, v0, v1;
  if (SSE) {
    int t[4], q[4];
#pragma unroll
    for (int i = 0; i < 4; i++) { t[i] = clamp255((us[i] + 255 * 255) >> 9); q[i] = clamp255((vs[i] + 255 * 255) >> 9); }
    u0 = (t[0] + t[1]) >> 1; u1 = (t[2] + t[3]) >> 1; v0 = (q[0] + q[1]) >> 1; v1 = (q[2] + q[3]) >> 1;
  } else {
    u0 = ((us[0] + us[1] + 512) >> 10) + 128; u1 = ((us[2] + us[3] + 512) >> 10) + 128;
    v0 = ((vs[0] + vs[1] + 512) >> 10) + 128; v1 = ((vs[2] + vs[3] + 512) >> 10) + 128;
  }
  const int crow = SSE ? h / 2 - 1 - ty : ty;
  *(uint16_t *)(ou + (size_t)crow * (w / 2) + tx * 2) = (uint16_t)((u0 & 255) | ((u1 & 255) << 8));
  *(uint16_t *)(ov + (size_t)crow * (w / 2) + tx * 2) = (uint16_t)((v0 & 255) | ((v1 & 255) << 8));
}

// variant: 1 = rgb_to_yuv420_i_c, 2 = rgb_to_yuv420_i_sse41 (vertical flip included); width % 4 == 0, height % 2 == 0
int convert_rgb32_to_i420(const uint8_t *rgb, uint8_t *i420, int w, int h, int variant, hipStream_t st)
{
  if (!rgb || !i420 || w < 4 || h < 2 || (w & 3) || (h & 1) || (variant != 1 && variant != 2)) return 0;
  const dim3 g((w / 4 + 255) / 256, h / 2);
  if (variant == 2) hipLaunchKernelGGL(k_rgb32_to_i420<true>, g, dim3(256), 0, st, rgb, i420, w, h);
  else hipLaunchKernelGGL(k_rgb32_to_i420<false>, g, dim3(256), 0, st, rgb, i420, w, h);
  return hipGetLastError() == hipSuccess ? 1 : 0;
}

}  // namespace kvzx

extern "C" {

KVZ_PUBLIC int kvzx_yuv420_to_rgb32_device(const void *d_y, const void *d_u, const void *d_v, int y_pitch, int c_pitch, void *d_rgb32,
                                            int width, int height, int variant, void *hip_stream)
{
  return kvzx::convert_i420_to_rgb32((const uint8_t *)d_y, (const uint8_t *)d_u, (const uint8_t *)d_v, y_pitch, c_pitch, (uint8_t *)d_rgb32,
                                     width, height, variant, (hipStream_t)hip_stream);
}

// Host buffers in and out (the signature shape of yuv420_to_rgb_i_*): upload, convert, download.  With the scalar
// arithmetic the output buffer is uploaded first, because that converter leaves every fourth byte as it found it.
KVZ_PUBLIC int kvzx_yuv420_to_rgb32(const uint8_t *i420, uint8_t *rgb32, int width, int height, int variant)
{
  if (!i420 || !rgb32 || width < 2 || height < 2) return 0;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return 0;            // no CPU fallback
  const size_t ny = (size_t)width * height, nin = ny * 3 / 2, nout = ny * 4;
  uint8_t *din = nullptr, *dout = nullptr;
  if (hipMalloc(&din, nin) != hipSuccess) return 0;
  if (hipMalloc(&dout, nout) != hipSuccess) { hipFree(din); return 0; }
  const bool simd = variant == 2 || (variant == 0 && (width % 16) == 0);
  int ok = hipMemcpy(din, i420, nin, hipMemcpyHostToDevice) == hipSuccess;
  if (ok && !simd) ok = hipMemcpy(dout, rgb32, nout, hipMemcpyHostToDevice) == hipSuccess;
  if (ok) ok = kvzx::convert_i420_to_rgb32(din, din + ny, din + ny + ny / 4, width, width / 2, dout, width, height, variant, nullptr);
  if (ok) ok = hipMemcpy(rgb32, dout, nout, hipMemcpyDeviceToHost) == hipSuccess;
  hipFree(din); hipFree(dout);
  return ok;
}

KVZ_PUBLIC int kvzx_rgb32_to_yuv420_device(const void *d_rgb32, void *d_i420, int width, int height, int variant, void *hip_stream)
{
  return kvzx::convert_rgb32_to_i420((const uint8_t *)d_rgb32, (uint8_t *)d_i420, width, height, variant, (hipStream_t)hip_stream);
}

// host buffers in and out, like rgb_to_yuv420_i_*(input, output, width, height): upload, convert, download
KVZ_PUBLIC int kvzx_rgb32_to_yuv420(const uint8_t *rgb32, uint8_t *i420, int width, int height, int variant)
{
  if (!rgb32 || !i420 || width < 4 || height < 2) return 0;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return 0;            // no CPU fallback
  const size_t ny = (size_t)width * height, nin = ny * 4, nout = ny * 3 / 2;
  uint8_t *din = nullptr, *dout = nullptr;
  if (hipMalloc(&din, nin) != hipSuccess) return 0;
  if (hipMalloc(&dout, nout) != hipSuccess) { hipFree(din); return 0; }
  int ok = hipMemcpy(din, rgb32, nin, hipMemcpyHostToDevice) == hipSuccess;
  if (ok) ok = kvzx::convert_rgb32_to_i420(din, dout, width, height, variant, nullptr);
  if (ok) ok = hipMemcpy(i420, dout, nout, hipMemcpyDeviceToHost) == hipSuccess;
  hipFree(din); hipFree(dout);
  return ok;
}

}  // extern "C"
