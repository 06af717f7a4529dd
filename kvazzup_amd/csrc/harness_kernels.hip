// kvazzup_amd/csrc/harness_kernels.hip -- what a measurement harness needs on the device and nothing of the codec: the uvgx-synth-v1
// test clip (numpy statement kvazzup_amd/synth.py, which tests/ check against the checker's C twin) generated
// where the encoder reads it, device buffers to keep it in, and a luma PSNR sum.  bench.py uses these instead of a tensor library:
// a process that loads one brings its own copy of the HIP runtime, and this library then runs on that copy instead of the system's.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/kvazzup_amd.h"

namespace {

__device__ __forceinline__ uint32_t fmix32(uint32_t h) { h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16; return h; }

// one thread per luma sample; the thread of an even (x, y) also writes the chroma pair
__global__ __launch_bounds__(256) void k_synth_frame(uint8_t *out, int kind, uint32_t seed, int w, int h, int t)
{
  const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
  if (x >= w || y >= h) return;
  uint8_t *Y = out, *U = out + (size_t)w * h, *V = U + (size_t)(w / 2) * (h / 2);
  const int cw = w / 2, ch = h / 2;
  const uint32_t tk = (uint32_t)t * 0x9E3779B1u;
  if (kind == 1) { Y[(size_t)y * w + x] = 128; if (!(x & 1) && !(y & 1)) { U[(size_t)(y / 2) * cw + x / 2] = 128; V[(size_t)(y / 2) * cw + x / 2] = 128; } return; }
  if (kind == 2) {
    const size_t i = (size_t)y * w + x;
    Y[i] = (uint8_t)(fmix32(seed ^ tk ^ ((uint32_t)i * 0x85EBCA77u)) & 255);
    if (!(x & 1) && !(y & 1)) {
      const size_t j = (size_t)(y / 2) * cw + x / 2, iu = (size_t)w * h + j, iv = iu + (size_t)cw * ch;
      U[j] = (uint8_t)(fmix32(seed ^ tk ^ ((uint32_t)iu * 0x85EBCA77u)) & 255);
      V[j] = (uint8_t)(fmix32(seed ^ tk ^ ((uint32_t)iv * 0x85EBCA77u)) & 255);
    }
    return;
  }
  const int S = h / 8;
  int v = 32 + (x * 160) / w + (y * 32) / h;
  for (int k = 0; k < 8; k++) {
    const int cx = (k * w / 8 + 5 * (k + 1) * t) % w, cy = (k * h / 8 + 3 * (k + 1) * t) % h;
    const int dx = x - cx, dy = y - cy;
    if (dx >= 0 && dx < S && dy >= 0 && dy < S) v = 64 + 16 * k + (((dx * 7) ^ (dy * 13)) & 63);
  }
  v += (int)(fmix32(seed ^ tk ^ ((uint32_t)(y * w + x) * 0x85EBCA77u)) & 7) - 3;
  Y[(size_t)y * w + x] = (uint8_t)(v < 16 ? 16 : (v > 235 ? 235 : v));
  if (!(x & 1) && !(y & 1)) {
    const int xc = x / 2, yc = y / 2;
    int u = 96 + (xc * 64) / cw; const int vv = 96 + (yc * 64) / ch;
    for (int k = 0; k < 8; k++) {
      const int cx = ((k * w / 8 + 5 * (k + 1) * t) % w) / 2, cy = ((k * h / 8 + 3 * (k + 1) * t) % h) / 2;
      const int dx = xc - cx, dy = yc - cy;
      if (dx >= 0 && dx < S / 2 && dy >= 0 && dy < S / 2) u = 96 + (xc * 64) / cw + 8 * k;
    }
    U[(size_t)yc * cw + xc] = (uint8_t)u; V[(size_t)yc * cw + xc] = (uint8_t)vv;
  }
}

// sum of squared luma differences of two pictures (a: packed, pitch w; b: pitch pb), one atomic per workgroup
__global__ __launch_bounds__(256) void k_sse(const uint8_t *a, const uint8_t *b, int w, int h, int pb, unsigned long long *out)
{
  __shared__ unsigned long long part[256];
  unsigned long long s = 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < (size_t)w * h; i += (size_t)gridDim.x * 256) {
    const int y = (int)(i / w), x = (int)(i % w);
    const int d = (int)a[i] - (int)b[(size_t)y * pb + x];
    s += (unsigned long long)(d * d);
  }
  part[threadIdx.x] = s;
  __syncthreads();
  for (int k = 128; k > 0; k >>= 1) { if ((int)threadIdx.x < k) part[threadIdx.x] += part[threadIdx.x + k]; __syncthreads(); }
  if (threadIdx.x == 0) atomicAdd(out, part[0]);
}

}  // namespace

extern "C" {

void *kvzx_harness_alloc(int device, size_t bytes)
{
  void *p = nullptr;
  if (hipSetDevice(device) != hipSuccess || hipMalloc(&p, bytes) != hipSuccess) return nullptr;
  return p;
}
void kvzx_harness_free(void *p) { if (p) hipFree(p); }
int kvzx_harness_sync(int device) { return hipSetDevice(device) == hipSuccess && hipDeviceSynchronize() == hipSuccess ? 1 : 0; }
int kvzx_harness_download(void *host, const void *dev, size_t bytes) { return hipMemcpy(host, dev, bytes, hipMemcpyDeviceToHost) == hipSuccess ? 1 : 0; }
int kvzx_harness_upload(void *dev, const void *host, size_t bytes) { return hipMemcpy(dev, host, bytes, hipMemcpyHostToDevice) == hipSuccess ? 1 : 0; }
int kvzx_harness_synth_frame(void *d_i420, int kind, uint32_t seed, int w, int h, int t)
{
  if (!d_i420 || w < 2 || h < 2 || (w & 1) || (h & 1) || kind < 0 || kind > 2) return 0;
  hipLaunchKernelGGL(k_synth_frame, dim3((w + 63) / 64, (h + 3) / 4), dim3(64, 4), 0, nullptr, (uint8_t *)d_i420, kind, seed, w, h, t);
  return hipGetLastError() == hipSuccess ? 1 : 0;
}
double kvzx_harness_luma_sse(const void *d_a, const void *d_b, int w, int h, int pitch_b)
{
  unsigned long long *d = nullptr, s = 0;
  if (hipMalloc(&d, sizeof(s)) != hipSuccess) return -1.0;
  hipMemset(d, 0, sizeof(s));
  hipLaunchKernelGGL(k_sse, dim3(1024), dim3(256), 0, nullptr, (const uint8_t *)d_a, (const uint8_t *)d_b, w, h, pitch_b, d);
  const bool ok = hipMemcpy(&s, d, sizeof(s), hipMemcpyDeviceToHost) == hipSuccess;
  hipFree(d);
  return ok ? (double)s : -1.0;
}
/* HBM peak of the device from what the runtime reports: bus width x memory clock x transfers per clock; 0 when unknown.  The HBM3E stacks of
 * gfx950 move FOUR bits per pin and reported clock (8 Gbit/s pins at the 2000 MHz hipDeviceProp_t::memoryClockRate names): 8192 bit x 2 GHz x 4 / 8
 * = 8192 GB/s, the figure MI355X_MICROARCH.md quotes as ~8 TB/s.  Older parts (HBM2/2e: two per clock) are not what this library is built for. */
double kvzx_harness_hbm_peak_gbs(int device)
{
  hipDeviceProp_t p;
  if (hipGetDeviceProperties(&p, device) != hipSuccess || p.memoryClockRate <= 0 || p.memoryBusWidth <= 0) return 0.0;
  return 4.0 * (double)p.memoryClockRate * 1e3 * ((double)p.memoryBusWidth / 8.0) / 1e9;
}
int kvzx_harness_device_info(int device, char *name, int name_cap, int *cus, int *clock_mhz, int *mem_clock_mhz, int *bus_bits)
{
  hipDeviceProp_t p;
  if (hipGetDeviceProperties(&p, device) != hipSuccess) return 0;
  if (name && name_cap > 0) { const char *src = p.name[0] ? p.name : p.gcnArchName; int i = 0; for (; i < name_cap - 1 && src[i]; i++) name[i] = src[i]; name[i] = 0; }
  if (cus) *cus = p.multiProcessorCount;
  if (clock_mhz) *clock_mhz = p.clockRate / 1000;
  if (mem_clock_mhz) *mem_clock_mhz = p.memoryClockRate / 1000;
  if (bus_bits) *bus_bits = p.memoryBusWidth;
  return 1;
}

}  // extern "C"
