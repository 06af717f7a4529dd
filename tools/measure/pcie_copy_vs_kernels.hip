// does a PCIe-bound copy slow down kernels running beside it?  victim: an HBM-streaming kernel and an ALU/LDS-ish kernel on stream s1;
// aggressor on s2: copy kernels (host-mapped reads / writes) with few or many workgroups, or the copy engine.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
__global__ void k_copy(const uint4 *__restrict__ src, uint4 *__restrict__ dst, size_t n)
{
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
__global__ void k_alu(const uint32_t *in, uint32_t *out, int iters)
{
  __shared__ uint32_t lds[1024];
  uint32_t v = in[blockIdx.x * 256 + threadIdx.x];
  for (int i = 0; i < iters; i++) { lds[threadIdx.x * 4 % 1024] = v; __syncthreads(); v = v * 1664525u + lds[(threadIdx.x * 7 + i) % 1024]; __syncthreads(); }
  out[blockIdx.x * 256 + threadIdx.x] = v;
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
  const size_t bytes = 3110400, vb = 32 << 20;
  uint8_t *d_a, *d_b, *h_a, *h_b, *hd_a, *hd_b, *v_a, *v_b;
  CK(hipMalloc(&d_a, bytes)); CK(hipMalloc(&d_b, bytes)); CK(hipMalloc(&v_a, vb)); CK(hipMalloc(&v_b, vb));
  CK(hipHostMalloc(&h_a, bytes, hipHostMallocMapped)); CK(hipHostMalloc(&h_b, bytes, hipHostMallocMapped));
  CK(hipHostGetDevicePointer((void **)&hd_a, h_a, 0)); CK(hipHostGetDevicePointer((void **)&hd_b, h_b, 0));
  CK(hipMemset(v_a, 1, vb));
  hipStream_t s1, s2, s3; CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s3, hipStreamNonBlocking));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const size_t n16 = bytes / 16;
  auto victim_hbm = [&] { hipLaunchKernelGGL(k_copy, dim3(2048), dim3(256), 0, s1, (const uint4 *)v_a, (uint4 *)v_b, vb / 16); };
  auto victim_alu = [&] { hipLaunchKernelGGL(k_alu, dim3(2048), dim3(256), 0, s1, (const uint32_t *)v_a, (uint32_t *)v_b, 200); };
  auto measure = [&](const char *name, auto victim, auto aggressor) {
    // aggressor copies keep running on s2 / s3 while 40 victims run on s1
    for (int i = 0; i < 3; i++) victim();
    CK(hipDeviceSynchronize());
    for (int i = 0; i < 200; i++) aggressor();
    CK(hipEventRecord(e0, s1)); for (int i = 0; i < 40; i++) victim(); CK(hipEventRecord(e1, s1));
    CK(hipEventSynchronize(e1)); float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    double t0 = now(); CK(hipDeviceSynchronize()); double rest = now() - t0;
    printf("%-58s victim %7.1f us each   (aggressors still running for %.1f ms after)\n", name, ms * 1e3 / 40, rest * 1e3);
  };
  auto none = [] {};
  for (int v = 0; v < 2; v++) {
    auto vic = [&] { if (v == 0) victim_hbm(); else victim_alu(); };
    const char *vn = v == 0 ? "HBM copy 32 MB" : "ALU+LDS kernel";
    char nm[128];
    snprintf(nm, sizeof nm, "%s alone", vn); measure(nm, vic, none);
    for (int blocks : {16, 64, 272, 1024}) {
      snprintf(nm, sizeof nm, "%s + kernel D2H (%d blocks)", vn, blocks);
      measure(nm, vic, [&] { hipLaunchKernelGGL(k_copy, dim3(blocks), dim3(256), 0, s2, (const uint4 *)d_b, (uint4 *)hd_b, n16); });
      snprintf(nm, sizeof nm, "%s + kernel H2D (%d blocks)", vn, blocks);
      measure(nm, vic, [&] { hipLaunchKernelGGL(k_copy, dim3(blocks), dim3(256), 0, s2, (const uint4 *)hd_a, (uint4 *)d_a, n16); });
      snprintf(nm, sizeof nm, "%s + kernel H2D and D2H (%d blocks)", vn, blocks);
      measure(nm, vic, [&] { hipLaunchKernelGGL(k_copy, dim3(blocks), dim3(256), 0, s2, (const uint4 *)hd_a, (uint4 *)d_a, n16); hipLaunchKernelGGL(k_copy, dim3(blocks), dim3(256), 0, s3, (const uint4 *)d_b, (uint4 *)hd_b, n16); });
    }
    snprintf(nm, sizeof nm, "%s + engine D2H", vn); measure(nm, vic, [&] { CK(hipMemcpyAsync(h_b, d_b, bytes, hipMemcpyDeviceToHost, s2)); });
    snprintf(nm, sizeof nm, "%s + engine H2D", vn); measure(nm, vic, [&] { CK(hipMemcpyAsync(d_a, h_a, bytes, hipMemcpyHostToDevice, s2)); });
    snprintf(nm, sizeof nm, "%s + engine H2D and D2H", vn); measure(nm, vic, [&] { CK(hipMemcpyAsync(d_a, h_a, bytes, hipMemcpyHostToDevice, s2)); CK(hipMemcpyAsync(h_b, d_b, bytes, hipMemcpyDeviceToHost, s3)); });
  }
  return 0;
}
