// PCIe copy microbenchmark: copy engine (hipMemcpyAsync) vs. kernels that read / write page-locked host memory, one direction and both at once.
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
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
  for (size_t bytes : {(size_t)3110400, (size_t)12441600}) {
    uint8_t *d_a, *d_b, *h_a, *h_b, *hd_a, *hd_b;
    CK(hipMalloc(&d_a, bytes)); CK(hipMalloc(&d_b, bytes));
    CK(hipHostMalloc(&h_a, bytes, hipHostMallocMapped)); CK(hipHostMalloc(&h_b, bytes, hipHostMallocMapped));
    CK(hipHostGetDevicePointer((void **)&hd_a, h_a, 0)); CK(hipHostGetDevicePointer((void **)&hd_b, h_b, 0));
    memset(h_a, 1, bytes); memset(h_b, 2, bytes);
    hipStream_t s1, s2; CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    const int R = 50; const size_t n16 = bytes / 16;
    auto run = [&](const char *name, auto up, auto down) {
      for (int i = 0; i < 3; i++) { up(); down(); }
      CK(hipDeviceSynchronize());
      double t0 = now();
      for (int i = 0; i < R; i++) { up(); down(); }
      CK(hipDeviceSynchronize());
      double dt = (now() - t0) / R;
      printf("%9zu  %-44s %7.1f us per pair\n", bytes, name, dt * 1e6);
    };
    auto none = [] {};
    auto sd_up = [&] { CK(hipMemcpyAsync(d_a, h_a, bytes, hipMemcpyHostToDevice, s1)); };
    auto sd_dn = [&] { CK(hipMemcpyAsync(h_b, d_b, bytes, hipMemcpyDeviceToHost, s2)); };
    auto sd_dn_s1 = [&] { CK(hipMemcpyAsync(h_b, d_b, bytes, hipMemcpyDeviceToHost, s1)); };
    for (int blocks : {64, 256, 1024}) {
      auto k_up = [&] { hipLaunchKernelGGL(k_copy, dim3(blocks), dim3(256), 0, s1, (const uint4 *)hd_a, (uint4 *)d_a, n16); };
      auto k_dn = [&] { hipLaunchKernelGGL(k_copy, dim3(blocks), dim3(256), 0, s2, (const uint4 *)d_b, (uint4 *)hd_b, n16); };
      char nm[96];
      snprintf(nm, sizeof nm, "kernel up only (%d blocks)", blocks); run(nm, k_up, none);
      snprintf(nm, sizeof nm, "kernel down only (%d blocks)", blocks); run(nm, none, k_dn);
      snprintf(nm, sizeof nm, "kernel up + kernel down (%d blocks)", blocks); run(nm, k_up, k_dn);
      snprintf(nm, sizeof nm, "kernel up + engine down (%d blocks)", blocks); run(nm, k_up, sd_dn);
      snprintf(nm, sizeof nm, "engine up + kernel down (%d blocks)", blocks); run(nm, sd_up, k_dn);
    }
    run("engine up only", sd_up, none);
    run("engine down only", none, sd_dn);
    run("engine up + engine down, two streams", sd_up, sd_dn);
    run("engine up + engine down, one stream", sd_up, sd_dn_s1);
  }
  return 0;
}
