#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ void k(uint64_t *out, uint64_t a, uint32_t b, int n) {
  uint64_t s[8]; uint32_t t[8];
  for (int i = 0; i < 8; i++) { s[i] = threadIdx.x + i; t[i] = threadIdx.x * 3 + i; }
  for (int i = 0; i < n; i++) {
#pragma unroll
    for (int j = 0; j < 8; j++) {
      if (MODE == 0) s[j] = __builtin_amdgcn_qsad_pk_u16_u8(a, b, s[j]);
      if (MODE == 1) t[j] = __builtin_amdgcn_sad_u8((uint32_t)a, b, t[j]);
      if (MODE == 2) t[j] = __builtin_amdgcn_sad_u8(__builtin_amdgcn_alignbyte((uint32_t)(a >> 32), (uint32_t)a, j & 3), b, t[j]);
      if (MODE == 3) s[j] = __builtin_amdgcn_mqsad_pk_u16_u8(a, b, s[j]);
    }
  }
  uint64_t r = 0; for (int i = 0; i < 8; i++) r += s[i] + t[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
template <int MODE> void run(const char *name, void *o, int waves_per_simd) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  int n = 20000; float ms;
  for (int rep = 0; rep < 2; rep++) {
    hipEventRecord(e0); hipLaunchKernelGGL(k<MODE>, dim3(256 * waves_per_simd), dim3(256), 0, 0, (uint64_t *)o, 0x123456789abcdef0ull, 0x9abcdef0u, n); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
  }
  printf("%-22s %d waves/SIMD: %.3f ms -> %.2f cycles per wave-instr per SIMD (2.4 GHz)\n", name, waves_per_simd, ms, ms * 1e-3 * 2.4e9 / ((double)waves_per_simd * n * 8));
}
int main() {
  void *o; hipMalloc(&o, 1 << 26);
  for (int w : {1, 4}) { run<0>("qsad_pk_u16_u8", o, w); run<1>("sad_u8", o, w); run<2>("alignbyte+sad_u8", o, w); run<3>("mqsad_pk_u16_u8", o, w); }
  return 0;
}
