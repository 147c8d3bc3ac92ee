// pure MFMA rate + effective shader clock under load: bf16 32x32x16 vs f32 32x32x2, random vs zero operands
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int MODE>
__global__ __launch_bounds__(512, 2) void k(const uint4* in, float* out, int iters, unsigned long long* clk) {
  const int tid = threadIdx.x + blockIdx.x * 512;
  uint4 ra[6], rb[6];
  for (int i = 0; i < 6; ++i) { ra[i] = in[(tid * 12 + i) & 0xfffff]; rb[i] = in[(tid * 12 + 6 + i) & 0xfffff]; }
  f32x16 acc[4];
  for (int j = 0; j < 4; ++j) for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
  unsigned long long t0 = __builtin_readcyclecounter(), w0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {
#pragma unroll
      for (int p = 0; p < 6; ++p)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ra[(p + j) % 6]), __builtin_bit_cast(bf16x8, rb[(p * 5 + j) % 6]), acc[j], 0, 0, 0);
    } else if (MODE == 2) {
#pragma unroll
      for (int p = 0; p < 6; ++p)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ra[(p + j) % 6]), __builtin_bit_cast(f16x8, rb[(p * 5 + j) % 6]), acc[j], 0, 0, 0);
    } else {
#pragma unroll
      for (int p = 0; p < 6; ++p)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(__builtin_bit_cast(float, ra[(p + j) % 6].x), __builtin_bit_cast(float, rb[(p * 5 + j) % 6].y), acc[j], 0, 0, 0);
    }
  }
  unsigned long long t1 = __builtin_readcyclecounter(), w1 = wall_clock64();
  float s = 0.f;
  for (int j = 0; j < 4; ++j) for (int e = 0; e < 16; ++e) s += acc[j][e];
  out[tid] = s;
  if (threadIdx.x == 0) { clk[blockIdx.x * 2] = t1 - t0; clk[blockIdx.x * 2 + 1] = w1 - w0; }
}
int main() {
  const int WG = 256 * 8, iters = 4000;
  std::vector<uint32_t> h(4 << 20);
  uint4* din; float* dout; unsigned long long* dclk;
  hipMalloc(&din, h.size() * 4); hipMalloc(&dout, WG * 512 * 4); hipMalloc(&dclk, WG * 16);
  for (int mode = 0; mode < 3; ++mode)
    for (int fill = 0; fill < 3; ++fill) {
      for (auto& v : h) {
        if (fill == 0) v = 0;
        else if (fill == 1) { float f = (float)rand() / RAND_MAX * 2.f - 1.f; uint32_t u; memcpy(&u, &f, 4); v = mode == 0 ? ((u >> 16) | (u & 0xffff0000u)) : u; if (mode == 2) { _Float16 hh = (_Float16)f; uint16_t hb; memcpy(&hb, &hh, 2); v = hb | ((uint32_t)hb << 16) ^ 0x00010000u; } }
        else v = ((uint32_t)rand() << 16) ^ rand();     // random bits incl. low planes (like lo-plane mantissas): clear exponent msb to avoid inf/nan
        if (fill == 2) v &= 0xbf7fbf7fu;
      }
      hipMemcpy(din, h.data(), h.size() * 4, hipMemcpyHostToDevice);
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(WG), dim3(512), 0, 0, din, dout, iters, dclk);
        else if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(WG), dim3(512), 0, 0, din, dout, iters, dclk);
        else hipLaunchKernelGGL(k<1>, dim3(WG), dim3(512), 0, 0, din, dout, iters / 2, dclk);
        hipEventRecord(e1); hipEventSynchronize(e1);
      }
      float ms; hipEventElapsedTime(&ms, e0, e1);
      std::vector<unsigned long long> c(WG * 2);
      hipMemcpy(c.data(), dclk, WG * 16, hipMemcpyDeviceToHost);
      double cyc = 0, wall = 0; for (int i = 0; i < WG; ++i) { cyc += c[2 * i]; wall += c[2 * i + 1]; }
      const double flops = mode != 1 ? 2.0 * 32 * 32 * 16 * 24.0 * iters * WG * 8 : 2.0 * 32 * 32 * 2 * 24.0 * (iters / 2) * WG * 8;
      printf("%s fill=%d: %.3f ms  %.1f TF  shader clock %.3f GHz (cyc/wall100MHz)\n", mode == 0 ? "bf16 32x32x16" : mode == 2 ? "f16  32x32x16" : "f32  32x32x2 ", fill, ms, flops / ms / 1e9, cyc / wall * 0.1);
    }
  return 0;
}
