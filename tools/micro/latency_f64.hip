// Dependent-chain latencies of a LONE wave on gfx950 (the quad forward kernel at B = 4096 is one wave per SIMD walking a chain of
// dependent operations): per-instruction cost of long unrolled chains (64 per loop trip, the ~32-clock loop overhead amortised).
// Build: hipcc --offload-arch=gfx950 -O3 -o latency_f64 latency_f64.hip ; run: ./latency_f64
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

template <int K> __device__ inline double bcast64(double x) {
  long v = __builtin_bit_cast(long, x);
  v = __builtin_amdgcn_update_dpp((long)0, v, 0x150 + K, 0xf, 0xf, false);
  return __builtin_bit_cast(double, v);
}
template <int CTRL> __device__ inline double dpp64(double x) {
  long v = __builtin_bit_cast(long, x);
  v = __builtin_amdgcn_update_dpp((long)0, v, CTRL, 0xf, 0xf, false);
  return __builtin_bit_cast(double, v);
}

__global__ void k(const double* in, double* out, uint64_t* clk, int n) {
  __shared__ double sh[256];
  const int l = threadIdx.x;
  const double x = in[l], y = in[64 + l];
  uint64_t t[16];
  double a = x, b = y, c = x + 1, d = y + 1;
  int s = 0;
  t[s++] = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < n; ++it) {  // 0: dependent v_fma_f64
#pragma unroll
    for (int i = 0; i < 32; ++i) { a = __builtin_fma(a, 1.0000001, y); a = __builtin_fma(a, 0.9999999, x); }
  }
  t[s++] = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < n; ++it) {  // 1: four independent chains of v_fma_f64 (per instruction)
#pragma unroll
    for (int i = 0; i < 16; ++i) { a = __builtin_fma(a, 1.0000001, y); b = __builtin_fma(b, 0.9999999, x); c = __builtin_fma(c, 1.0000001, y); d = __builtin_fma(d, 0.9999999, x); }
  }
  t[s++] = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < n; ++it) {  // 2: dependent v_mul_f64
#pragma unroll
    for (int i = 0; i < 32; ++i) { a = a * 1.0000001; a = a * 0.9999999; }
  }
  t[s++] = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < n; ++it) {  // 3: dependent v_rsq_f64 + v_add_f64 pairs (per pair)
#pragma unroll
    for (int i = 0; i < 32; ++i) a = __builtin_amdgcn_rsq(a) + 2.0;
  }
  t[s++] = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < n; ++it) {  // 4: independent v_rsq_f64 (4 chains, per instruction)
#pragma unroll
    for (int i = 0; i < 8; ++i) { a = __builtin_amdgcn_rsq(a + 2.0); b = __builtin_amdgcn_rsq(b + 2.0); c = __builtin_amdgcn_rsq(c + 2.0); d = __builtin_amdgcn_rsq(d + 2.0); }
  }
  t[s++] = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < n; ++it) {  // 5: dependent 4x4x4 MFMA -> v_add_f64 -> MFMA (per pair)
#pragma unroll
    for (int i = 0; i < 16; ++i) { a = __builtin_amdgcn_mfma_f64_4x4x4f64(a, 1e-3, 0.0, 0, 0, 0); a = a + 1.0; }
  }
  t[s++] = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < n; ++it) {  // 6: dependent MFMA -> MFMA through the accumulator (per MFMA)
#pragma unroll
    for (int i = 0; i < 32; ++i) a = __builtin_amdgcn_mfma_f64_4x4x4f64(x, 1e-3, a, 0, 0, 0);
  }
  t[s++] = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < n; ++it) {  // 7: dependent MFMA -> MFMA through the B operand (per MFMA)
#pragma unroll
    for (int i = 0; i < 32; ++i) a = __builtin_amdgcn_mfma_f64_4x4x4f64(1e-3, a, y, 0, 0, 0);
  }
  t[s++] = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < n; ++it) {  // 8: LDS round trip: write b64, fence, six broadcast reads, use (per trip)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      sh[l] = a;
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
      const int g = (l >> 2) & 3;
      a = (sh[g * 16] + sh[g * 16 + 4]) + (sh[g * 16 + 5] + sh[g * 16 + 8]) + (sh[g * 16 + 10] + sh[g * 16 + 15]) * 1e-3;
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
  }
  t[s++] = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < n; ++it) {  // 9: dependent 64-bit quad_perm broadcast + add (per pair)
#pragma unroll
    for (int i = 0; i < 32; ++i) a = dpp64<0x55>(a) + 1e-3;
  }
  t[s++] = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < n; ++it) {  // 10: dependent v_mov_b64_dpp row_newbcast + add (per pair)
#pragma unroll
    for (int i = 0; i < 32; ++i) a = bcast64<3>(a) + 1e-3;
  }
  t[s++] = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < n; ++it) {  // 11: v_cndmask pair (fp64 select) dependent chain
#pragma unroll
    for (int i = 0; i < 32; ++i) a = (l & (1 << (i & 3))) ? a : b;
    asm volatile("" : "+v"(a));
  }
  t[s++] = __builtin_amdgcn_s_memtime();
  out[l] = a + b + c + d;
  if (l == 0)
    for (int i = 0; i + 1 < s; ++i) clk[i] = t[i + 1] - t[i];
}

int main() {
  double hx[128], *X, *O;
  uint64_t *C, h[16];
  for (int i = 0; i < 128; ++i) hx[i] = 1.0 + 1e-3 * i;
  hipMalloc(&X, 1024); hipMalloc(&O, 512); hipMalloc(&C, 128);
  hipMemcpy(X, hx, 1024, hipMemcpyHostToDevice);
  const int n = 200;
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, X, O, C, n);
  hipDeviceSynchronize();
  hipMemcpy(h, C, 128, hipMemcpyDeviceToHost);
  const char* names[12] = {"dependent v_fma_f64", "independent v_fma_f64 (4 chains)", "dependent v_mul_f64", "dependent v_rsq_f64 + v_add_f64 (pair)",
                           "independent v_rsq_f64 + add (4 chains, per pair)", "dependent 4x4x4 MFMA + v_add_f64 (pair)", "dependent MFMA via accumulator",
                           "dependent MFMA via B operand", "LDS round trip: write, six reads, ~6 flops, fences", "dependent 64-bit quad_perm + add (pair)",
                           "dependent v_mov_b64_dpp row_newbcast + add (pair)", "dependent fp64 select (2 x v_cndmask_b32)"};
  const double per[12] = {64, 64, 64, 32, 32, 16, 32, 32, 8, 32, 32, 32};
  for (int i = 0; i < 12; ++i) printf("%-56s %7.1f clocks\n", names[i], h[i] / (n * per[i]));
  return 0;
}
