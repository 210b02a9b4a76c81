// Microbenchmark + layout probe for the fp64 matrix instruction the wave kernels are built on (csrc/i2c_wave.hpp):
//   v_mfma_f64_16x16x4_f64   D(16x16) += A(16x4) B(4x16), one wavefront
// (1) operand / result layout against a host product; (2) issue cost back to back on independent accumulators and the
// latency of a dependent chain; (3) the cross-lane moves the kernels use beside it: DPP row_newbcast of an fp64 value,
// v_permlane32_swap / v_permlane16_swap all-reduce over the four 16-lane rows, an LDS round trip.
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_f64 mfma_f64.hip ; run: ./mfma_f64
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

typedef double d4 __attribute__((ext_vector_type(4)));

__global__ void k_layout(const double* A, const double* B, double* D) {  // A [16][4], B [4][16] row-major, D [16][16]
  const int l = threadIdx.x, q = l >> 4, j = l & 15;
  const double a = A[j * 4 + q];   // assumed A operand: lane (q, i) holds A[i][k = q]
  const double b = B[q * 16 + j];  // assumed B operand: lane (q, j) holds B[k = q][j]
  d4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  for (int v = 0; v < 4; ++v) D[(q + 4 * v) * 16 + j] = c[v];  // assumed D: register v of lane (q, j) = D[q + 4 v][j]
}

template <int NACC> __global__ void k_time(double* out, uint64_t* clk, int n) {
  const int l = threadIdx.x;
  double a = 1.0 + 1e-3 * l, b = 1.0 - 1e-3 * l;
  d4 c[NACC];
  for (int i = 0; i < NACC; ++i) c[i] = d4{0, 0, 0, 0};
  const uint64_t t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < n; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
      for (int i = 0; i < NACC; ++i) c[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c[i], 0, 0, 0);
  }
  const uint64_t t1 = __builtin_amdgcn_s_memtime();
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += c[i][0] + c[i][1] + c[i][2] + c[i][3];
  out[blockIdx.x * 64 + l] = s;
  if (l == 0) clk[blockIdx.x] = t1 - t0;
}

// MFMA interleaved with independent fp64 FMAs: how much vector work hides behind one matrix instruction
template <int NFMA> __global__ void k_mix(double* out, uint64_t* clk, int n) {
  const int l = threadIdx.x;
  double a = 1.0 + 1e-3 * l, b = 1.0 - 1e-3 * l;
  d4 c = {0, 0, 0, 0};
  double f[8];
  for (int i = 0; i < 8; ++i) f[i] = 1.0 + 1e-9 * (l + i);
  const uint64_t t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < n; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
#pragma unroll
      for (int i = 0; i < NFMA; ++i) f[i & 7] = __builtin_fma(f[i & 7], 1.0000001, 1e-7);
    }
  }
  const uint64_t t1 = __builtin_amdgcn_s_memtime();
  double s = c[0] + c[1] + c[2] + c[3];
  for (int i = 0; i < 8; ++i) s += f[i];
  out[blockIdx.x * 64 + l] = s;
  if (l == 0) clk[blockIdx.x] = t1 - t0;
}

template <int K> __device__ inline double bcast_row(double x) {  // lane K of my 16-lane row
  int lo = __double2loint(x), hi = __double2hiint(x);
  lo = __builtin_amdgcn_update_dpp(0, lo, 0x150 + K, 0xf, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(0, hi, 0x150 + K, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
__device__ inline double rows_allreduce(double x) {  // sum over the four 16-lane rows, result in every row
  int lo = __double2loint(x), hi = __double2hiint(x);
  auto l32 = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
  auto h32 = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
  double y = __hiloint2double(h32[0], l32[0]) + __hiloint2double(h32[1], l32[1]);
  lo = __double2loint(y), hi = __double2hiint(y);
  auto l16 = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
  auto h16 = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
  return __hiloint2double(h16[0], l16[0]) + __hiloint2double(h16[1], l16[1]);
}
__global__ void k_cross(const double* in, double* out_b, double* out_r, uint64_t* clk, int n) {
  __shared__ double sh[64];
  const int l = threadIdx.x;
  double x = in[l];
  out_b[l] = bcast_row<5>(x);
  out_r[l] = rows_allreduce(x);
  // timing: dependent chains of each cross-lane primitive
  double y = x;
  uint64_t t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < n; ++it) y = bcast_row<3>(y) * 1.0000001;
  uint64_t t1 = __builtin_amdgcn_s_memtime();
  double z = x;
  for (int it = 0; it < n; ++it) z = rows_allreduce(z) * 0.25;
  uint64_t t2 = __builtin_amdgcn_s_memtime();
  double w = x;
  for (int it = 0; it < n; ++it) {  // LDS transpose-style round trip
    sh[l] = w;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    w = sh[(l * 17 + 1) & 63] * 1.0000001;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
  uint64_t t3 = __builtin_amdgcn_s_memtime();
  out_b[64 + l] = y + z + w;
  if (l == 0) { clk[0] = t1 - t0; clk[1] = t2 - t1; clk[2] = t3 - t2; }
}

int main() {
  double hA[64], hB[64], hD[256], *A, *B, *D;
  srand(1);
  for (int i = 0; i < 64; ++i) { hA[i] = rand() / (double)RAND_MAX - 0.5; hB[i] = rand() / (double)RAND_MAX - 0.5; }
  hipMalloc(&A, 512); hipMalloc(&B, 512); hipMalloc(&D, 2048);
  hipMemcpy(A, hA, 512, hipMemcpyHostToDevice); hipMemcpy(B, hB, 512, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k_layout, dim3(1), dim3(64), 0, 0, A, B, D);
  hipMemcpy(hD, D, 2048, hipMemcpyDeviceToHost);
  double err = 0;
  for (int i = 0; i < 16; ++i)
    for (int j = 0; j < 16; ++j) {
      double r = 0;
      for (int k = 0; k < 4; ++k) r += hA[i * 4 + k] * hB[k * 16 + j];
      err = fmax(err, fabs(r - hD[i * 16 + j]));
    }
  printf("layout: A lane(q,i)=A[i][q], B lane(q,j)=B[q][j], D reg v lane(q,j)=D[q+4v][j] : max |err| = %.3e  %s\n", err, err < 1e-14 ? "OK" : "MISMATCH");

  double* out; uint64_t* clk;
  hipMalloc(&out, 64 * 8 * 4096); hipMalloc(&clk, 8 * 4096);
  const int n = 1000;
  static uint64_t h[4096];
  auto avg = [&](int blocks) { hipDeviceSynchronize(); hipMemcpy(h, clk, 8 * blocks, hipMemcpyDeviceToHost); double a = 0; for (int i = 0; i < blocks; ++i) a += double(h[i]); return a / blocks; };
  for (int blocks : {1, 1024}) {
    hipLaunchKernelGGL(k_time<1>, dim3(blocks), dim3(64), 0, 0, out, clk, n);
    printf("v_mfma_f64_16x16x4_f64 dependent chain (1 accumulator),  %4d waves: %.1f clocks per MFMA\n", blocks, avg(blocks) / (n * 8.0));
    hipLaunchKernelGGL(k_time<4>, dim3(blocks), dim3(64), 0, 0, out, clk, n);
    printf("v_mfma_f64_16x16x4_f64 independent (4 accumulators),     %4d waves: %.1f clocks per MFMA\n", blocks, avg(blocks) / (n * 8.0 * 4));
    hipLaunchKernelGGL(k_mix<0>, dim3(blocks), dim3(64), 0, 0, out, clk, n);
    double base = avg(blocks) / (n * 8.0);
    hipLaunchKernelGGL(k_mix<4>, dim3(blocks), dim3(64), 0, 0, out, clk, n);
    double m4 = avg(blocks) / (n * 8.0);
    hipLaunchKernelGGL(k_mix<8>, dim3(blocks), dim3(64), 0, 0, out, clk, n);
    double m8 = avg(blocks) / (n * 8.0);
    hipLaunchKernelGGL(k_mix<16>, dim3(blocks), dim3(64), 0, 0, out, clk, n);
    double m16 = avg(blocks) / (n * 8.0);
    printf("one MFMA + k independent v_fma_f64 per step,             %4d waves: k=0 %.1f  k=4 %.1f  k=8 %.1f  k=16 %.1f clocks per step\n", blocks, base, m4, m8, m16);
  }
  // two waves per SIMD (2048 workgroups of one wave on 1024 SIMDs ... 4 per SIMD at 4096)
  for (int blocks : {2048, 4096}) {
    hipLaunchKernelGGL(k_time<4>, dim3(blocks), dim3(64), 0, 0, out, clk, n);
    hipDeviceSynchronize();
  }
  double hx[64], hb[128], hr[64], *X, *OB, *OR;
  for (int i = 0; i < 64; ++i) hx[i] = 1.0 + i;
  hipMalloc(&X, 512); hipMalloc(&OB, 1024); hipMalloc(&OR, 512);
  hipMemcpy(X, hx, 512, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k_cross, dim3(1), dim3(64), 0, 0, X, OB, OR, clk, n);
  hipDeviceSynchronize();
  hipMemcpy(hb, OB, 1024, hipMemcpyDeviceToHost); hipMemcpy(hr, OR, 512, hipMemcpyDeviceToHost); hipMemcpy(h, clk, 24, hipMemcpyDeviceToHost);
  bool okb = true, okr = true;
  for (int l = 0; l < 64; ++l) {
    okb = okb && hb[l] == hx[(l & ~15) + 5];
    okr = okr && hr[l] == hx[l & 15] + hx[16 + (l & 15)] + hx[32 + (l & 15)] + hx[48 + (l & 15)];
  }
  printf("DPP row_newbcast of an fp64 value: %s, %.1f clocks per dependent (bcast + mul)\n", okb ? "OK" : "MISMATCH", h[0] / double(n));
  printf("permlane32/16 swap all-reduce over the four rows: %s, %.1f clocks per dependent (allreduce + mul)\n", okr ? "OK" : "MISMATCH", h[1] / double(n));
  printf("LDS write -> fence -> permuted read round trip: %.1f clocks\n", h[2] / double(n));
  return 0;
}
