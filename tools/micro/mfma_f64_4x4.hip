// Microbenchmark + layout probe for the cross-lane machinery of the quad kernels (csrc/i2c_quad.hpp): FOUR trajectories per
// wavefront, one per 16-lane row, every matrix cut into 4 x 4 blocks held one element per lane.
//   (1) v_mfma_f64_4x4x4_4b_f64: which lanes supply A[i][k], B[k][j] and receive D[i][j] of which block (probed with one-hot
//       operands, nothing assumed), then issue cost back to back / dependent, alone and next to independent v_fma_f64;
//   (2) v_mov_b64_dpp row_newbcast (ONE instruction per fp64 broadcast inside a 16-lane row): cost, independent and dependent;
//   (3) v_fmac_f64_dpp row_newbcast (the broadcast fused into the multiply-add; inline asm, hipcc does not form it): result and
//       cost, and whether it needs wait states after a VALU write of its DPP source (hipcc cannot see inside the asm);
//   (4) v_readlane pair, ds_bpermute, 64-bit DPP quad_perm / row_ror moves for comparison.
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_f64_4x4 mfma_f64_4x4.hip ; run: ./mfma_f64_4x4
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

__global__ void k_probe(uint64_t* mask) {  // mask[la * 64 + lb] = lanes whose D is non-zero when only lane la has a = 1, lane lb has b = 1
  const int l = threadIdx.x;
  for (int la = 0; la < 64; ++la)
    for (int lb = 0; lb < 64; ++lb) {
      const double a = l == la ? 1.0 : 0.0, b = l == lb ? 1.0 : 0.0;
      const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
      const uint64_t m = __ballot(d != 0.0);
      if (l == 0) mask[la * 64 + lb] = m;
    }
}

template <int NACC> __global__ void k_time(double* out, uint64_t* clk, int n) {
  const int l = threadIdx.x;
  double a = 1.0 + 1e-3 * l, b = 1.0 - 1e-3 * l;
  double c[NACC];
  for (int i = 0; i < NACC; ++i) c[i] = 0.0;
  const uint64_t t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < n; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
      for (int i = 0; i < NACC; ++i) c[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c[i], 0, 0, 0);
  }
  const uint64_t t1 = __builtin_amdgcn_s_memtime();
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += c[i];
  out[blockIdx.x * 64 + l] = s;
  if (l == 0) clk[blockIdx.x] = t1 - t0;
}
template <int NFMA> __global__ void k_mix(double* out, uint64_t* clk, int n) {
  const int l = threadIdx.x;
  double a = 1.0 + 1e-3 * l, b = 1.0 - 1e-3 * l;
  double c = 0.0;
  double f[8];
  for (int i = 0; i < 8; ++i) f[i] = 1.0 + 1e-9 * (l + i);
  const uint64_t t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < n; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      c = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0);
#pragma unroll
      for (int i = 0; i < NFMA; ++i) f[i & 7] = __builtin_fma(f[i & 7], 1.0000001, 1e-7);
    }
  }
  const uint64_t t1 = __builtin_amdgcn_s_memtime();
  double s = c;
  for (int i = 0; i < 8; ++i) s += f[i];
  out[blockIdx.x * 64 + l] = s;
  if (l == 0) clk[blockIdx.x] = t1 - t0;
}

template <int K> __device__ inline double bcast64(double x) {  // one v_mov_b64_dpp
  long v = __builtin_bit_cast(long, x);
  v = __builtin_amdgcn_update_dpp((long)0, v, 0x150 + K, 0xf, 0xf, false);
  return __builtin_bit_cast(double, v);
}
template <int K> __device__ inline double fmac_bc(double acc, double b, double c) {  // acc += b[lane K of the row] * c
  asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(b), "v"(c), "n"(K));
  return acc;
}
template <int K> __device__ inline double fmac_bc_nop(double acc, double b, double c) {
  asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(b), "v"(c), "n"(K));
  return acc;
}
template <int CTRL> __device__ inline double dpp64(double x) {  // any other control: two v_mov_b32_dpp
  long v = __builtin_bit_cast(long, x);
  v = __builtin_amdgcn_update_dpp((long)0, v, CTRL, 0xf, 0xf, false);
  return __builtin_bit_cast(double, v);
}

__global__ void k_cross(const double* in, double* res, uint64_t* clk, int n) {
  const int l = threadIdx.x;
  const double x = in[l], y = in[64 + l];
  // results: (0) mov bcast, (1) fused fmac bcast on a source read from memory, (2) fused fmac on a source just written by a VALU
  // instruction (hazard probe, no nop), (3) the same with s_nop 1, (4) quad_perm [1,0,3,2] 64-bit, (5) row_ror:8 64-bit
  res[0 * 64 + l] = bcast64<5>(x);
  res[1 * 64 + l] = fmac_bc<6>(1.0, x, y);
  {
    double s = x * 3.0 + y;  // VALU write immediately before the DPP read of s
    res[2 * 64 + l] = fmac_bc<7>(0.5, s, y);
    double s2 = x * 5.0 - y;
    res[3 * 64 + l] = fmac_bc_nop<9>(0.25, s2, y);
  }
  res[4 * 64 + l] = dpp64<0xB1>(x);
  res[5 * 64 + l] = dpp64<0x128>(x);
  // ---- timings ----
  uint64_t t[12];
  double a0 = x, a1 = y, a2 = x + 1, a3 = y + 1;
  t[0] = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < n; ++it) a0 = bcast64<3>(a0) * 1.0000001;  // dependent: mov_dpp -> mul -> mov_dpp ...
  t[1] = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < n; ++it) {  // independent: 4 mov_dpp + 4 fma per step (sources not recently written)
    const double b0 = bcast64<1>(x), b1 = bcast64<2>(x), b2 = bcast64<3>(y), b3 = bcast64<4>(y);
    a0 = __builtin_fma(b0, 1.0000001, a0), a1 = __builtin_fma(b1, 1.0000001, a1), a2 = __builtin_fma(b2, 1.0000001, a2), a3 = __builtin_fma(b3, 1.0000001, a3);
    asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
  }
  t[2] = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < n; ++it) {  // fused, independent accumulators, constant sources: 4 per step
    a0 = fmac_bc<1>(a0, x, y), a1 = fmac_bc<2>(a1, x, y), a2 = fmac_bc<3>(a2, y, x), a3 = fmac_bc<4>(a3, y, x);
  }
  t[3] = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < n; ++it) {  // fused, dependent through the accumulator: 4 per step
    a0 = fmac_bc<1>(a0, x, y), a0 = fmac_bc<2>(a0, x, y), a0 = fmac_bc<3>(a0, y, x), a0 = fmac_bc<4>(a0, y, x);
  }
  t[4] = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < n; ++it) {  // fused, dependent through the DPP source (needs the nop): 2 per step
    a1 = fmac_bc_nop<1>(a1, a0, y);
    a0 = fmac_bc_nop<2>(a0, a1, x);
  }
  t[5] = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < n; ++it) {  // plain v_fma_f64 dependent, 4 per step (the yardstick)
    a0 = __builtin_fma(a0, 1.0000001, y), a0 = __builtin_fma(a0, 0.9999999, x), a0 = __builtin_fma(a0, 1.0000001, y), a0 = __builtin_fma(a0, 0.9999999, x);
    asm volatile("" : "+v"(a0));
  }
  t[6] = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < n; ++it) {  // readlane pair -> scalar operand -> fma, 4 per step
    int lo = __double2loint(a1), hi = __double2hiint(a1);
    const double s0 = __hiloint2double(__builtin_amdgcn_readlane(hi, 3), __builtin_amdgcn_readlane(lo, 3));
    const double s1 = __hiloint2double(__builtin_amdgcn_readlane(hi, 17), __builtin_amdgcn_readlane(lo, 17));
    const double s2 = __hiloint2double(__builtin_amdgcn_readlane(hi, 33), __builtin_amdgcn_readlane(lo, 33));
    const double s3 = __hiloint2double(__builtin_amdgcn_readlane(hi, 49), __builtin_amdgcn_readlane(lo, 49));
    a2 = __builtin_fma(s0, 1e-9, a2), a3 = __builtin_fma(s1, 1e-9, a3), a2 = __builtin_fma(s2, 1e-9, a2), a3 = __builtin_fma(s3, 1e-9, a3);
    asm volatile("" : "+v"(a2), "+v"(a3));
  }
  t[7] = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < n; ++it) {  // 64-bit quad_perm / row_ror moves (two v_mov_b32_dpp each) + fma, 2 per step
    a2 = __builtin_fma(dpp64<0xB1>(a3), 1e-9, a2);
    a3 = __builtin_fma(dpp64<0x128>(a2), 1e-9, a3);
  }
  t[8] = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < n; ++it) {  // v_rsq_f64 dependent chain
    a1 = __builtin_amdgcn_rsq(a1 + 2.0);
  }
  t[9] = __builtin_amdgcn_s_memtime();
  res[6 * 64 + l] = a0 + a1 + a2 + a3;
  if (l == 0)
    for (int i = 0; i < 9; ++i) clk[i] = t[i + 1] - t[i];
}

int main() {
  uint64_t *mask, *clk;
  double* out;
  hipMalloc(&mask, 4096 * 8);
  hipMalloc(&clk, 8 * 8192);
  hipMalloc(&out, 64 * 8 * 8192);
  hipLaunchKernelGGL(k_probe, dim3(1), dim3(64), 0, 0, mask);
  static uint64_t hm[4096];
  hipMemcpy(hm, mask, sizeof(hm), hipMemcpyDeviceToHost);
  // decode: for every D lane, the (la, lb) pairs that reach it
  printf("v_mfma_f64_4x4x4_4b: D lane <- list of (A lane, B lane) pairs (k = 0..3)\n");
  for (int ld = 0; ld < 64; ++ld) {
    printf("  D lane %2d:", ld);
    int cnt = 0;
    for (int la = 0; la < 64; ++la)
      for (int lb = 0; lb < 64; ++lb)
        if (hm[la * 64 + lb] >> ld & 1) {
          printf(" (%d,%d)", la, lb);
          ++cnt;
        }
    printf("   [%d]\n", cnt);
  }
  const int n = 1000;
  static uint64_t h[8192];
  auto avg = [&](int blocks) { hipDeviceSynchronize(); hipMemcpy(h, clk, 8 * blocks, hipMemcpyDeviceToHost); double a = 0; for (int i = 0; i < blocks; ++i) a += double(h[i]); return a / blocks; };
  for (int blocks : {1, 1024, 2048, 4096}) {
    hipLaunchKernelGGL(k_time<1>, dim3(blocks), dim3(64), 0, 0, out, clk, n);
    printf("v_mfma_f64_4x4x4 dependent chain (1 accumulator),  %4d waves: %.1f clocks per MFMA\n", blocks, avg(blocks) / (n * 8.0));
    hipLaunchKernelGGL(k_time<4>, dim3(blocks), dim3(64), 0, 0, out, clk, n);
    printf("v_mfma_f64_4x4x4 independent (4 accumulators),     %4d waves: %.1f clocks per MFMA\n", blocks, avg(blocks) / (n * 8.0 * 4));
    hipLaunchKernelGGL(k_mix<0>, dim3(blocks), dim3(64), 0, 0, out, clk, n);
    double base = avg(blocks) / (n * 8.0);
    hipLaunchKernelGGL(k_mix<2>, dim3(blocks), dim3(64), 0, 0, out, clk, n);
    double m2 = avg(blocks) / (n * 8.0);
    hipLaunchKernelGGL(k_mix<4>, dim3(blocks), dim3(64), 0, 0, out, clk, n);
    double m4 = avg(blocks) / (n * 8.0);
    hipLaunchKernelGGL(k_mix<8>, dim3(blocks), dim3(64), 0, 0, out, clk, n);
    double m8 = avg(blocks) / (n * 8.0);
    printf("one 4x4x4 MFMA + k independent v_fma_f64 per step, %4d waves: k=0 %.1f  k=2 %.1f  k=4 %.1f  k=8 %.1f clocks per step\n", blocks, base, m2, m4, m8);
  }
  double hx[128], hr[7 * 64], *X, *RES;
  for (int i = 0; i < 64; ++i) hx[i] = 1.0 + i, hx[64 + i] = 0.5 + 0.25 * i;
  hipMalloc(&X, 1024);
  hipMalloc(&RES, sizeof(hr));
  hipMemcpy(X, hx, 1024, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k_cross, dim3(1), dim3(64), 0, 0, X, RES, clk, n);
  hipDeviceSynchronize();
  hipMemcpy(hr, RES, sizeof(hr), hipMemcpyDeviceToHost);
  hipMemcpy(h, clk, 9 * 8, hipMemcpyDeviceToHost);
  bool ok[6] = {true, true, true, true, true, true};
  for (int l = 0; l < 64; ++l) {
    const int r = l & ~15;
    const double x = hx[l], y = hx[64 + l];
    (void)x;
    ok[0] = ok[0] && hr[l] == hx[r + 5];
    ok[1] = ok[1] && hr[64 + l] == fma(hx[r + 6], y, 1.0);
    ok[2] = ok[2] && hr[128 + l] == fma(hx[r + 7] * 3.0 + hx[64 + r + 7], y, 0.5);
    ok[3] = ok[3] && hr[192 + l] == fma(hx[r + 9] * 5.0 - hx[64 + r + 9], y, 0.25);
    ok[4] = ok[4] && hr[256 + l] == hx[l ^ 1];
    ok[5] = ok[5] && hr[320 + l] == hx[r + ((l & 15) + 8) % 16];
  }
  printf("v_mov_b64_dpp row_newbcast: %s\n", ok[0] ? "OK" : "MISMATCH");
  printf("v_fmac_f64_dpp row_newbcast (source from memory): %s\n", ok[1] ? "OK" : "MISMATCH");
  printf("v_fmac_f64_dpp right after a VALU write of its DPP source, no nop: %s\n", ok[2] ? "OK" : "MISMATCH (hazard)");
  printf("v_fmac_f64_dpp right after a VALU write of its DPP source, s_nop 1: %s\n", ok[3] ? "OK" : "MISMATCH");
  printf("64-bit quad_perm [1,0,3,2]: %s   64-bit row_ror:8: %s\n", ok[4] ? "OK" : "MISMATCH", ok[5] ? "OK" : "MISMATCH");
  printf("clocks: dependent (mov_dpp + mul) %.1f | 4 x (mov_dpp + fma) independent %.1f per pair | fused fmac_dpp independent %.1f each | fused dependent on acc %.1f each | fused dependent on DPP source (+s_nop 1) %.1f each\n",
         h[0] / double(n), h[1] / (4.0 * n), h[2] / (4.0 * n), h[3] / (4.0 * n), h[4] / (2.0 * n));
  printf("clocks: plain dependent v_fma_f64 %.1f each | readlane pair + fma %.1f per value | 64-bit quad_perm/row_ror + fma dependent %.1f each | v_rsq_f64 + add dependent %.1f\n",
         h[5] / (4.0 * n), h[6] / (4.0 * n), h[7] / (2.0 * n), h[8] / double(n));
  return 0;
}
