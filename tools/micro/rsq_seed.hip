// Accuracy of the fp64 seed instructions v_rsq_f64 / v_rcp_f64 on gfx950 (how many refinement steps r_rsqrt / r_rcp need),
// and of one / two Newton steps and the cubic correction of i2c_linalg.hpp::r_rsqrt on top of them.
// Build: hipcc --offload-arch=gfx950 -O3 -o rsq_seed rsq_seed.hip ; run: ./rsq_seed
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ void k(const double* x, double* o, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double p = x[i];
  const double y = __builtin_amdgcn_rsq(p);
  o[i] = y;
  o[n + i] = __builtin_amdgcn_rcp(p);
  {  // one Newton step: y (1.5 - 0.5 p y^2)
    const double e = fma(-(p * y), y, 1.0);
    o[2 * n + i] = fma(0.5 * y, e, y);
  }
  {  // cubic correction (r_rsqrt)
    const double e = fma(-(p * y), y, 1.0);
    o[3 * n + i] = fma(y * e, fma(e, 0.375, 0.5), y);
  }
  {  // rcp: one / two Newton steps
    double r = __builtin_amdgcn_rcp(p);
    r = fma(fma(-p, r, 1.0), r, r);
    o[4 * n + i] = r;
    r = fma(fma(-p, r, 1.0), r, r);
    o[5 * n + i] = r;
  }
}

int main() {
  const int n = 1 << 20;
  std::vector<double> hx(n), ho(6 * n);
  srand(3);
  for (int i = 0; i < n; ++i) hx[i] = exp(log(1e-8) + (log(1e8) - log(1e-8)) * (rand() / (double)RAND_MAX)) * (1.0 + rand() / (double)RAND_MAX);
  double *x, *o;
  hipMalloc(&x, n * 8); hipMalloc(&o, 6 * n * 8);
  hipMemcpy(x, hx.data(), n * 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, x, o, n);
  hipMemcpy(ho.data(), o, 6 * n * 8, hipMemcpyDeviceToHost);
  const char* names[6] = {"v_rsq_f64 seed", "v_rcp_f64 seed", "rsq + 1 Newton step", "rsq + cubic correction (r_rsqrt)", "rcp + 1 Newton step", "rcp + 2 Newton steps (r_rcp)"};
  for (int j = 0; j < 6; ++j) {
    long double worst = 0;
    for (int i = 0; i < n; ++i) {
      const long double ref = (j == 1 || j >= 4) ? 1.0L / (long double)hx[i] : 1.0L / sqrtl((long double)hx[i]);
      const long double e = fabsl(((long double)ho[j * n + i] - ref) / ref);
      if (e > worst) worst = e;
    }
    printf("%-36s max relative error %.3Le = 2^%.1Lf\n", names[j], worst, log2l(worst));
  }
  return 0;
}
