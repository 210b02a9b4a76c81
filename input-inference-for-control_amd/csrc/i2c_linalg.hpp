// Register-resident small dense linear algebra for one trajectory per lane.
//
// Everything is templated on compile-time sizes and fully unrolled so that the packed
// covariance blocks, Cholesky factors and triangular solves live in VGPRs for the whole sweep
// (no runtime-indexed arrays -> no scratch). Symmetric matrices are packed lower, row-major.
#pragma once

#ifdef I2C_HOST_SIM
// Host build of the SAME math, used only by tests/ to check the kernels on a CPU-only box.
#include <cmath>
#define I2C_HD
#else
#include <hip/hip_runtime.h>
#define I2C_HD __host__ __device__
#endif
#define I2C_FN I2C_HD static inline __attribute__((always_inline))

namespace i2c {

constexpr int sym(int n) { return n * (n + 1) / 2; }
// packed index of (i, j) with i >= j
constexpr int tri(int i, int j) { return i * (i + 1) / 2 + j; }
// packed index of (i, j) for any order
constexpr int tri_any(int i, int j) { return i >= j ? tri(i, j) : tri(j, i); }

// scalar math, overloaded on precision (device: OCML via the HIP math headers; host sim: libm)
#ifdef I2C_HOST_SIM
I2C_FN double r_rsqrt(double x) { return 1.0 / std::sqrt(x); }
I2C_FN float r_rsqrt(float x) { return 1.0f / std::sqrt(x); }
I2C_FN double r_exp(double x) { return std::exp(x); }
I2C_FN float r_exp(float x) { return std::exp(x); }
I2C_FN double r_sin(double x) { return std::sin(x); }
I2C_FN float r_sin(float x) { return std::sin(x); }
I2C_FN void r_sincos(double x, double* s, double* c) { *s = std::sin(x); *c = std::cos(x); }
I2C_FN void r_sincos(float x, float* s, float* c) { *s = std::sin(x); *c = std::cos(x); }
#else
I2C_FN double r_rsqrt(double x) { return 1.0 / sqrt(x); }
I2C_FN float r_rsqrt(float x) { return 1.0f / sqrtf(x); }
I2C_FN double r_exp(double x) { return exp(x); }
I2C_FN float r_exp(float x) { return expf(x); }
I2C_FN double r_sin(double x) { return sin(x); }
I2C_FN float r_sin(float x) { return sinf(x); }
I2C_FN void r_sincos(double x, double* s, double* c) { sincos(x, s, c); }
I2C_FN void r_sincos(float x, float* s, float* c) { sincosf(x, s, c); }
#endif
template <typename R> I2C_FN R r_clip(R x, R lo, R hi) { return x < lo ? lo : (x > hi ? hi : x); }

// In-place Cholesky of a packed SPD matrix: a <- L (lower), rinv[j] = 1 / L[j][j].
// Returns false if a pivot is not strictly positive (or NaN): the covariance is not PD.
template <int N, typename R> I2C_FN bool chol(R* a, R* rinv) {
  bool ok = true;
#pragma unroll
  for (int j = 0; j < N; ++j) {
    R s = a[tri(j, j)];
#pragma unroll
    for (int k = 0; k < j; ++k) s -= a[tri(j, k)] * a[tri(j, k)];
    ok = ok && (s > R(0));
    const R r = r_rsqrt(s);
    rinv[j] = r;
    a[tri(j, j)] = s * r;
#pragma unroll
    for (int i = j + 1; i < N; ++i) {
      R v = a[tri(i, j)];
#pragma unroll
      for (int k = 0; k < j; ++k) v -= a[tri(i, k)] * a[tri(j, k)];
      a[tri(i, j)] = v * r;
    }
  }
  return ok;
}

// Solve L y = b in place (forward substitution); `stride` lets b be a row of a row-major matrix.
template <int N, typename R> I2C_FN void fsub(const R* L, const R* rinv, R* b) {
#pragma unroll
  for (int i = 0; i < N; ++i) {
    R v = b[i];
#pragma unroll
    for (int k = 0; k < i; ++k) v -= L[tri(i, k)] * b[k];
    b[i] = v * rinv[i];
  }
}

// Solve L^T x = y in place (back substitution).
template <int N, typename R> I2C_FN void bsub(const R* L, const R* rinv, R* y) {
#pragma unroll
  for (int i = N - 1; i >= 0; --i) {
    R v = y[i];
#pragma unroll
    for (int k = i + 1; k < N; ++k) v -= L[tri(k, i)] * y[k];
    y[i] = v * rinv[i];
  }
}

// y = A x for packed symmetric A (N x N)
template <int N, typename R> I2C_FN void symv(const R* A, const R* x, R* y) {
#pragma unroll
  for (int i = 0; i < N; ++i) {
    R v = R(0);
#pragma unroll
    for (int j = 0; j < N; ++j) v += A[tri_any(i, j)] * x[j];
    y[i] = v;
  }
}

// C (packed sym, M x M) += J (M x N row-major) * D (packed sym N x N) * J^T
template <int M, int N, typename R> I2C_FN void add_JDJt(const R* J, const R* D, R* C) {
  R JD[M * N];
#pragma unroll
  for (int i = 0; i < M; ++i)
#pragma unroll
    for (int k = 0; k < N; ++k) {
      R v = R(0);
#pragma unroll
      for (int l = 0; l < N; ++l) v += J[i * N + l] * D[tri_any(l, k)];
      JD[i * N + k] = v;
    }
#pragma unroll
  for (int i = 0; i < M; ++i)
#pragma unroll
    for (int j = 0; j <= i; ++j) {
      R v = C[tri(i, j)];
#pragma unroll
      for (int k = 0; k < N; ++k) v += JD[i * N + k] * J[j * N + k];
      C[tri(i, j)] = v;
    }
}

}  // namespace i2c
